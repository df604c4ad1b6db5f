"""GEMM unit cases driven straight through the C ABI (ghn3_run with a one-op program)."""

import numpy as np
import torch

from ghn3_amd import _lib as L


def _ref(buf, off=0):
    return (buf, off)


def run_gemm_case(ctx, M, N, K, a_mode, b_mode, tile=0, ctype=L.CT_F32, gather=False, qs_map=False, epilogue='none',
                  accum=False, seed=0, dev='cuda'):
    """Runs one GEMM through libghn3_hip.so and returns (got, expected fp64) as numpy arrays."""
    rs = np.random.RandomState(seed)
    lda_rows = (M if a_mode == L.MODE_ROW else K)
    lda_cols = (K if a_mode == L.MODE_ROW else M)
    ldb_rows = (N if b_mode == L.MODE_ROW else K)
    ldb_cols = (K if b_mode == L.MODE_ROW else N)
    pad = lambda v: (v + 3) // 4 * 4
    extra = 5 if gather else 0
    XA = rs.standard_normal((lda_rows + extra, pad(lda_cols) + 4)).astype(np.float32)
    XB = rs.standard_normal(((ldb_rows + extra) * (2 if qs_map else 1), pad(ldb_cols) + 8)).astype(np.float32)
    lda, ldb = XA.shape[1], XB.shape[1]
    ldc = pad(N) + 4
    c_rows = M + extra
    C0 = rs.standard_normal((c_rows, ldc)).astype(np.float32)
    bias = rs.standard_normal(N * 3 + 7).astype(np.float32)
    resid = rs.standard_normal((c_rows, ldc)).astype(np.float32)
    aux = rs.standard_normal((c_rows, ldc)).astype(np.float32)
    ga = rs.permutation(lda_rows + extra)[:lda_rows].astype(np.int32) if gather else None
    gb = None
    gc = rs.permutation(c_rows)[:M].astype(np.int32) if gather else None
    bq, bs = (0, 0)
    if qs_map:
        bq = max(1, ldb_rows // 3)
        bs = bq + 2

    def rowmap(r, g, q, s):
        r = np.asarray(r)
        if g is not None:
            r = g[r]
        if q > 0:
            r = (r // q) * s + (r % q)
        return r

    # expected (fp64)
    a_idx = rowmap(np.arange(lda_rows), ga, 0, 0)
    b_idx = rowmap(np.arange(ldb_rows), gb, bq, bs)
    A = XA[a_idx][:, :lda_cols].astype(np.float64)
    Bm = XB[b_idx][:, :ldb_cols].astype(np.float64)
    A = A if a_mode == L.MODE_ROW else A.T            # (M, K)
    Bm = Bm.T if b_mode == L.MODE_ROW else Bm         # (K, N)
    acc = A @ Bm
    alpha = 0.5 if epilogue == 'full' else 1.0
    v = acc * alpha
    c_idx = rowmap(np.arange(M), gc, 0, 0)
    aux_out_expected = None
    bias_stride = 1
    if epilogue in ('bias_relu', 'full', 'gelu', 'dgelu', 'drelu'):
        bias_stride = 3 if epilogue == 'full' else 1
        v = v + bias[np.arange(N) * bias_stride][None, :]
    if epilogue in ('gelu', 'full'):
        aux_out_expected = v.copy()
        tt = torch.from_numpy(v)
        v = torch.nn.functional.gelu(tt).numpy()
    if epilogue == 'bias_relu':
        v = np.maximum(v, 0)
    if epilogue == 'drelu':
        v = v * (aux[c_idx][:, :N] > 0)
    if epilogue == 'dgelu':
        z = torch.from_numpy(aux[c_idx][:, :N].astype(np.float64)).requires_grad_(True)
        torch.nn.functional.gelu(z).sum().backward()
        v = v * z.grad.numpy()
    if epilogue == 'full':
        v = v + resid[c_idx][:, :N]
    if accum:
        v = v + C0[c_idx][:, :N]
    bias_expected = None
    if epilogue == 'biasgrad':
        assert a_mode == L.MODE_COL
        bias_stride = 2
        bias = rs.standard_normal(c_rows * 2 + 7).astype(np.float32)
        bias_expected = bias.astype(np.float64).copy()
        np.add.at(bias_expected, c_idx * bias_stride, A.sum(1))
    expected = C0.astype(np.float64).copy()
    expected[c_idx, :N] = v

    # device buffers
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    dA, dB, dC, dbias, dres, daux = t(XA), t(XB), t(C0), t(bias), t(resid), t(aux)
    daux_out = torch.zeros_like(dC)
    dga = t(ga) if ga is not None else None
    dgc = t(gc) if gc is not None else None
    bufs = [dA, dB, dC, dbias, dres, daux, daux_out, dga, dgc]
    ptrs = np.asarray([b.data_ptr() if b is not None else 0 for b in bufs], dtype=np.uint64)
    p = np.zeros(1, dtype=L.PROBLEM_DT)
    for name in ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather'):
        p[name]['buf'] = -1
    p['A']['buf'], p['B']['buf'], p['C']['buf'] = 0, 1, 2
    if epilogue in ('bias_relu', 'full', 'gelu', 'dgelu', 'drelu', 'biasgrad'):
        p['bias']['buf'] = 3
    if epilogue == 'full':
        p['residual']['buf'] = 4
    if epilogue in ('drelu', 'dgelu'):
        p['aux_in']['buf'] = 5
    if epilogue in ('gelu', 'full'):
        p['aux_out']['buf'] = 6
    if gather:
        p['a_gather']['buf'] = 7
        p['c_gather']['buf'] = 8
    p['M'], p['N'], p['K'], p['lda'], p['ldb'], p['ldc'] = M, N, K, lda, ldb, ldc
    p['a_mode'], p['b_mode'] = a_mode, b_mode
    p['b_q'], p['b_s'] = bq, bs
    p['bias_stride'] = bias_stride
    p['act'] = {'bias_relu': L.ACT_RELU, 'gelu': L.ACT_GELU, 'full': L.ACT_GELU}.get(epilogue, L.ACT_NONE)
    p['dact'] = {'drelu': L.DACT_RELU, 'dgelu': L.DACT_GELU}.get(epilogue, L.DACT_NONE)
    p['flags'] = (L.GEMM_ACCUM if accum else 0) | (L.GEMM_BIASGRAD if epilogue == 'biasgrad' else 0)
    p['alpha'] = alpha
    op = np.zeros(1, dtype=L.OP_DT)
    op['kind'] = L.OP_GEMM
    op['flags'] = 1 + ctype
    op['i'][0][:3] = (0, 1, tile)
    op['r']['buf'][:] = -1
    ctx.run(op, p, ptrs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dC.cpu().numpy()
    extra_out = None
    if bias_expected is not None:
        extra_out = (dbias.cpu().numpy(), bias_expected)
    if aux_out_expected is not None:
        e2 = np.zeros_like(expected)
        e2[c_idx, :N] = aux_out_expected
        extra_out = (daux_out.cpu().numpy()[c_idx][:, :N], aux_out_expected)
    return got, expected, extra_out


CASES = []
for a_mode in (L.MODE_ROW, L.MODE_COL):
    for b_mode in (L.MODE_ROW, L.MODE_COL):
        CASES.append(dict(M=200, N=136, K=100, a_mode=a_mode, b_mode=b_mode, tile=64))
        CASES.append(dict(M=300, N=260, K=72, a_mode=a_mode, b_mode=b_mode, tile=128))
        CASES.append(dict(M=37, N=50, K=19, a_mode=a_mode, b_mode=b_mode, tile=0, gather=True, accum=True))
CASES += [
    dict(M=130, N=70, K=33, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=64, qs_map=True, epilogue='bias_relu'),
    dict(M=130, N=70, K=64, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=128, epilogue='full', gather=True),
    dict(M=65, N=129, K=40, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, tile=64, epilogue='dgelu'),
    dict(M=65, N=129, K=40, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=64, epilogue='drelu', qs_map=True),
    dict(M=9, N=1000, K=384, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=0, epilogue='gelu'),
    dict(M=1, N=5, K=3, a_mode=L.MODE_COL, b_mode=L.MODE_ROW, tile=0),
    dict(M=300, N=200, K=150, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=128, epilogue='biasgrad', accum=True),
    dict(M=70, N=33, K=45, a_mode=L.MODE_COL, b_mode=L.MODE_ROW, tile=64, epilogue='biasgrad', gather=True),
    dict(M=1100, N=1300, K=64, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=128),
    dict(M=40, N=2100, K=40, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, tile=64),
]
# the 32x32 split-K-in-workgroup kernel (tile code 32), every addressing mode and epilogue
for a_mode in (L.MODE_ROW, L.MODE_COL):
    for b_mode in (L.MODE_ROW, L.MODE_COL):
        CASES.append(dict(M=100, N=77, K=391, a_mode=a_mode, b_mode=b_mode, tile=32))
        CASES.append(dict(M=33, N=65, K=19, a_mode=a_mode, b_mode=b_mode, tile=32, gather=True, accum=True))
CASES += [
    dict(M=256, N=1152, K=384, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=32, epilogue='full', gather=True),
    dict(M=256, N=384, K=1536, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, tile=32, epilogue='dgelu'),
    dict(M=384, N=1536, K=256, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=32, epilogue='biasgrad', accum=True),
    dict(M=70, N=40, K=130, a_mode=L.MODE_COL, b_mode=L.MODE_ROW, tile=32, epilogue='drelu', qs_map=True),
    dict(M=50, N=90, K=70, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=32, epilogue='bias_relu', qs_map=True),
    dict(M=2, N=3, K=1, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=32),
]
