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
    for name in ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather', 'lim', 'alpha_amax', 'B2', 'mtiles'):
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
# split-bf16 weight-gradient kernel (tile code 48, gemm_wg.hip): fp32 activations reduced over rows (COL / COL), fused bias
# gradient, accumulate; Graphormer shapes of ghn3xlm16 / ghn3tm8, ragged M / N (multiples of 4), K tails, a problem the kernel
# cannot take (N % 4 != 0: falls back to the exact 64 x 64 tiles)
WG_CASES = [
    dict(M=384, N=1536, K=256, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48, epilogue='biasgrad', accum=True),
    dict(M=1152, N=384, K=256, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48, accum=True),
    dict(M=1536, N=384, K=512, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48, epilogue='biasgrad'),
    dict(M=64, N=256, K=128, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48, epilogue='biasgrad'),
    dict(M=100, N=76, K=300, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48, epilogue='biasgrad', accum=True),
    dict(M=4, N=8, K=1, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48),
    dict(M=132, N=72, K=33, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48),
    dict(M=130, N=70, K=33, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=48, epilogue='biasgrad'),
    # tile code 49: the same kernel on 128 x 128 tiles (8 waves), used for the grouped launch of all layers' weight gradients
    dict(M=384, N=1536, K=256, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=49, epilogue='biasgrad', accum=True),
    dict(M=1152, N=384, K=256, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=49, accum=True),
    dict(M=384, N=384, K=512, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=49, epilogue='biasgrad'),
    dict(M=100, N=76, K=300, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=49, epilogue='biasgrad', accum=True),
    dict(M=260, N=132, K=70, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=49, epilogue='biasgrad'),
    dict(M=4, N=8, K=1, a_mode=L.MODE_COL, b_mode=L.MODE_COL, tile=49),
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
    # deep-prefetch variant (ROW-mode A, <= 256 tiles, K >= 256, K % 4 == 0, no gather / row map): ragged M / N, K tails
    # inside and across the 128-wide chunks, trip counts that are not multiples of the prefetch depth
    dict(M=250, N=380, K=388, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=32, epilogue='full'),
    dict(M=256, N=384, K=384, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=32, epilogue='bias_relu'),
    dict(M=100, N=76, K=1156, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, tile=32, accum=True),
    dict(M=33, N=36, K=644, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, tile=32, epilogue='drelu'),
    dict(M=31, N=64, K=256, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, tile=32),
    dict(M=512, N=500, K=1152, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, tile=32),
]


# ---------------------------------------------------------------------------------------------------------------
# 16-bit-operand pipeline: GHN3_OP_CAST16 (straight / transposed / column sums) feeding a GHN3_GEMM_OP16 problem.
# The expectation is computed in fp64 from the operands rounded on the CPU exactly as the cast kernel rounds them,
# so the tolerance only has to cover fp32 accumulation order.
# ---------------------------------------------------------------------------------------------------------------
def round16(x, ctype):
    x = np.asarray(x, dtype=np.float32)
    if ctype == L.CT_F16:
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def run_op16_case(ctx, M, N, K, ctype=L.CT_F16, tile=0, grid_cap=0, a_transposed=False, b_transposed=False, kmap=None, ksplit=1,
                  epilogue='none', accum=False, colsum=False, cmap=False, seed=0, dev='cuda', pins=None, mtiles=None,
                  ragged=0):
    """mtiles: row-tile table of tile code 28 as a list of (m0, mi[, extent]); ragged = 1: the extents are valid COLUMNS
    (columns of a tile beyond its extent are don't-care), 2: valid reduction lengths (A holds zeros beyond them).
    pins: a list (one entry per problem of ONE launch) of XCD numbers or None: problem r multiplies the first
    M - 29 r rows of the same A into its own C (ghn3_gemm_problem::xcd_pin)."""
    rs = np.random.RandomState(seed)
    r64 = lambda v: (v + 63) // 64 * 64
    mt_rows = lambda mi_: 32 * mi_                                 # height code = rows / 32 (2, 4, 6 .. 10)
    # fp32 sources.  A: [M][K] (or [K][M] when the cast transposes it); B physical k extent Kp (k-map) or K.
    kq, ks = kmap if kmap else (0, 0)
    Kp = ((K + kq - 1) // kq) * ks if kmap else K
    A_log = rs.standard_normal((M, K)).astype(np.float32)
    if ragged == 2:
        for (m0_, mi_, e_) in mtiles:
            A_log[m0_:m0_ + mt_rows(mi_), e_:] = 0
    B_phys = rs.standard_normal((N, Kp)).astype(np.float32)
    kk = np.arange(K)
    kphys = (kk // kq) * ks + kk % kq if kmap else kk
    B_log = B_phys[:, kphys]                                      # [N][K]
    srcA = np.ascontiguousarray(A_log.T if a_transposed else A_log)
    srcB = np.ascontiguousarray(B_phys.T if b_transposed else B_phys)
    assert not (b_transposed and kmap)
    pad4 = lambda v: (v + 3) // 4 * 4
    ld_sa, ld_sb = pad4(srcA.shape[1]) + 4, pad4(srcB.shape[1]) + 8
    SA = np.zeros((srcA.shape[0], ld_sa), np.float32); SA[:, :srcA.shape[1]] = srcA
    SB = np.zeros((srcB.shape[0], ld_sb), np.float32); SB[:, :srcB.shape[1]] = srcB
    src = np.concatenate([SA.reshape(-1), SB.reshape(-1)])
    offB = SA.size
    assert offB % 4 == 0
    # 16-bit destinations (elements)
    lda = r64(K) + 8
    ldb = r64(Kp) + 64 + 8
    dA_off, dB_off = 0, (M * lda + 63) // 64 * 64
    total16 = dB_off + N * ldb + 64
    descs = np.zeros(2, dtype=L.CAST_DT)
    bf = ctype == L.CT_BF16
    blocks = 0
    for d, (off, rows, cols, ld_s, trans, dst_off, ld_d) in enumerate((
            (0, srcA.shape[0], srcA.shape[1], ld_sa, a_transposed, dA_off, lda),
            (offB, srcB.shape[0], srcB.shape[1], ld_sb, b_transposed, dB_off, ldb))):
        D = descs[d]
        D['src_off'], D['rows'], D['cols'], D['ld_src'] = off, rows, cols, ld_s
        if trans:
            D['dstT_off'], D['ld_dstT'] = dst_off, ld_d
            D['flags'] = L.CAST_TRANSPOSED | (L.CAST_TRANSPOSED_BF16 if bf else 0)
        else:
            D['dst_off'], D['ld_dst'] = dst_off, ld_d
            D['flags'] = L.CAST_STRAIGHT | (L.CAST_STRAIGHT_BF16 if bf else 0)
        D['block_start'] = blocks
        blocks += ((rows + 63) // 64) * ((cols + 63) // 64)
    ldc = pad4(N) + 4
    c_rows = M * 2 if cmap else M
    cq, cs = (max(1, M // 3), max(1, M // 3) + 2) if cmap else (0, 0)
    m = np.arange(M)
    c_idx = (m // cq) * cs + m % cq if cmap else m
    c_rows = int(c_idx.max()) + 1
    C0 = rs.standard_normal((c_rows, ldc)).astype(np.float32)
    if ksplit > 1 and not accum:
        C0[:] = 0
    bias = rs.standard_normal(max(N, c_rows) + 3).astype(np.float32)
    dbias0 = rs.standard_normal(c_rows + 3).astype(np.float32)
    if colsum:
        assert a_transposed
        descs[0]['flags'] |= L.CAST_COLSUM
        descs[0]['bias_q'], descs[0]['bias_s'] = cq, cs
    # expectation
    Ar, Br = round16(A_log, ctype).astype(np.float64), round16(B_log, ctype).astype(np.float64)
    v = Ar @ Br.T
    aux = rs.standard_normal((c_rows, ldc)).astype(np.float32)
    res = rs.standard_normal((c_rows, ldc)).astype(np.float32)
    pre = None
    if epilogue == 'bias_relu':
        v = np.maximum(v + bias[:N][None, :], 0)
    if epilogue == 'drelu_res':                       # alpha, bias, aux_out, dReLU(aux_in), residual
        pre = 0.5 * v + bias[:N][None, :]
        v = np.where(aux[c_idx][:, :N] > 0, pre, 0.0) + res[c_idx][:, :N]
    if accum or ksplit > 1:
        v = v + C0[c_idx][:, :N]
    expected = C0.astype(np.float64).copy()
    expected[c_idx, :N] = v
    dbias_exp = dbias0.astype(np.float64).copy()
    if colsum:
        np.add.at(dbias_exp, c_idx, A_log.astype(np.float64).sum(1))

    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_src, d_16 = t(src), torch.full((total16,), 0x7e00 if not bf else 0x7fc0, dtype=torch.int16, device=dev)  # NaNs
    # the regions the GEMM reads as padding beyond what the cast writes must be finite: B rows are read up to
    # round64(K) logical k -> zero the B area like the program does for its workspace
    d_16[dB_off:] = 0
    d_desc, d_C, d_bias, d_dbias = t(descs.view(np.uint8)), t(C0), t(bias), t(dbias0)
    d_aux, d_res, d_pre = t(aux), t(res), torch.zeros(c_rows, ldc, device=dev)
    mt = np.zeros((1, 3), np.int32)
    if mtiles is not None:
        mt = np.asarray([(m_[0], m_[1], m_[2] if len(m_) > 2 else (1 << 30)) for m_ in mtiles], dtype=np.int32)
    d_mt = t(mt)
    bufs = [d_src, d_16, d_desc, d_C, d_bias, d_dbias, d_aux, d_res, d_pre, d_mt]
    ptrs = np.asarray([b.data_ptr() for b in bufs], dtype=np.uint64)
    p = np.zeros(1, dtype=L.PROBLEM_DT)
    for name in ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather', 'lim', 'alpha_amax',
                 'B2', 'mtiles'):
        p[name]['buf'] = -1
    p['ln_p']['buf'] = -1
    if mtiles is not None:
        p['mtiles']['buf'], p['n_mtiles'], p['lim_kind'] = 9, len(mt), ragged
    p['A']['buf'], p['A']['off'] = 1, 2 * dA_off
    p['B']['buf'], p['B']['off'] = 1, 2 * dB_off
    p['C']['buf'] = 3
    if epilogue == 'bias_relu':
        p['bias']['buf'] = 4
        p['act'] = L.ACT_RELU
    p['M'], p['N'], p['K'], p['lda'], p['ldb'], p['ldc'] = M, N, K, lda, ldb, ldc
    p['c_q'], p['c_s'] = cq, cs
    p['flags'] = L.GEMM_OP16 | (L.GEMM_ACCUM if accum else 0)
    p['alpha'] = 1.0
    if epilogue == 'drelu_res':
        p['bias']['buf'], p['aux_in']['buf'], p['residual']['buf'], p['aux_out']['buf'] = 4, 6, 7, 8
        p['dact'], p['alpha'] = L.DACT_RELU, 0.5
    p['ksplit'] = ksplit
    p['b_kq'], p['b_ks'] = kq, ks
    p['bias_stride'] = 1
    ops = np.zeros(2, dtype=L.OP_DT)
    ops['r']['buf'][:] = -1
    ops[0]['kind'] = L.OP_CAST16
    ops[0]['i'][:2] = (2, blocks)
    ops[0]['r']['buf'][:4] = (0, 1, 2, 5 if colsum else -1)
    ops[1]['kind'] = L.OP_GEMM
    ops[1]['flags'] = 1 + ctype
    ops[1]['i'][:4] = (0, 1, tile, grid_cap)
    if pins is not None:
        assert not (cmap or colsum or accum or ksplit > 1)
        R = len(pins)
        p = np.repeat(p, R)
        d_CR = torch.from_numpy(np.ascontiguousarray(np.stack([C0] * R))).to(dev)
        ptrs[3] = d_CR.data_ptr()
        for r in range(R):
            p[r]['M'] = max(1, M - 29 * r)
            p[r]['C']['off'] = 4 * r * C0.size
            p[r]['xcd_pin'] = 0 if pins[r] is None else 1 + pins[r]
        ops[1]['i'][1] = R
        ctx.run(ops, p, ptrs, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = d_CR.cpu().numpy()
        out = []
        for r in range(R):
            e = C0.astype(np.float64).copy()
            mr = max(1, M - 29 * r)
            e[:mr, :N] = expected[:mr, :N]
            out.append((got[r], e))
        return out
    ctx.run(ops, p, ptrs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_C.cpu().numpy()
    if ragged == 1:                                   # columns beyond a tile's extent are don't-care
        for (m0_, mi_, e_) in mtiles:
            rows_ = c_idx[m0_:min(m0_ + mt_rows(mi_), M)]
            expected[rows_, e_:N] = got[rows_, e_:N]
    out = [(got, expected)]
    if pre is not None:
        e_pre = np.zeros((c_rows, ldc))
        e_pre[c_idx, :N] = pre
        out.append((d_pre.cpu().numpy(), e_pre))
    if colsum:
        out.append((d_dbias.cpu().numpy(), dbias_exp))
    return out


OP16_CASES = [
    dict(M=300, N=200, K=150),
    dict(M=128, N=128, K=64),
    dict(M=1100, N=1300, K=512, epilogue='bias_relu'),
    dict(M=300, N=200, K=150, a_transposed=True, b_transposed=True, colsum=True, cmap=True, accum=True),
    dict(M=70, N=3072 // 8, K=120, kmap=(24, 40), ksplit=2),
    dict(M=533, N=384, K=64 * 48, kmap=(64, 96), ksplit=5),
    dict(M=200, N=130, K=1000, a_transposed=True, b_transposed=True, ksplit=3, accum=True),
    dict(M=5, N=7, K=9),
    # the 256 x 256 tile variant (tile code 24)
    dict(M=300, N=520, K=150, tile=24),
    dict(M=1100, N=1300, K=512, epilogue='bias_relu', tile=24),
    dict(M=300, N=200, K=150, a_transposed=True, b_transposed=True, colsum=True, cmap=True, accum=True, tile=24),
    dict(M=533, N=384, K=64 * 48, kmap=(64, 96), ksplit=5, tile=24),
    dict(M=5, N=7, K=9, tile=24),
    # the 256 x 128 three-stage variant (tile code 20): deep K loops (ring wrap-around), k-map + split-K, ragged M / N
    dict(M=300, N=520, K=150, tile=20),
    dict(M=1100, N=1300, K=512, epilogue='bias_relu', tile=20),
    dict(M=533, N=384, K=64 * 48, kmap=(64, 96), ksplit=5, tile=20),
    dict(M=768, N=200, K=64 * 21, ksplit=3, tile=20),
    dict(M=5, N=7, K=9, tile=20),
    dict(M=300, N=200, K=64, tile=20, accum=True),
    # the persistent output-heavy variant (tile code 25): transposed products + direct float4 stores, the next tile's
    # first k-tile issued from the last iteration of the current one; ragged M / N, one and many k-tiles, odd and even
    # k-tile counts (stage parity carried across tiles), a row map of C, more tiles than workgroups (grid cap)
    dict(M=300, N=520, K=150, tile=25),
    dict(M=1100, N=1300, K=512, tile=25, grid_cap=3),
    dict(M=1100, N=780, K=64 * 3, tile=25, grid_cap=2),
    dict(M=700, N=3072 // 4, K=64, tile=25, grid_cap=2, cmap=True),
    dict(M=5, N=7, K=9, tile=25),
    dict(M=2048, N=1024, K=768, a_transposed=True, b_transposed=True, tile=25, grid_cap=5),
    # tile code 30 (round 6, an experiment kept behind GHN3_WGRAD_TILE=30): the persistent stream with two accumulator sets
    # (256 x 128 tiles; the stores of a tile leave during the next one): short / long reductions (fewer / more than the eight
    # k-tiles the drain is spread over), row map, ragged edges, grid caps
    dict(M=300, N=520, K=150, tile=30),
    dict(M=1100, N=1300, K=512, tile=30, grid_cap=3),
    dict(M=1100, N=780, K=64 * 3, tile=30, grid_cap=2),
    dict(M=700, N=3072 // 4, K=64, tile=30, grid_cap=2, cmap=True),
    dict(M=5, N=7 * 4, K=9, tile=30),
    dict(M=2048, N=1024, K=768, a_transposed=True, b_transposed=True, tile=30, grid_cap=5),
    dict(M=2100, N=1156, K=64 * 9 + 24, tile=30, cmap=True),
    dict(M=520, N=384, K=64 * 24, tile=30, grid_cap=8),
    # the persistent 8-phase stream (tile code 29): the DMA stream continues across tile boundaries, a tile's stores are
    # deferred into the next tile's first k-tile; one / two / many k-tiles per tile (stream shorter than the look-ahead),
    # more tiles than workgroups, a row map of C, ragged M / N
    dict(M=300, N=520, K=150, tile=29),
    dict(M=1100, N=1300, K=512, tile=29, grid_cap=3),
    dict(M=1100, N=780, K=64 * 3, tile=29, grid_cap=2),
    dict(M=700, N=3072 // 4, K=64, tile=29, grid_cap=2, cmap=True),
    dict(M=5, N=8, K=9, tile=29),
    dict(M=2048, N=1024, K=768, a_transposed=True, b_transposed=True, tile=29, grid_cap=5),
    dict(M=4096, N=512, K=64 * 9, tile=29, grid_cap=8),
    dict(M=2100, N=512, K=100, tile=29, grid_cap=16, cmap=True),
    # ... and with a bounded number of tiles per workgroup (grid_cap = workers | tiles << 16): chunks of 1 / 2 / 3 tiles
    dict(M=4096, N=512, K=64 * 9, tile=29, grid_cap=8 | (1 << 16)),
    dict(M=2100, N=1024, K=100, tile=29, grid_cap=8 | (2 << 16), cmap=True),
    dict(M=4096, N=1300, K=200, tile=29, grid_cap=16 | (3 << 16)),
    # XCD-pinned problems of one launch (xcd_pin): several problems on one XCD, XCDs without a problem, unpinned problems
    # behind the pinned ids, fewer than 8 problems (pins ignored), the 256 x 128 variant and the persistent grid
    dict(M=300, N=520, K=150, pins=[0, 1, 2, 3, 4, 5, 6, 7, 0, 3, None, 5]),
    dict(M=300, N=200, K=200, epilogue='bias_relu', pins=[2, 2, 2, None, 7, None, 2, 0, None]),
    dict(M=533, N=384, K=64 * 12, kmap=(64, 96), tile=20, pins=[0, 1, 2, 3, 4, 5, 6, 7]),
    dict(M=700, N=300, K=130, tile=20, pins=[1, None, 6, 6, 3, None, None, 1, 4, 4], grid_cap=3),
    dict(M=300, N=200, K=64, pins=[3, 5, None]),
    # the 8-phase kernel (tile code 28): default 256-row tiles, ragged M / N, one / few / many k-tiles (ring wrap-around and
    # the tail waits), K = 0 is covered by the pinned planes of the program; row-tile tables with 192 / 256 / 320-row tiles,
    # extents as columns and as reduction lengths, k-map, row map of C, accumulate, the full epilogue, XCD pins, grid cap
    dict(M=300, N=520, K=150, tile=28),
    dict(M=1100, N=1300, K=512, epilogue='bias_relu', tile=28),
    dict(M=533, N=384, K=64 * 48, kmap=(64, 96), tile=28),
    dict(M=260, N=250, K=64, tile=28, accum=True),
    dict(M=200, N=7, K=9, tile=28),
    dict(M=300, N=200, K=150, a_transposed=True, b_transposed=True, cmap=True, accum=True, tile=28),
    dict(M=700, N=600, K=64 * 7, tile=28, mtiles=[(0, 8), (256, 6), (448, 8)]),
    dict(M=533, N=1000, K=64 * 5 + 8, tile=28, mtiles=[(0, 6), (192, 6), (384, 6)], epilogue='drelu_res'),
    dict(M=700, N=900, K=64 * 3, tile=28, mtiles=[(0, 6, 900), (192, 6, 900), (384, 8, 520), (640, 6, 64)], ragged=1, epilogue='bias_relu'),
    dict(M=768, N=300, K=64 * 9, tile=28, mtiles=[(0, 8, 576), (256, 8, 300), (512, 8, 64)], ragged=2, kmap=(64, 96)),
    dict(M=640, N=512, K=64 * 2, tile=28, mtiles=[(0, 6), (192, 8), (448, 6)], grid_cap=3),
    # round 6: height codes 6 .. 10 = 32 mi rows -- 224- and 288-row tiles (sub-tiles of 4 + 3 / 5 + 4 MFMA rows per wave), so
    # that the row tiles of one streamed panel can be equal: 533 rows = 288 + 288, 660 rows = 3 x 224
    dict(M=533, N=1000, K=64 * 5 + 8, tile=28, mtiles=[(0, 9), (288, 9)], epilogue='drelu_res'),
    dict(M=660, N=700, K=64 * 6, tile=28, mtiles=[(0, 7), (224, 7), (448, 7)], epilogue='bias_relu'),
    dict(M=575, N=520, K=64 * 11 + 24, tile=28, mtiles=[(0, 9, 520), (288, 9, 300)], ragged=1, kmap=(64, 96)),
    dict(M=672, N=300, K=64 * 9, tile=28, mtiles=[(0, 7, 576), (224, 7, 320), (448, 7, 64)], ragged=2, kmap=(64, 96), cmap=True),
    dict(M=1000, N=260, K=64 * 2 + 3, tile=28, mtiles=[(0, 6), (192, 8), (448, 10), (768, 7), (992, 6)], accum=True),
    dict(M=288, N=256, K=64, tile=28, mtiles=[(0, 9)]),
    # skinny row tiles (64 / 128 rows): the inference forward's families
    dict(M=109, N=1000, K=64 * 5 + 8, tile=28, mtiles=[(0, 4)], epilogue='bias_relu'),
    dict(M=40, N=520, K=64 * 48, tile=28, mtiles=[(0, 2, 520)], ragged=1, kmap=(64, 96)),
    dict(M=230, N=300, K=64 * 3, tile=28, mtiles=[(0, 2), (64, 4), (192, 2)], cmap=True),
    dict(M=100, N=260, K=130, tile=28, mtiles=[(0, 2, 130), (64, 2, 64)], ragged=2, epilogue='drelu_res'),
    dict(M=533, N=384, K=64 * 12, kmap=(64, 96), tile=28, mtiles=[(0, 9), (288, 9)], pins=[0, 1, 2, 3, 4, 5, 6, 7]),
    dict(M=533, N=384, K=64 * 12, kmap=(64, 96), tile=28, pins=[0, 1, 2, 3, 4, 5, 6, 7]),
    dict(M=700, N=300, K=130, tile=28, pins=[1, None, 6, 6, 3, None, None, 1, 4, 4], grid_cap=3),
    # persistent workgroups (grid cap): 3 workgroups stride over 99 / 30 tiles
    dict(M=1100, N=1300, K=512, epilogue='bias_relu', grid_cap=3),
    dict(M=1100, N=1300, K=200, tile=24, grid_cap=3, accum=True),
]


# ------------------------------------------------------------------------------------------------------------------
# LayerNorm row prologue of the small-problem kernel (ghn3_gemm_problem::ln_kind)
# ------------------------------------------------------------------------------------------------------------------
LN_CASES = [  # K <= 512: register-cached single pass; K > 512: two-pass path; ragged M / N; both B modes
    dict(kind=1, M=256, N=1152, K=384, b_mode=L.MODE_ROW),
    dict(kind=1, M=70, N=100, K=384, b_mode=L.MODE_ROW),
    dict(kind=1, M=45, N=64, K=640, b_mode=L.MODE_ROW),
    dict(kind=1, M=33, N=40, K=24, b_mode=L.MODE_COL),
    dict(kind=2, M=256, N=384, K=384, b_mode=L.MODE_COL),
    dict(kind=2, M=70, N=100, K=128, b_mode=L.MODE_COL, res=False),
    dict(kind=2, M=45, N=64, K=640, b_mode=L.MODE_ROW),
]


def run_ln_case(ctx, kind, M, N, K, b_mode, res=True, seed=0, dev='cuda'):
    """C = LN(A) @ B (kind 1) or LN'(dy = A) @ B (kind 2) through the C ABI; returns [(got, expected fp64), ...] for C
    and every by-product the prologue writes."""
    rs = np.random.RandomState(seed)
    lda = K + 8
    A = rs.standard_normal((M, lda)).astype(np.float32) * 1.5 + 0.3
    XB = rs.standard_normal((N, K + 4) if b_mode == L.MODE_ROW else (K, (N + 3) // 4 * 4 + 4)).astype(np.float32)
    ldb = XB.shape[1]
    ldc = (N + 3) // 4 * 4
    gamma = (1.0 + 0.2 * rs.standard_normal(K)).astype(np.float32)
    beta = (0.1 * rs.standard_normal(K)).astype(np.float32)
    X = rs.standard_normal((M, lda)).astype(np.float32)
    RES = rs.standard_normal((M, lda)).astype(np.float32)
    eps = 1e-5
    A64, X64 = A[:, :K].astype(np.float64), X[:, :K].astype(np.float64)
    if kind == 1:
        mu = A64.mean(1, keepdims=True)
        rstd = 1.0 / np.sqrt(((A64 - mu) ** 2).mean(1, keepdims=True) + eps)
        T = (A64 - mu) * rstd * gamma + beta
        mean_in = rstd_in = None
    else:
        mu = X64.mean(1, keepdims=True)
        rstd = 1.0 / np.sqrt(((X64 - mu) ** 2).mean(1, keepdims=True) + eps)
        mean_in, rstd_in = mu[:, 0].astype(np.float32), rstd[:, 0].astype(np.float32)
        xh = (X64 - mean_in[:, None].astype(np.float64)) * rstd_in[:, None].astype(np.float64)
        dg = A64 * gamma
        T = rstd_in[:, None].astype(np.float64) * (dg - dg.mean(1, keepdims=True) - xh * (dg * xh).mean(1, keepdims=True))
        if res:
            T = T + RES[:, :K]
    Bm = XB[:, :K].astype(np.float64).T if b_mode == L.MODE_ROW else XB[:, :N].astype(np.float64)
    expC = T @ Bm
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    dA, dB, dC = t(A), t(XB), torch.zeros(M, ldc, device=dev)
    dgam, dbet, dX, dRES = t(gamma), t(beta), t(X), t(RES)
    dmean = t(mean_in) if kind == 2 else torch.zeros(M, device=dev)
    drstd = t(rstd_in) if kind == 2 else torch.zeros(M, device=dev)
    dout = torch.zeros(M, lda, device=dev)
    bufs = [dA, dB, dC, dgam, dbet, dX, dRES, dmean, drstd, dout]
    ptrs = np.asarray([b.data_ptr() for b in bufs], dtype=np.uint64)
    p = np.zeros(1, dtype=L.PROBLEM_DT)
    for name in ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather', 'lim',
                 'alpha_amax'):
        p[name]['buf'] = -1
    p['ln_p']['buf'] = -1
    p['A']['buf'], p['B']['buf'], p['C']['buf'] = 0, 1, 2
    p['M'], p['N'], p['K'], p['lda'], p['ldb'], p['ldc'] = M, N, K, lda, ldb, ldc
    p['a_mode'], p['b_mode'], p['alpha'], p['ksplit'] = L.MODE_ROW, b_mode, 1.0, 1
    p['ln_kind'], p['ln_eps'] = kind, eps
    if kind == 1:
        p['ln_p']['buf'][0, :5] = (3, 4, 7, 8, 9)
    else:
        p['ln_p']['buf'][0, :6] = (3, 5, 7, 8, 6 if res else -1, 9)
    op = np.zeros(1, dtype=L.OP_DT)
    op['kind'] = L.OP_GEMM
    op['i'][0][:3] = (0, 1, 0)
    op['r']['buf'][:] = -1
    ctx.run(op, p, ptrs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = [(dC.cpu().numpy()[:, :N], expC), (dout.cpu().numpy()[:, :K], T)]
    if kind == 1:
        out += [(dmean.cpu().numpy(), mu[:, 0]), (drstd.cpu().numpy(), rstd[:, 0])]
    return out


# ---------------------------------------------------------------------------------------------------------------
# Split-bf16 ("x3") GEMM: GHN3_OP_CAST16 with GHN3_CAST_SPLIT (hi / lo copies, straight + transposed) feeding
# GHN3_GEMM_X3 problems.  Expectation: fp64 product of the fp32 operands (the deviation is the dropped lo.lo term and
# the bf16 rounding of the lo halves: ~1e-5 relative).
# ---------------------------------------------------------------------------------------------------------------
def x3_setup(W, dev='cuda', frag=False, f16=False):
    """fp32 weight W [R][Cc] (numpy) -> (device fp32 weight, shadow buffer (uint16), layout dict) through one cast op.
    frag: fragment-major copies (GHN3_CAST_FRAG, the operand layout of tile codes 44 / 45)."""
    R, Cc = W.shape
    n = R * Cc
    lay = dict(hi=0, hiT=n, lo=2 * n)
    dW = torch.from_numpy(np.ascontiguousarray(W)).to(dev)
    shadow = torch.zeros(4 * n + 256, dtype=torch.int16, device=dev)
    desc = np.zeros(1, dtype=L.CAST_DT)
    desc['rows'], desc['cols'], desc['ld_src'] = R, Cc, Cc
    desc['dst_off'], desc['ld_dst'] = lay['hi'], Cc
    desc['dstT_off'], desc['ld_dstT'] = lay['hiT'], R
    desc['lo_off'] = lay['lo']
    desc['flags'] = L.CAST_STRAIGHT | L.CAST_TRANSPOSED | L.CAST_SPLIT | (L.CAST_FRAG if frag else 0) | \
        (L.CAST_SPLIT_F16 if f16 else 0)
    return dW, shadow, lay, desc


def run_x3_case(ctx, M, N, K, transposed=False, tile=40, ksplit=1, slice_=0, epilogue='none', seed=0, dev='cuda', ln=0,
                ln_res=True, f16=False, big=False):
    """C = A W^T (W [N][K]) or, transposed, C = A W (W [K][N]).  Returns [(got, expected)].
    ln = 1 / 2 (tile codes 44 / 45, gemm_x3d.hip): the LayerNorm forward / backward row prologue of A
    (ghn3_gemm_problem::ln_kind); the by-products (normalised rows, mean, rstd / the propagated gradient) are returned too."""
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((M, K)).astype(np.float32)
    W = (rs.standard_normal((K, N) if transposed else (N, K)) * 0.05).astype(np.float32)
    if big:
        # out-of-range operands of the f16-piece path (|activation| > 65504, |W| * 2^6 > 65504): row 1 of A and one weight;
        # the kernel saturates the pieces -- results of that row / column are finite, everything else is unaffected
        A[1] *= 1e6
        W[3, 5] = 2000.0
    bias = rs.standard_normal(N).astype(np.float32)
    resid = rs.standard_normal((M, N)).astype(np.float32)
    aux = rs.standard_normal((M, N)).astype(np.float32)
    A_eff = A.astype(np.float64)
    ln_extra = []
    if ln:
        A = (A * 1.5 + 0.3).astype(np.float32)
        gamma = (1.0 + 0.2 * rs.standard_normal(K)).astype(np.float32)
        beta = (0.1 * rs.standard_normal(K)).astype(np.float32)
        X = rs.standard_normal((M, K)).astype(np.float32)
        RES = rs.standard_normal((M, K)).astype(np.float32)
        A64, X64 = A.astype(np.float64), X.astype(np.float64)
        if ln == 1:
            mu = A64.mean(1, keepdims=True)
            rstd = 1.0 / np.sqrt(((A64 - mu) ** 2).mean(1, keepdims=True) + 1e-5)
            A_eff = (A64 - mu) * rstd * gamma + beta
            mean_in = rstd_in = None
        else:
            mu = X64.mean(1, keepdims=True)
            rstd = 1.0 / np.sqrt(((X64 - mu) ** 2).mean(1, keepdims=True) + 1e-5)
            mean_in, rstd_in = mu[:, 0].astype(np.float32), rstd[:, 0].astype(np.float32)
            xh = (X64 - mean_in[:, None].astype(np.float64)) * rstd_in[:, None].astype(np.float64)
            dg = A64 * gamma
            A_eff = rstd_in[:, None].astype(np.float64) * (dg - dg.mean(1, keepdims=True) - xh * (dg * xh).mean(1, keepdims=True))
            if ln_res:
                A_eff = A_eff + RES
    acc = A_eff @ (W.astype(np.float64) if transposed else W.astype(np.float64).T)
    v = acc.copy()
    aux_expected = None
    if epilogue in ('full', 'gelu', 'bias_relu', 'bias_res'):
        v = v + bias[None, :]
    if epilogue == 'bias_res':
        v = v + resid
    if epilogue in ('full', 'gelu'):
        aux_expected = v.copy()
        v = torch.nn.functional.gelu(torch.from_numpy(v)).numpy()
    if epilogue == 'bias_relu':
        v = np.maximum(v, 0)
    if epilogue == 'dgelu':
        z = torch.from_numpy(aux.astype(np.float64)).requires_grad_(True)
        torch.nn.functional.gelu(z).sum().backward()
        v = v * z.grad.numpy()
    if epilogue == 'drelu':
        v = v * (aux > 0)
    if epilogue == 'full':
        v = v + resid
    assert not (f16 and transposed)                  # (f16 pieces: the straight copies / forward products only)
    dW, shadow, lay, desc = x3_setup(W, dev, frag=tile in (44, 45), f16=f16)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    dA, dbias, dres, daux = t(A), t(bias), t(resid), t(aux)
    dC = torch.full((max(ksplit, 1), M, N), 7.0, dtype=torch.float32, device=dev)
    daux_out = torch.zeros(M, N, dtype=torch.float32, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
    bufs = [dA, shadow, dC, dbias, dres, daux, daux_out, dW, ddesc]
    if ln:
        dmean = t(mean_in) if ln == 2 else torch.zeros(M, device=dev)
        drstd = t(rstd_in) if ln == 2 else torch.zeros(M, device=dev)
        dlnout = torch.zeros(M, K, device=dev)
        bufs += [t(gamma), t(beta), t(X), t(RES), dmean, drstd, dlnout]          # 9 .. 15
    ptrs = np.asarray([b.data_ptr() for b in bufs], dtype=np.uint64)
    hi = lay['hiT'] if transposed else lay['hi']
    kc = K // ksplit
    sl = slice_ or 64 * max(d for d in (6, 4, 3, 2, 1) if (kc // 64) % d == 0)
    p = np.zeros(ksplit, dtype=L.PROBLEM_DT)
    for name in L._REF_NAMES + ('lim', 'alpha_amax', 'B2', 'mtiles'):
        p[name]['buf'] = -1
    p['ln_p']['buf'] = -1
    for j in range(ksplit):
        p[j]['A']['buf'], p[j]['A']['off'] = 0, 4 * j * kc
        p[j]['B']['buf'], p[j]['B']['off'] = 1, 2 * (hi + j * kc)
        p[j]['B2']['buf'], p[j]['B2']['off'] = 1, 2 * (hi + lay['lo'] + j * kc)
        p[j]['C']['buf'], p[j]['C']['off'] = 2, 4 * j * M * N
        p[j]['M'], p[j]['N'], p[j]['K'], p[j]['lda'], p[j]['ldb'], p[j]['ldc'] = M, N, kc, K, K, N
        p[j]['flags'], p[j]['alpha'], p[j]['x3_slice'] = L.GEMM_X3, 1.0, sl
        if f16:
            p[j]['flags'], p[j]['alpha'] = L.GEMM_X3 | L.GEMM_X3F16, 2.0 ** -L.X3F16_WSHIFT
    if epilogue in ('full', 'gelu', 'bias_relu', 'bias_res'):
        p[0]['bias']['buf'] = 3
    if epilogue in ('full', 'bias_res'):
        p[0]['residual']['buf'] = 4
    if epilogue in ('dgelu', 'drelu'):
        p[0]['aux_in']['buf'] = 5
    if epilogue in ('full', 'gelu'):
        p[0]['aux_out']['buf'] = 6
    p[0]['act'] = {'bias_relu': L.ACT_RELU, 'gelu': L.ACT_GELU, 'full': L.ACT_GELU}.get(epilogue, L.ACT_NONE)
    p[0]['dact'] = {'drelu': L.DACT_RELU, 'dgelu': L.DACT_GELU}.get(epilogue, L.DACT_NONE)
    if ln:
        p[0]['ln_kind'], p[0]['ln_eps'] = ln, 1e-5
        if ln == 1:
            p['ln_p']['buf'][0, :5] = (9, 10, 13, 14, 15)
        else:
            p['ln_p']['buf'][0, :6] = (9, 11, 13, 14, 12 if ln_res else -1, 15)
    ops = np.zeros(2, dtype=L.OP_DT)
    ops['r']['buf'][:] = -1
    ops[0]['kind'] = L.OP_CAST16
    ops[0]['r']['buf'][:3] = (7, 1, 8)
    ops[0]['i'][:2] = (1, ((W.shape[0] + 63) // 64) * ((W.shape[1] + 63) // 64))
    ops[1]['kind'] = L.OP_GEMM
    ops[1]['i'][:3] = (0, ksplit, tile)
    ctx.run(ops, p, ptrs, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dC.cpu().numpy().astype(np.float64).sum(0) if ksplit > 1 else dC[0].cpu().numpy()
    out = [(got, v)]
    if aux_expected is not None:
        out.append((daux_out.cpu().numpy(), aux_expected))
    if ln:
        out.append((dlnout.cpu().numpy(), A_eff))
        if ln == 1:
            out += [(dmean.cpu().numpy(), mu[:, 0]), (drstd.cpu().numpy(), rstd[:, 0])]
    return out


X3_CASES = [
    dict(M=256, N=1152, K=384, tile=40),                                        # to_qkv forward (ghn3xlm16)
    dict(M=256, N=384, K=384, tile=42, ksplit=2, epilogue='bias_res'),          # to_out forward, 2 K planes (linear epilogue)
    dict(M=256, N=1536, K=384, tile=40, epilogue='gelu'),                       # ff.net.0 forward
    dict(M=256, N=384, K=1536, tile=42, ksplit=8),                              # ff.net.3 forward, 8 planes of 192
    dict(M=256, N=384, K=1536, tile=40, ksplit=4, epilogue='bias_res'),
    dict(M=256, N=1536, K=384, tile=40, transposed=True, epilogue='dgelu'),     # ff.net.3 dgrad
    dict(M=256, N=384, K=1152, tile=42, transposed=True, ksplit=6),             # to_qkv dgrad
    dict(M=256, N=384, K=384, tile=42, transposed=True),                        # to_out dgrad on 32 x 32 tiles
    dict(M=200, N=256, K=1024, tile=41, slice_=256),                            # 64 x 64 tiles, four slices in the workgroup
    dict(M=70, N=64, K=64, tile=40, epilogue='bias_relu'),                      # ghn3tm8 width, ragged M
    dict(M=48, N=192, K=64, tile=42, epilogue='drelu'),
    dict(M=1421, N=3072, K=1536, tile=41, slice_=256, epilogue='bias_relu'),    # decoder.conv.0 forward shape
    dict(M=33, N=128, K=512, tile=40, slice_=128, epilogue='full'),             # walks 4 slices of 128
    # staged kernels (gemm_x3d.hip: fragment-major weights, no partial planes).  45 = 16 x 32 tiles, K split over the waves
    dict(M=256, N=384, K=384, tile=45, epilogue='bias_res'),                    # to_out forward: 4 parts of 96
    dict(M=256, N=384, K=1536, tile=45, epilogue='bias_res'),                   # ff.net.3 forward: 8 parts of 192
    dict(M=256, N=384, K=1536, tile=45, transposed=True),                       # ff.net.0 dgrad
    dict(M=256, N=384, K=1152, tile=45, transposed=True),                       # to_qkv dgrad: 6 parts of 192
    dict(M=200, N=256, K=1024, tile=45, epilogue='full'),                       # ghn3lm8 widths, ragged M: 8 parts of 128
    dict(M=200, N=256, K=768, tile=45, transposed=True),                        # 8 parts of 96
    dict(M=70, N=64, K=64, tile=45, epilogue='bias_relu'),                      # ghn3tm8: 2 parts of 32
    dict(M=48, N=64, K=192, tile=45, epilogue='drelu'),                         # 2 parts of 96
    dict(M=33, N=112, K=256, tile=45, epilogue='dgelu'),                        # 4 parts of 64, N = 7 x 16 (half a column tile)
    dict(M=512, N=128, K=512, tile=45, epilogue='gelu'),                        # ghn3sm8, two graphs: 8 parts of 64
    # 44 = 32 x 48 tiles, two K halves (wide outputs, K = C)
    dict(M=256, N=1536, K=384, tile=44, transposed=True, epilogue='dgelu'),     # ff.net.3 dgrad of the top layer (no prologue)
    dict(M=100, N=192, K=64, tile=44, epilogue='gelu'),
    # LayerNorm row prologues (tile codes 44 = 32 x 32 tiles, 45 = 16 x 32 tiles): forward (ln=1) and backward (ln=2)
    dict(M=256, N=1152, K=384, tile=44, ln=1),                                  # LN1 -> to_qkv
    dict(M=256, N=1536, K=384, tile=44, ln=1, epilogue='gelu'),                 # LN2 -> ff.net.0
    dict(M=256, N=1536, K=384, tile=44, ln=2, transposed=True, epilogue='dgelu'),   # LN1' -> ff.net.3 dgrad of the layer below
    dict(M=256, N=384, K=384, tile=45, ln=2, transposed=True),                  # LN2' -> to_out dgrad
    dict(M=256, N=384, K=384, tile=45, ln=2, transposed=True, ln_res=False),
    dict(M=200, N=768, K=256, tile=44, ln=1),                                   # ghn3lm8
    dict(M=200, N=256, K=256, tile=45, ln=2, transposed=True),
    dict(M=70, N=192, K=64, tile=44, ln=1, epilogue='bias_relu'),               # ghn3tm8, ragged M
    dict(M=70, N=64, K=64, tile=45, ln=2, transposed=True),
    dict(M=45, N=512, K=128, tile=44, ln=2, transposed=True, epilogue='dgelu'),  # ghn3sm8
    dict(M=45, N=128, K=128, tile=45, ln=1),
    dict(M=33, N=96, K=192, tile=45, ln=2, transposed=True),
    # f16 pieces (GHN3_GEMM_X3F16, round 5): the forward linears, fp32-grade products
    dict(M=256, N=1152, K=384, tile=44, ln=1, f16=True),                        # LN1 -> to_qkv
    dict(M=256, N=1536, K=384, tile=44, ln=1, epilogue='gelu', f16=True),       # LN2 -> ff.net.0
    dict(M=256, N=384, K=384, tile=45, epilogue='bias_res', f16=True),          # to_out
    dict(M=256, N=384, K=1536, tile=45, epilogue='bias_res', f16=True),         # ff.net.3
    dict(M=70, N=192, K=64, tile=44, ln=1, epilogue='bias_relu', f16=True),
    dict(M=33, N=96, K=192, tile=44, ln=1),
]
