#!/usr/bin/env python3
"""Latency of the Graphormer GEMM shapes: split-bf16 kernel (gemm_x3.hip) against the exact-fp32 small-problem kernel
(gemm_small.hip), each launch preceded by a dependent predecessor as on the real chain (GPU box only).
Weights rotate over 24 copies (one per layer) so that they come from HBM / L2 like in the model."""
import os
import sys
import numpy as np
import torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from ghn3_amd import _lib as L   # noqa: E402


def bench(ctx, M, N, K, kind, tile=40, ksplit=1, slice_=0, transposed=False, reps=10, layers=24, name='', ln=0):
    dev = 'cuda'
    A = torch.randn(M, K, device=dev)
    Ws = torch.randn(layers, N * K, device=dev) * 0.05                   # (nn.Linear weights [N][K] or [K][N])
    C = torch.zeros(max(ksplit, 1), M, N, device=dev)
    n = N * K
    shadow = torch.zeros(layers, 4 * n, dtype=torch.int16, device=dev)
    bufs = [A, shadow, C, Ws]
    if ln:      # gamma, beta / x, res, mean, rstd, by-product
        bufs += [torch.ones(K, device=dev), torch.zeros(K, device=dev), torch.randn(M, K, device=dev), torch.randn(M, K, device=dev),
                 torch.zeros(M, device=dev), torch.ones(M, device=dev), torch.zeros(M, K, device=dev)]
    ptrs = np.asarray([b.data_ptr() for b in bufs], dtype=np.uint64)
    probs, ops = [], []
    R, Cc = (K, N) if transposed else (N, K)
    if kind == 'x3':
        # shadows: [hi straight | hi transposed | lo straight | lo transposed] per layer
        desc = np.zeros(layers, dtype=L.CAST_DT)
        for l in range(layers):
            d = desc[l]
            d['src_off'], d['rows'], d['cols'], d['ld_src'] = l * n, R, Cc, Cc
            d['dst_off'], d['ld_dst'], d['dstT_off'], d['ld_dstT'], d['lo_off'] = l * 4 * n, Cc, l * 4 * n + n, R, 2 * n
            d['flags'] = L.CAST_STRAIGHT | L.CAST_TRANSPOSED | L.CAST_SPLIT | (L.CAST_FRAG if tile in (44, 45) else 0)
            d['block_start'] = l * ((R + 63) // 64) * ((Cc + 63) // 64)
        ddesc = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        op = np.zeros(1, dtype=L.OP_DT)
        op['r']['buf'][:] = -1
        op['kind'] = L.OP_CAST16
        op['r']['buf'][0][:3] = (3, 1, len(bufs))
        op['i'][0][:2] = (layers, layers * ((R + 63) // 64) * ((Cc + 63) // 64))
        ctx.run(op, np.zeros(0, dtype=L.PROBLEM_DT), np.append(ptrs, np.uint64(ddesc.data_ptr())),
                torch.cuda.current_stream().cuda_stream)
    kc = K // ksplit
    sl = slice_ or 64 * max(d for d in (6, 4, 3, 2, 1) if (kc // 64) % d == 0)
    p = np.zeros(layers * ksplit, dtype=L.PROBLEM_DT)
    for nm in L._REF_NAMES + ('lim', 'alpha_amax', 'B2', 'mtiles'):
        p[nm]['buf'] = -1
    p['ln_p']['buf'] = -1
    for l in range(layers):
        for j in range(ksplit):
            q = p[l * ksplit + j]
            q['A']['buf'], q['A']['off'] = 0, 4 * j * kc
            q['C']['buf'], q['C']['off'] = 2, 4 * j * M * N
            q['M'], q['N'], q['K'], q['lda'], q['ldc'], q['alpha'] = M, N, kc, K, N, 1.0
            if kind == 'x3':
                hi = l * 4 * n + (n if transposed else 0)
                q['B']['buf'], q['B']['off'] = 1, 2 * (hi + j * kc)
                q['B2']['buf'], q['B2']['off'] = 1, 2 * (hi + 2 * n + j * kc)
                q['ldb'], q['flags'], q['x3_slice'] = K, L.GEMM_X3, sl
                if ln:
                    q['ln_kind'], q['ln_eps'] = ln, 1e-5
                    q['ln_p']['buf'][:6] = (4, 5, 8, 9, 10, -1) if ln == 1 else (4, 6, 8, 9, 7, 10)
            else:
                q['B']['buf'] = 3
                if transposed:
                    q['B']['off'], q['ldb'], q['b_mode'] = 4 * (l * n + j * kc * N), N, L.MODE_COL
                else:
                    q['B']['off'], q['ldb'], q['b_mode'] = 4 * (l * n + j * kc), K, L.MODE_ROW
    ops = np.zeros(layers, dtype=L.OP_DT)
    ops['r']['buf'][:] = -1
    ops['kind'] = L.OP_GEMM
    for l in range(layers):
        ops[l]['i'][:3] = (l * ksplit, ksplit, tile if kind == 'x3' else 32)
    stream = torch.cuda.current_stream().cuda_stream
    pp = ptrs
    for _ in range(2):
        ctx.run(ops, p, pp, stream)
    torch.cuda.synchronize()
    e0, e1 = L.Event(), L.Event()
    e0.record(stream)
    for _ in range(reps):
        ctx.run(ops, p, pp, stream)
    e1.record(stream)
    us = 1e3 * e0.elapsed_ms(e1) / (reps * layers)
    print('%-22s %-5s M=%4d N=%4d K=%4d tile=%d ksplit=%d slice=%3d ln=%d : %6.2f us per launch (back to back, 24 weights)'
          % (name, kind, M, N, K, tile if kind == 'x3' else 32, ksplit, sl, ln, us), flush=True)
    return us


def bench_ln(ctx, M, C, bwd=False, reps=10, layers=24):
    """the standalone LayerNorm launches the prologue kernels replace (no addend planes)"""
    dev = 'cuda'
    x, y = torch.randn(M, C, device=dev), torch.zeros(M, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mean, rstd, res = torch.zeros(M, device=dev), torch.ones(M, device=dev), torch.randn(M, C, device=dev)
    bufs = [y, x, g, b, mean, rstd, res]
    ptrs = np.asarray([t.data_ptr() for t in bufs], dtype=np.uint64)
    ops = np.zeros(layers, dtype=L.OP_DT)
    ops['r']['buf'][:] = -1
    for l in range(layers):
        if bwd:
            ops[l]['kind'] = L.OP_LAYERNORM_BWD
            ops[l]['r']['buf'][:7] = (0, 6, 1, 2, 4, 5, 6)
        else:
            ops[l]['kind'] = L.OP_LAYERNORM_FWD
            ops[l]['r']['buf'][:6] = (0, 1, 2, 3, 4, 5)
            ops[l]['f'][0] = 1e-5
        ops[l]['i'][:2] = (M, C)
    stream = torch.cuda.current_stream().cuda_stream
    pr = np.zeros(0, dtype=L.PROBLEM_DT)
    ctx.run(ops, pr, ptrs, stream)
    torch.cuda.synchronize()
    e0, e1 = L.Event(), L.Event()
    e0.record(stream)
    for _ in range(reps):
        ctx.run(ops, pr, ptrs, stream)
    e1.record(stream)
    print('%-22s M=%4d C=%4d : %6.2f us per launch' % ('layernorm_bwd' if bwd else 'layernorm_fwd', M, C,
                                                       1e3 * e0.elapsed_ms(e1) / (reps * layers)), flush=True)


if __name__ == '__main__':
    ctx = L.context(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'new':
        # round 4: direct (43) and LayerNorm-prologue (44 / 45) kernels against what they replace
        for rows in (256, 512):
            C = 384
            print('--- rows %d, C %d' % (rows, C))
            bench_ln(ctx, rows, C)
            bench_ln(ctx, rows, C, bwd=True)
            bench(ctx, rows, 3 * C, C, 'x3', tile=42, slice_=192, name='to_qkv fwd (old)')
            bench(ctx, rows, 3 * C, C, 'x3', tile=44, name='to_qkv staged')
            bench(ctx, rows, 3 * C, C, 'x3', tile=44, ln=1, name='LN1 + to_qkv')
            bench(ctx, rows, 4 * C, C, 'x3', tile=42, slice_=192, name='ff0 fwd (old)')
            bench(ctx, rows, 4 * C, C, 'x3', tile=44, ln=1, name='LN2 + ff0')
            bench(ctx, rows, C, C, 'x3', tile=42, ksplit=2, name='to_out fwd (old)')
            bench(ctx, rows, C, C, 'x3', tile=45, name='to_out fwd staged')
            bench(ctx, rows, C, 4 * C, 'x3', tile=42, ksplit=8, name='ff3 fwd (old)')
            bench(ctx, rows, C, 4 * C, 'x3', tile=45, name='ff3 fwd staged')
            bench(ctx, rows, 4 * C, C, 'x3', tile=42, slice_=192, transposed=True, name='ff3 dgrad (old)')
            bench(ctx, rows, 4 * C, C, 'x3', tile=44, transposed=True, name='ff3 dgrad staged')
            bench(ctx, rows, 4 * C, C, 'x3', tile=44, ln=2, transposed=True, name="LN1' + ff3 dgrad")
            bench(ctx, rows, C, 4 * C, 'x3', tile=42, ksplit=8, transposed=True, name='ff0 dgrad (old)')
            bench(ctx, rows, C, 4 * C, 'x3', tile=45, transposed=True, name='ff0 dgrad staged')
            bench(ctx, rows, C, C, 'x3', tile=42, transposed=True, name='to_out dgrad (old)')
            bench(ctx, rows, C, C, 'x3', tile=45, ln=2, transposed=True, name="LN2' + to_out dgrad")
            bench(ctx, rows, C, C, 'x3', tile=44, ln=2, transposed=True, name="LN2' + to_out dgrad")
            bench(ctx, rows, C, 3 * C, 'x3', tile=42, ksplit=6, transposed=True, name='to_qkv dgrad (old)')
            bench(ctx, rows, C, 3 * C, 'x3', tile=45, transposed=True, name='to_qkv dgrad staged')
        sys.exit(0)
    for C in (384, 256):
        for (nm, N, K, tr, splits) in (('to_qkv fwd', 3 * C, C, False, (1,)), ('to_out fwd', C, C, False, (1, 2, 3)),
                                       ('ff0 fwd', 4 * C, C, False, (1,)), ('ff3 fwd', C, 4 * C, False, (4, 8) if C == 384 else (4,)),
                                       ('ff3 dgrad', 4 * C, C, True, (1,)), ('ff0 dgrad', C, 4 * C, True, (4,)),
                                       ('to_out dgrad', C, C, True, (1,)), ('to_qkv dgrad', C, 3 * C, True, (3,))):
            bench(ctx, 256, N, K, 'f32', ksplit=1, transposed=tr, name=nm)
            if K >= 768 and C == 384:
                bench(ctx, 256, N, K, 'f32', ksplit=2, transposed=tr, name=nm)
            for ks in splits:
                for tile in (40, 41, 42):
                    kc = K // ks
                    if kc % 64:
                        continue
                    nkt = max(d for d in (6, 4, 3, 2, 1) if (kc // 64) % d == 0)
                    if kc // 64 != nkt or (tile == 41 and nkt > 4):
                        continue
                    bench(ctx, 256, N, K, 'x3', tile=tile, ksplit=ks, transposed=tr, name=nm)
