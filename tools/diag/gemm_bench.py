#!/usr/bin/env python3
"""Micro-benchmark of the grouped GEMM kernels on shapes of the decoder W2 GEMMs (GPU box only).
    python tools/diag/gemm_bench.py [f32|f16]
"""
import os
import sys
import time

import numpy as np
import torch

import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from ghn3_amd import _lib as L   # noqa: E402


def bench(ctx, M, N, K, a_mode, b_mode, tile=0, ctype=L.CT_F32, ksplit=1, accum=False, reps=5, qs=None, name=''):
    dev = 'cuda'
    a_rows, a_cols = (M, K) if a_mode == L.MODE_ROW else (K, M)
    b_rows, b_cols = (N, K) if b_mode == L.MODE_ROW else (K, N)
    lda, ldb, ldc = a_cols + 64, b_cols + 64, N + 64
    A = torch.randn(a_rows, lda, device=dev)
    B = torch.randn(b_rows, ldb, device=dev)
    C = torch.zeros(M, ldc, device=dev)
    bufs = np.asarray([A.data_ptr(), B.data_ptr(), C.data_ptr()], dtype=np.uint64)
    p = np.zeros(1, dtype=L.PROBLEM_DT)
    for nme in ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather', 'lim', 'alpha_amax',
                'B2', 'mtiles'):
        p[nme]['buf'] = -1
    p['ln_p']['buf'] = -1
    p['A']['buf'], p['B']['buf'], p['C']['buf'] = 0, 1, 2
    p['M'], p['N'], p['K'], p['lda'], p['ldb'], p['ldc'] = M, N, K, lda, ldb, ldc
    p['a_mode'], p['b_mode'] = a_mode, b_mode
    p['alpha'] = 1.0
    p['ksplit'] = ksplit
    p['flags'] = L.GEMM_ACCUM if accum else 0
    if qs:
        p['b_q'], p['b_s'] = qs
    op = np.zeros(1, dtype=L.OP_DT)
    op['kind'] = L.OP_GEMM
    op['flags'] = 1 + ctype
    op['i'][0][:3] = (0, 1, tile)
    op['r']['buf'][:] = -1
    stream = torch.cuda.current_stream().cuda_stream
    ctx.run(op, p, bufs, stream)
    torch.cuda.synchronize()
    e0, e1 = L.Event(), L.Event()
    e0.record(stream)
    for _ in range(reps):
        ctx.run(op, p, bufs, stream)
    e1.record(stream)
    ms = e0.elapsed_ms(e1) / reps
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print('%-28s M=%6d N=%6d K=%6d modes=%d%d tile=%3d ksplit=%2d acc=%d  %8.3f ms  %7.1f TF' %
          (name, M, N, K, a_mode, b_mode, tile, ksplit, accum, ms, tf), flush=True)
    return ms


if __name__ == '__main__':
    ct = {'f32': L.CT_F32, 'f16': L.CT_F16, 'bf16': L.CT_BF16}[sys.argv[1] if len(sys.argv) > 1 else 'f32']
    ctx = L.context(0)
    R, Cc = L.MODE_ROW, L.MODE_COL
    # square-ish references
    bench(ctx, 4096, 4096, 4096, R, R, 128, ct, name='square RR')
    bench(ctx, 4096, 4096, 4096, R, Cc, 128, ct, name='square RC')
    bench(ctx, 4096, 4096, 4096, Cc, Cc, 128, ct, name='square CC')
    bench(ctx, 4096, 4096, 4096, Cc, R, 128, ct, name='square CR')
    # decoder W2 shapes (ghn3xlm16, 512 rows)
    bench(ctx, 512, 147456, 3072, R, R, 128, ct, name='w2 fwd')
    bench(ctx, 512, 147456, 3072, R, R, 64, ct, name='w2 fwd t64')
    for ks in (1, 4, 9, 16):
        bench(ctx, 512, 3072, 147456, R, Cc, 128, ct, ksplit=ks, name='w2 dgrad')
    bench(ctx, 512, 3072, 147456, R, Cc, 64, ct, ksplit=9, name='w2 dgrad t64')
    bench(ctx, 147456, 3072, 512, Cc, Cc, 128, ct, name='w2 wgrad')
    bench(ctx, 147456, 3072, 512, Cc, Cc, 128, ct, accum=True, name='w2 wgrad accum')
    bench(ctx, 147456, 3072, 512, Cc, Cc, 64, ct, name='w2 wgrad t64')
    bench(ctx, 147456, 3072, 512, R, R, 128, ct, name='w2 wgrad as RR (transposed ops)')
    bench(ctx, 147456, 3072, 2048, Cc, Cc, 128, ct, name='w2 wgrad K=2048')
    # transformer-sized
    bench(ctx, 256, 1152, 384, R, R, 0, ct, name='qkv (auto)')
    bench(ctx, 256, 1152, 384, R, R, 64, ct, name='qkv t64')
    bench(ctx, 256, 384, 1536, R, Cc, 0, ct, name='ffn dgrad (auto)')
    bench(ctx, 1536, 384, 256, Cc, Cc, 0, ct, name='ffn wgrad (auto)')


def bench16(ctx, M, N, K, ctype, ksplit=1, accum=False, kmap=None, reps=5, name='', tile=0, a_qs=None, c_qs=None):
    """GHN3_GEMM_OP16: 16-bit operands resident in HBM (random bit patterns of small magnitude)."""
    dev = 'cuda'
    kq, ks = kmap if kmap else (0, 0)
    Kp = ((K + kq - 1) // kq) * ks if kmap else K
    lda, ldb, ldc = (K + 63) // 64 * 64, (Kp + 63) // 64 * 64 + 64, N
    mk = (lambda r, c: torch.randn(r, c, device=dev).to(torch.float16 if ctype == L.CT_F16 else torch.bfloat16))
    A, B = mk(M, lda), mk(N, ldb)
    C = torch.zeros(M, ldc, device=dev)
    bufs = np.asarray([A.data_ptr(), B.data_ptr(), C.data_ptr()], dtype=np.uint64)
    p = np.zeros(1, dtype=L.PROBLEM_DT)
    for nme in ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather', 'lim', 'alpha_amax',
                'B2', 'mtiles'):
        p[nme]['buf'] = -1
    p['ln_p']['buf'] = -1
    p['A']['buf'], p['B']['buf'], p['C']['buf'] = 0, 1, 2
    p['M'], p['N'], p['K'], p['lda'], p['ldb'], p['ldc'] = M, N, K, lda, ldb, ldc
    p['alpha'] = 1.0
    p['ksplit'] = ksplit
    p['b_kq'], p['b_ks'] = kq, ks
    if a_qs:
        p['a_q'], p['a_s'] = a_qs
    if c_qs:
        p['c_q'], p['c_s'] = c_qs
    p['flags'] = L.GEMM_OP16 | (L.GEMM_ACCUM if accum else 0)
    op = np.zeros(1, dtype=L.OP_DT)
    op['kind'] = L.OP_GEMM
    op['flags'] = 1 + ctype
    op['i'][0][:3] = (0, 1, tile)
    op['r']['buf'][:] = -1
    stream = torch.cuda.current_stream().cuda_stream
    ctx.run(op, p, bufs, stream)
    torch.cuda.synchronize()
    e0, e1 = L.Event(), L.Event()
    e0.record(stream)
    for _ in range(reps):
        ctx.run(op, p, bufs, stream)
    e1.record(stream)
    ms = e0.elapsed_ms(e1) / reps
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print('%-28s M=%6d N=%6d K=%6d OP16 tile=%2d ksplit=%2d acc=%d  %8.3f ms  %7.1f TF' %
          (name, M, N, K, tile, ksplit, accum, ms, tf), flush=True)
    return ms


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'wgrad':
    ct = {'f16': L.CT_F16, 'bf16': L.CT_BF16}[sys.argv[1]]
    ctx = L.context(0)
    for tile in (16, 20, 24, 25, 28, 29):
        bench16(ctx, 147456, 3072, 768, ct, name='wgrad K=768', tile=tile)
        bench16(ctx, 65536, 3072, 576, ct, name='wgrad band 65536 x 576', tile=tile)
        bench16(ctx, 147456, 3072, 512, ct, name='wgrad', tile=tile)
        if tile in (25, 28, 29):
            bench16(ctx, 147456, 3072, 2048, ct, name='wgrad K=2048', tile=tile)
            continue
        bench16(ctx, 147456, 3072, 512, ct, name='wgrad A rows -> 256 (L2)', tile=tile, a_qs=(256, 0))
        bench16(ctx, 147456, 3072, 512, ct, name='wgrad C rows -> 256 (L2)', tile=tile, c_qs=(256, 0))
        bench16(ctx, 147456, 3072, 512, ct, name='wgrad A,C rows -> 256', tile=tile, a_qs=(256, 0), c_qs=(256, 0))
        bench16(ctx, 147456, 3072, 2048, ct, name='wgrad K=2048', tile=tile)
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] in ('f16', 'bf16'):
    ct = {'f16': L.CT_F16, 'bf16': L.CT_BF16}[sys.argv[1]]
    ctx = L.context(0)
    for tile in (16, 20, 24):
        bench16(ctx, 4096, 4096, 4096, ct, name='d16 square', tile=tile)
        bench16(ctx, 8192, 8192, 8192, ct, name='d16 square 8k', tile=tile)
        bench16(ctx, 512, 147456, 3072, ct, name='d16 w2 fwd', tile=tile)
        for ks in (4, 9, 16, 32):
            bench16(ctx, 512, 3072, 147456, ct, ksplit=ks, name='d16 w2 dgrad', tile=tile)
        bench16(ctx, 512, 3072, 147456, ct, ksplit=16, kmap=(128, 384), name='d16 w2 dgrad kmap', tile=tile)
        bench16(ctx, 147456, 3072, 512, ct, name='d16 w2 wgrad', tile=tile)
        bench16(ctx, 147456, 3072, 512, ct, accum=True, name='d16 w2 wgrad accum', tile=tile)
