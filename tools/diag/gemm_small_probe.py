#!/usr/bin/env python3
"""Latency probe of the small-problem GEMM kernels (GPU box only)."""
import os
import sys
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from gemm_bench import bench, L   # noqa: E402

if __name__ == '__main__':
    ctx = L.context(0)
    R, Cc = L.MODE_ROW, L.MODE_COL
    for tile in (32, 64):
        for (M, N, K) in ((32, 32, 32), (256, 1152, 32), (256, 1152, 384), (256, 384, 384), (256, 1536, 384),
                          (256, 384, 1536), (2048, 1152, 384)):
            bench(ctx, M, N, K, R, R, tile, L.CT_F32, reps=20, name='fwd RR')
        for (M, N, K) in ((256, 384, 1152), (256, 384, 1536), (256, 1536, 384)):
            bench(ctx, M, N, K, R, Cc, tile, L.CT_F32, reps=20, name='dgrad RC')
        for (M, N, K) in ((1152, 384, 256), (1536, 384, 256), (384, 1536, 256)):
            bench(ctx, M, N, K, Cc, Cc, tile, L.CT_F32, reps=20, name='wgrad CC')
