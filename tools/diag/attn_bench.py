#!/usr/bin/env python3
"""Latency of the attention kernels at the bench shape (one 256-node graph, 16 heads of 24, ghn3xlm16): `layers` launches
back to back on one stream, each on its own qkv / P / out buffers (as in the model: fresh data from the previous kernel),
one shared edge bias.  GPU box only.   python tools/diag/attn_bench.py [N] [H] [C]"""
import os
import sys
import numpy as np
import torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from ghn3_amd import _lib as L   # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 384
    B, layers = 1, 24
    dev = 'cuda'
    ctx = L.context(0)
    stream = torch.cuda.current_stream().cuda_stream
    qkv = torch.randn(layers, B * N, 3 * C, device=dev)
    bias = torch.randn(B, H, N, N, device=dev)
    P = torch.zeros(layers, B, H, N, N, device=dev)
    out = torch.zeros(layers, B * N, C, device=dev)
    dO = torch.randn(layers, B * N, C, device=dev)
    dqkv = torch.zeros(layers, B * N, 3 * C, device=dev)
    dBias = torch.zeros(B, H, N, N, device=dev)
    nn_ = torch.full((B,), N, dtype=torch.int32, device=dev)
    bufs = [qkv, bias, P, out, dO, dqkv, dBias, nn_]
    ptrs = np.asarray([b.data_ptr() for b in bufs], dtype=np.uint64)
    probs = np.zeros(0, dtype=L.PROBLEM_DT)

    def program(kind, save_p=True, with_bias=True):
        ops = np.zeros(layers, dtype=L.OP_DT)
        ops['r']['buf'][:] = -1
        for l in range(layers):
            o = ops[l]
            o['kind'] = kind
            o['i'][:4] = (B, N, C, H)
            if kind == L.OP_ATTN_FWD:
                refs = [(3, 4 * l * B * N * C), (0, 4 * l * B * N * 3 * C), (1, 0) if with_bias else (-1, 0),
                        (2, 4 * l * B * H * N * N) if save_p else (-1, 0), (7, 0)]
            else:
                refs = [(5, 4 * l * B * N * 3 * C), (4, 4 * l * B * N * C), (0, 4 * l * B * N * 3 * C),
                        (2, 4 * l * B * H * N * N), (3, 4 * l * B * N * C), (-1, 0), (6, 0), (7, 0)]
            for j, (b, off) in enumerate(refs):
                o['r']['buf'][j], o['r']['off'][j] = b, off
        return ops

    def time_it(ops, name, reps=20):
        for _ in range(3):
            ctx.run(ops, probs, ptrs, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.run(ops, probs, ptrs, stream)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * layers)
        print('%-44s N=%d H=%d C=%d  %7.2f us per launch' % (name, N, H, C, us), flush=True)

    time_it(program(L.OP_ATTN_FWD), 'attn fwd (bias, P saved)')
    time_it(program(L.OP_ATTN_FWD, save_p=False), 'attn fwd (bias, no P)')
    time_it(program(L.OP_ATTN_FWD, save_p=False, with_bias=False), 'attn fwd (no bias, no P)')
    time_it(program(L.OP_ATTN_BWD), 'attn bwd')


if __name__ == '__main__':
    main()
