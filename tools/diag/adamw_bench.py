"""Micro-benchmark (GPU box): GHN3_OP_SUMSQ + GHN3_OP_ADAMW over ghn3xlm16's 654 M parameters, 16-byte path against the
scalar path (selected by misaligning the buffers by one float)."""
import os, sys
import numpy as np
import torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from ghn3_amd import _lib as L
from ghn3_amd.optim import _dbits

n = 654365312
dev = 'cuda'
bufs_t = [torch.randn(n + 4, device=dev) * s for s in (1.0, 1e-3, 0.0, 0.0)]
scal = torch.zeros(64, device=dev)
parts = torch.zeros(1 << 16, device=dev)
ctx = L.context(0)
st = torch.cuda.current_stream().cuda_stream


def run(shift, with_sumsq=True, reps=10):
    ops = np.zeros(3, dtype=L.OP_DT)
    ops['r']['buf'][:] = -1
    ops[0]['kind'] = L.OP_MEMSET0
    ops[0]['r']['buf'][0] = 4
    ops[0]['i'][0] = 4
    ops[1]['kind'] = L.OP_SUMSQ if with_sumsq else L.OP_NOP
    ops[1]['r']['buf'][:3] = (4, 1, 5)
    ops[1]['i'][0] = n
    ops[2]['kind'] = L.OP_ADAMW
    ops[2]['r']['buf'][:5] = (0, 1, 2, 3, 4 if with_sumsq else -1)
    ops[2]['i'][0] = n
    for k, h in enumerate((1e-6, 0.9, 0.999, 1e-8, 1e-2, 1.0 - 0.9 ** 3, 1.0 - 0.999 ** 3)):
        ops[2]['i'][1 + k] = _dbits(h)
    ops[2]['f'][0] = 5.0 if with_sumsq else 0.0
    ops[2]['f'][1] = 1.0
    bufs = np.asarray([t.data_ptr() + 4 * shift for t in bufs_t] + [scal.data_ptr(), parts.data_ptr()], dtype=np.uint64)
    none = np.zeros(0, dtype=L.PROBLEM_DT)
    for _ in range(2):
        ctx.run(ops, none, bufs, st)
    a, b = L.Event(), L.Event()
    a.record(st)
    for _ in range(reps):
        ctx.run(ops, none, bufs, st)
    b.record(st)
    torch.cuda.synchronize()
    return a.elapsed_ms(b) / reps


for name, shift in (('16-byte path', 0), ('scalar path (misaligned by one float)', 1), ('16-byte path', 0), ('scalar path', 1)):
    t = run(shift) if shift == 0 else float('nan')       # (GHN3_OP_SUMSQ needs 16-byte alignment)
    t2 = run(shift, with_sumsq=False)
    print('%-40s sumsq + adamw %.3f ms (%.2f TB/s over 32 B/param)   adamw alone %.3f ms (%.2f TB/s over 28 B/param)' % (
        name, t, 32 * n / t / 1e9, t2, 28 * n / t2 / 1e9))
