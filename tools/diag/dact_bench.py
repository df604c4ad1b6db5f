"""GPU box: GHN3_OP_DACT with partial planes alone (the plane sum + dReLU behind the W2 dgrad of the bench workload:
1421 x 3072 floats, 15 planes), microseconds per launch and the rate over X (read + write), the mask and the planes."""
import sys
import numpy as np
import torch
import _paths  # noqa: F401
from ghn3_amd import _lib as L

M, N = 1421, 3072
ctx = L.context(0)
st = torch.cuda.current_stream().cuda_stream
for n_parts in (15, 7, 3, 0):
    X = torch.randn(M * N, device='cuda')
    aux = torch.randn(M * N, device='cuda')
    parts = torch.randn(max(n_parts, 1) * M * N, device='cuda')
    ops = np.zeros(1, dtype=L.OP_DT)
    ops['r']['buf'][:] = -1
    ops[0]['kind'] = L.OP_DACT
    ops[0]['r']['buf'][:2] = (0, 1)
    if n_parts:
        ops[0]['r']['buf'][3] = 2
    ops[0]['i'][:7] = (M, N, N, L.DACT_RELU, n_parts, M * N, M)
    bufs = np.asarray([X.data_ptr(), aux.data_ptr(), parts.data_ptr()], dtype=np.uint64)
    none = np.zeros(0, dtype=L.PROBLEM_DT)
    for _ in range(3):
        ctx.run(ops, none, bufs, st)
    a, b = L.Event(), L.Event()
    a.record(st)
    for _ in range(20):
        ctx.run(ops, none, bufs, st)
    b.record(st)
    torch.cuda.synchronize()
    us = 1e3 * a.elapsed_ms(b) / 20
    gb = 4.0 * M * N * (3 + n_parts) / 1e9
    print('%2d planes: %.1f us per launch, %.2f GB -> %.2f TB/s' % (n_parts, us, gb, gb / us * 1e-3 * 1e3 / 1e0 / 1e3 * 1e3))
