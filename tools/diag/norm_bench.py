"""Micro-benchmark (GPU box): the loss passes over the flat predicted-parameter buffer of the bench workload
(GHN3_OP_PARAM_NORM_FWD / BWD) and the tile kernels, from the per-op profile of a replayed plan."""
import os, sys
import numpy as np
import torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from ghn3_amd import GHN3, _lib as L
from ghn3_amd.synthetic import synthetic_batch
import bench

hip = GHN3(**bench.model_cfg('ghn3xlm16'), compute='f16').to('cuda').train()
gb, nets = synthetic_batch([256], 256000)
plan = hip.compile(nets, gb, training=True)
prog = plan.program
hip._run_forward(plan)
dout = torch.empty(prog.out_numel, dtype=torch.float32, device='cuda')
hip._fill_bufs(plan, out=plan.out, dout=dout)
f_norm, b_norm = prog.norm_ops(1.0)
ctx = hip._ctx()
st = torch.cuda.current_stream().cuda_stream
for name, ops in (('param_norm_fwd (sq + sqrt + loss)', f_norm), ('param_norm_bwd', b_norm)):
    for _ in range(3):
        ctx.run(ops, prog.problems, plan.bufs, st)
    a, b = L.Event(), L.Event()
    a.record(st)
    for _ in range(20):
        ctx.run(ops, prog.problems, plan.bufs, st)
    b.record(st)
    torch.cuda.synchronize()
    ms = a.elapsed_ms(b) / 20
    nbytes = 4 * prog.out_numel * (1 if 'fwd' in name else 2)
    print('%-36s %.1f us  (%.2f TB/s over %d MB)' % (name, 1e3 * ms, nbytes / ms / 1e9, nbytes >> 20))
