"""Diagnostic (GPU box): where the host time of a FRESH architecture per step goes -- device half of the compile
(GHN3.plan) and the enqueue of the step -- single-threaded under cProfile."""
import cProfile
import os
import pstats
import sys
import time
import torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from bench import model_cfg, _loader_worker
from ghn3_amd import GHN3, _lib as L

torch.manual_seed(0)
ghn = GHN3(**model_cfg('ghn3xlm16'), compute='f16').to('cuda')
ghn.train()
pcfg = ghn.program_config()
items = [_loader_worker((256, 1, 256000 + 7919 * (k + 1), pcfg)) for k in range(24)]
ctx = L.context(0)
stream = torch.cuda.current_stream().cuda_stream
dout = None


def one(item, t):
    global dout
    gb, nets, prog = item
    t0 = time.perf_counter()
    plan = ghn.plan(prog, gb, nets)
    t1 = time.perf_counter()
    f_norm, b_norm = prog.norm_ops(1.0)
    if dout is None or dout.numel() < prog.out_numel:
        dout = torch.empty(prog.out_numel, dtype=torch.float32, device='cuda')
    ghn._run_forward(plan)
    ghn._fill_bufs(plan, out=plan.out, dout=dout)
    ctx.run(f_norm, prog.problems, plan.bufs, stream)
    ctx.run(b_norm, prog.problems, plan.bufs, stream)
    ghn._run_backward(plan, dout)
    t2 = time.perf_counter()
    t[0] += t1 - t0
    t[1] += t2 - t1


for it in items[:12]:
    one(it, [0, 0])
torch.cuda.synchronize()
t = [0.0, 0.0]
pr = cProfile.Profile()
pr.enable()
for it in items[12:]:
    one(it, t)
pr.disable()
torch.cuda.synchronize()
n = len(items) - 12
torch.cuda.synchronize()
print('per fresh step: plan() %.2f ms, enqueue %.2f ms (the enqueue time includes waiting for the GPU once the host is two '
      'steps ahead)' % (t[0] / n * 1e3, t[1] / n * 1e3))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
