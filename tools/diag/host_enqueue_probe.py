"""Diagnostic (not a test): host time to ENQUEUE one step of the bench workload vs the GPU time of that step."""
import os, sys, time
import torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from bench import model_cfg
from ghn3_amd import GHN3, _lib as L
from ghn3_amd.synthetic import synthetic_batch

torch.manual_seed(0)
ghn = GHN3(**model_cfg('ghn3xlm16'), compute='f16').to('cuda')
ghn.train()
gb, nets = synthetic_batch([256], 256000)
plan = ghn.compile(nets, gb, training=True)
prog = plan.program
f_norm, b_norm = prog.norm_ops(1.0)
ctx = L.context(0)
stream = torch.cuda.current_stream().cuda_stream
dout = torch.empty(prog.out_numel, dtype=torch.float32, device='cuda')
print('ops fwd %d bwd %d problems %d' % (len(prog.fwd_ops), sum(len(p) for p in prog.bwd_parts), len(prog.problems)))


def step(t):
    t0 = time.perf_counter(); ghn._run_forward(plan)
    t1 = time.perf_counter(); ghn._fill_bufs(plan, out=plan.out, dout=dout)
    ctx.run(f_norm, prog.problems, plan.bufs, stream); ctx.run(b_norm, prog.problems, plan.bufs, stream)
    t2 = time.perf_counter(); ghn._run_backward(plan, dout)
    t3 = time.perf_counter()
    t[0] += t1 - t0; t[1] += t2 - t1; t[2] += t3 - t2


for _ in range(5):
    step([0, 0, 0])
torch.cuda.synchronize()
for rep in range(3):
    t = [0.0, 0.0, 0.0]
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    for _ in range(2):
        step(t)
    w1 = time.perf_counter()
    torch.cuda.synchronize()
    w2 = time.perf_counter()
    print('2 steps from idle: host enqueue %.2f ms/step (fwd %.2f norms %.2f bwd %.2f), until GPU done %.2f ms/step'
          % ((w1 - w0) * 500, t[0] * 500, t[1] * 500, t[2] * 500, (w2 - w0) * 500))
