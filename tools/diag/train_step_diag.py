"""GPU box: where the milliseconds of a TRAINING step go (bench.py's `train_step` extra): events behind the forward (incl.
the refresh of the 16-bit weight copies), behind the backward and behind the fused clip + AdamW, in the pipelined loop."""
import os, sys, time
import numpy as np
import torch
import _paths  # noqa: F401
import bench
from ghn3_amd import GHN3, _lib as L
from ghn3_amd.optim import FusedAdamW
from ghn3_amd.synthetic import synthetic_batch

dev = 'cuda'
torch.manual_seed(0)
ghn = GHN3(**bench.model_cfg('ghn3xlm16'), compute='f16').to(dev)
ghn.train()
gb, nets = synthetic_batch([256], 256000)
plan = ghn.compile(nets, gb, training=True)
prog = plan.program
ctx = L.context(0)
stream = torch.cuda.current_stream().cuda_stream
fin = prog.norm_fin_ops()
one = torch.ones(1, dtype=torch.float32, device=dev)
opt = FusedAdamW(ghn, lr=1e-6, max_grad_norm=5.0)
fused = os.environ.get('DIAG_FUSED', '1') != '0'       # FusedAdamW.step with the plan: the W2 update writes its 16-bit copies


def fwd():
    ghn._run_forward(plan)
    ctx.run(fin, prog.problems, plan.bufs, stream)


def bwd():
    ghn._run_backward(plan, None, norm_g=one)


def upd():
    opt.step(plan.gflat, plan=plan if fused else None, local_grads=fused)


for _ in range(3):
    fwd(); bwd(); upd()
torch.cuda.synchronize()
n = 20
ev = [[L.Event() for _ in range(4)] for _ in range(n)]
t0 = time.perf_counter()
for k in range(n):
    ev[k][0].record(stream); fwd()
    ev[k][1].record(stream); bwd()
    ev[k][2].record(stream); upd()
    ev[k][3].record(stream)
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t0) / n
f = np.mean([e[0].elapsed_ms(e[1]) for e in ev[2:]])
b = np.mean([e[1].elapsed_ms(e[2]) for e in ev[2:]])
u = np.mean([e[2].elapsed_ms(e[3]) for e in ev[2:]])
print('train step %.3f ms wall: forward (+ shadow refresh) %.3f, backward %.3f, clip + AdamW %.3f (sum %.3f)' % (wall, f, b, u, f + b + u))
if os.environ.get('DIAG_ONLY_TRAIN', '0') != '0':
    sys.exit(0)
# the same without the update (weights unchanged -> no refresh)
for _ in range(2):
    fwd(); bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(n):
    ev[k][0].record(stream); fwd()
    ev[k][1].record(stream); bwd()
    ev[k][2].record(stream)
torch.cuda.synchronize()
wall = 1e3 * (time.perf_counter() - t0) / n
f = np.mean([e[0].elapsed_ms(e[1]) for e in ev[2:]])
b = np.mean([e[1].elapsed_ms(e[2]) for e in ev[2:]])
print('fwd + bwd    %.3f ms wall: forward %.3f, backward %.3f' % (wall, f, b))
