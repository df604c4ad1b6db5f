"""GPU box: bench.py's `train_step` extra alone (ghn3xlm16, one 256-node graph: forward + loss + backward + fused clip / AdamW), for
A/B runs of optimizer switches:   GHN3_ADAMW_NT=0 python tools/diag/train_step_ab.py [steps]
Prints: fwd+bwd alone, training step with the serial optimizer, with the overlapped optimizer, the optimizer pass alone."""
import sys, time
import torch
import _paths  # noqa: F401
import bench
from ghn3_amd import GHN3, _lib as L
from ghn3_amd.optim import FusedAdamW
from ghn3_amd.synthetic import synthetic_batch

n_t = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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


def step():
    ghn._run_forward(plan)
    ctx.run(fin, prog.problems, plan.bufs, stream)
    ghn._run_backward(plan, None, norm_g=one)


def loop(update, overlap=False):
    for _ in range(3):
        step()
        if update:
            opt.step(plan.gflat, plan=plan, local_grads=True, overlap=overlap)
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    for _ in range(n_t):
        step()
        if update:
            opt.step(plan.gflat, plan=plan, local_grads=True, overlap=overlap)
    opt.wait()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t_) / n_t


res = []
for rep in range(2):
    t0, t1, t2 = loop(False), loop(True, False), loop(True, True)
    ea, eb = L.Event(), L.Event()
    ea.record(stream)
    for _ in range(n_t):
        opt.step(plan.gflat, plan=plan, local_grads=True)
    eb.record(stream)
    torch.cuda.synchronize()
    res.append((t0, t1, t2, ea.elapsed_ms(eb) / n_t))
for r in res:
    print('fwd+bwd %.3f ms | train step, serial optimizer %.3f | overlapped %.3f | clip + AdamW alone %.3f' % r, flush=True)
