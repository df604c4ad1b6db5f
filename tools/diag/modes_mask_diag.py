"""GPU box: one batch of the f16-vs-f32 sweep (tools/diag/modes_sweep.py) -- per-parameter gradient differences and the ReLU
masks of decoder_1d's hidden layer (post-activation h1d in the workspace) in both modes: a unit whose pre-activation sits at
~1e-6 of its scale flips between any two arithmetics that differ at 1e-5, and with it one row's share of the gradients.
  python modes_mask_diag.py ghn3xlm16 93 27        (nodes, index k of the sweep: seed 5000 + 31 k)"""
import sys, torch, numpy as np
import _paths  # noqa: F401
import recipe
from test_gpu_configs import _cfg, _bench_step
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch
name = sys.argv[1]; nodes = [int(v) for v in sys.argv[2].split(',')]; k = int(sys.argv[3])
shapes = {n: tuple(v.shape) for n, v in GHN3(**_cfg(name)).state_dict().items()}
sd = {n: torch.from_numpy(v) for n, v in recipe.seeded_state_dict(shapes, seed=7).items()}
res = {}
for compute in ('f16', 'f32'):
    m = GHN3(**_cfg(name), compute=compute); m.load_state_dict(sd); hip = m.to('cuda').train()
    gb, nets = synthetic_batch(nodes, 5000 + 31 * k)
    plan = hip.compile(nets, gb, training=True)
    dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
    out, gflat, loss = _bench_step(hip, plan, dout)
    prog = plan.program
    off = prog._ws_names['h1d']
    h1d = plan.ws[off:off + 4 * prog.n1 * 2 * prog.C].view(torch.float32).view(prog.n1, 2 * prog.C).clone()
    res[compute] = (hip, plan, gflat.clone(), h1d)
hip, plan, g16, h16 = res['f16']; _, _, g32, h32 = res['f32']
rows = []
params = dict(hip.named_parameters())
for pname, off in zip(plan.program.names, hip._offs):
    n = params[pname].numel()
    a, b = g16[int(off):int(off) + n], g32[int(off):int(off) + n]
    rows.append((float((a - b).norm()) / (float(b.norm()) + 1e-12), pname))
rows.sort(reverse=True)
print('worst gradients:', ['%.2e %s' % r for r in rows[:6]])
flip = (h16 > 0) != (h32 > 0)
print('decoder_1d hidden units: %d x %d, mask flips %d' % (h16.shape[0], h16.shape[1], int(flip.sum())))
idx = torch.nonzero(flip)
for r, c in idx[:10].tolist():
    col = h32[:, c]
    print('  row %d unit %d: f16-mode %.3e  f32-mode %.3e   (rms of the unit over rows %.3e)' % (r, c, float(h16[r, c]), float(h32[r, c]), float(col.pow(2).mean().sqrt())))
