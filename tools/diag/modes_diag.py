import sys, torch, numpy as np
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from test_gpu_configs import _cfg, _bench_step
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch
name = sys.argv[1]; nodes = [int(v) for v in sys.argv[2].split(',')]; seed = int(sys.argv[3])
res = {}
for compute in ('f16', 'f32'):
    torch.manual_seed(0)
    hip = GHN3(**_cfg(name), compute=compute).to('cuda').train()
    gb, nets = synthetic_batch(nodes, 1000 * seed + 17)
    plan = hip.compile(nets, gb, training=True)
    dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
    res[compute] = (hip, plan) + _bench_step(hip, plan, dout)
hip, plan, out, gflat, loss = res['f16']
_, _, out32, g32, loss32 = res['f32']
params = dict(hip.named_parameters())
rows = []
for pname, off in zip(plan.program.names, hip._offs):
    n = params[pname].numel()
    a, b = gflat[int(off):int(off) + n], g32[int(off):int(off) + n]
    rows.append((float((a - b).norm()) / (float(b.norm()) + 1e-12), pname, float(b.norm())))
rows.sort(reverse=True)
for r in rows[:14]: print('%.2e  %-40s ref norm %.3e' % r)
worst = 0
for p in plan.program.predicted:
    a, b = out[p['offset']:p['offset'] + p['numel']], out32[p['offset']:p['offset'] + p['numel']]
    worst = max(worst, float((a - b).norm() / (b.norm() + 1e-12)))
print('worst forward', worst, 'decoder rows', plan.program.M)
