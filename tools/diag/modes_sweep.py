"""One-off robustness sweep (GPU box): f16 mode vs exact-fp32 mode on many random batches; prints the worst pairs."""
import sys, torch, numpy as np
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import recipe
from test_gpu_configs import _cfg, _bench_step
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch
name = sys.argv[1]; n_batches = int(sys.argv[2])
shapes = {k: tuple(v.shape) for k, v in GHN3(**_cfg(name)).state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=7).items()}
models = {}
for compute in ('f16', 'f32'):
    m = GHN3(**_cfg(name), compute=compute); m.load_state_dict(sd); models[compute] = m.to('cuda').train()
rs = np.random.RandomState(123)
worst = []
for k in range(n_batches):
    B = int(rs.choice([1, 1, 2, 3, 4]))
    nodes = [int(rs.randint(6, 330)) for _ in range(B)]
    res = {}
    for compute in ('f16', 'f32'):
        hip = models[compute]
        gb, nets = synthetic_batch(nodes, 5000 + 31 * k)
        plan = hip.compile(nets, gb, training=True)
        dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
        res[compute] = (hip, plan) + _bench_step(hip, plan, dout)
    hip, plan, out, gflat, loss = res['f16']; _, _, out32, g32, loss32 = res['f32']
    assert torch.isfinite(gflat).all() and torch.isfinite(out[:plan.program.out_numel]).all(), nodes
    wf = max(float((out[p['offset']:p['offset'] + p['numel']] - out32[p['offset']:p['offset'] + p['numel']]).norm() /
                   (out32[p['offset']:p['offset'] + p['numel']].norm() + 1e-12)) for p in plan.program.predicted)
    params = dict(hip.named_parameters()); wg = (0.0, '')
    for pname, off in zip(plan.program.names, hip._offs):
        n = params[pname].numel()
        a, b = gflat[int(off):int(off) + n], g32[int(off):int(off) + n]
        ref = float(b.norm())
        if ref > 1e-3:
            wg = max(wg, (float((a - b).norm()) / ref, pname))
    worst.append((wg[0], wg[1], wf, nodes, plan.program.M))
    print('%2d nodes %-22s rows %5d  worst fwd %.2e  worst grad %.2e %s' % (k, nodes, plan.program.M, wf, wg[0], wg[1]), flush=True)
    del res
worst.sort(reverse=True)
print('WORST', worst[:3])
