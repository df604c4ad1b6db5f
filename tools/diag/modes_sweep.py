"""Robustness sweep (GPU box): the 16-bit mode against the exact-fp32 mode on many random batches -- how often does ANY GHN parameter
gradient differ by more than 1e-3 (a ReLU-mask element on the knife edge flipped by the ~1e-5 deviation of the forward)?

    python tools/diag/modes_sweep.py ghn3xlm16 500 [variant]

variant: 'x3off' = the 16-bit decoders with an exact-fp32 Graphormer (graphormer_x3=False) instead of the split-bf16 one.
Prints one line per batch above 1e-3, a histogram and the fractions."""
import sys, torch, numpy as np
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import recipe
from test_gpu_configs import _cfg, _bench_step
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch
name = sys.argv[1]; n_batches = int(sys.argv[2]); variant = sys.argv[3] if len(sys.argv) > 3 else 'default'
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # (skip the batches in front of this one: same random stream)
shapes = {k: tuple(v.shape) for k, v in GHN3(**_cfg(name)).state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=7).items()}
models = {}
for compute in ('f16', 'f32'):
    kw = dict(graphormer_x3=False) if (variant == 'x3off' and compute == 'f16') else {}
    m = GHN3(**_cfg(name), compute=compute, **kw); m.load_state_dict(sd); models[compute] = m.to('cuda').train()
rs = np.random.RandomState(123)
worst, fwd = [], []
for k in range(n_batches):
    B = int(rs.choice([1, 1, 2, 3, 4]))
    nodes = [int(rs.randint(6, 330)) for _ in range(B)]
    if k < first:
        continue
    res = {}
    for compute in ('f16', 'f32'):
        hip = models[compute]
        gb, nets = synthetic_batch(nodes, 5000 + 31 * k)
        plan = hip.compile(nets, gb, training=True)
        dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
        res[compute] = (hip, plan) + _bench_step(hip, plan, dout)
    hip, plan, out, gflat, loss = res['f16']; _, _, out32, g32, loss32 = res['f32']
    # (the flat output has 16-float alignment gaps between the tensors that nothing writes: only the tensors are checked)
    out_ok = all(bool(torch.isfinite(out[p['offset']:p['offset'] + p['numel']]).all()) for p in plan.program.predicted)
    if not (torch.isfinite(gflat).all() and out_ok):
        pr = dict(hip.named_parameters())
        bad = [n for n, off in zip(plan.program.names, hip._offs) if not torch.isfinite(gflat[int(off):int(off) + pr[n].numel()]).all()]
        print('NON-FINITE at batch %d nodes %s: out finite %s, gradients: %s' % (k, nodes, out_ok, bad[:10]), flush=True)
        amax = plan.ws[plan.program.r_amax[1]:plan.program.r_amax[1] + 32].view(torch.float32)
        print('   amax slots', amax.tolist(), 'scal[60:64]', plan.scal[240:256].view(torch.float32).tolist(), flush=True)
        continue
    wf = max(float((out[p['offset']:p['offset'] + p['numel']] - out32[p['offset']:p['offset'] + p['numel']]).norm() /
                   (out32[p['offset']:p['offset'] + p['numel']].norm() + 1e-12)) for p in plan.program.predicted)
    params = dict(hip.named_parameters()); wg = (0.0, '')
    for pname, off in zip(plan.program.names, hip._offs):
        n = params[pname].numel()
        a, b = gflat[int(off):int(off) + n], g32[int(off):int(off) + n]
        ref = float(b.norm())
        if ref > 1e-3:
            wg = max(wg, (float((a - b).norm()) / ref, pname))
    worst.append((wg[0], wg[1], wf, nodes, int(plan.program.M)))
    fwd.append(wf)
    if wg[0] > 1e-3:
        print('%3d nodes %-22s rows %5d  worst fwd %.2e  worst grad %.2e %s' % (k, nodes, plan.program.M, wf, wg[0], wg[1]), flush=True)
    del res
g = np.asarray([w[0] for w in worst])
print('# %s, %d random batches of 1-4 graphs (6-329 nodes), variant %s: f16 mode against the exact-fp32 mode' % (name, n_batches, variant))
print('# worst forward difference over all batches %.2e (median %.2e)' % (max(fwd), float(np.median(fwd))))
print('# worst gradient per batch: median %.2e, 90 %% %.2e, 99 %% %.2e, max %.2e' % tuple(np.percentile(g, [50, 90, 99, 100])))
for thr in (5e-4, 7.5e-4, 1e-3, 1.5e-3, 2e-3, 3e-3):
    print('# batches with a gradient above %.1e: %d of %d = %.2f %%' % (thr, int((g > thr).sum()), n_batches, 100.0 * (g > thr).mean()))
by = {}
for w in worst:
    if w[0] > 1e-3:
        by[w[1]] = by.get(w[1], 0) + 1
print('# parameters that carry the excess:', by)
worst.sort(reverse=True)
print('WORST', worst[:5])
