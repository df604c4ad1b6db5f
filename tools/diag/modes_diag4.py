"""Diagnostic: 1-D decoder backward intermediates, f16 vs f32 mode, default-initialised weights."""
import sys
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import torch, numpy as np
from util_parity import synthetic_case, ws_tensor
from test_gpu_configs import _cfg
from ghn3_amd import GHN3
name = sys.argv[1]; nodes = [int(v) for v in sys.argv[2].split(',')]; seed = int(sys.argv[3])
torch.manual_seed(0)
sd = {k: v.detach().clone() for k, v in GHN3(**_cfg(name), compute='f32').state_dict().items()}
res = {}
for compute in ('f32', 'f16'):
    hip = GHN3(**_cfg(name), compute=compute); hip.load_state_dict(sd); hip = hip.to('cuda').train()
    nh, gh, _, _ = synthetic_case(nodes, seed)
    hip(nh, gh, keep_grads=True)
    hip.predicted_param_norm().backward()
    torch.cuda.synchronize()
    plan = hip.last_plan; prog = plan.program
    M, C, n1, mc = prog.M, prog.C, prog.n1, prog.mc
    d = {}
    for n_, shape in (('h1d', (n1, 2 * C)), ('w1d', (n1, 2 * mc)), ('d_w1d', (n1, 2 * mc)), ('d_h1d', (n1, 2 * C))):
        if n_ in prog._ws_names:
            d[n_] = ws_tensor(plan, n_, shape).cpu().double()
    full = ws_tensor(plan, 'd_xrows', (M + n1, C)).cpu().double()
    d['d_rows_1d'] = full[M:]
    res[compute] = (prog, d, {k: p.grad.detach().cpu().double() for k, p in hip.named_parameters()})
prog, a, ga = res['f16']; _, b, gb_ = res['f32']
print('n1', prog.n1, 'mc', prog.mc)
for k in a:
    x, y = a[k], b[k]
    e = (x - y).norm(dim=1) / (y.norm(dim=1) + 1e-30)
    print('%-10s rel %.2e (max row %.2e, median %.1e) | abs max %.2e  value range: median |y| %.1e min nonzero %.1e' % (
        k, float((x - y).norm() / (y.norm() + 1e-30)), float(e.max()), float(e.median()), float((x - y).abs().max()),
        float(y.abs().median()), float(y.abs()[y.abs() > 0].min()) if (y.abs() > 0).any() else 0))
for k in ('decoder_1d.fc.0.weight', 'decoder_1d.fc.2.weight', 'bias_class.1.weight', 'decoder_1d.fc.0.bias'):
    if k in ga:
        print('%-26s grad rel %.2e (norm %.2e)' % (k, float((ga[k] - gb_[k]).norm() / (gb_[k].norm() + 1e-30)), float(gb_[k].norm())))
fa, fb = a['h1d'] > 0, b['h1d'] > 0
flips = (fa != fb)
print('h1d mask flips', int(flips.sum()), 'of', flips.numel(), 'rows with flips', int(flips.any(dim=1).sum()))
vals = torch.where(flips, torch.maximum(a['h1d'], b['h1d']), torch.zeros_like(a['h1d']))
print('largest value among flipped elements %.2e, rms of positive h1d %.2e' % (float(vals.max()), float(b['h1d'][fb].pow(2).mean().sqrt())))
small = (b['h1d'] > 0) & (b['h1d'] < 1e-4 * b['h1d'][fb].pow(2).mean().sqrt())
print('positive elements below 1e-4 rms:', int(small.sum()))
# per-row error of d_h1d explained by flips?
e = (a['d_h1d'] - b['d_h1d']).norm(dim=1) / (b['d_h1d'].norm(dim=1) + 1e-30)
print('rows with d_h1d err > 1e-3:', int((e > 1e-3).sum()), 'of which have flips:', int(((e > 1e-3) & flips.any(dim=1)).sum()))
cols = torch.nonzero(flips)[:, 1]
print('flipped columns (unique):', sorted(set(cols.tolist()))[:20], 'count per column', np.bincount(cols.numpy())[np.unique(cols.numpy())][:20])
j = int(cols[0])
print('column', j, 'f32-mode values (first 10 rows):', ['%.2e' % float(v) for v in b['h1d'][:10, j]], 'f16-mode:', ['%.2e' % float(v) for v in a['h1d'][:10, j]])
W1 = sd['decoder_1d.fc.0.weight'].double(); b1 = sd['decoder_1d.fc.0.bias'].double()
print('W1 row', j, 'norm %.3e mean %.3e std %.3e | bias %.3e | typical row norm %.3e' % (float(W1[j].norm()), float(W1[j].mean()), float(W1[j].std()), float(b1[j]), float(W1.norm(dim=1).median())))
