"""Diagnostic: f16 mode vs f32 mode with default-initialised weights, Graphormer backward intermediates."""
import sys, os
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import torch
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
    rows = prog.B * prog.N
    d = {}
    for n_, w in [('d_xe', prog.C), ('dxa', prog.C)] + [('gout_%d' % l, prog.C) for l in range(prog.Lyr)] + \
            [('x%d' % l, prog.C) for l in range(prog.Lyr + 1)] + [('xe', prog.C)]:
        if n_ in prog._ws_names:
            d[n_] = ws_tensor(plan, n_, (rows, w)).cpu().double()
    res[compute] = (prog, d)
prog, a = res['f16']; _, b = res['f32']
nn = prog.n_nodes; N = prog.N
print('B', prog.B, 'N', N, 'n_nodes', nn)
def rel(x, y): return float((x - y).norm() / (y.norm() + 1e-30))
for k in ['xe', 'x%d' % prog.Lyr, 'd_xe', 'dxa'] + ['gout_%d' % l for l in reversed(range(prog.Lyr))]:
    if k in a:
        x, y = a[k], b[k]
        per_graph = []
        for g in range(prog.B):
            valid = slice(g * N, g * N + nn[g]); pad = slice(g * N + nn[g], (g + 1) * N)
            per_graph.append('g%d valid %.1e (norm %.1e) pad %.1e (norm %.1e)' % (g, rel(x[valid], y[valid]), float(y[valid].norm()),
                             rel(x[pad], y[pad]) if nn[g] < N else 0.0, float(y[pad].norm()) if nn[g] < N else 0.0))
        print('%-8s total %.2e | %s' % (k, rel(x, y), ' | '.join(per_graph)))
