"""Diagnostic (not a test): f16-mode backward against the exact-fp32 mode of the HIP path itself, per node of d_xe.
    python tools/diag/gpu_diag_modes.py ghn3lm8 25 40"""
import sys, os
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import numpy as np
import torch
from util_parity import make_models, synthetic_case, ws_tensor
from test_gpu_configs import _cfg

name = sys.argv[1]
nodes = [int(v) for v in sys.argv[2:]]
res = {}
for compute in ('f32', 'f16'):
    hip, _ = make_models(_cfg(name), 7, compute=compute)
    nets_h, gb_h, _, _ = synthetic_case(nodes, nodes[0] * 1000)
    hip.train()
    nets_h = hip(nets_h, gb_h, keep_grads=True)
    loss = hip.predicted_param_norm()
    loss.backward()
    torch.cuda.synchronize()
    plan = hip.last_plan
    prog = plan.program
    rows = prog.B * prog.N
    M, C_ = prog.M, prog.C
    extra = {n: ws_tensor(plan, n, (M, w)).cpu().double() for n, w in
             (('u', 8 * C_), ('t', 4 * C_), ('d_u', 8 * C_), ('d_t', 4 * C_), ('d_xrows', C_))}
    res[compute] = dict(prog=prog, extra=extra, d_xe=ws_tensor(plan, 'd_xe', (rows, prog.C)).cpu().double(),
                        g={k: p.grad.detach().cpu().double() for k, p in hip.named_parameters()})
a, b = res['f16'], res['f32']
e = (a['d_xe'] - b['d_xe']).norm(dim=1) / (b['d_xe'].norm(dim=1) + 1e-30)
print('d_xe per-node rel err: median %.2e max %.2e' % (e.median(), e.max()))
prog = a['prog']
node_group = {}
for g in prog.conv_groups:
    for ind in g['inds']:
        node_group[prog._src_row(ind)] = ('conv', g['key'], g['o'], g['i'], g['kh'], g['kw'])
for key, inds in prog.oned_plain + prog.oned_clsb:
    for ind in inds:
        node_group.setdefault(prog._src_row(ind), ('1d', key))
order = torch.argsort(e, descending=True)
for r in order[:25].tolist():
    print('row %4d err %.2e norm %.2e  %s' % (r, e[r], b['d_xe'][r].norm(), node_group.get(r)))
print('gemm groups:', [(g['kind'], g['o'], g['i_ld'], g['rows'], g.get('nc'), len(g['subs'])) for g in prog.gemm_groups])
for k in ('decoder.conv.2.weight', 'decoder.conv.2.bias', 'decoder.conv.0.weight', 'decoder.fc.0.weight', 'embed.weight'):
    d = (a['g'][k] - b['g'][k])
    print('%-28s rel %.2e' % (k, d.norm() / b['g'][k].norm()))
# dW2 per (o', i') row block: where is the error?
C = prog.C
d = (a['g']['decoder.conv.2.weight'] - b['g']['decoder.conv.2.weight']).view(C, C, 8 * C)
n = b['g']['decoder.conv.2.weight'].view(C, C, 8 * C)
blk = 32
err = d.pow(2).sum(2).view(C // blk, blk, C // blk, blk).sum((1, 3)).sqrt()
nrm = n.pow(2).sum(2).view(C // blk, blk, C // blk, blk).sum((1, 3)).sqrt()
np.set_printoptions(linewidth=200, precision=1)
print('dW2 rel err x1e4 per (o-block, i-block) of %d:' % blk)
print((1e4 * err / (nrm + 1e-30)).numpy())
print('block norms x1e2:')
print((1e2 * nrm).numpy())
print('abs err x1e4:')
print((1e4 * err).numpy())

# ---- per decoder row: u, t (forward), d_u, d_t, d_xrows (backward), f16 mode vs f32 mode ------------------
def rowkeys(prog):
    keys = {}
    for g in prog.conv_groups:
        for n_idx, ind in enumerate(g['inds']):
            for p in range(g['hw']):
                keys[(ind, p)] = (g['row0'] + n_idx * g['hw'] + p, g['key'])
    return keys

ka, kb = rowkeys(a['prog']), rowkeys(b['prog'])
common = sorted(ka)
ia = torch.tensor([ka[k][0] for k in common])
ib = torch.tensor([kb[k][0] for k in common])
for nm in ('u', 't', 'd_u', 'd_t', 'd_xrows'):
    xa, xb = a['extra'][nm][ia], b['extra'][nm][ib]
    er = (xa - xb).norm(dim=1) / (xb.norm(dim=1) + 1e-30)
    tot = (xa - xb).norm() / xb.norm()
    w = torch.argsort(er, descending=True)[:6].tolist()
    print('%-8s total rel %.2e; worst rows: %s' % (nm, tot, [(common[i], ka[common[i]][1], '%.1e' % er[i], 'norm %.1e' % xb[i].norm(),
                                                          'amax %.1e' % xb[i].abs().max()) for i in w]))
am = a['extra']['d_t'][ia].abs()
print('d_t abs: max %.2e median %.2e min-nonzero %.2e' % (am.max(), am[am > 0].median(), am[am > 0].min()))
am = a['extra']['d_u'][ia].abs()
print('d_u abs: max %.2e median %.2e' % (am.max(), am[am > 0].median()))
