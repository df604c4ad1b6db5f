#!/usr/bin/env python3
"""Diagnostic (GPU box): staged split-bf16 plan (tile codes 44 / 45, LayerNorm prologues) against the round-3 plan (planes +
LayerNorm launches) on the same model / batch: per-layer backward intermediates and per-parameter gradients.
    python tools/diag/x3s_vs_x3.py ghn3xlm16 90 170"""
import os
import sys
import numpy as np
import torch
import _paths  # noqa: F401
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch
import recipe

name = sys.argv[1] if len(sys.argv) > 1 else 'ghn3lm8'
nodes = [int(v) for v in sys.argv[2:]] or [40, 70]
hid, layers, heads = recipe.VARIANTS[name]
cfg = dict(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers, weight_norm=True, ve=True,
           layernorm=True)
res = {}
for mode in ('0', '1'):
    os.environ['GHN3_X3S'] = mode
    hip = GHN3(**cfg, compute='f16')
    shapes = {k: tuple(v.shape) for k, v in hip.state_dict().items()}
    hip.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=31337).items()})
    hip = hip.to('cuda').train()
    gb, nets = synthetic_batch(nodes, 777000)
    plan = hip.compile(nets, gb, training=True)
    prog = plan.program
    hip._run_forward(plan)
    hip._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, torch.cuda.current_stream().cuda_stream)
    hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
    torch.cuda.synchronize()
    rows = prog.B * prog.N
    C = prog.C
    inter = {}
    for l in range(layers):
        for nm, w in (('x%d' % (l + 1), C), ('dz_%d' % l, 4 * C), ('dhA_%d' % l, C), ('gmid_%d' % l, C), ('dqkv_%d' % l, 3 * C),
                      ('dhB_%d' % l, C), ('gout_%d' % l, C)):
            if nm in prog._ws_names:
                off = prog._ws_names[nm]
                inter[nm] = plan.ws[off:off + 4 * rows * w].view(torch.float32).clone()
    for nm, n in (('d_xe', rows * C), ('dxa', rows * C)):
        off = prog._ws_names[nm]
        inter[nm] = plan.ws[off:off + 4 * n].view(torch.float32).clone()
    res[mode] = (plan.gflat.clone(), inter, prog.names, hip._offs.copy(), dict((k, v.numel()) for k, v in hip.named_parameters()))
    del hip, plan
g0, i0, names, offs, numel = res['0']
g1, i1 = res['1'][:2]
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
print('intermediates (staged vs round-3 plan):')
for nm in ['d_xe', 'dxa'] + [k for k in i0 if k not in ('d_xe', 'dxa')]:
    if nm in i1:
        r = rel(i1[nm], i0[nm])
        if r > 2e-5 or nm in ('d_xe', 'dxa'):
            print('  %-10s %.2e' % (nm, r))
print('parameter gradients:')
for nm, off in zip(names, offs):
    n = numel[nm]
    r = rel(g1[int(off):int(off) + n], g0[int(off):int(off) + n])
    if r > 5e-5:
        print('  %-40s %.2e' % (nm, r))
# where does d_xe differ?  (quirk Q1: bias / norm nodes of graph 1 read PADDED rows of graph 0, identical rows whose decoder_1d
# pre-activations can sit on a ReLU knife edge)
C = hid
a, b = i1['d_xe'].view(-1, C), i0['d_xe'].view(-1, C)
rowdiff = (a - b).norm(dim=1)
top = torch.argsort(rowdiff, descending=True)[:12]
print('d_xe rows with the largest difference (row, |diff|, |row|):', [(int(r), float(rowdiff[r]), float(b[r].norm())) for r in top])
print('total |diff| %.3e, of which rows >= %d of graph 0 (padding): %.3e' % (float(rowdiff.norm()), nodes[0],
      float(rowdiff[nodes[0]:max(nodes)].norm())))
