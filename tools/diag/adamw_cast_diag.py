"""GPU box: FusedAdamW.step with and without the plan (GHN3_OP_ADAMW_CAST16 against GHN3_OP_ADAMW): where do parameters differ?"""
import sys
import numpy as np
import torch
import _paths  # noqa: F401
from util_parity import make_models, synthetic_case
from ghn3_amd import FusedAdamW

cfg = dict(max_shape=(64, 64, 11, 11), num_classes=10, hid=128, heads=8, layers=8, weight_norm=True, ve=True, layernorm=True)
nets_h, gb_h, _, _ = synthetic_case([48], 4800)
res = {}
for fused in (True, False):
    hip, _ = make_models(cfg, 7, compute='f16')
    hip.train()
    plan = hip.compile(nets_h, gb_h, training=True)
    opt = FusedAdamW(hip, lr=1e-2, weight_decay=0.05, max_grad_norm=1.0)
    torch.manual_seed(3)
    hip._run_forward(plan)
    dout = torch.randn(plan.program.out_numel, device='cuda') * 1e-3
    hip._run_backward(plan, dout)
    g = plan.gflat.clone()
    p0 = hip._flat.clone()
    opt.step(plan.gflat, plan=plan if fused else None)
    torch.cuda.synchronize()
    res[fused] = (hip._flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), g, p0, hip, plan)
a, b = res[True], res[False]
prog = a[6].program
lo = int(a[5]._offs[prog.slot['decoder.conv.2.weight']])
it = prog.shadow_w2['item']
hi = lo + it['rows'] * it['cols']
print('W2 range', lo, hi, 'of', a[0].numel(), 'item', {k: v for k, v in it.items()})
print('grads equal', torch.equal(a[3], b[3]), 'params before equal', torch.equal(a[4], b[4]))
for name, x, y in (('p', a[0], b[0]), ('m', a[1], b[1]), ('v', a[2], b[2])):
    d = (x - y).abs()
    print(name, 'max diff inside W2 %.3e outside %.3e | n differing inside %d outside %d' % (
        float(d[lo:hi].max()), float(torch.cat([d[:lo], d[hi:]]).max()), int((d[lo:hi] > 0).sum()), int((torch.cat([d[:lo], d[hi:]]) > 0).sum())))
d = (a[0] - b[0]).abs()[lo:hi].view(it['rows'], it['cols'])
idx = torch.nonzero(d > 0)
print('first differing (row, col):', idx[:8].tolist(), 'rel', float(d.max() / b[0][lo:hi].abs().max()))
outs = {}
for fused in (True, False):
    hip, plan = res[fused][5], res[fused][6]
    outs[fused] = hip._run_forward(plan).clone()
    torch.cuda.synchronize()
sa, sb = a[5]._shadow.view(torch.int16), b[5]._shadow.view(torch.int16)
lay = prog.shadow_lay
print('forward outputs equal', torch.equal(outs[True], outs[False]), 'max diff %.3e' % float((outs[True] - outs[False]).abs().max()))
w2h, w2hT, ldT = lay['w2h'], lay['w2hT'], lay['w2hT_ld']
n1 = it['rows'] * it['cols']
print('straight copies equal', torch.equal(sa[w2h:w2h + n1], sb[w2h:w2h + n1]), int((sa[w2h:w2h + n1] != sb[w2h:w2h + n1]).sum()))
ta, tb = sa[w2hT:w2hT + it['cols'] * ldT].view(it['cols'], ldT), sb[w2hT:w2hT + it['cols'] * ldT].view(it['cols'], ldT)
neq = ta != tb
print('transposed copies equal', not bool(neq.any()), int(neq.sum()), 'first', torch.nonzero(neq)[:6].tolist())
rest = torch.ones_like(sa, dtype=torch.bool); rest[w2h:w2h + n1] = False; rest[w2hT:w2hT + it['cols'] * ldT] = False
print('other copies differing', int((sa[rest] != sb[rest]).sum()))
from ghn3_amd import GHN3
hip = a[5]
m = GHN3(**cfg, compute='f16')
m.load_state_dict({k: v.detach().cpu().clone() for k, v in hip.state_dict().items()})
m = m.to('cuda').train()
want = m._run_forward(m.compile(nets_h, gb_h, training=True))
torch.cuda.synchronize()
for fused in (True, False):
    bad = [(p['attr'], tuple(p['shape'])) for p in prog.predicted
           if not torch.equal(outs[fused][p['offset']:p['offset'] + p['numel']], want[p['offset']:p['offset'] + p['numel']])]
    print('fused' if fused else 'plain', 'step: predicted tensors differing from a fresh model:', len(bad), 'of', len(prog.predicted), bad[:6])
again = a[5]._run_forward(a[6]).clone()
print('second forward of the fused model equals its first', torch.equal(again, outs[True]))
