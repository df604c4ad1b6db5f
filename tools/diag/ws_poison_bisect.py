"""GPU box: which workspace regions must read as zeros before a plan's first run?  The workspace of a tiny f16-mode plan is
filled with NaN bytes, the regions Program.ws_zero names are zeroed, and the forward + backward (upstream-gradient route) is
repeated with ONE further named region zeroed at a time: the regions that turn a non-finite gradient finite."""
import sys, os
import numpy as np
import torch
import _paths  # noqa: F401
import recipe
from util_parity import make_models, tiny_case

case = sys.argv[1] if len(sys.argv) > 1 else 'ragged3'
hip, _ = make_models(dict(recipe.TINY_CFG), recipe.TINY_SEED, compute='f16')
hip.train()
nets_h, gb_h, _, _ = tiny_case(case)
plan = hip.compile(nets_h, gb_h, training=True)
prog = plan.program
names = sorted(prog._ws_names, key=lambda k: prog._ws_names[k])
offs = [prog._ws_names[k] for k in names] + [prog.ws_bytes]
req = set(o for o, _ in list(prog.ws_zero) + list(getattr(prog, 'ws_zero_dout', ())))


def run(extra=()):
    plan.ws.fill_(0xff)
    for off, n in list(prog.ws_zero) + list(getattr(prog, 'ws_zero_dout', ())):
        plan.ws[off:off + n].zero_()
    for k in extra:
        i = names.index(k)
        plan.ws[offs[i]:offs[i + 1]].zero_()
    hip._run_forward(plan)
    dout = torch.randn(prog.out_numel, device='cuda') * 1e-3
    hip._run_backward(plan, dout)
    torch.cuda.synchronize()
    out_ok = all(bool(torch.isfinite(plan.out[p['offset']:p['offset'] + p['numel']]).all()) for p in prog.predicted)
    return out_ok, bool(torch.isfinite(plan.gflat).all())


print('case', case, 'regions', len(names), 'baseline (ws_zero only): forward finite %s, gradients finite %s' % run())
for k in names:
    if prog._ws_names[k] in req:
        continue
    f, g = run((k,))
    if f and g:
        print('  zeroing %-14s (%d bytes) makes everything finite' % (k, offs[names.index(k) + 1] - offs[names.index(k)]))
print('all zero:', run(tuple(names)))
