"""Diagnostic: are repeated steps of one plan bit-identical, with and without ghn3_run's table store (GHN3_RUN_CACHE)?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from util_parity import make_models, synthetic_case
from ghn3_amd import _lib as L

T_CFG = dict(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3, weight_norm=True, ve=True, layernorm=True)
hip, _ = make_models(T_CFG, 7)
hip.train()
ctx = L.context(0)
nets_a, gb_a, _, _ = synthetic_case([40], 4100)
plan = hip.compile(nets_a, gb_a, training=True)
outs = []
for k in range(10):
    torch.manual_seed(3)
    out = hip._run_forward(plan)
    tok = plan.tok.clone()
    o = out.detach().clone()
    grads = hip._run_backward(plan, torch.ones_like(out))
    torch.cuda.synchronize()
    outs.append((o, tok, plan.gflat.clone()))
    if len(outs) > 1:
        last = outs.pop()
    else:
        last = outs[0]
    print(k, 'out ptr %x tok ptr %x' % (out.data_ptr(), plan.tok.data_ptr()), 'stats', ctx.cache_stats(),
          'out==first', torch.equal(o, outs[0][0]), 'tok==first', torch.equal(tok, outs[0][1]),
          'maxdiff', float((o - outs[0][0]).abs().max()), 'gflat rel', float((last[2] - outs[0][2]).norm() / outs[0][2].norm()))
    del last, o, tok, out, grads
