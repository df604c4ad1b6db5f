"""Diagnostic (not a test): per-tensor forward / per-parameter gradient errors of the HIP path vs the oracle at released
widths.   python tools/diag/gpu_diag_configs.py ghn3xlm16 40 f16 [more node counts...]"""
import sys, os
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import torch
from util_parity import rel_l2, make_models, synthetic_case, predicted_dict_hip
from test_gpu_configs import _cfg

name, compute = sys.argv[1], sys.argv[2]
nodes = [int(v) for v in sys.argv[3:]]
hip, oracle = make_models(_cfg(name), 7, compute=compute)
nets_h, gb_h, nets_o, gb_o = synthetic_case(nodes, nodes[0] * 1000)
hip.train()
nets_h = hip(nets_h, gb_h, keep_grads=True)
loss = hip.predicted_param_norm()
loss.backward()
torch.cuda.synchronize()
oracle.train()
nets_o, pred_o = oracle(nets_o, gb_o, keep_grads=True)
loss_o = sum(torch.norm(t, p='fro') for (_, _, _, t) in pred_o)
loss_o.backward()
print(name, compute, nodes, 'loss', loss.item(), loss_o.item())
pred_h = predicted_dict_hip(hip.last_plan, hip.last_plan.out)
errs = []
for k, (ind, attr, m, t) in enumerate(pred_o):
    errs.append((rel_l2(pred_h[k].detach().cpu(), t.detach()), k, attr, tuple(t.shape)))
errs.sort(reverse=True)
print('forward worst:', errs[:6])
po = dict(oracle.named_parameters())
g = []
for k, p in hip.named_parameters():
    go = po[k].grad
    err = float((p.grad.cpu().double() - go.double()).norm())
    g.append((err / (float(go.norm()) + 1e-12), k, err, float(go.norm())))
g.sort(reverse=True)
for row in g[:25]:
    print('grad %.3e  %-45s err %.3e norm %.3e' % row)
