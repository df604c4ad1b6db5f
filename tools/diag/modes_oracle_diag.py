import sys, torch
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from test_gpu_configs import _cfg
from util_parity import synthetic_case
from oracle import ghn3_ref as R
from ghn3_amd import GHN3
name = sys.argv[1]; nodes = [int(v) for v in sys.argv[2].split(',')]; seed = int(sys.argv[3])
torch.manual_seed(0)
ref_model = GHN3(**_cfg(name), compute='f32')
sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
oracle = R.GHN3Ref(**_cfg(name))
oracle.load_state_dict(sd)
nets_h, gb_h, nets_o, gb_o = synthetic_case(nodes, seed)
oracle.train()
nets_o, pred_o = oracle(nets_o, gb_o, keep_grads=True)
loss_o = sum(torch.norm(t, p='fro') for (_, _, _, t) in pred_o)
loss_o.backward()
po = dict(oracle.named_parameters())
for compute in ('f16', 'f32'):
    hip = GHN3(**_cfg(name), compute=compute)
    hip.load_state_dict(sd)
    hip = hip.to('cuda').train()
    nh, gh, _, _ = synthetic_case(nodes, seed)
    hip(nh, gh, keep_grads=True)
    loss = hip.predicted_param_norm()
    loss.backward()
    torch.cuda.synchronize()
    rows = []
    for k, p in hip.named_parameters():
        go = po[k].grad
        rows.append((float((p.grad.cpu().double() - go.double()).norm()) / (float(go.norm()) + 1e-12), k, float(go.norm())))
    rows.sort(reverse=True)
    print(compute, 'loss', loss.item(), loss_o.item())
    for r in rows[:6]: print('   %.2e  %-40s ref norm %.3e' % r)
