"""GPU box: three training steps with and without the fused W2 update; after each step, where do parameters / moments / gradients differ?"""
import torch
import _paths  # noqa: F401
from util_parity import make_models, synthetic_case
from ghn3_amd import FusedAdamW

cfg = dict(max_shape=(64, 64, 11, 11), num_classes=10, hid=128, heads=8, layers=8, weight_norm=True, ve=True, layernorm=True)
nets_h, gb_h, _, _ = synthetic_case([48], 4800)
models = {}
for fused in (True, False):
    hip, _ = make_models(cfg, 7, compute='f16')
    hip.train()
    plan = hip.compile(nets_h, gb_h, training=True)
    models[fused] = (hip, plan, FusedAdamW(hip, lr=1e-2, weight_decay=0.05, max_grad_norm=1.0))
prog = models[True][1].program
lo = int(models[True][0]._offs[prog.slot['decoder.conv.2.weight']])
it = prog.shadow_w2['item']
hi = lo + it['rows'] * it['cols']
gen = torch.Generator(device='cuda'); gen.manual_seed(3)
for k in range(3):
    dout = torch.randn(prog.out_numel, device='cuda', generator=gen) * 1e-3
    st = {}
    for fused in (True, False):
        hip, plan, opt = models[fused]
        out = hip._run_forward(plan).clone()
        hip._run_backward(plan, dout)
        g = plan.gflat.clone()
        opt.step(plan.gflat, plan=plan if fused else None)
        torch.cuda.synchronize()
        st[fused] = (out, g, hip._flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.scal[:1].clone(), hip._shadow.clone())
    a, b = st[True], st[False]
    for name, x, y in (('out', a[0], b[0]), ('grad', a[1], b[1]), ('p', a[2], b[2]), ('m', a[3], b[3]), ('v', a[4], b[4]), ('sumsq', a[5], b[5])):
        d = (x - y).abs()
        if name in ('grad', 'p', 'm', 'v'):
            print('step', k, name, 'differing inside W2 %d outside %d max %.3e' % (int((d[lo:hi] > 0).sum()), int((d[:lo] > 0).sum() + (d[hi:] > 0).sum()), float(d.max())))
        elif name == 'out':
            bad = sum(1 for p in prog.predicted if not torch.equal(x[p['offset']:p['offset'] + p['numel']], y[p['offset']:p['offset'] + p['numel']]))
            print('step', k, 'out: predicted tensors differing', bad)
        else:
            print('step', k, name, float(x), float(y))
    print('step', k, 'shadow bytes differing', int((a[6] != b[6]).sum()))
