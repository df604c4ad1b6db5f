"""Which parameters of a light-flavour network get different gradients when the fused layers keep their NHWC output (lazy layout)?
    python tools/diag/lazy_layout_diag.py [case]"""
import os, sys
import numpy as np, torch
import _paths  # noqa: F401
import network_cases, recipe
from ghn3_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else 'cifar_darts'
geno, kw, img = network_cases.CASES[name]
g = ops.Genotype(**geno)
res = {}
for mode in ('stock', 'fused'):
    os.environ['GHN3_NATIVE_OPS'] = '0' if mode == 'stock' else '1'
    torch.manual_seed(0)
    kws = {k: ('bn' if (k == 'norm' and v) else v) for k, v in kw.items()}
    net = ops.NetworkLight(genotype=g, **kws)
    x = torch.from_numpy(recipe.seeded_images(img, seed=7)).cuda()
    table = {}
    for cell in net._layered_modules:
        table.update(cell)
    shapes = [(n, tuple(e['sz'])) for n, e in table.items()]
    params = recipe.seeded_net_params(shapes, seed=len(name))
    leaves = {}
    for n, e in table.items():
        t = torch.from_numpy(params[n]).cuda().requires_grad_(True)
        leaves[n] = t
        setattr(e['module'], 'weight' if e['is_w'] else 'bias', t)
    for attr in ('auxiliary_head',):
        if hasattr(net, attr):
            getattr(net, attr).cuda()
    net.train()
    torch.manual_seed(123)
    logits, aux = net(x)
    loss = logits.square().mean() + (aux.square().mean() if aux is not None else 0.)
    loss.backward()
    torch.cuda.synchronize()
    res[mode] = (logits.detach().cpu(), {n: (t.grad.detach().cpu() if t.grad is not None else None) for n, t in leaves.items()})
l0, g0 = res['stock']; l1, g1 = res['fused']
print('logits rel', float((l1 - l0).norm() / l0.norm()))
bad = []
for n in g0:
    a, b = g1[n], g0[n]
    if a is None or b is None or float(b.norm()) == 0:
        continue
    e = float((a - b).norm() / b.norm())
    bad.append((e, n, tuple(b.shape)))
bad.sort(reverse=True)
for e, n, sh in sorted(bad, key=lambda t: list(g0).index(t[1])):
    print('%.3e %s %s' % (e, n, sh))
