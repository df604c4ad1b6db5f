"""Where a network on the fused target-network layers departs from the stock path: per-parameter gradient differences of one
tests/golden/network_cases.py case under (a) stock NCHW, (b) stock layers on a channels_last input, (c) fused layers +
channels_last, (d) fused layers, NCHW between them.    python tools/diag/target_ops_diag.py [case]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import network_cases      # noqa: E402
import recipe             # noqa: E402
from ghn3_amd import ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cifar_darts'
geno, kw, img = network_cases.CASES[name]
g = ops.Genotype(**geno)


def run(native, cl, force_cl_input=False):
    os.environ['GHN3_NATIVE_OPS'] = native
    os.environ['GHN3_NATIVE_CL'] = cl
    torch.manual_seed(0)
    net = ops.Network(genotype=g, **kw).cuda()
    params = recipe.seeded_net_params([(n, tuple(p.shape)) for n, p in net.named_parameters()], seed=len(name))
    with torch.no_grad():
        for n, p in net.named_parameters():
            p.copy_(torch.from_numpy(params[n]))
    x = torch.from_numpy(recipe.seeded_images(img, seed=7)).cuda()
    if force_cl_input:
        x = x.contiguous(memory_format=torch.channels_last)
    net.train()
    torch.manual_seed(123)
    logits, aux = net(x)
    loss = logits.square().mean() + (aux.square().mean() if aux is not None else 0.)
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().cpu().double(), {n: p.grad.detach().cpu().double() for n, p in net.named_parameters() if p.grad is not None}


base_l, base_g = run('0', '0')
for tag, args in (('stock, NCHW again', ('0', '0')), ('stock layers, channels_last input', ('0', '0', True)),
                  ('fused + channels_last', ('1', '1')), ('fused, NCHW between', ('1', '0'))):
    l, gr = run(*args)
    errs = sorted(((float((gr[n] - base_g[n]).norm() / (base_g[n].norm() + 1e-30)), n) for n in base_g), reverse=True)
    print('%-36s logits %.2e   worst gradients: %s' % (tag, float((l - base_l).norm() / base_l.norm()),
                                                     ', '.join('%s %.1e' % (n, e) for e, n in errs[:4])))
