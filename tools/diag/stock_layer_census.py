"""Diagnostic: which layers of the training loop's target networks still run on stock ATen / MIOpen kernels?
Runs Trainer.update on the architecture stream of examples/train_ghn_ddp.py and counts, per step, the stock Conv2d / BatchNorm2d /
Linear / pooling calls by configuration and by the module class that issued them."""
import collections
import inspect
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import torch
from ghn3_amd import GHN3, Trainer, light_ops
from ghn3_amd.deepnets1m import SampledNets
from ghn3_amd.graph import GraphBatch

STEPS = int(os.environ.get('CENSUS_STEPS', '12'))
counts = collections.Counter()
host = collections.Counter()


def caller():
    for fr in inspect.stack()[2:8]:
        slf = fr.frame.f_locals.get('self')
        if slf is not None and type(slf).__module__.endswith('ops') and not type(slf).__name__.startswith(('Sequential', 'Conv2d')):
            return type(slf).__name__
    return '?'


def wrap(cls, describe):
    orig = cls.forward

    def fwd(self, x):
        key = (cls.__name__, caller(), describe(self, x))
        t0 = time.perf_counter()
        y = orig(self, x)
        host[key] += time.perf_counter() - t0
        counts[key] += 1
        return y
    cls.forward = fwd


wrap(light_ops.Conv2d, lambda m, x: 'w%s s%s p%s d%s g%d x%s' % (tuple(m.weight.shape), m.stride, m.padding, m.dilation, m.groups,
                                                                 tuple(x.shape[1:])))
wrap(light_ops.BatchNorm2d, lambda m, x: 'x%s' % (tuple(x.shape[1:]),))
wrap(light_ops.Linear, lambda m, x: 'w%s x%s' % (tuple(m.weight.shape), tuple(x.shape)))
wrap(light_ops.AvgPool2d, lambda m, x: '')
wrap(light_ops.MaxPool2d, lambda m, x: '')
wrap(light_ops.AdaptiveAvgPool2d, lambda m, x: '')

hid, layers, heads = 64, 3, 8
config = {'max_shape': (hid, hid, 11, 11), 'num_classes': 10, 'weight_norm': True, 've': True, 'layernorm': True, 'hid': hid,
          'layers': layers, 'heads': heads}
torch.manual_seed(0)
ghn = GHN3(**config, compute='f16')
trainer = Trainer(ghn, opt='adamw', opt_args={'lr': 4e-4, 'weight_decay': 1e-2}, scheduler='cosine', n_batches=STEPS, grad_clip=5,
                  device='cuda', log_interval=100, amp=False, predparam_wd=3e-5, verbose=False)
gen = torch.Generator().manual_seed(1)
images = torch.randn(64, 3, 32, 32, generator=gen).cuda()
targets = torch.randint(0, 10, (64,), generator=gen).cuda()
nets = SampledNets(large_images=False, seed=0, max_nodes=400)
for step in range(STEPS):
    gb = GraphBatch([nets[step * 8 + k] for k in range(8)], dense=True)
    trainer.update(images, targets, graphs=gb)
torch.cuda.synchronize()
by_kind = collections.Counter()
for (kind, who, cfg), n in counts.items():
    by_kind[(kind, who)] += n
print('stock layer calls per step, by (layer, issuing class):')
for (kind, who), n in by_kind.most_common():
    t = sum(v for (k2, w2, _), v in host.items() if (k2, w2) == (kind, who))
    print('  %-20s %-24s %7.1f calls/step  %7.2f ms host/step (forward only)' % (kind, who, n / STEPS, 1e3 * t / STEPS))
print('top Conv2d configurations:')
for (kind, who, cfg), n in counts.most_common(400):
    if kind == 'Conv2d':
        print('  %6.2f /step  %-18s %s' % (n / STEPS, who, cfg))
