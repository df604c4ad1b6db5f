"""Micro-benchmark of the dense-convolution op (ghn3_conv_bn_fwd / _bwd) on the shapes the training loop's architecture stream
produces (batch 64, 32 x 32 images): microseconds per forward / backward, and the same for the stock MIOpen layers."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from ghn3_amd import target_ops as T

SHAPES = [  # C_in, C_out, H, (kh, kw), stride, pad
    (32, 32, 16, (3, 3), 1, 1), (64, 64, 8, (3, 3), 1, 1), (128, 128, 4, (3, 3), 1, 1), (256, 256, 4, (3, 3), 1, 1),
    (64, 64, 16, (5, 5), 1, 2), (128, 128, 8, (5, 5), 1, 2), (64, 64, 8, (1, 7), 1, (0, 3)), (128, 128, 8, (7, 1), 1, (3, 0)),
    (48, 48, 32, (3, 3), 2, 1), (96, 96, 16, (7, 7), 1, 3), (256, 256, 8, (3, 3), 1, 1), (64, 128, 16, (2, 2), 2, 0),
]
REPS = int(os.environ.get('REPS', '30'))


def timed(fn):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS


print('%-34s %10s %10s %10s %10s   %s' % ('shape', 'fwd us', 'bwd us', 'stock fwd', 'stock bwd', 'GFLOP fwd'))
tot = [0.0, 0.0, 0.0, 0.0]
for ci, co, hw, ks, st, pad in SHAPES:
    x = torch.randn(64, ci, hw, hw, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, *ks, device='cuda') / (ci * ks[0] * ks[1]) ** 0.5).requires_grad_(True)
    g, b = torch.ones(co, device='cuda', requires_grad=True), torch.zeros(co, device='cuda', requires_grad=True)
    out, _ = T.conv_bn(x, w, g, b, st, pad, 1, True)
    up = torch.randn_like(out)
    xs = x.detach().contiguous(memory_format=torch.contiguous_format).requires_grad_(True)

    def fwd():
        with torch.no_grad():
            T.conv_bn(x, w, g, b, st, pad, 1, True)

    def fwdbwd():
        o, _ = T.conv_bn(x, w, g, b, st, pad, 1, True)
        o.backward(up)

    def sfwd():
        with torch.no_grad():
            F.batch_norm(F.conv2d(F.relu(xs), w, None, st, pad), None, None, g, b, True, 0.1, 1e-5)

    ups = up.contiguous(memory_format=torch.contiguous_format)

    def sfwdbwd():
        o = F.batch_norm(F.conv2d(F.relu(xs), w, None, st, pad), None, None, g, b, True, 0.1, 1e-5)
        o.backward(ups)

    tf, tfb, sf, sfb = timed(fwd), timed(fwdbwd), timed(sfwd), timed(sfwdbwd)
    gf = 2.0 * out.numel() * ci * ks[0] * ks[1] / 1e9
    for k, v in enumerate((tf, tfb - tf, sf, sfb - sf)):
        tot[k] += v
    print('%-34s %10.1f %10.1f %10.1f %10.1f   %.2f' % ('%d->%d %dx%d k%s s%s' % (ci, co, hw, hw, ks, st), tf, tfb - tf, sf, sfb - sf, gf))
print('%-34s %10.1f %10.1f %10.1f %10.1f' % ('sum', *tot))
