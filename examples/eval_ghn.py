#!/usr/bin/env python3
"""
Counterpart of the reference's eval_ghn.py (call sequence of /root/reference/eval_ghn.py:41-42,107-183) on an MI355X:

    ghn = from_pretrained(ckpt).to(device).eval()            # or a randomly initialised GHN-3 of a released size
    for every architecture in the queue:
        with torch.no_grad():
            model = ghn(model, graphs=graphs, bn_track_running_stats=True, reduce_graph=True)   # predict parameters
        total norm of the predicted parameters  ->  norm_check against the known answers (ghn3_results.json)
        evaluation forward passes with BatchNorm in batch-statistics mode (eval_ghn.py:155-160)

    python examples/eval_ghn.py [--ckpt ghn3xlm16.pt] [--ghn ghn3xlm16] [--arch resnet50,resnet_small] [--save_ckpt out.pt]

Stand-ins where the image has no data: torchvision / DeepNets-1M / ImageNet are not installed, so the architecture queue
holds the hand-written networks of this directory and "evaluation" runs on a random batch (top-1 against random labels
is chance level by construction; the point is the call sequence, the norm check and the timing).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ghn3_amd import GHN3, Graph, GraphBatch, from_pretrained, norm_check, get_metadata          # noqa: E402
from ghn_single_model import ResNet50, Bottleneck, MODELS                                          # noqa: E402


class ResNetSmall(nn.Module):
    """A short bottleneck ResNet for CIFAR-sized inputs."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32)
        self.layer1 = nn.Sequential(Bottleneck(32, 16, 1), Bottleneck(64, 16, 1))
        self.layer2 = nn.Sequential(Bottleneck(64, 32, 2))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(128, num_classes)

    def forward(self, x):
        x = torch.relu(self.bn1(self.conv1(x)))
        return self.fc(torch.flatten(self.avgpool(self.layer2(self.layer1(x))), 1))


ARCHS = {'resnet50': (ResNet50, 224), 'resnet_small': (ResNetSmall, 32)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ghn', default='ghn3xlm16', choices=sorted(MODELS))
    ap.add_argument('--ckpt', default=None, help='GHN-3 checkpoint (ghn3xlm16.pt ...); default: random init')
    ap.add_argument('--arch', default='resnet50,resnet_small')
    ap.add_argument('--compute', default='f16', choices=['f32', 'f16', 'bf16'])
    ap.add_argument('--save_ckpt', default=None)
    ap.add_argument('--batch', type=int, default=8)
    args = ap.parse_args()
    device = 'cuda'
    if args.ckpt:
        ghn = from_pretrained(args.ckpt, compute=args.compute).to(device).eval()           # eval_ghn.py:41
        norms = get_metadata(os.path.basename(args.ckpt), attr='paramnorm')                  # eval_ghn.py:54
    else:
        hid, layers, heads = MODELS[args.ghn]
        torch.manual_seed(0)
        ghn = GHN3(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
                   weight_norm=True, ve=True, layernorm=True, compute=args.compute).to(device).eval()
        norms = None
    queue = [a for a in args.arch.split(',') if a]
    matched = []
    t_all = time.time()
    for k, arch in enumerate(queue):
        cls, imsize = ARCHS[arch]
        model = cls().to(device)
        n_params = sum(p.numel() for p in model.parameters()) / 1e6
        print('\n%d/%d: %s with %.2fM parameters' % (k + 1, len(queue), arch.upper(), n_params), end='...', flush=True)
        graphs = GraphBatch([Graph(model, ve_cutoff=50 if ghn.ve else 1)], dense=True)
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            model = ghn(model, graphs=graphs, bn_track_running_stats=True, reduce_graph=True)  # eval_ghn.py:147-148
            if args.save_ckpt is not None:
                torch.save({'state_dict': model.state_dict()}, args.save_ckpt)
            model.eval()

            def bn_set_train(module):                                                         # eval_ghn.py:155-160
                if isinstance(module, nn.BatchNorm2d):
                    module.track_running_stats = False
                    module.training = True
            model.apply(bn_set_train)
        torch.cuda.synchronize()
        t_pred = time.time() - t0
        total, expected, ok = norm_check(model, arch=arch, ghn3_name=os.path.basename(args.ckpt or 'randinit'),
                                         expected=(norms or {}).get(arch))
        if ok is not None:
            matched.append(ok)
        print('done in %.3f sec' % t_pred)
        with torch.no_grad():
            x = torch.randn(args.batch, 3, imsize, imsize, device=device)
            y = model(x)
            top1 = (y.argmax(1) == torch.randint(0, y.shape[1], (args.batch,), device=device)).float().mean().item()
        print('evaluation forward on a random batch: logits %s, finite=%s, top1 vs random labels=%.3f'
              % (tuple(y.shape), bool(torch.isfinite(y).all()), top1), flush=True)
    print('\n%d architectures in %.2f s; norm checks: %s' % (len(queue), time.time() - t_all,
                                                             ('%d/%d matched' % (sum(matched), len(matched)))
                                                             if matched else 'no known answers for these weights'))


if __name__ == '__main__':
    main()
