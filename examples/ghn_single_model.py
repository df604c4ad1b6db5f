#!/usr/bin/env python3
"""
Predict all parameters of one PyTorch model with GHN-3 on an MI355X -- the call sequence of the reference's
examples/ghn_single_model.py:27-32 and eval_ghn.py:41-42,147-169:

    ghn = from_pretrained(ckpt).to(device).eval()          # or a randomly initialised GHN3 of a released size
    model = ghn(model)                                      # graph built on the fly (ghn3_amd.Graph(model))
    print total parameter norm

    python examples/ghn_single_model.py [--ghn ghn3xlm16 | --ckpt path/to/ghn3xlm16.pt] [--arch resnet50]

torchvision is not installed in this image, so --arch selects one of the hand-written networks below; with torchvision
present any ``torchvision.models.<arch>()`` works the same way.
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ghn3_amd import GHN3, Graph, GraphBatch, from_pretrained                          # noqa: E402

MODELS = {'ghn3tm8': (64, 3, 8), 'ghn3sm8': (128, 5, 16), 'ghn3lm8': (256, 12, 16), 'ghn3xlm16': (384, 24, 16)}


class Bottleneck(nn.Module):
    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    """Layer shapes and topology of torchvision.models.resnet50 (25,557,032 parameters)."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin, layers = 64, []
        for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            seq = []
            for b in range(blocks):
                seq.append(Bottleneck(cin, planes, stride if b == 0 else 1))
                cin = planes * 4
            layers.append(nn.Sequential(*seq))
        self.layer1, self.layer2, self.layer3, self.layer4 = layers
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, num_classes)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ghn', default='ghn3xlm16', choices=sorted(MODELS))
    ap.add_argument('--ckpt', default=None, help='GHN-3 checkpoint (ghn3xlm16.pt ...); default: random init')
    ap.add_argument('--arch', default='resnet50')
    ap.add_argument('--compute', default='f32', choices=['f32', 'f16', 'bf16'])
    args = ap.parse_args()
    device = 'cuda'
    if args.ckpt:
        ghn = from_pretrained(args.ckpt, compute=args.compute).to(device)
    else:
        hid, layers, heads = MODELS[args.ghn]
        torch.manual_seed(0)
        ghn = GHN3(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
                   weight_norm=True, ve=True, layernorm=True, compute=args.compute).to(device)
    ghn.eval()
    model = ResNet50().to(device)
    n_params = sum(p.numel() for p in model.parameters())
    t0 = time.time()
    graph = Graph(model, ve_cutoff=50 if ghn.ve else 1)
    t_graph = time.time() - t0
    with torch.no_grad():
        for it in range(2):                                           # second pass: warm kernels / allocator
            torch.cuda.synchronize()
            t0 = time.time()
            model = ghn(model, GraphBatch([graph], dense=True), bn_track_running_stats=True, reduce_graph=True)
            torch.cuda.synchronize()
            t_pred = time.time() - t0
    total_norm = torch.norm(torch.stack([p.norm() for p in model.parameters()]), 2)
    print('%s: %d nodes, graph built in %.2f s (CPU); %d parameters predicted in %.4f s (%.1f M params/s); '
          'total param norm %.4f' % (args.arch, graph.n_nodes, t_graph, n_params, t_pred, n_params / t_pred / 1e6,
                                     total_norm.item()))
    model.eval()
    with torch.no_grad():
        y = model(torch.randn(2, 3, 224, 224, device=device))
    print('forward with the predicted parameters:', tuple(y.shape), 'finite' if torch.isfinite(y).all() else 'NOT finite')


if __name__ == '__main__':
    main()
