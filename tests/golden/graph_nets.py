"""
Small hand-written networks with real forward passes, used to pin ``ghn3_amd.Graph(model)`` (automatic graph
construction) against the reference's ``ghn3.Graph(model)``: tests/golden/make_golden.py runs the reference on them
and stores node types, adjacency (with virtual edges) and node_info in graphs.npz; the tests rebuild the same
networks and compare.  No reference code here -- the architectures are generic PyTorch (residual blocks,
squeeze-excitation, depthwise / dilated convolutions, concatenation, multi-head attention).

``bases``: dict of the classes the reference recognises by isinstance (torchvision stand-ins when generating the
goldens, local classes of the same NAME in the tests): 'VisionTransformer', 'Encoder'.
"""

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNetTiny(nn.Module):
    expected_input_sz = 32

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 16, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(16)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_Block(16, 16, 1), _Block(16, 16, 1))
        self.layer2 = nn.Sequential(_Block(16, 32, 2), _Block(32, 32, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(32, 10)

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        x = self.layer2(self.layer1(x))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class _SE(nn.Module):
    def __init__(self, c, hard):
        super().__init__()
        self.fc1 = nn.Conv2d(c, c // 4, 1)
        self.fc2 = nn.Conv2d(c // 4, c, 1)
        self.hard = hard

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = self.fc2(F.relu(self.fc1(s)))
        s = F.hardsigmoid(s) if self.hard else torch.sigmoid(s)
        return x * s


class MobileSE(nn.Module):
    """Depthwise / dilated convolutions, squeeze-excitation (sigmoid and hard-sigmoid), average pooling,
    concatenation, 'classifier' head with two linear layers."""
    expected_input_sz = 32

    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 16, 3, 2, 1, bias=False), nn.BatchNorm2d(16), nn.Hardswish())
        self.dw = nn.Conv2d(16, 16, 3, 1, 1, groups=16, bias=False)
        self.dw_bn = nn.BatchNorm2d(16)
        self.se1 = _SE(16, hard=False)
        self.pw = nn.Conv2d(16, 24, 1, bias=False)
        self.pw_bn = nn.BatchNorm2d(24)
        self.dil = nn.Conv2d(24, 24, 3, 1, 2, dilation=2, groups=24, bias=True)
        self.se2 = _SE(24, hard=True)
        self.branch_a = nn.Conv2d(24, 8, 1)
        self.branch_b = nn.Conv2d(24, 8, 3, 1, 1)
        self.pool = nn.AvgPool2d(2, 2)
        self.classifier = nn.Sequential(nn.Linear(16, 32), nn.Hardswish(), nn.Dropout(0.0), nn.Linear(32, 10))

    def forward(self, x):
        x = self.features(x)
        y = F.relu(self.dw_bn(self.dw(x)))
        y = self.se1(y)
        x = x + y
        x = self.pw_bn(self.pw(x))
        x = self.se2(F.relu(self.dil(x)))
        x = torch.cat([self.branch_a(x), self.branch_b(x)], 1)
        x = self.pool(x)
        x = x.mean((2, 3))
        return self.classifier(x)


class AlexTiny(nn.Module):
    expected_input_sz = 32

    def __init__(self):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 16, 5, 2, 2), nn.ReLU(), nn.MaxPool2d(3, 2),
            nn.Conv2d(16, 32, 3, padding=1), nn.ReLU(),
            nn.Conv2d(32, 32, 3, padding=1), nn.ReLU(), nn.MaxPool2d(3, 2))
        self.avgpool = nn.AdaptiveAvgPool2d((2, 2))
        self.classifier = nn.Sequential(nn.Dropout(0.0), nn.Linear(32 * 4, 64), nn.ReLU(), nn.Dropout(0.0),
                                        nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 10))

    def forward(self, x):
        x = self.avgpool(self.features(x))
        return self.classifier(torch.flatten(x, 1))


class _EncoderBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.self_attention = nn.MultiheadAttention(d, heads, batch_first=True)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(nn.Linear(d, 2 * d), nn.GELU(), nn.Linear(2 * d, d))

    def forward(self, x):
        y = self.ln_1(x)
        y, _ = self.self_attention(y, y, y, need_weights=False)
        x = x + y
        return x + self.mlp(self.ln_2(x))


class _SoftmaxAttention(nn.Module):
    """Explicit softmax(q k^T) v attention (the 'msa' primitive; exercises the softmax edge repair)."""

    def __init__(self, d, heads):
        super().__init__()
        self.heads = heads
        self.to_qkv = nn.Linear(d, 3 * d, bias=False)
        self.to_out = nn.Linear(d, d)

    def forward(self, x):
        n, l, d = x.shape
        q, k, v = self.to_qkv(x).reshape(n, l, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        a = torch.softmax(q @ k.transpose(-2, -1) * (d // self.heads) ** -0.5, dim=-1)
        return self.to_out((a @ v).transpose(1, 2).reshape(n, l, d))


class AttnTiny(nn.Module):
    expected_input_sz = 32

    def __init__(self):
        super().__init__()
        self.stem = nn.Conv2d(3, 16, 4, 4)
        self.norm1 = nn.LayerNorm(16)
        self.attn = _SoftmaxAttention(16, 2)
        self.norm2 = nn.LayerNorm(16)
        self.mlp = nn.Sequential(nn.Linear(16, 32), nn.GELU(), nn.Linear(32, 16))
        self.norm = nn.LayerNorm(16)
        self.head = nn.Linear(16, 10)

    def forward(self, x):
        x = self.stem(x).flatten(2).transpose(1, 2)
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return self.head(self.norm(x).mean(1))


class TwoHeads(nn.Module):
    """Dict output with two heads, a weight shared by two layers (tied), BatchNorm1d / GroupNorm (no GHN-3 primitive:
    pruned), an embedding-like parameter of the top module."""
    expected_input_sz = 32

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 16, 3, 2, 1)
        self.gn = nn.GroupNorm(4, 16)
        self.conv2 = nn.Conv2d(16, 16, 3, 1, 1, bias=False)
        self.scale = nn.Parameter(torch.ones(1, 16, 1, 1))
        self.fc_a = nn.Linear(16, 16)
        self.fc_b = nn.Linear(16, 16)
        self.fc_b.weight = self.fc_a.weight                      # tied
        self.bn1d = nn.BatchNorm1d(16)
        self.head = nn.Linear(16, 10)
        self.aux_head = nn.Linear(16, 5)

    def forward(self, x):
        x = F.relu(self.gn(self.conv(x)))
        x = self.conv2(x) * self.scale
        x = F.adaptive_avg_pool2d(x, 1).flatten(1)
        y = self.bn1d(self.fc_b(F.relu(self.fc_a(x))))
        return {'out': self.head(y), 'aux': self.aux_head(x)}


def make_vit(bases):
    """A 2-layer ViT shaped like torchvision's (conv_proj, class_token, encoder.pos_embedding, encoder.layers,
    encoder.ln, heads.head).  The encoder instance has EXACTLY the type bases['Encoder'] (the reference looks the
    primitive up by exact type), so its forward is attached to the instance."""
    VT, Enc = bases['VisionTransformer'], bases['Encoder']
    d, heads, seq = 24, 2, 16 + 1

    class ViTTiny(VT):
        expected_input_sz = 32

        def __init__(self):
            nn.Module.__init__(self)
            self.conv_proj = nn.Conv2d(3, d, 8, 8)
            self.class_token = nn.Parameter(torch.zeros(1, 1, d))
            enc = Enc.__new__(Enc)
            nn.Module.__init__(enc)
            enc.pos_embedding = nn.Parameter(torch.empty(1, seq, d).normal_(std=0.02))
            enc.layers = nn.Sequential(_EncoderBlock(d, heads), _EncoderBlock(d, heads))
            enc.ln = nn.LayerNorm(d)
            enc.forward = lambda x, enc=enc: enc.ln(enc.layers(x + enc.pos_embedding))
            self.encoder = enc
            self.heads = nn.Sequential()
            self.heads.add_module('head', nn.Linear(d, 10))

        def forward(self, x):
            n = x.shape[0]
            x = self.conv_proj(x).reshape(n, d, -1).permute(0, 2, 1)
            x = torch.cat([self.class_token.expand(n, -1, -1), x], dim=1)
            x = self.encoder(x)
            return self.heads(x[:, 0])

    return ViTTiny


def local_bases():
    """Stand-ins with the class names the builder looks for (tests; torchvision is not installed)."""
    class VisionTransformer(nn.Module):
        pass

    class Encoder(nn.Module):
        pass
    return {'VisionTransformer': VisionTransformer, 'Encoder': Encoder}


def all_nets(bases):
    torch.manual_seed(0)
    return {'resnet_tiny': ResNetTiny(), 'mobile_se': MobileSE(), 'alex_tiny': AlexTiny(),
            'vit_tiny': make_vit(bases)(), 'attn_tiny': AttnTiny(), 'two_heads': TwoHeads()}
