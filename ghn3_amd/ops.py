"""
Target networks of the DeepNets-1M search space, in two flavours (behaviour of
/root/reference/ghn3/ops.py:143-585):

  * ``Network``      -- ``torch.nn`` layers with their own parameters (evaluation, graph construction);
  * ``NetworkLight`` -- the light layers of ``ghn3_amd.light_ops``: shapes only, until ``GHN3.forward`` assigns the
    predicted tensors (the networks ``train_ghn_ddp.py`` runs on images every step, trainer.py:308-319).

A network is a stem, ``n_cells`` DARTS-style cells built from a ``Genotype`` (pairs of (op name, input index) per
step, concatenation of the listed states) and a classifier.  Every definition below is written once as a plain class
whose layers come from ``self.L`` (the layer namespace of the flavour); ``_flavours`` derives the two concrete classes
at module level, so light networks pickle like any object (DataLoader workers, the loader pool of ``bench.py``).

The helpers the reference imports from the third-party ``ppuda`` package (absent from /root/reference: genotype
tuple, ``parse_op_ks``, ``drop_path``, ``_is_none``, auxiliary heads) are restated here from that package's published
behaviour -- parity for them is unpinned, see DESIGN.md 8.
"""

import os
import sys
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import light_ops
from . import target_ops
from .bookkeeping import named_layered_modules

Genotype = namedtuple('Genotype', 'normal normal_concat reduce reduce_concat')


def from_dict(d):
    """Genotype from the json form of the DeepNets-1M meta files ({'normal': [[op, idx], ...], ...})."""
    return Genotype(normal=[tuple(p) for p in d['normal']], normal_concat=list(d['normal_concat']),
                    reduce=[tuple(p) for p in d['reduce']], reduce_concat=list(d['reduce_concat']))


def parse_op_ks(op):
    """'sep_conv_5x5' -> ('sep_conv', 5); 'conv_7x1_1x7' -> ('conv2', 7); names without a size keep ks = 3."""
    parts = op.split('_')
    sizes = [p for p in parts if 'x' in p and p.replace('x', '').isdigit()]
    if not sizes:
        return op, 3
    name = '_'.join(p for p in parts if p not in sizes)
    if len(sizes) == 2:
        return name + '2', max(int(v) for v in sizes[0].split('x'))
    return name, int(sizes[0].split('x')[0])


def drop_path(x, drop_prob):
    """Per-sample stochastic depth: zero the branch with probability drop_prob, rescale the survivors."""
    if drop_prob > 0.:
        keep = 1. - drop_prob
        mask = torch.empty(x.size(0), 1, 1, 1, device=x.device, dtype=x.dtype).bernoulli_(keep)
        x = x / keep * mask
    return x


def _is_none(mod):
    """True when a layer below ``mod`` has no weight (the GHN was told not to predict it): the op is skipped."""
    if mod is None:
        return True
    if not hasattr(mod, 'named_modules'):
        return False
    for _, m in mod.named_modules():
        if hasattr(m, 'weight') and m.weight is None:
            return True
    return False


class _TorchLayers:
    """Layer namespace of the ``torch.nn`` flavour."""
    light = False
    Module = ModuleEmpty = nn.Module
    for _n in ('ModuleList', 'Sequential', 'Dropout', 'Identity', 'Linear', 'Conv2d', 'BatchNorm2d', 'LayerNorm',
               'AvgPool2d', 'MaxPool2d', 'AdaptiveAvgPool2d', 'ReLU', 'GELU', 'Hardswish'):
        locals()[_n] = getattr(nn, _n)
    del _n


class _LightLayers:
    """Layer namespace of the light flavour."""
    light = True
    for _n in light_ops.__all__:
        locals()[_n] = getattr(light_ops, _n)
    del _n


def bn_layer(Lyr, norm, C):
    if norm in (None, '', 'none'):
        return Lyr.Identity()
    if norm.startswith('bn'):
        return Lyr.BatchNorm2d(C, track_running_stats=norm.find('track') >= 0)
    raise NotImplementedError(norm)


# ======================================================================================================
# definitions (flavour-free): `self.L` is the layer namespace, `self.T` the table of sibling classes
# ======================================================================================================
class _Stride:
    _base = 'ModuleEmpty'

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        return x if self.stride == 1 else x[:, :, ::self.stride, ::self.stride]


class _Zero:
    _base = 'ModuleEmpty'

    def __init__(self, stride):
        super().__init__()
        self.stride = stride

    def forward(self, x):
        return (x if self.stride == 1 else x[:, :, ::self.stride, ::self.stride]).mul(0.)


class _FactorizedReduce:
    """Two 1x1 convolutions of stride 2 on the even and the odd pixel grid, concatenated (ops.py:163-178)."""

    def __init__(self, C_in, C_out, norm='bn', stride=2):
        super().__init__()
        assert C_out % 2 == 0
        Lyr = self.L
        self.stride = stride
        self.relu = Lyr.ReLU(inplace=False)
        self.conv_1 = Lyr.Conv2d(C_in, C_out // 2, 1, stride=stride, padding=0, bias=False)
        self.conv_2 = Lyr.Conv2d(C_in, C_out // 2, 1, stride=stride, padding=0, bias=False)
        self.bn = bn_layer(Lyr, norm, C_out)

    def forward(self, x):
        if x.is_cuda and self.stride == 2:               # (round 6: one dense-convolution node, target_ops.run_factorized_reduce)
            y = target_ops.run_factorized_reduce(self.relu, self.conv_1, self.conv_2, self.bn, x, self.stride)
            if y is not None:
                return y
        x = self.relu(x)
        shifted = x[:, :, 1:, 1:] if self.stride > 1 else x
        return self.bn(torch.cat([self.conv_1(x), self.conv_2(shifted)], dim=1))


class _ReLUConvBN:
    """ReLU - conv (or a 1 x ks, ks x 1 pair) - norm (ops.py:180-198)."""

    def __init__(self, C_in, C_out, ks=1, stride=1, padding=0, norm='bn', double=False):
        super().__init__()
        Lyr = self.L
        self.stride = stride
        if double:
            convs = [Lyr.Conv2d(C_in, C_in, (1, ks), stride=(1, stride), padding=(0, padding), bias=False),
                     Lyr.Conv2d(C_in, C_out, (ks, 1), stride=(stride, 1), padding=(padding, 0), bias=False)]
        else:
            convs = [Lyr.Conv2d(C_in, C_out, ks, stride=stride, padding=padding, bias=False)]
        self.op = Lyr.Sequential(Lyr.ReLU(inplace=False), *convs, bn_layer(Lyr, norm, C_out))
        self._pointwise = (not double) and ks == 1 and padding == 0

    def forward(self, x):
        if self._pointwise and x.is_cuda:                # (ReLU -> 1x1 conv -> norm: the fused HIP op without a depthwise stage)
            return target_ops.run_pointwise_block(list(self.op), x)
        if x.is_cuda and len(self.op) == 3:              # (round 6: ReLU -> k x k conv -> norm on the dense-convolution op)
            return target_ops.run_conv_block(list(self.op), x)
        if x.is_cuda and len(self.op) == 4 and os.environ.get('GHN3_NATIVE_PAIR', '1') != '0':   # (the 1 x k / k x 1 pair: two nodes of the same op)
            return target_ops.run_conv_pair_block(list(self.op), x)
        return self.op(x)


class _DilConv:
    """ReLU - dilated depthwise conv - 1x1 conv - norm (ops.py:200-216)."""

    def __init__(self, C_in, C_out, ks, stride, padding, dilation, norm='bn'):
        super().__init__()
        Lyr = self.L
        self.stride = stride
        self.op = Lyr.Sequential(
            Lyr.ReLU(inplace=False),
            Lyr.Conv2d(C_in, C_in, kernel_size=ks, stride=stride, padding=padding, dilation=dilation, groups=C_in,
                       bias=False),
            Lyr.Conv2d(C_in, C_out, kernel_size=1, padding=0, bias=False),
            bn_layer(Lyr, norm, C_out))

    def forward(self, x):
        # (one HIP node for the whole chain, forward and backward, where it applies: ghn3_amd/target_ops.py)
        return target_ops.run_block(list(self.op), x) if x.is_cuda else self.op(x)


class _SepConv:
    """Two (ReLU - depthwise conv - 1x1 conv - norm) blocks, the stride in the first (ops.py:218-237)."""

    def __init__(self, C_in, C_out, ks, stride, padding, norm='bn'):
        super().__init__()
        Lyr = self.L
        self.stride = stride

        def block(c_out, s):
            return [Lyr.ReLU(inplace=False),
                    Lyr.Conv2d(C_in, C_in, kernel_size=ks, stride=s, padding=padding, groups=C_in, bias=False),
                    Lyr.Conv2d(C_in, c_out, kernel_size=1, padding=0, bias=False),
                    bn_layer(Lyr, norm, c_out)]
        self.op = Lyr.Sequential(*block(C_in, stride), *block(C_out, 1))

    def forward(self, x):
        if not x.is_cuda:
            return self.op(x)
        layers = list(self.op)
        # (the intermediate activation only travels from one fused block to the next: it keeps the kernels' NHWC layout)
        return target_ops.run_block(layers[4:], target_ops.run_block(layers[:4], x, keep_layout=True))


class _ChannelSELayer:
    """Squeeze-and-excitation with a hard-swish gate (ops.py:239-274)."""

    def __init__(self, num_channels, reduction_ratio=2, dim_out=None, stride=1):
        super().__init__()
        if dim_out is not None:
            assert dim_out == num_channels, (dim_out, num_channels, 'only same dimensionality is supported')
        Lyr = self.L
        self.reduction_ratio, self.stride = reduction_ratio, stride
        self.fc1 = Lyr.Linear(num_channels, num_channels // reduction_ratio, bias=True)
        self.fc2 = Lyr.Linear(num_channels // reduction_ratio, num_channels, bias=True)
        self.relu = Lyr.ReLU(inplace=True)
        self.sigmoid = Lyr.Hardswish()

    def forward(self, x):
        b, c = x.shape[:2]
        y = target_ops.run_se_layer(self.fc1, self.fc2, x) if x.is_cuda else None     # (round 6: one fused node)
        if y is None:
            squeeze = x.reshape(b, c, -1).mean(dim=2)
            gate = self.sigmoid(self.fc2(self.relu(self.fc1(squeeze))))
            y = torch.mul(x, gate.view(b, c, 1, 1))
        return y[:, :, ::self.stride, ::self.stride] if self.stride > 1 else y


class _PosEnc:
    """Learned positional encoding added to the patch grid (ops.py:278-291)."""

    def __init__(self, C, ks):
        super().__init__()
        self.weight = [1, C, ks, ks] if self.L.light else nn.Parameter(torch.randn(1, C, ks, ks))

    def forward(self, x):
        return x + self.weight


class _FeedForward:
    """graphormer.py:22-48."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=None, drop=0):
        super().__init__()
        Lyr = self.L
        act_layer = act_layer or Lyr.GELU
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features

        def dropout():
            return Lyr.Dropout(drop) if drop > 0 else Lyr.Identity()
        self.net = Lyr.Sequential(Lyr.Linear(in_features, hidden_features), act_layer(), dropout(),
                                  Lyr.Linear(hidden_features, out_features), dropout())

    def forward(self, x):
        return self.net(x)


class _MultiHeadSelfAttention:
    """graphormer.py:66-142 without the edge branch (edge_dim = 0: target networks have no graph)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        Lyr = self.L
        self.num_heads, self.dim, self.edge_dim = num_heads, dim, 0
        self.scale = (dim // num_heads) ** -0.5
        self.to_qkv = Lyr.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = Lyr.Dropout(attn_drop) if attn_drop > 0 else Lyr.Identity()
        self.to_out = Lyr.Sequential(Lyr.Linear(dim, dim), Lyr.Dropout(proj_drop) if proj_drop > 0 else Lyr.Identity())

    def forward(self, x, edges=None, mask=None):
        B, N, C = x.shape
        q, k, v = self.to_qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4).unbind(0)
        attn = (q @ k.transpose(-2, -1)) * self.scale
        if edges is not None:
            attn = attn + edges.permute(0, 3, 1, 2)
        if mask is not None:
            attn = attn.masked_fill(~mask.unsqueeze(1), -2 ** 15)
        attn = self.attn_drop(attn.softmax(dim=-1))
        return self.to_out((attn @ v).transpose(1, 2).reshape(B, N, C)), edges


class _TransformerLayer:
    """Pre-LN transformer layer on a (B, C, H, W) feature map, the 'msa' op (graphormer.py:144-248 with edge_dim = 0)."""

    def __init__(self, dim, num_heads=8, mlp_ratio=1, qkv_bias=False, act_layer=None, eps=1e-5, stride=1):
        super().__init__()
        Lyr, T = self.L, self.T
        self.stride, self.edge_dim, self.return_edges = stride, 0, False
        self.ln1 = Lyr.LayerNorm(dim, eps=eps)
        self.attn = T['MultiHeadSelfAttention'](dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.ln2 = Lyr.LayerNorm(dim, eps=eps)
        self.ff = T['FeedForward'](in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, x, edges=None, mask=None):
        sz = x.shape
        if len(sz) == 2:
            x = x.unsqueeze(0)
        elif len(sz) == 4:
            x = x.reshape(sz[0], sz[1], -1).permute(0, 2, 1)
        assert x.dim() == 3, x.shape
        x = x + self.attn(self.ln1(x), edges, mask)[0]
        x = x + self.ff(self.ln2(x))
        if len(sz) == 4:
            x = x.permute(0, 2, 1).view(sz[0], x.shape[2], sz[2], sz[3])
            if self.stride > 1:
                x = x[:, :, ::self.stride, ::self.stride]
        return x


class _AuxiliaryHeadCIFAR:
    """DARTS auxiliary classifier on the 2/3-depth feature map (8 x 8 input assumed)."""

    def __init__(self, C, num_classes, norm='bn', pool_sz=5):
        super().__init__()
        Lyr = self.L
        self.features = Lyr.Sequential(
            Lyr.ReLU(inplace=True), Lyr.AvgPool2d(pool_sz, stride=3, padding=0, count_include_pad=False),
            Lyr.Conv2d(C, 128, 1, bias=False), bn_layer(Lyr, norm, 128), Lyr.ReLU(inplace=True),
            Lyr.Conv2d(128, 768, 2, bias=False), bn_layer(Lyr, norm, 768), Lyr.ReLU(inplace=True))
        self.classifier = Lyr.Linear(768, num_classes)

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))


class _AuxiliaryHeadImageNet:
    """DARTS auxiliary classifier for 224 x 224 inputs (14 x 14 feature map assumed)."""

    def __init__(self, C, num_classes, norm='bn'):
        super().__init__()
        Lyr = self.L
        self.features = Lyr.Sequential(
            Lyr.ReLU(inplace=True), Lyr.AvgPool2d(5, stride=2, padding=0, count_include_pad=False),
            Lyr.Conv2d(C, 128, 1, bias=False), bn_layer(Lyr, norm, 128), Lyr.ReLU(inplace=True),
            Lyr.Conv2d(128, 768, 2, bias=False), bn_layer(Lyr, norm, 768), Lyr.ReLU(inplace=True))
        self.classifier = Lyr.Linear(768, num_classes)

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))


def _make_op(T, Lyr, name, i, o, k, s, n):
    """The op table of the search space (ops.py:293-304): i, o = channels in / out, k = kernel size, s = stride,
    n = norm."""
    if name == 'none':
        return T['Zero'](s)
    if name == 'skip_connect':
        return Lyr.Identity() if s == 1 else T['FactorizedReduce'](i, o, norm=n)
    if name == 'avg_pool':
        return Lyr.AvgPool2d(k, stride=s, padding=k // 2, count_include_pad=False)
    if name == 'max_pool':
        return Lyr.MaxPool2d(k, stride=s, padding=k // 2)
    if name == 'conv':
        return T['ReLUConvBN'](i, o, k, s, k // 2, norm=n)
    if name == 'sep_conv':
        return T['SepConv'](i, o, k, s, k // 2, norm=n)
    if name == 'dil_conv':
        return T['DilConv'](i, o, k, s, k - k % 2, 2, norm=n)
    if name == 'conv2':
        return T['ReLUConvBN'](i, o, k, s, k // 2, norm=n, double=True)
    if name == 'conv_stride':
        return Lyr.Conv2d(i, o, k, stride=k, bias=False, padding=int(k < 4))
    if name == 'msa':
        return T['TransformerLayer'](i, stride=s)
    if name == 'cse':
        return T['ChannelSELayer'](i, dim_out=o, stride=s)
    raise KeyError(name)


class _Cell:
    """One cell: two preprocessed inputs, `steps` intermediate states (each the sum of two ops applied to earlier states),
    concatenation of the states the genotype lists (ops.py:306-401)."""

    def __init__(self, genotype, C_prev_prev, C_prev, C_in, C_out, reduction, reduction_prev, norm='bn', preproc=True,
                 is_vit=False, cell_ind=0):
        super().__init__()
        Lyr, T = self.L, self.T
        self._is_vit, self._cell_ind, self.genotype = is_vit, cell_ind, genotype
        self._has_none = any(n[0] == 'none' for n in genotype.normal + genotype.reduce)
        halve = reduction_prev and not is_vit
        if preproc:
            self.preprocess0 = T['FactorizedReduce'](C_prev_prev, C_out, norm=norm) if halve else \
                T['ReLUConvBN'](C_prev_prev, C_out, norm=norm)
            self.preprocess1 = T['ReLUConvBN'](C_prev, C_out, norm=norm)
        else:
            self.preprocess0 = T['Stride'](stride=2) if halve else Lyr.Identity()
            self.preprocess1 = Lyr.Identity()
        pairs = genotype.reduce if reduction else genotype.normal
        self._concat = genotype.reduce_concat if reduction else genotype.normal_concat
        self.multiplier = len(self._concat)
        self._steps = len(pairs) // 2
        self._indices = tuple(index for _, index in pairs)
        self._ops = Lyr.ModuleList()
        for name, index in pairs:
            stride = 2 if (reduction and index < 2 and not is_vit) else 1
            name, ks = parse_op_ks(name)
            self._ops.append(_make_op(T, Lyr, name, C_in if index <= 1 else C_out, C_out, ks, stride, norm))

    def _branch(self, op, h, drop_path_prob):
        """One op applied to one earlier state; None when the op or its input is absent."""
        if h is None or isinstance(op, self.T['Zero']) or _is_none(op):
            return None
        h = op(h)
        if self.training and drop_path_prob > 0 and not isinstance(op, self.L.Identity):
            h = drop_path(h, drop_path_prob)
        return h

    def forward(self, s0, s1, drop_path_prob=0):
        s0 = None if (s0 is None or _is_none(self.preprocess0)) else self.preprocess0(s0)
        s1 = None if (s1 is None or _is_none(self.preprocess1)) else self.preprocess1(s1)
        states = [s0, s1]
        for k in range(self._steps):
            a = self._branch(self._ops[2 * k], states[self._indices[2 * k]], drop_path_prob)
            b = self._branch(self._ops[2 * k + 1], states[self._indices[2 * k + 1]], drop_path_prob)
            states.append(a if b is None else (b if a is None else a + b))
        outs = [states[k] for k in self._concat]
        if any(s is None for s in outs):
            # states that received nothing ('none' ops) are replaced by zeros of a live state's shape
            assert self._has_none, self.genotype
            live = next((s for s in outs if s is not None), None)
            if live is None:
                return None
            outs = [live * 0 if s is None else s for s in outs]
        return torch.cat(outs, dim=1)


def _layer_seq(Lyr, norm, spec):
    """Sequential from a table of rows: ('conv', c_in, c_out, kernel, stride) -- bias-free, 'same' padding --, ('bn', c),
    ('relu',), ('maxpool',), ('id',), ('linear', c_in, c_out), ('dropout',)."""
    make = {'conv': lambda ci, co, k, st: Lyr.Conv2d(ci, co, k, stride=st, padding=k // 2, bias=False),
            'bn': lambda c: bn_layer(Lyr, norm, c),
            'relu': lambda: Lyr.ReLU(inplace=True),
            'maxpool': lambda: Lyr.MaxPool2d(3, stride=2, padding=1, ceil_mode=False),
            'id': lambda: Lyr.Identity(),
            'linear': lambda ci, co: Lyr.Linear(ci, co),
            'dropout': lambda: Lyr.Dropout(p=0.5, inplace=False)}
    return Lyr.Sequential(*[make[row[0]](*row[1:]) for row in spec])


def network_plan(C, num_classes, n_steps, n_cells, ks, is_imagenet_input, stem_pool, stem_type, imagenet_stride, is_vit,
                 preproc, C_mult, fc_layers, fc_dim, glob_avg, multiplier):
    """The widths of a DeepNets-1M network as data (the bookkeeping of ops.py:403-521, independent of the layer flavour):
    {'stems': {attribute: layer table}, 'cells': [per cell: C_prev_prev, C_prev, C_in, C_out, reduction, reduction_prev],
    'aux_at', 'head': layer table of the classifier}.  `multiplier` = (normal, reduction) cell: states it concatenates."""
    hi_res = bool(is_imagenet_input)
    stems, width0 = {}, C                                   # width of the two states entering cell 0
    if is_vit:
        pass                                                # (patch embedding + positional encoding: built by the caller)
    elif stem_type == 0:
        width0 = C * 3 if (preproc and not hi_res) else C
        stems['stem'] = [('conv', 3, width0, ks, imagenet_stride if hi_res else 1), ('bn', width0),
                         ('maxpool',) if stem_pool else ('id',)]
    else:
        st = 2 if hi_res else 1
        stems['stem0'] = [('conv', 3, C // 2, ks, st), ('bn', C // 2), ('relu',), ('conv', C // 2, C, 3, st), ('bn', C)]
        stems['stem1'] = [('relu',), ('conv', C, C, 3, 2), ('bn', C)]
    reductions = {c for c in (n_cells // 3, 2 * n_cells // 3) if c > 0}
    cells, width, older, newer = [], C, width0, width0
    for c in range(n_cells):
        red = c in reductions
        width = width * C_mult if red else width
        # (a single-step cell without preprocessing widens one cell ahead of the next reduction)
        ahead = (c + 1) in reductions and n_steps == 1 and not preproc
        cells.append(dict(C_prev_prev=older, C_prev=newer, C_in=width if preproc else newer,
                          C_out=width * C_mult if ahead else width, reduction=red, width=width,
                          out_width=multiplier[int(red)] * width,
                          reduction_prev=(stem_type == 1) if c == 0 else cells[-1]['reduction']))
        older, newer = newer, cells[-1]['out_width']
    feat = newer
    if not glob_avg:
        small = stem_type == 1 or stem_pool
        side = (7 if small else 14) if hi_res else (4 if small else 8)
        feat *= side * side
    dims = [feat] + [fc_dim] * (fc_layers - 1) + [num_classes]
    assert fc_layers <= 1 or fc_dim > 0, fc_dim
    head = []
    for k in range(len(dims) - 1):
        head += ([('relu',), ('dropout',)] if k else []) + [('linear', dims[k], dims[k + 1])]
    return dict(stems=stems, cells=cells, aux_at=2 * n_cells // 3, head=head)


class _Network:
    """Stem + cells + classifier (ops.py:403-569): built from network_plan()'s tables."""

    def __init__(self, C, num_classes, genotype, n_cells, ks=3, is_imagenet_input=True, stem_pool=False, stem_type=0,
                 imagenet_stride=4, is_vit=None, norm='bn-track', preproc=True, C_mult=2, fc_layers=0, fc_dim=0,
                 glob_avg=True, auxiliary=False):
        super().__init__()
        Lyr, T = self.L, self.T
        assert stem_type in (0, 1), ('either 0 (simple) or 1 (imagenet-style) stem must be chosen', stem_type)
        n_steps = len(genotype.normal_concat)
        assert preproc or (n_steps <= 1 and C_mult <= 1), 'preprocessing layers must be used in this case'
        self.genotype, self._C, self._auxiliary, self._stem_type = genotype, C, auxiliary, stem_type
        self.drop_path_prob = 0
        self.expected_input_sz = 224 if is_imagenet_input else 32
        self._is_vit = is_vit if is_vit is not None else any(n[0] == 'msa' for n in genotype.normal + genotype.reduce)
        self._n_cells, self._glob_avg = n_cells, glob_avg
        plan = network_plan(C, num_classes, n_steps, n_cells, ks, is_imagenet_input, stem_pool, stem_type, imagenet_stride,
                            self._is_vit, preproc, C_mult, fc_layers, fc_dim, glob_avg,
                            multiplier=(len(genotype.normal_concat), len(genotype.reduce_concat)))
        if self._is_vit:
            self.stem0 = _make_op(T, Lyr, 'conv_stride', 3, C, 16 if is_imagenet_input else 3, None, None)
            self.pos_enc = T['PosEnc'](C, 14 if is_imagenet_input else 11)
        for attr, spec in plan['stems'].items():
            setattr(self, attr, _layer_seq(Lyr, norm, spec))
        self._auxiliary_cell_ind = plan['aux_at']
        self.cells = Lyr.ModuleList()
        for c, w in enumerate(plan['cells']):
            cell = T['Cell'](genotype, w['C_prev_prev'], w['C_prev'], C_in=w['C_in'], C_out=w['C_out'],
                             reduction=w['reduction'], reduction_prev=w['reduction_prev'], norm=norm, is_vit=self._is_vit,
                             preproc=preproc, cell_ind=c)
            # (the planned width of the cell's output = what the built cell concatenates: ops.py:503 `multiplier * C_curr`)
            assert cell.multiplier * w['width'] == w['out_width'], (c, cell.multiplier, w)
            self.cells.append(cell)
            if auxiliary and c == plan['aux_at']:
                width = w['out_width']
                if is_imagenet_input:
                    self.auxiliary_head = T['AuxiliaryHeadImageNet'](width, num_classes, norm=norm)
                else:
                    self.auxiliary_head = T['AuxiliaryHeadCIFAR'](width, num_classes, norm=norm,
                                                                  pool_sz=2 if (stem_type == 1 or stem_pool) else 5)
        if glob_avg:
            self.global_pooling = Lyr.AdaptiveAvgPool2d(1)
        self.classifier = _layer_seq(Lyr, norm, plan['head'])
        if Lyr.light:
            # the parameter table GHN3.forward walks (nn.py:612); built once, here
            self.__dict__['_layered_modules'] = named_layered_modules(self)

    @staticmethod
    def _run_stem(stem, x):
        """The stem's conv -> norm windows on the fused dense-convolution op (round 6), the rest layer by layer."""
        if torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and target_ops.enabled() and hasattr(stem, '__iter__') and \
                os.environ.get('GHN3_NATIVE_STEM', '1') != '0':
            return target_ops.run_layer_seq(stem, x)
        return stem(x)

    def forward(self, x):
        if x.is_cuda and x.dim() == 4 and not self._is_vit and target_ops.enabled() and \
                os.environ.get('GHN3_NATIVE_CL', '0') == '1':
            # NHWC in memory from the stem on, so that the fused HIP layers (target_ops) never convert.  OFF by default: on
            # this ROCm build the stock ATen / MIOpen layers return WRONG GRADIENTS for channels_last activations (same
            # logits, parameter gradients up to 86 % off in the preprocessing conv / BatchNorm layers:
            # profiles/r05c_target_ops_channels_last_diag.txt), so the fused layers hand NCHW tensors to their neighbours
            x = x.contiguous(memory_format=torch.channels_last)
        if self._is_vit:
            patches = target_ops.run_conv_layer(self.stem0, x) if (x.is_cuda and target_ops.enabled() and
                                                                   os.environ.get('GHN3_NATIVE_STEM', '1') != '0') else self.stem0(x)
            s0 = s1 = self.pos_enc(patches)
        elif self._stem_type == 1:
            s0 = self._run_stem(self.stem0, x)
            s1 = None if _is_none(self.stem1) else self._run_stem(self.stem1, s0)
        else:
            s0 = s1 = self._run_stem(self.stem, x)
        logits_aux = None
        for c, cell in enumerate(self.cells):
            s0, s1 = s1, cell(s0, s1, self.drop_path_prob)
            if self._auxiliary and c == self._auxiliary_cell_ind and self.training:
                small_vit = self._is_vit and self.expected_input_sz == 32
                logits_aux = self.auxiliary_head(F.adaptive_avg_pool2d(s1, 8) if small_vit else s1)
        if s1 is None:
            raise ValueError('the network has invalid configuration: the output is None')
        out = self.global_pooling(s1) if self._glob_avg else s1
        with torch.autocast(out.device.type, enabled=False):       # the classifier always runs in fp32
            logits = self.classifier(out.float().reshape(out.size(0), -1))
        return logits, logits_aux


# ======================================================================================================
# the two flavours
# ======================================================================================================
_DEFS = (_Stride, _Zero, _FactorizedReduce, _ReLUConvBN, _DilConv, _SepConv, _ChannelSELayer, _PosEnc, _FeedForward,
         _MultiHeadSelfAttention, _TransformerLayer, _AuxiliaryHeadCIFAR, _AuxiliaryHeadImageNet, _Cell, _Network)


def _flavours():
    this = sys.modules[__name__]
    tables = {}
    for Lyr, suffix in ((_TorchLayers, ''), (_LightLayers, 'Light')):
        T = {}
        for d in _DEFS:
            short = d.__name__[1:]
            base = getattr(Lyr, getattr(d, '_base', 'Module'))
            body = {'L': Lyr, 'T': T, '__module__': __name__, '__qualname__': short + suffix, '__doc__': d.__doc__}
            if Lyr.light and short.startswith('AuxiliaryHead'):
                # the auxiliary classifiers stay torch.nn modules with parameters of their own in a light network too
                # (the reference takes them from ppuda, ops.py:21,506-510): the GHN does not predict them
                T[short] = tables[False][short]
                continue
            cls = type(short + suffix, (d, base), body)
            T[short] = cls
            setattr(this, short + suffix, cls)
        tables[Lyr.light] = T
    return tables


_T = _flavours()
types_torch_nn, types_light = _T[False], _T[True]       # name -> class, as the reference's two tables (ops.py:572-573)
del _T
