"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED.

Restatement of the pieces of the third-party package ``ppuda``
(github.com/facebookresearch/ppuda, installed by the reference from git HEAD with no
version pin -- /root/reference/README.md:47-52) that sit under the GHN-3 hot path.
``ppuda`` is not vendored in /root/reference and is not installed in this image, so these
classes restate its published structure; what the reference itself forces about them:

  * Sequential index positions        -- ghn3/nn.py:167-169 (decoder_1d.fc[-2], decoder.conv[-2],
                                          decoder.class_layer_predictor[-1]) and nn.py:727-733
                                          (conv[0], conv[2], class_layer_predictor[1] are 1x1 convs)
  * key-name / shape inference        -- ghn3/nn.py:69-88 (``embed.weight`` last dim = hid,
                                          ``decoder.conv.2.weight`` rows = max_shape**2,
                                          ``shape_enc.embed_spatial.weight`` has 9 rows iff s == 11)
  * total parameter count             -- 654,365,184 for ghn3xlm16
                                          (examples/ghn_all_pytorch.ipynb:109); checked in
                                          tests/test_oracle.py::test_param_counts
  * 15 node-type primitives           -- ghn3/graph.py:811,1008-1010
  * named_layered_modules contract    -- ghn3/nn.py:612-613,625,650,686-690
"""

import copy
import numpy as np
import torch
import torch.nn as nn

# Order as published in ppuda/deepnets1m/genotypes.py (length 15 pinned by the parameter count).
PRIMITIVES_DEEPNETS1M = [
    'max_pool', 'avg_pool', 'sep_conv', 'dil_conv', 'conv', 'msa', 'cse', 'sum',
    'concat', 'input', 'bias', 'bn', 'ln', 'pos_enc', 'glob_avg',
]


def shape_lookup_tables(num_classes, max_shape):
    """Channel / spatial vocabularies of ppuda's ShapeEncoder and the nearest-value lookups."""
    ch_steps = (2 ** 3, 2 ** 6, 2 ** 12, 2 ** 13)
    channels = np.unique([1, 3, num_classes] +
                         list(range(ch_steps[0], ch_steps[1], 2 ** 3)) +
                         list(range(ch_steps[1], ch_steps[2], 2 ** 4)) +
                         list(range(ch_steps[2], ch_steps[3] + 1, 2 ** 5)))
    spatial = np.unique(list(range(1, max(12, max_shape[3]), 2)) + [14, 16])

    channels_lookup = {int(c): i for i, c in enumerate(channels)}
    for c in range(4, int(channels[0])):
        channels_lookup[c] = channels_lookup[int(channels[0])]
    # 4-7 channels are treated as 8 channels (channels[0] == 1, so the loop above is empty
    # unless num_classes etc. change; the explicit rule below is the published behaviour)
    for c in range(4, 8):
        channels_lookup[c] = channels_lookup[8]
    for c in range(1, int(channels[-1])):
        if c not in channels_lookup:
            channels_lookup[c] = channels_lookup[int(channels[np.argmin(abs(channels - c))])]

    spatial_lookup = {int(c): i for i, c in enumerate(spatial)}
    spatial_lookup[2] = spatial_lookup[3]  # 2x2 treated as 3x3
    for c in range(1, int(spatial[-1])):
        if c not in spatial_lookup:
            spatial_lookup[c] = spatial_lookup[int(spatial[np.argmin(abs(spatial - c))])]
    return channels, spatial, channels_lookup, spatial_lookup


class ShapeEncoder(nn.Module):
    def __init__(self, hid, num_classes, max_shape, debug_level=0):
        super().__init__()
        assert max_shape[2] == max_shape[3], max_shape
        self.debug_level = debug_level
        self.num_classes = num_classes
        (self.channels, self.spatial,
         self.channels_lookup, self.spatial_lookup) = shape_lookup_tables(num_classes, max_shape)
        n_ch, n_s = len(self.channels), len(self.spatial)
        self.embed_spatial = nn.Embedding(n_s + 1, hid // 4)
        self.embed_channel = nn.Embedding(n_ch + 1, hid // 4)
        self.register_buffer('dummy_ind', torch.tensor([n_ch, n_ch, n_s, n_s], dtype=torch.long).view(1, 4),
                             persistent=False)

    def shape_indices(self, n_rows, params_map, predict_class_layers=True):
        shape_ind = self.dummy_ind.repeat(n_rows, 1)
        for node_ind in params_map:
            sz = params_map[node_ind][0]['sz']
            if sz is None:
                continue
            if len(sz) == 1:
                sz = (sz[0], 1)
            if len(sz) == 2:
                sz = (sz[0], sz[1], 1, 1)
            if len(sz) == 3:
                # unverified: same 3-D -> 4-D rule as ghn3/graph.py:874-878, nn.py:669-671
                if sz[0] == 1 and min(sz[1:]) > 1:
                    s_ = int(np.floor(sz[1] ** 0.5))
                    sz = (1, sz[2], s_, s_)
                else:
                    sz = (sz[0], sz[1], sz[2], 1)
            assert len(sz) == 4, sz
            if not predict_class_layers and params_map[node_ind][1] in ['cls_w', 'cls_b']:
                sz = (self.num_classes, *sz[1:])
            for i in range(4):
                if i < 2:
                    shape_ind[node_ind, i] = self.channels_lookup[
                        int(sz[i]) if int(sz[i]) in self.channels_lookup else int(self.channels[-1])]
                else:
                    shape_ind[node_ind, i] = self.spatial_lookup[
                        int(sz[i]) if int(sz[i]) in self.spatial_lookup else int(self.spatial[-1])]
        return shape_ind

    def forward(self, x, params_map, predict_class_layers=True):
        shape_ind = self.shape_indices(len(x), params_map, predict_class_layers)
        shape_embed = torch.cat((self.embed_channel(shape_ind[:, 0]),
                                 self.embed_channel(shape_ind[:, 1]),
                                 self.embed_spatial(shape_ind[:, 2]),
                                 self.embed_spatial(shape_ind[:, 3])), dim=1)
        return x + shape_embed


def _act(name):
    if name is None:
        return nn.Identity()
    assert name == 'relu', name
    return nn.ReLU()


class MLP(nn.Module):
    def __init__(self, in_features=32, hid=(32, 32), activation='relu', last_activation='same'):
        super().__init__()
        assert len(hid) > 0, hid
        fc = []
        for j, n in enumerate(hid):
            fc.extend([nn.Linear(in_features if j == 0 else hid[j - 1], n),
                       _act(last_activation if (j == len(hid) - 1 and last_activation != 'same') else activation)])
        self.fc = nn.Sequential(*fc)

    def forward(self, x, *args, **kwargs):
        if isinstance(x, tuple):
            x = x[0]
        return self.fc(x)


class ConvDecoder(nn.Module):
    def __init__(self, in_features=64, hid=(128, 256), out_shape=None, num_classes=None):
        super().__init__()
        assert len(hid) > 0, hid
        self.out_shape = out_shape
        self.num_classes = num_classes
        self.fc = nn.Sequential(nn.Linear(in_features, hid[0] * int(np.prod(out_shape[2:]))), nn.ReLU())
        conv = []
        for j, n_hid in enumerate(hid):
            n_out = int(np.prod(out_shape[:2])) if j == len(hid) - 1 else hid[j + 1]
            conv.extend([nn.Conv2d(n_hid, n_out, 1), _act(None if j == len(hid) - 1 else 'relu')])
        self.conv = nn.Sequential(*conv)
        self.class_layer_predictor = nn.Sequential(nn.ReLU(), nn.Conv2d(out_shape[0], num_classes, 1))


class _GatedGNNPlaceholder(nn.Module):
    """GHN-2's GatedGNN; GHN3.__init__ replaces it immediately (ghn3/nn.py:148), so only a stub."""

    def __init__(self, in_features):
        super().__init__()


class GHN(nn.Module):
    def __init__(self, max_shape, num_classes, hypernet='gatedgnn', decoder='conv', weight_norm=False, ve=False,
                 layernorm=False, hid=32, debug_level=0):
        super().__init__()
        assert len(max_shape) == 4, max_shape
        self.layernorm = layernorm
        self.weight_norm = weight_norm
        self.ve = ve
        self.debug_level = debug_level
        self.num_classes = num_classes
        self.max_shape = max_shape
        if layernorm:
            self.ln = nn.LayerNorm(hid)
        self.embed = nn.Embedding(len(PRIMITIVES_DEEPNETS1M), hid)
        self.shape_enc = ShapeEncoder(hid=hid, num_classes=num_classes, max_shape=max_shape,
                                      debug_level=debug_level)
        assert hypernet == 'gatedgnn' and decoder == 'conv', (hypernet, decoder)
        self.gnn = _GatedGNNPlaceholder(hid)
        self.decoder = ConvDecoder(in_features=hid, hid=(hid * 4, hid * 8), out_shape=max_shape,
                                   num_classes=num_classes)
        max_ch = max(max_shape[:2])
        self.decoder_1d = MLP(hid, hid=(hid * 2, 2 * max_ch), last_activation=None)
        self.bias_class = nn.Sequential(nn.ReLU(), nn.Linear(max_ch, num_classes))


def named_layered_modules(model):
    """param_name -> {'param_name','module','is_w','sz'} per cell (contract: ghn3/nn.py:612-690)."""
    if hasattr(model, 'module'):
        model = model.module
    layers = model._n_cells if hasattr(model, '_n_cells') else 1
    layered_modules = [{} for _ in range(layers)]
    cell_ind = 0
    for module_name, m in model.named_modules():
        cell_ind = m._cell_ind if hasattr(m, '_cell_ind') else cell_ind
        is_w = hasattr(m, 'weight') and m.weight is not None
        is_b = hasattr(m, 'bias') and m.bias is not None
        is_proj_w = hasattr(m, 'in_proj_weight') and m.in_proj_weight is not None
        is_proj_b = hasattr(m, 'in_proj_bias') and m.in_proj_bias is not None
        is_pos = hasattr(m, 'pos_embedding') and m.pos_embedding is not None

        def _sz(p):
            return tuple(p) if isinstance(p, (list, tuple)) else tuple(p.shape)

        if is_w:
            key = module_name + '.weight'
            layered_modules[cell_ind][key] = {'param_name': key, 'module': m, 'is_w': True, 'sz': _sz(m.weight)}
        if is_b:
            key = module_name + '.bias'
            layered_modules[cell_ind][key] = {'param_name': key, 'module': m, 'is_w': False, 'sz': _sz(m.bias)}
        if is_proj_w:
            key = module_name + '.in_proj_weight'
            layered_modules[cell_ind][key] = {'param_name': key, 'module': m, 'is_w': True,
                                              'sz': _sz(m.in_proj_weight)}
        if is_proj_b:
            key = module_name + '.in_proj_bias'
            layered_modules[cell_ind][key] = {'param_name': key, 'module': m, 'is_w': False,
                                              'sz': _sz(m.in_proj_bias)}
        if is_pos:
            key = module_name + '.pos_embedding.weight'
            layered_modules[cell_ind][key] = {'param_name': key, 'module': m, 'is_w': True,
                                              'sz': _sz(m.pos_embedding)}
    return layered_modules


def capacity(model, is_grad=True):
    c, n = 0, 0
    for p in model.parameters():
        if (is_grad and p.requires_grad) or not is_grad:
            c += 1
            n += int(np.prod(p.shape))
    return c, int(n)


# ---- ppuda.deepnets1m.{ops,net} helpers the reference's target networks import (ghn3/ops.py:20-21) ----------------
# Restated from the published ppuda package (absent offline): parity unpinned for these few functions; the reference's
# own Cell / Network / op classes that call them are pinned by tests/golden/networks.npz.

def parse_op_ks(op):
    """'sep_conv_5x5' -> ('sep_conv', 5), 'conv_7x1_1x7' -> ('conv2', 7), names without a size -> (name, 3)."""
    toks = op.split('_')
    ks = [t for t in toks if t.count('x') == 1 and all(v.isdigit() for v in t.split('x'))]
    if len(ks) == 0:
        return op, 3
    base = '_'.join(t for t in toks if t not in ks)
    if len(ks) > 1:
        return base + '2', max(int(v) for v in ks[0].split('x'))
    return base, int(ks[0].split('x')[0])


def drop_path(x, drop_prob):
    if drop_prob > 0.:
        keep_prob = 1. - drop_prob
        mask = torch.empty(x.size(0), 1, 1, 1, device=x.device).bernoulli_(keep_prob)
        x = x.div(keep_prob) * mask
    return x


def is_none(mod):
    if mod is None:
        return True
    for _, m in (mod.named_modules() if hasattr(mod, 'named_modules') else ()):
        if hasattr(m, 'weight') and m.weight is None:
            return True
    return False


def _norm(norm, C):
    if norm in [None, '', 'none']:
        return nn.Identity()
    return nn.BatchNorm2d(C, track_running_stats=norm.find('track') >= 0)


class AuxiliaryHeadCIFAR(nn.Module):
    def __init__(self, C, num_classes, norm='bn', pool_sz=5):
        super().__init__()
        self.features = nn.Sequential(
            nn.ReLU(inplace=True), nn.AvgPool2d(pool_sz, stride=3, padding=0, count_include_pad=False),
            nn.Conv2d(C, 128, 1, bias=False), _norm(norm, 128), nn.ReLU(inplace=True),
            nn.Conv2d(128, 768, 2, bias=False), _norm(norm, 768), nn.ReLU(inplace=True))
        self.classifier = nn.Linear(768, num_classes)

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))


class AuxiliaryHeadImageNet(nn.Module):
    def __init__(self, C, num_classes, norm='bn'):
        super().__init__()
        self.features = nn.Sequential(
            nn.ReLU(inplace=True), nn.AvgPool2d(5, stride=2, padding=0, count_include_pad=False),
            nn.Conv2d(C, 128, 1, bias=False), _norm(norm, 128), nn.ReLU(inplace=True),
            nn.Conv2d(128, 768, 2, bias=False), _norm(norm, 768), nn.ReLU(inplace=True))
        self.classifier = nn.Linear(768, num_classes)

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))
