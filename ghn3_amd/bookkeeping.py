"""
CPU-side shape bookkeeping of GHN3.forward (stays on the host per the north star):

  * node <-> target-module matching and parameter-group keys      /root/reference/ghn3/nn.py:594-692
  * shape -> embedding-index lookup of ppuda's ShapeEncoder        (third-party ppuda)
  * enumeration of a target network's parameters                  (ppuda named_layered_modules contract,
                                                                    visible at nn.py:612-613,625,650)
  * tile / normalise rules turned into kernel descriptors         nn.py:422-506,554-592,508-552
"""

import math
import numpy as np
import torch.nn as nn

PRIMITIVES_DEEPNETS1M = ['max_pool', 'avg_pool', 'sep_conv', 'dil_conv', 'conv', 'msa', 'cse', 'sum',
                         'concat', 'input', 'bias', 'bn', 'ln', 'pos_enc', 'glob_avg']


# ---- ShapeEncoder vocabulary (ppuda) ---------------------------------------------------------------
class ShapeVocab:
    _cache = {}

    def __new__(cls, num_classes, max_shape):
        # the lookup tables depend on (num_classes, max spatial size) only: built once per process
        key = (int(num_classes), int(max_shape[3]))
        inst = cls._cache.get(key)
        if inst is None:
            inst = super().__new__(cls)
            inst._build(num_classes, max_shape)
            cls._cache[key] = inst
        return inst

    def _build(self, num_classes, max_shape):
        ch = sorted(set([1, 3, num_classes]) | set(range(8, 64, 8)) | set(range(64, 4096, 16)) |
                    set(range(4096, 8193, 32)))
        sp = sorted(set(range(1, max(12, max_shape[3]), 2)) | {14, 16})
        self.channels = np.asarray(ch)
        self.spatial = np.asarray(sp)
        self.n_ch, self.n_sp = len(ch), len(sp)
        self.ch_lookup = {int(c): i for i, c in enumerate(ch)}
        for c in range(4, 8):                       # 4-7 channels are treated as 8
            self.ch_lookup[c] = self.ch_lookup[8]
        for c in range(1, ch[-1]):
            if c not in self.ch_lookup:
                self.ch_lookup[c] = self.ch_lookup[int(self.channels[np.argmin(abs(self.channels - c))])]
        self.sp_lookup = {int(c): i for i, c in enumerate(sp)}
        self.sp_lookup[2] = self.sp_lookup[3]        # 2x2 treated as 3x3
        for c in range(1, sp[-1]):
            if c not in self.sp_lookup:
                self.sp_lookup[c] = self.sp_lookup[int(self.spatial[np.argmin(abs(self.spatial - c))])]
        self.num_classes = num_classes

    def indices(self, n_rows, params_map, predict_class_layers=True):
        """(n_rows, 4) int32: [out-ch, in-ch, kh, kw] embedding rows; dummy row (last) where no shape."""
        out = np.empty((n_rows, 4), dtype=np.int32)
        out[:, :2] = self.n_ch
        out[:, 2:] = self.n_sp
        for node_ind, (info, key, _) in params_map.items():
            sz = info['sz']
            if sz is None:
                continue
            sz = tuple(int(v) for v in sz)
            if len(sz) == 1:
                sz = (sz[0], 1)
            if len(sz) == 2:
                sz = (sz[0], sz[1], 1, 1)
            if len(sz) == 3:
                if sz[0] == 1 and min(sz[1:]) > 1:
                    s_ = int(math.floor(sz[1] ** 0.5))
                    sz = (1, sz[2], s_, s_)
                else:
                    sz = (sz[0], sz[1], sz[2], 1)
            if not predict_class_layers and key in ('cls_w', 'cls_b'):
                sz = (self.num_classes,) + sz[1:]
            for i in range(4):
                if i < 2:
                    out[node_ind, i] = self.ch_lookup.get(sz[i], self.ch_lookup[int(self.channels[-1])])
                else:
                    out[node_ind, i] = self.sp_lookup.get(sz[i], self.sp_lookup[int(self.spatial[-1])])
        return out


# ---- target network enumeration ----------------------------------------------------------------
def _sz(p):
    return tuple(int(v) for v in p) if isinstance(p, (list, tuple)) else tuple(int(v) for v in p.shape)


def named_layered_modules(model):
    """param_name -> {'param_name','module','is_w','sz'} per cell."""
    if hasattr(model, 'module'):
        model = model.module
    layers = model._n_cells if hasattr(model, '_n_cells') else 1
    out = [{} for _ in range(layers)]
    cell_ind = 0
    for module_name, m in model.named_modules():
        cell_ind = m._cell_ind if hasattr(m, '_cell_ind') else cell_ind
        for attr, is_w, suffix in (('weight', True, '.weight'), ('bias', False, '.bias'),
                                   ('in_proj_weight', True, '.in_proj_weight'),
                                   ('in_proj_bias', False, '.in_proj_bias'),
                                   ('pos_embedding', True, '.pos_embedding.weight')):
            p = getattr(m, attr, None)
            if p is None:
                continue
            key = module_name + suffix
            out[cell_ind][key] = {'param_name': key, 'module': m, 'is_w': is_w, 'sz': _sz(p), 'attr': attr}
    return out


def target_attr(module, is_w):
    """nn.py:519-524."""
    if isinstance(module, nn.MultiheadAttention):
        return 'in_proj_weight' if is_w else 'in_proj_bias'
    if hasattr(module, 'pos_embedding') and not hasattr(module, 'weight'):
        return 'pos_embedding'
    return 'weight' if is_w else 'bias'


# ---- parameter groups ------------------------------------------------------------------------------
def group_key(sz, max_shape, last_weight, last_bias):
    """nn.py:652-675."""
    def min_sz(j):
        n = min(sz[j], max_shape[j])
        if n % 3 == 0:
            n = n // 3 * 4
        if n >= max_shape[j] / 2:
            n = max_shape[j]
        return n

    if len(sz) == 1:
        return (min_sz(0), -1) if last_bias else (min_sz(0), 0)
    if last_weight:
        return (min_sz(0), min_sz(1))
    if len(sz) == 2:
        return (min_sz(0), min_sz(1), 1, 1)
    if len(sz) == 3:
        if sz[0] == 1 and min(sz[1:]) > 1:
            s = int(math.floor(sz[1] ** 0.5))
            return (1, sz[2], s, s)
        return (min_sz(0), min_sz(1), min_sz(2))
    return (min_sz(0), min_sz(1), sz[2], sz[3])


def map_net_params(node_infos, n_nodes, nets, max_shape, reduce_graph=False):
    """
    nn.py:594-692.  node_infos: per graph, per cell, list of node tuples; n_nodes: python ints.
    Returns (param_groups: key -> [sparse-flat node index], params_map: index -> (info, key, pos)).
    """
    mapping, params_map = {}, {}
    offset = 0
    for b, (node_info, net) in enumerate(zip(node_infos, nets)):
        target_modules = net.__dict__['_layered_modules'] if hasattr(net, '_layered_modules') \
            else named_layered_modules(net)
        for cell_id in range(len(node_info)):
            for (node_ind, p_, name, sz, last_weight, last_bias) in node_info[cell_id]:
                p_name = p_ if p_.endswith(('.weight', '.bias', 'in_proj_weight', 'in_proj_bias')) else p_ + '.weight'
                matched = None
                for cand in (p_name, p_name.replace('to_qkv', 'attn.to_qkv').replace('to_out', 'attn.to_out')):
                    if cand in target_modules[cell_id]:
                        matched, param_name = target_modules[cell_id][cand], cand
                        break
                if matched is None:
                    if sz is not None:
                        params_map[offset + node_ind] = ({'sz': sz}, None, None)
                    continue
                key = group_key(tuple(matched['sz']), max_shape, last_weight, last_bias)
                mapping.setdefault(key, [])
                params_map[offset + node_ind] = (matched, key, len(mapping[key]))
                mapping[key].append(offset + node_ind)
                if reduce_graph:
                    del target_modules[cell_id][param_name]
            if reduce_graph:
                # nn.py:684-690: prune ops the graph does not reference (training-time speed-up)
                for m in target_modules[cell_id].values():
                    if m['is_w']:
                        m['module'].weight = None
                        if hasattr(m['module'], 'bias') and m['module'].bias is not None:
                            m['module'].bias = None
        offset += int(n_nodes[b])
    return mapping, params_map


# ---- normalisation rule (nn.py:554-592) -------------------------------------------------------------
def norm_rule(shape, is_w):
    """Returns (mode, scale) for a predicted tensor of the given target shape."""
    shape = tuple(int(v) for v in shape)
    if len(shape) > 1:
        if len(shape) > 2 and shape[2] >= 11 and shape[0] == 1:
            return 0, 1.0
        no_relu = len(shape) > 2 and (shape[1] == 1 or (len(shape) > 3 and shape[2] < shape[3]))
        beta = 1.0 if no_relu else 2.0
        fan_in = int(math.prod(shape[1:]))
        return 0, float((beta / fan_in) ** 0.5)
    return (1, 1.0) if is_w else (2, 1.0)
