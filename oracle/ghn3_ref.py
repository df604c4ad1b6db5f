"""
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain-PyTorch CPU restatement of the GHN-3 parameter-prediction path.
Follows /root/reference/ghn3/nn.py and the dense-batch half of ghn3/graph.py:

  GraphBatchRef          graph.py:38-88,155-185,243-269   (dense=True path only)
  GHN3Ref.__init__       nn.py:140-184                    (module tree / state-dict names)
  GHN3Ref.forward        nn.py:186-349
  map_net_params         nn.py:594-692
  conv_decoder3          nn.py:735-762
  tile_params            nn.py:422-506
  normalize              nn.py:554-592
  set_params             nn.py:508-552

The ppuda-defined parts come from oracle/ppuda_base.py (PARITY UNPINNED); everything nn.py defines is
pinned by golden vectors generated from the reference's own GHN3 class (tests/golden/make_golden.py).

Quirk Q1: after the Graphormer the rows are dense-flat (b*N_max + i) but the group
indices are sparse-flat (sum_{b'<b} n_b' + i), nn.py:259-260,275,615.  ``index_mode='reference'``
reproduces that; ``index_mode='correct'`` gathers the rows the indices were meant to address.
"""

import math
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ppuda_base
from . import graphormer_ref as G


# ----------------------------------------------------------------------------------------------
# Graph containers (dense path only)
# ----------------------------------------------------------------------------------------------

class GraphRef:
    """graph.py:336-352 -- the explicit (node_feat, node_info, A) constructor with dense=True."""

    def __init__(self, node_feat, node_info, A, net_args=None, net_idx=None):
        self.n_nodes = len(node_feat)
        self.node_feat = node_feat          # (N,1) int64 primitive ids
        self.node_info = node_info          # list (per cell) of (node_ind, param_name, name, sz, last_w, last_b)
        self._Adj = A                       # (N,N) int64 shortest-path lengths (0 = no edge)
        self.net_args = net_args
        self.net_idx = net_idx


class GraphBatchRef:
    """graph.py:48-88 (append) and 243-269 (_cat) for dense=True."""

    def __init__(self, graphs):
        self.graphs = graphs
        self.dense = True
        self.n_nodes = torch.tensor([len(g.node_feat) for g in graphs], dtype=torch.long)
        self.node_info = [g.node_info for g in graphs]
        self.net_args = [g.net_args for g in graphs]
        m = int(self.n_nodes.max())
        B = len(graphs)
        self.mask = torch.zeros(B, m, 1, dtype=torch.bool)
        self.node_feat = torch.zeros(B, m, 1, dtype=torch.long)
        self.edges = torch.zeros(B, m, m, dtype=torch.long)
        for b, g in enumerate(graphs):
            n = len(g.node_feat)
            self.node_feat[b, :n] = g.node_feat
            self.edges[b, :n, :n] = g._Adj
            self.mask[b, :n] = True

    def to_sparse(self, x):                       # graph.py:183-185
        return torch.cat([x[b, :self.n_nodes[b]] for b in range(len(self.n_nodes))])

    def to_dense(self, x):                        # graph.py:172-181
        B, M, C = len(self.n_nodes), int(self.n_nodes.max()), x.shape[-1]
        out = torch.zeros(B, M, C, dtype=x.dtype)
        off = 0
        for b in range(B):
            n = int(self.n_nodes[b])
            out[b, :n] = x[off:off + n]
            off += n
        return out

    def __len__(self):
        return len(self.n_nodes)


# ----------------------------------------------------------------------------------------------
# Host-side bookkeeping  (nn.py:594-692)
# ----------------------------------------------------------------------------------------------

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
            s = int(np.floor(sz[1] ** 0.5))
            return (1, sz[2], s, s)
        return (min_sz(0), min_sz(1), min_sz(2))
    return (min_sz(0), min_sz(1), sz[2], sz[3])


def map_net_params(graphs, nets, max_shape):
    """nn.py:594-692 without the reduce_graph pruning side effect (684-690) and the debug checks."""
    mapping, params_map = {}, {}
    for b, (node_info, net) in enumerate(zip(graphs.node_info, nets)):
        target_modules = net.__dict__['_layered_modules'] if hasattr(net, '_layered_modules') \
            else ppuda_base.named_layered_modules(net)
        param_ind = int(torch.sum(graphs.n_nodes[:b]).item())
        for cell_id in range(len(node_info)):
            for (node_ind, p_, name, sz, last_weight, last_bias) in node_info[cell_id]:
                p_name = p_ if p_.endswith(('.weight', '.bias', 'in_proj_weight', 'in_proj_bias')) else p_ + '.weight'
                matched = None
                for param_name in [p_name, p_name.replace('to_qkv', 'attn.to_qkv').replace('to_out', 'attn.to_out')]:
                    if param_name in target_modules[cell_id]:
                        matched = target_modules[cell_id][param_name]
                        break
                if matched is None:
                    if sz is not None:
                        params_map[param_ind + node_ind] = ({'sz': sz}, None, None)
                    continue
                key = group_key(tuple(matched['sz']), max_shape, last_weight, last_bias)
                mapping.setdefault(key, [])
                params_map[param_ind + node_ind] = (matched, key, len(mapping[key]))
                mapping[key].append(param_ind + node_ind)
    return mapping, params_map


# ----------------------------------------------------------------------------------------------
# Decoders and tiling
# ----------------------------------------------------------------------------------------------

def conv_decoder3(x, p, out_shape, max_shape, class_pred):
    """nn.py:735-762 (GHN-3 branch).  x (n,C) -> (n,o,i,h,w) or (n,K,i) for class_pred."""
    n = x.shape[0]
    t = F.relu(F.linear(x, p['decoder.fc.0.weight'], p['decoder.fc.0.bias'])).view(n, -1, *out_shape[2:])
    off = out_shape[2] // 2
    t = t[:, :,
          max(0, off - max_shape[2] // 2): off + int(math.ceil(max_shape[2] / 2)),
          max(0, off - max_shape[3] // 2): off + int(math.ceil(max_shape[3] / 2))]
    oshape = (out_shape[0], out_shape[1], min(out_shape[2], max_shape[2]), min(out_shape[3], max_shape[3]))
    t = t.permute(0, 2, 3, 1)
    t = F.relu(F.linear(t, p['decoder.conv.0.weight'], p['decoder.conv.0.bias']))
    t = F.linear(t, p['decoder.conv.2.weight'], p['decoder.conv.2.bias']).permute(0, 3, 1, 2)
    t = t.reshape(n, oshape[0], oshape[1], oshape[2], oshape[3])
    t = t[:, :, :max_shape[1], :max_shape[2], :max_shape[3]]
    if min(max_shape[2:]) > min(oshape[2:]):
        assert t.shape[0] == 1, t.shape
        t = F.interpolate(t[0], max_shape[2:], mode='bilinear').unsqueeze(0)
    if class_pred:
        assert t.shape[-2] == t.shape[-1]
        k = t.shape[-1] // 2
        c = F.relu(t[:, :, :, k, k].permute(0, 2, 1))
        t = F.linear(c, p['decoder.class_layer_predictor.1.weight'],
                     p['decoder.class_layer_predictor.1.bias']).permute(0, 2, 1)
    else:
        t = t[:, :max_shape[0]]
    return t


def tile_params(w, target_shape, gen=None):
    """nn.py:422-506 (GHN-3 branch: centre crop)."""
    t, s = tuple(target_shape), tuple(w.shape)
    if len(t) == 1:
        if len(s) == 1:
            w = w[:min(t[0], s[0])]
        elif len(s) == 2:
            w = w[:min(t[0], s[0]), 0]
        else:
            w = w[:min(t[0], s[0]), 0, s[-2] // 2, s[-1] // 2]
    elif len(t) == 2:
        if len(s) == 2:
            w = w[:min(t[0], s[0]), :min(t[1], s[1])]
        else:
            w = w[:min(t[0], s[0]), :min(t[1], s[1]), s[-2] // 2, s[-1] // 2]
    elif len(t) == 3:
        if len(s) == 3:
            w = w[:min(t[0], s[0]), :min(t[1], s[1]), :min(t[2], s[2])]
        else:
            w = w.reshape(*s[:2], -1).permute(0, 2, 1)
            w = w[:min(t[0], w.shape[0]), :min(t[1], w.shape[1]), :min(t[2], w.shape[2])]
            tok = torch.normal(mean=0, std=0.02, size=(1, 1, w.shape[2]), generator=gen)   # Q3: random row
            w = torch.cat((tok.to(w.dtype), w), dim=1)
    else:
        s2 = min(t[2], s[2]) if len(s) > 2 else 1
        s3 = min(t[3], s[3]) if len(s) > 3 else 1
        if len(s) > 2:
            o2, o3 = s[-2] // 2, s[-1] // 2
            w = w[:min(t[0], s[0]), :min(t[1], s[1]),
                  o2 - s2 // 2: o2 + int(math.ceil(s2 / 2)),
                  o3 - s3 // 2: o3 + int(math.ceil(s3 / 2))]
        else:
            w = w[:min(t[0], s[0]), :min(t[1], s[1])].unsqueeze(2).unsqueeze(3)
    s = tuple(w.shape)
    assert len(s) == len(t), (s, t)
    if t[0] > s[0]:
        reps = int(math.ceil(t[0] / s[0]))
        w = w.repeat((reps,) + (1,) * (len(t) - 1))[:t[0]]
    if len(t) > 1:
        if t[1] > s[1]:
            reps = int(math.ceil(t[1] / s[1]))
            w = w.repeat((1, reps) + (1,) * (len(t) - 2))[:, :t[1]]
        elif len(t) == 3 and len(s) == 3 and t[2] > s[2]:
            reps = int(math.ceil(t[2] / s[2]))
            w = w.repeat((1, 1, reps))[:, :, :t[2]]
    if len(t) == 1:
        w = w[:t[0]]
    elif len(t) == 2:
        w = w[:t[0], :t[1]]
    elif len(t) == 3:
        w = w[:t[0], :t[1], :t[2]]
    else:
        o2, o3 = w.shape[-2] // 2, w.shape[-1] // 2
        w = w[:t[0], :t[1],
              o2 - t[2] // 2: o2 + int(math.ceil(t[2] / 2)),
              o3 - t[3] // 2: o3 + int(math.ceil(t[3] / 2))]
    return w


def normalize(p, is_w):
    """nn.py:554-592."""
    if p.dim() > 1:
        sz = p.shape
        if len(sz) > 2 and sz[2] >= 11 and sz[0] == 1:
            return p
        no_relu = len(sz) > 2 and (sz[1] == 1 or sz[2] < sz[3])
        beta = 1. if no_relu else 2.
        return p * (beta / p[0].numel()) ** 0.5
    if is_w:
        return 2 * torch.sigmoid(0.5 * p)
    return torch.tanh(0.2 * p)


def target_attr(module, is_w):
    """nn.py:519-524 (isinstance checks replaced by attribute checks: torchvision is absent)."""
    if isinstance(module, nn.MultiheadAttention):
        return 'in_proj_weight' if is_w else 'in_proj_bias'
    if hasattr(module, 'pos_embedding') and not hasattr(module, 'weight'):
        return 'pos_embedding'
    return 'weight' if is_w else 'bias'


# ----------------------------------------------------------------------------------------------
# The model
# ----------------------------------------------------------------------------------------------

class _Layer(nn.Module):
    """Parameter container with the reference's names for one Graphormer layer (graphormer.py:168-206)."""

    def __init__(self, dim, heads, layer0, mlp_ratio=4):
        super().__init__()
        self.ln1 = nn.LayerNorm(dim)
        attn = nn.Module()
        attn.to_qkv = nn.Linear(dim, 3 * dim, bias=False)
        attn.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Identity())
        if layer0:
            ee = nn.Module()
            ee.embed = nn.Embedding(257, dim)
            attn.edge_embed = ee
            attn.proj_e = nn.Sequential(nn.Linear(2 * dim, dim), nn.ReLU(), nn.Linear(dim, heads))
        self.attn = attn
        self.ln2 = nn.LayerNorm(dim)
        ff = nn.Module()
        ff.net = nn.Sequential(nn.Linear(dim, mlp_ratio * dim), nn.GELU(), nn.Identity(),
                               nn.Linear(mlp_ratio * dim, dim), nn.Identity())
        self.ff = ff
        if layer0:
            self.centrality_embed_in = nn.Embedding(G.MAX_DEGREE + 1, dim)
            self.centrality_embed_out = nn.Embedding(G.MAX_DEGREE + 1, dim)
            self.input_dist_embed = nn.Embedding(G.MAX_INPUT_DIST + 1, dim)


class GHN3Ref(ppuda_base.GHN):

    def __init__(self, max_shape, num_classes, hid, heads=8, layers=3, index_mode='reference', **kwargs):
        kwargs.pop('is_ghn2', None)
        kwargs.pop('pretrained', None)
        super().__init__(max_shape, num_classes, hid=hid, **kwargs)
        self.heads, self.layers, self.hid = heads, layers, hid
        self.index_mode = index_mode
        self.gnn = nn.Sequential(*[_Layer(hid, heads, layer0=(l == 0)) for l in range(layers)])
        dec = self.decoder
        dec.conv[0] = nn.Linear(hid * 4, hid * 8)                       # nn.py:727-729
        dec.conv[2] = nn.Linear(hid * 8, max_shape[0] * max_shape[1])
        dec.class_layer_predictor[1] = nn.Linear(max_shape[0], num_classes)   # nn.py:730-733
        # fresh-init policy of nn.py:167-170,694-713 (not part of parity: tests load explicit weights)
        for m in (self.decoder_1d.fc[-2], dec.conv[-2], dec.class_layer_predictor[-1]):
            m.weight.data /= 5.0
            m.bias.data *= 0
        for m in self.modules():
            if isinstance(m, nn.Embedding):
                nn.init.trunc_normal_(m.weight.data, std=m.weight.shape[1] ** (-0.5))

    # -- forward pieces, each returns plain tensors so tests can pin them individually ------------

    def node_embeddings(self, graphs, params_map, predict_class_layers=True):
        """nn.py:248-249."""
        x = graphs.to_sparse(graphs.node_feat)[:, 0]
        return self.shape_enc(self.embed(x), params_map, predict_class_layers=predict_class_layers)

    def graphormer(self, x_sparse, graphs, return_bias=False):
        """nn.py:251-263: to_dense, pair mask, L layers, flatten, final LN."""
        p = dict(self.named_parameters())
        x = graphs.to_dense(x_sparse)
        mask = graphs.mask & graphs.mask.permute(0, 2, 1)
        bias = graphs.edges
        for l in range(self.layers):
            x, bias = G.transformer_layer(x, bias, mask, p, 'gnn.%d.' % l, self.heads, layer0=(l == 0))
        x = x.reshape(-1, x.shape[-1])
        if self.layernorm:
            x = F.layer_norm(x, (x.shape[-1],), p['ln.weight'], p['ln.bias'], 1e-5)
        return (x, bias) if return_bias else x

    def gather_rows(self, x, inds, graphs):
        """nn.py:275 (reference = Q1 bug-compatible) or the intended rows."""
        inds = torch.as_tensor(inds, dtype=torch.long)
        if self.index_mode == 'reference':
            return x[inds]
        n_max = int(graphs.n_nodes.max())
        offs = torch.cumsum(graphs.n_nodes, 0) - graphs.n_nodes
        b = torch.searchsorted(torch.cumsum(graphs.n_nodes, 0), inds, right=True)
        return x[b * n_max + (inds - offs[b])]

    def decode_group(self, key, x_):
        """nn.py:277-299.  Returns (w, is_cls)."""
        p = dict(self.named_parameters())
        sz = key
        n = x_.shape[0]
        if len(sz) in (2, 3):
            if len(sz) == 2 and sz[1] > 0:
                return conv_decoder3(x_, p, self.max_shape, (sz[0], sz[1], 1, 1), class_pred=True), True
            h = F.relu(F.linear(x_, p['decoder_1d.fc.0.weight'], p['decoder_1d.fc.0.bias']))
            w = F.linear(h, p['decoder_1d.fc.2.weight'], p['decoder_1d.fc.2.bias'])
            if len(sz) == 3:
                return w.view(n, -1, 1, 1), False
            w = w.view(n, 2, -1)
            if sz[1] < 0:
                return F.linear(F.relu(w), p['bias_class.1.weight'], p['bias_class.1.bias']), True
            return w, False
        assert len(sz) == 4, sz
        return conv_decoder3(x_, p, self.max_shape, sz, class_pred=False), False

    def forward(self, nets, graphs, return_embeddings=False, predict_class_layers=True, keep_grads=True,
                assign=True, gen=None):
        """
        nn.py:186-349.  ``nets``: list of target containers (see ppuda_base.named_layered_modules).
        Returns (nets, predicted) where ``predicted`` is the ordered list of
        (graph-global node index, attribute name, module, tensor) in the reference's assignment order.
        """
        is_lst = isinstance(nets, (list, tuple))
        if not is_lst:
            nets = [nets]
        param_groups, params_map = map_net_params(graphs, nets, self.max_shape)
        x = self.node_embeddings(graphs, params_map, predict_class_layers)
        x = self.graphormer(x, graphs)
        predicted = []
        for key, inds in param_groups.items():
            if len(inds) == 0:
                continue
            w, is_cls = self.decode_group(key, self.gather_rows(x, inds, graphs).float())
            if not predict_class_layers and is_cls:
                continue
            for ind in inds:
                matched, _, w_ind = params_map[ind]
                if w_ind is None:
                    continue
                m, sz, is_w = matched['module'], matched['sz'], matched['is_w']
                for it in range(2 if (len(sz) == 1 and is_w) else 1):
                    w_ = w[w_ind][1 - is_w + it] if len(sz) == 1 else w[w_ind]
                    t = tile_params(w_, sz, gen=gen)
                    w_flag = bool(is_w) and not it
                    if self.weight_norm:
                        t = normalize(t, w_flag)
                    attr = target_attr(m, w_flag)
                    if assign:
                        set_params(m, attr, t, keep_grads)
                    predicted.append((ind, attr, m, t))
        out = nets if is_lst else nets[0]
        return (out, predicted, x) if return_embeddings else (out, predicted)


def set_params(module, key, tensor, keep_grads):
    """nn.py:525-552."""
    target = getattr(module, key)
    sz_target = tuple(target) if isinstance(target, (list, tuple)) else tuple(target.shape)
    if len(sz_target) == 4 and tensor.dim() == 2:
        tensor = tensor.unsqueeze(2).unsqueeze(3)
    if keep_grads:
        if isinstance(target, (list, tuple)) or not isinstance(module, nn.Module):
            setattr(module, key, tensor)         # light modules (shape lists, light_ops.py:236-240)
        else:
            module.__dict__[key] = tensor
            module._parameters[key] = tensor
    else:
        assert isinstance(target, nn.Parameter), type(target)
        target.data = tensor.clone()
    assert sz_target == tuple(getattr(module, key).shape), (sz_target, tensor.shape, key)
    return sz_target
