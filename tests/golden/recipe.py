"""
Deterministic recipes shared by tests/golden/make_golden.py (dev container, reference present) and the
tests (anywhere).  Contains NO reference code: only seeded inputs (GHN weights, synthetic target
networks and their graphs) so that fixtures need to store expected outputs only.
"""

import zlib
import numpy as np
import torch
import torch.nn as nn

PRIMITIVES = ['max_pool', 'avg_pool', 'sep_conv', 'dil_conv', 'conv', 'msa', 'cse', 'sum',
              'concat', 'input', 'bias', 'bn', 'ln', 'pos_enc', 'glob_avg']
PRIM_ID = {n: i for i, n in enumerate(PRIMITIVES)}

VARIANTS = {  # name: (hid, layers, heads)   /root/reference/README.md:23, ghn3/nn.py:93
    'ghn3tm8': (64, 3, 8), 'ghn3sm8': (128, 5, 16), 'ghn3lm8': (256, 12, 16), 'ghn3xlm16': (384, 24, 16)}

TINY_CFG = dict(max_shape=(32, 32, 16, 16), num_classes=10, hid=32, heads=8, layers=2,
                weight_norm=True, ve=True, layernorm=True)
TINY_SEED = 4242


# ------------------------------------------------------------------------------------------------
# seeded GHN weights
# ------------------------------------------------------------------------------------------------

def seeded_tensor(name, shape, seed):
    rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
    v = rs.standard_normal(shape).astype(np.float32)
    leaf = name.split('.')[-1]
    is_norm = ('ln' in name.split('.')[-2]) if len(name.split('.')) > 1 else False
    if is_norm and leaf == 'weight':
        return (1.0 + 0.1 * v).astype(np.float32)
    if leaf == 'bias':
        return (0.05 * v).astype(np.float32)
    if len(shape) == 2 and ('embed' in name):
        return (shape[1] ** -0.5 * v).astype(np.float32)
    fan_in = shape[-1] if len(shape) >= 2 else shape[0]
    return (v / np.sqrt(fan_in)).astype(np.float32)


def seeded_state_dict(shapes, seed):
    """shapes: {name: shape}.  Returns {name: float32 ndarray}; depends only on (name, shape, seed)."""
    return {k: seeded_tensor(k, tuple(shapes[k]), seed) for k in sorted(shapes)}


def sample_indices(n, k, seed):
    rs = np.random.RandomState(seed)
    return np.sort(rs.randint(0, n, size=min(k, n)))


def count_params(hid, layers, heads, num_classes, s=16):
    C, K, H = hid, num_classes, heads
    ch = set([1, 3, K]) | set(range(8, 64, 8)) | set(range(64, 4096, 16)) | set(range(4096, 8193, 32))
    sp = set(range(1, max(12, s), 2)) | {14, 16}
    n = 2 * C + 15 * C + (len(sp) + 1) * (C // 4) + (len(ch) + 1) * (C // 4)
    n += layers * (12 * C * C + 10 * C)
    n += 257 * C + (2 * C * C + C) + (H * C + H) + 2 * 101 * C + 1001 * C
    n += (4 * C * s * s) * C + 4 * C * s * s + 8 * C * 4 * C + 8 * C + C * C * 8 * C + C * C + K * C + K
    n += 2 * C * C + 2 * C + 2 * C * 2 * C + 2 * C + K * C + K
    return n


# ------------------------------------------------------------------------------------------------
# graphs
# ------------------------------------------------------------------------------------------------

def spd_from_edges(n, edges, cutoff=50):
    """Directed shortest-path lengths (1-hop edges given), 0 = unreachable / self, values <= cutoff."""
    adj = [[] for _ in range(n)]
    for a, b in edges:
        adj[a].append(b)
    A = np.zeros((n, n), dtype=np.int64)
    for s in range(n):
        dist = {s: 0}
        frontier = [s]
        d = 0
        while frontier and d < cutoff:
            d += 1
            nxt = []
            for u in frontier:
                for v in adj[u]:
                    if v not in dist:
                        dist[v] = d
                        nxt.append(v)
            frontier = nxt
        for v, dv in dist.items():
            if dv > 0:
                A[s, v] = dv
    return A


def random_dag_spd(n, seed, cutoff=50):
    rs = np.random.RandomState(seed)
    edges = [(i, i + 1) for i in range(n - 1)]
    for i in range(2, n):
        if rs.rand() < 0.3:
            edges.append((int(rs.randint(0, i - 1)), i))
    return spd_from_edges(n, edges, cutoff)


# ------------------------------------------------------------------------------------------------
# tiny target networks: (primitive, param_name, shape) per node + extra skip edges
# ------------------------------------------------------------------------------------------------

TINY_NETS = [
    dict(nodes=[('input', None, None),
                ('conv', 'stem.weight', (16, 3, 7, 7)),
                ('bn', 'bn1.weight', (16,)),
                ('conv', 'c2.weight', (24, 16, 3, 3)),
                ('bias', 'c2.bias', (24,)),
                ('conv', 'c3.weight', (8, 24, 1, 1)),
                ('conv', 'c4.weight', (40, 8, 5, 5)),
                ('sum', None, None),
                ('conv', 'c5.weight', (12, 48, 3, 3)),
                ('dil_conv', 'dw.weight', (16, 1, 3, 3)),
                ('conv', 'asym.weight', (8, 8, 1, 3)),
                ('ln', 'ln.weight', (12,)),
                ('conv', 'lin.weight', (20, 12)),
                ('bias', 'lin.bias', (20,)),
                ('pos_enc', 'enc.pos_embedding', (1, 17, 12)),
                ('glob_avg', None, None),
                ('conv', 'fc.weight', (10, 20)),
                ('bias', 'fc.bias', (10,))],
         skips=[(3, 7), (1, 5), (9, 14)]),
    dict(nodes=[('input', None, None),
                ('conv', 'stem.weight', (32, 3, 3, 3)),
                ('bn', 'bn1.weight', (32,)),
                ('conv', 'c2.weight', (64, 32, 3, 3)),
                ('sum', None, None),
                ('conv', 'c3.weight', (32, 64, 1, 1)),
                ('glob_avg', None, None),
                ('conv', 'fc.weight', (10, 32)),
                ('bias', 'fc.bias', (10,))],
         skips=[(1, 4)]),
]
TINY_CASES = {'b1': [0], 'b2': [1, 0], 'b2r': [0, 1]}

# Extra fixtures (ghn3_tiny_extra.npz): kernels larger than the 16x16 decoder grid (bilinear branch of nn.py:751-753,
# e.g. the 32x32 patch embedding of ViT-B/32) and the weight_norm=False / layernorm=False configurations.
EXTRA_NETS = [
    dict(nodes=[('input', None, None),
                ('conv', 'patch.weight', (24, 3, 20, 20)),
                ('bias', 'patch.bias', (24,)),
                ('ln', 'ln.weight', (24,)),
                ('conv', 'big2.weight', (8, 24, 18, 18)),
                ('conv', 'c3.weight', (24, 8, 3, 3)),
                ('sum', None, None),
                ('glob_avg', None, None),
                ('conv', 'fc.weight', (10, 24)),
                ('bias', 'fc.bias', (10,))],
         skips=[(2, 6)]),
]
# Degenerate batches (no golden file: HIP path / host compiler against the oracle only): a network without any 2-D / 4-D
# weight (the decoder GEMMs have zero rows), a single 1x1 convolution, three graphs of very different sizes in one batch.
EDGE_NETS = {
    'only1d': dict(nodes=[('input', None, None), ('bn', 'bn.weight', (8,)), ('bias', 'bn.bias', (8,)),
                          ('ln', 'ln.weight', (8,)), ('bias', 'ln.bias', (8,)), ('glob_avg', None, None)]),
    'single': dict(nodes=[('input', None, None), ('conv', 'c.weight', (4, 3, 1, 1)), ('glob_avg', None, None)]),
}
EDGE_CASES = {'only1d': ['only1d'], 'single': ['single'], 'ragged3': [1, 'single', 0], 'mixed1d': ['only1d', 1]}


def edge_specs(case):
    return [EDGE_NETS[k] if isinstance(k, str) else TINY_NETS[k] for k in EDGE_CASES[case]]


EXTRA_CASES = {  # name: (spec list, overrides of TINY_CFG)
    'big': ([EXTRA_NETS[0]], {}),
    'big_b2': ([TINY_NETS[1], EXTRA_NETS[0]], {}),
    'nonorm': ([TINY_NETS[0]], dict(weight_norm=False)),
    'noln': ([TINY_NETS[0]], dict(layernorm=False)),
}


def graph_arrays(spec):
    """(node_feat (N,1) int64, node_info [[(ind, param_name, prim, sz, last_w, last_b)...]], A (N,N) int64)."""
    nodes = spec['nodes']
    n = len(nodes)
    node_feat = np.asarray([[PRIM_ID[p]] for p, _, _ in nodes], dtype=np.int64)
    info = []
    for i, (prim, pname, sz) in enumerate(nodes):
        if pname is None and prim.find('pool') < 0:
            continue
        info.append((i, pname if pname is not None else prim, prim, sz,
                     i == n - 2 and pname is not None and pname.endswith('.weight'),
                     i == n - 1 and pname is not None and pname.endswith('.bias')))
    if 'edges' in spec:
        edges = list(spec['edges'])
    else:
        edges = [(i, i + 1) for i in range(n - 1)] + list(spec.get('skips', []))
    return node_feat, [info], spd_from_edges(n, edges, 50)


class _Bag(nn.Module):
    pass


def build_torch_net(spec, encoder_cls=None):
    """An nn.Module whose sub-module names match the param_name fields (no forward needed)."""
    net = _Bag()
    have_bias = {p[:-5] for _, p, _ in spec['nodes'] if p is not None and p.endswith('.bias')}
    for prim, pname, sz in spec['nodes']:
        if pname is None or pname.endswith('.bias') or pname.endswith('.in_proj_bias'):
            continue
        if pname.endswith('.in_proj_weight'):           # torch.nn.MultiheadAttention (torchvision ViT encoder block)
            net.add_module(pname.rsplit('.', 1)[0], nn.MultiheadAttention(sz[1], spec.get('heads', 12)))
            continue
        if pname.endswith('.out_proj.weight'):          # child of the MultiheadAttention module above
            continue
        mname = pname.rsplit('.', 1)[0]
        if prim == 'bn':
            m = nn.BatchNorm2d(sz[0])
        elif prim == 'ln':
            m = nn.LayerNorm(sz[0])
        elif prim == 'pos_enc':
            base = encoder_cls if encoder_cls is not None else nn.Module

            class _Enc(base):
                def __init__(self):
                    nn.Module.__init__(self)
                    self.pos_embedding = nn.Parameter(torch.zeros(*sz))
            m = _Enc()
        elif len(sz) == 2:
            m = nn.Linear(sz[1], sz[0], bias=mname in have_bias)
        else:
            groups = sz[0] if sz[1] == 1 else 1
            m = nn.Conv2d(sz[1] * groups, sz[0], (sz[2], sz[3]), groups=groups, bias=mname in have_bias)
            assert tuple(m.weight.shape) == tuple(sz), (m.weight.shape, sz)
        net.add_module(mname, m)
    return net


# ------------------------------------------------------------------------------------------------
# torchvision-shaped ResNets (BASELINE configs 1 and 4): layer shapes and topology of torchvision.models.resnet18 /
# resnet50 written out by hand (torchvision is not installed); 53 / 127 graph nodes, 11,689,512 / 25,557,032 params.
# ------------------------------------------------------------------------------------------------

def resnet_spec(depth):
    assert depth in (18, 50)
    bottleneck = depth == 50
    blocks = [2, 2, 2, 2] if depth == 18 else [3, 4, 6, 3]
    exp = 4 if bottleneck else 1
    nodes, edges = [('input', None, None)], []

    def add(prim, name, shape, srcs):
        nodes.append((prim, name, shape))
        for s_ in srcs:
            edges.append((s_, len(nodes) - 1))
        return len(nodes) - 1

    x = add('conv', 'conv1.weight', (64, 3, 7, 7), [0])
    x = add('bn', 'bn1.weight', (64,), [x])
    x = add('max_pool', None, None, [x])
    cin = 64
    for li, (nb, planes) in enumerate(zip(blocks, (64, 128, 256, 512))):
        for b in range(nb):
            pre = 'layer%d_%d_' % (li + 1, b)
            y = x
            if bottleneck:
                convs = [(planes, cin, 1, 1), (planes, planes, 3, 3), (planes * 4, planes, 1, 1)]
            else:
                convs = [(planes, cin, 3, 3), (planes, planes, 3, 3)]
            for k, sz in enumerate(convs):
                y = add('conv', '%sconv%d.weight' % (pre, k + 1), sz, [y])
                y = add('bn', '%sbn%d.weight' % (pre, k + 1), (sz[0],), [y])
            srcs = [y]
            if b == 0 and (li > 0 or bottleneck):
                d = add('conv', pre + 'downsample_0.weight', (planes * exp, cin, 1, 1), [x])
                d = add('bn', pre + 'downsample_1.weight', (planes * exp,), [d])
                srcs.append(d)
            else:
                srcs.append(x)
            x = add('sum', None, None, srcs)
            cin = planes * exp
    x = add('glob_avg', None, None, [x])
    x = add('conv', 'fc.weight', (1000, cin), [x])
    add('bias', 'fc.bias', (1000,), [x])
    return dict(nodes=nodes, edges=edges)


def vit_b16_spec():
    """torchvision.models.vit_b_16 layer shapes (hidden 768, 12 layers, 12 heads, mlp 3072, 16x16 patches on 224^2).
    The class token is a bare nn.Parameter of the top module and is not a graph node."""
    D, L_, H = 768, 12, 12
    nodes, edges = [('input', None, None)], []

    def add(prim, name, shape, srcs):
        nodes.append((prim, name, shape))
        for s_ in srcs:
            edges.append((s_, len(nodes) - 1))
        return len(nodes) - 1

    x = add('conv', 'conv_proj.weight', (D, 3, 16, 16), [0])
    x = add('bias', 'conv_proj.bias', (D,), [x])
    x = add('pos_enc', 'encoder.pos_embedding', (1, 197, D), [x])
    for l in range(L_):
        pre = 'enc%d_' % l
        y = add('ln', pre + 'ln_1.weight', (D,), [x])
        y = add('conv', pre + 'self_attention.in_proj_weight', (3 * D, D), [y])
        y = add('bias', pre + 'self_attention.in_proj_bias', (3 * D,), [y])
        y = add('msa', None, None, [y])
        y = add('conv', pre + 'self_attention.out_proj.weight', (D, D), [y])
        y = add('bias', pre + 'self_attention.out_proj.bias', (D,), [y])
        x = add('sum', None, None, [x, y])
        y = add('ln', pre + 'ln_2.weight', (D,), [x])
        y = add('conv', pre + 'mlp_0.weight', (4 * D, D), [y])
        y = add('bias', pre + 'mlp_0.bias', (4 * D,), [y])
        y = add('conv', pre + 'mlp_3.weight', (D, 4 * D), [y])
        y = add('bias', pre + 'mlp_3.bias', (D,), [y])
        x = add('sum', None, None, [x, y])
    x = add('ln', 'encoder_ln.weight', (D,), [x])
    x = add('conv', 'heads_head.weight', (1000, D), [x])
    add('bias', 'heads_head.bias', (1000,), [x])
    return dict(nodes=nodes, edges=edges, heads=H)


RESNET_SEED = 777
RESNET_SAMPLES = 2048


def named_predicted(net):
    for mname, m in net.named_modules():
        if mname == '':
            continue
        for attr in ('weight', 'bias', 'pos_embedding', 'in_proj_weight', 'in_proj_bias'):
            t = m.__dict__.get(attr, None)
            if t is None:
                t = m._parameters.get(attr, None)
            if t is not None:
                yield '%s.%s' % (mname, attr), t


# (source tile shape, target shape) pairs exercising every branch of ghn3/nn.py:422-506
TILE_CASES = [
    ((32,), (20,)), ((32,), (70,)), ((2, 32), (24,)), ((32, 32, 3, 3), (50,)),
    ((10, 32), (10, 20)), ((10, 32), (25, 70)), ((32, 32, 1, 1), (20, 12)), ((32, 8, 1, 1), (48, 20)),
    ((64, 1, 1), (20, 1, 1)), ((64, 1, 1), (40, 1, 3)),
    ((1, 12, 4, 4), (1, 17, 12)),
    ((32, 4, 7, 7), (16, 3, 7, 7)), ((32, 32, 3, 3), (24, 16, 3, 3)), ((32, 32, 3, 3), (70, 48, 3, 3)),
    ((32, 8, 5, 5), (40, 8, 5, 5)), ((32, 4, 3, 3), (16, 1, 3, 3)), ((8, 8, 1, 3), (8, 8, 1, 3)),
    ((8, 12, 16, 16), (8, 8, 16, 16)), ((32, 32, 5, 5), (20, 20, 3, 3)), ((16, 16), (8, 40, 1, 1)),
    ((1, 32, 14, 14), (1, 32, 14, 14)), ((32, 32, 4, 4), (16, 40, 4, 4)), ((32, 32, 2, 2), (48, 16, 2, 2)),
]


# ------------------------------------------------------------------------------------------------
# seeded target-network parameters / images (tests/golden/networks.npz)
# ------------------------------------------------------------------------------------------------

def seeded_net_params(named_shapes, seed):
    """[(name, shape)] -> {name: float32 ndarray}: 1-D weights of norm layers near one, biases small, everything else
    He-like; depends only on (name, shape, seed)."""
    out = {}
    for name, shape in named_shapes:
        rs = np.random.RandomState((seed * 7919 + zlib.crc32(name.encode())) % (2 ** 31 - 1))
        v = rs.standard_normal(shape).astype(np.float32)
        if name.endswith('.bias'):
            out[name] = (0.1 * v).astype(np.float32)
        elif len(shape) == 1:
            out[name] = (1.0 + 0.2 * v).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = (v * np.sqrt(2.0 / fan_in)).astype(np.float32)
    return out


def seeded_images(shape, seed):
    return np.random.RandomState(seed).standard_normal(shape).astype(np.float32)
