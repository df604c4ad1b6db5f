"""
Seeded synthetic computational graphs + shape-only target networks (SURVEY 8(d) generator).

The DeepNets-1M hdf5 files, h5py and ppuda's loader are not available offline, so benchmarks and tests use
graphs of the same statistical shape: a chain of conv / bn / sum nodes with skip edges into every sum node,
a global-average-pool node and an ImageNet classifier head; A holds directed shortest-path lengths cut at 50
(the "virtual edges" of /root/reference/ghn3/graph.py:755-798).
"""

import numpy as np

from .bookkeeping import PRIMITIVES_DEEPNETS1M
from .graph import Graph, GraphBatch

PRIM_ID = {n: i for i, n in enumerate(PRIMITIVES_DEEPNETS1M)}


class LightParam:
    """Shape-only stand-in for a layer (the light-module contract of ghn3/light_ops.py:236-240: ``weight`` /
    ``bias`` are shape lists until the GHN assigns tensors with ``module.weight = tensor``)."""

    def __init__(self, weight=None, bias=None):
        self.weight = list(weight) if weight is not None else None
        self.bias = list(bias) if bias is not None else None


class LightNet:
    """Bag of LightParam layers exposing the ``_layered_modules`` table GHN3.forward looks for (nn.py:612)."""

    def __init__(self):
        self.layers = {}
        self._layered_modules = [{}]

    def add(self, name, weight=None, bias=None):
        m = LightParam(weight, bias)
        self.layers[name] = m
        if weight is not None:
            k = name + '.weight'
            self._layered_modules[0][k] = {'param_name': k, 'module': m, 'is_w': True, 'sz': tuple(weight)}
        if bias is not None:
            k = name + '.bias'
            self._layered_modules[0][k] = {'param_name': k, 'module': m, 'is_w': False, 'sz': tuple(bias)}
        return m

    def parameters(self):
        for m in self.layers.values():
            for t in (m.weight, m.bias):
                if t is not None and not isinstance(t, list):
                    yield t

    def num_params(self):
        n = 0
        for tab in self._layered_modules:
            for e in tab.values():
                n += int(np.prod(e['sz']))
        return n


def shortest_paths(n, edges, cutoff=50):
    """A[s][v] = length of the shortest directed path s -> v when it is in 1..cutoff, else 0 (what the reference takes from
    networkx's all_pairs_shortest_path_length(cutoff=50), graph.py: the Graphormer's distance feature)."""
    try:
        from scipy.sparse import csr_matrix
        from scipy.sparse.csgraph import dijkstra
    except ImportError:                                    # pragma: no cover  (pure-Python breadth-first search below)
        dijkstra = None
    if dijkstra is not None and len(edges):
        e = np.asarray(edges, dtype=np.int64)
        g = csr_matrix((np.ones(len(e), dtype=np.int8), (e[:, 0], e[:, 1])), shape=(n, n))
        d = dijkstra(g, directed=True, unweighted=True, limit=cutoff)
        return np.where(np.isfinite(d) & (d > 0), d, 0).astype(np.int64)
    adj = [[] for _ in range(n)]
    for a, b in edges:
        adj[a].append(b)
    A = np.zeros((n, n), dtype=np.int64)
    for s in range(n):
        dist = np.full(n, -1, dtype=np.int64)
        dist[s] = 0
        frontier, d = [s], 0
        while frontier and d < cutoff:
            d += 1
            nxt = []
            for u in frontier:
                for v in adj[u]:
                    if dist[v] < 0:
                        dist[v] = d
                        nxt.append(v)
            frontier = nxt
        reach = dist > 0
        A[s, reach] = dist[reach]
    return A


def synthetic_graph(n_nodes, seed, num_classes=1000, channels=(32, 64, 96, 128, 192, 256, 384, 512),
                    kernels=(1, 1, 3, 3, 3, 5, 7)):
    """Returns (Graph, LightNet).  Deterministic in (n_nodes, seed)."""
    assert n_nodes >= 6
    rs = np.random.RandomState(seed)
    net = LightNet()
    prims, info, edges = ['input'], [], []
    c_prev = 3
    have_conv = False
    for i in range(1, n_nodes - 3):
        u = rs.rand()
        kind = 'conv' if (u < 0.45 or not have_conv) else ('bn' if u < 0.85 else 'sum')
        edges.append((i - 1, i))
        if kind == 'conv':
            c_out = int(channels[rs.randint(len(channels))])
            k = int(kernels[rs.randint(len(kernels))])
            sz = (c_out, c_prev, k, k)
            net.add('n%d' % i, weight=sz)
            info.append((i, 'n%d.weight' % i, 'conv', sz, False, False))
            c_prev = c_out
            have_conv = True
        elif kind == 'bn':
            net.add('n%d' % i, weight=(c_prev,), bias=(c_prev,))
            info.append((i, 'n%d.weight' % i, 'bn', (c_prev,), False, False))
        else:
            lo, hi = max(0, i - 12), i - 2
            if hi >= lo:
                edges.append((int(rs.randint(lo, hi + 1)), i))
        prims.append(kind)
    n = n_nodes
    prims += ['glob_avg', 'conv', 'bias']
    edges += [(n - 4, n - 3), (n - 3, n - 2), (n - 2, n - 1)]
    net.add('fc', weight=(num_classes, c_prev), bias=(num_classes,))
    info.append((n - 2, 'fc.weight', 'conv', (num_classes, c_prev), True, False))
    info.append((n - 1, 'fc.bias', 'bias', (num_classes,), False, True))
    node_feat = np.asarray([[PRIM_ID[p]] for p in prims], dtype=np.int64)
    A = shortest_paths(n, edges, 50)
    g = Graph(node_feat=node_feat, node_info=[info], A=A, net_args={'seed': seed, 'n_nodes': n_nodes})
    g.net = net
    return g, net


def synthetic_batch(n_nodes_list, seed0):
    graphs, nets = [], []
    for k, n in enumerate(n_nodes_list):
        g, net = synthetic_graph(int(n), seed0 + k)
        graphs.append(g)
        nets.append(net)
    return GraphBatch(graphs, dense=True), nets
