"""
The DeepNets-1M on-disk format and the graph repairs of the reference loader (SURVEY 8(f) row 2;
/root/reference/ghn3/deepnets1m.py:84-279).

On disk (ppuda's published layout, read at deepnets1m.py:90-95,144-146):

    <nets_dir>/deepnets1m_<train|search|eval>.hdf5        group <split>/<idx>: 'adj' (N, N) int, 'nodes' (N, 3) int
    <nets_dir>/deepnets1m_<train|search|eval>_meta.json   {split: {'nets': [net arguments ...],
                                                                   'meta': {'primitives_ext': {id: name},
                                                                            'unique_op_names': {id: name}}}}

``nodes[k] = (extended primitive id, cell index, op-name id)``; ``adj`` holds 1 for an edge and the shortest-path
length (<= 50) for a virtual edge.  ``h5py`` is optional: without it (this image has none) the same arrays are read from /
written to ``deepnets1m_<...>.npz`` with keys ``<split>/<idx>/adj|nodes`` -- same contents, another container.

``init_graph`` restates ``DeepNets1MDDP._init_graph`` (deepnets1m.py:155-269): the two repairs of the stored graphs -- the
stem of ``stem_type=1`` networks wired to the second cell through the wrong stem node, and layers with more than one
producer -- followed by a recomputation of the virtual edges, the name normalisation of the older generator versions, and
the node features / ``node_info`` the GHN consumes.  Pinned by ``tests/golden/deepnets1m_cases.npz``: outputs of the
reference's own method run in the dev container on records written by ``record_from_graph`` (tests/golden/make_golden.py
deepnets1m).  ``rand_choice`` / the width ranges are ppuda's (``ppuda.utils``, ``ppuda.deepnets1m.loader``; restated from the
published package, not under /root/reference: unpinned).
"""

import json
import os

import numpy as np
import torch

from .bookkeeping import PRIMITIVES_DEEPNETS1M
from .graph import Graph

try:                                               # optional: the reference's container
    import h5py
except ImportError:                                # pragma: no cover  (this image)
    h5py = None


def split_file(split):
    return 'deepnets1m_%s' % (split if split in ('train', 'search') else 'eval')


# ---------------------------------------------------------------------------------------------------------------------
# reading
# ---------------------------------------------------------------------------------------------------------------------
class NetStore:
    """The arrays of one DeepNets-1M file, opened lazily in the process that reads (one handle per loader worker, as
    deepnets1m.py:90-91 does)."""

    def __init__(self, nets_dir, split):
        base = os.path.join(nets_dir, split_file(split))
        self.h5_file, self.npz_file, self.split = base + '.hdf5', base + '.npz', split
        self.meta_file = base + '_meta.json'
        self._h = None

    def exists(self):
        return os.path.exists(self.meta_file) and (os.path.exists(self.npz_file) or
                                                   (h5py is not None and os.path.exists(self.h5_file)))

    def partial(self):
        """Why a directory that holds SOME of the split's files cannot be read (None when nothing of the split is there, or
        when the store is complete): a user who points `nets_dir` at DeepNets-1M must not silently train / evaluate on the
        sampled stand-in stream -- the reference raises when it cannot open its files (deepnets1m.py:38-46,90-91)."""
        have = [f for f in (self.h5_file, self.npz_file, self.meta_file) if os.path.exists(f)]
        if not have or self.exists():
            return None
        if not os.path.exists(self.meta_file):
            return '%s is missing (found %s)' % (self.meta_file, ', '.join(have))
        if os.path.exists(self.h5_file) and h5py is None:
            return ('%s is there but h5py is not installed: install h5py or convert the file to %s '
                    '(ghn3_amd.deepnets1m_io.Writer)' % (self.h5_file, self.npz_file))
        return 'neither %s nor %s exists (found %s)' % (self.h5_file, self.npz_file, ', '.join(have))

    def load_meta(self):
        with open(self.meta_file) as fh:
            meta = json.load(fh)[self.split]
        to_list = lambda d: [d[str(k)] if str(k) in d else d[k] for k in range(1 + max(int(q) for q in d))]
        return meta['nets'], to_list(meta['meta']['primitives_ext']), to_list(meta['meta']['unique_op_names'])

    def get(self, idx):
        if self._h is None:
            if h5py is not None and os.path.exists(self.h5_file):
                self._h = ('h5', h5py.File(self.h5_file, mode='r'))
            else:
                self._h = ('npz', np.load(self.npz_file))
        kind, h = self._h
        if kind == 'h5':
            grp = h[self.split][str(idx)]
            return grp['adj'][()], grp['nodes'][()]
        return h['%s/%d/adj' % (self.split, idx)], h['%s/%d/nodes' % (self.split, idx)]

    def __getstate__(self):                        # (a DataLoader worker re-opens the file)
        d = dict(self.__dict__)
        d['_h'] = None
        return d


# ---------------------------------------------------------------------------------------------------------------------
# writing (what the DeepNets-1M generator stored; used to make datasets / fixtures in this format from sampled networks)
# ---------------------------------------------------------------------------------------------------------------------
def record_from_graph(graph, old_names=False):
    """(adj, [(extended primitive name, cell, op name)]) of a ``Graph(model)`` in the stored convention: the primitive name
    carries the kernel size, the op name is relative to its cell.  old_names: the naming of the first generator version
    ('_ops.<k>.<m>' without 'op', attention weights without 'attn.') that init_graph normalises."""
    nodes = []
    n = graph.n_nodes
    cells = {}
    for c, cell in enumerate(graph.node_info):
        for entry in cell:
            cells[int(entry[0])] = c
    cell = 0
    for k, node in enumerate(graph._nodes):
        prim = PRIMITIVES_DEEPNETS1M[int(graph.node_feat[k])]
        cell = cells.get(k, cell)
        sz = graph._param_shapes[k]
        ext = prim
        if prim in ('conv', 'sep_conv', 'dil_conv'):
            ks = tuple(sz[2:]) if (sz is not None and len(sz) == 4) else (1, 1)
            ext = '%s_%dx%d' % (prim, ks[0], ks[1])
        elif prim in ('max_pool', 'avg_pool'):
            ext = prim + '_3x3'
        elif prim == 'bias' and k == n - 1:
            ext = 'fc-b'
        name = node.name if node.module is not None else ''
        p = name.find('cells.')
        if p >= 0:
            name = name[p:].split('.', 2)[2]
        if old_names:
            name = name.replace('.op.', '.') if '_ops.' in name else name
            name = name.replace('attn.to_qkv', 'to_qkv').replace('attn.to_out', 'to_out')
        nodes.append((ext, int(cell), name))
    return np.asarray(graph._Adj, dtype=np.int64).copy(), nodes


class Writer:
    """Collects records and writes one DeepNets-1M file pair (hdf5 when h5py exists, npz otherwise)."""

    def __init__(self):
        self.splits = {}
        self.prims, self.names = {}, {}

    def add(self, split, net_args, adj, nodes, num_params=None):
        recs = self.splits.setdefault(split, [])
        ids = np.zeros((len(nodes), 3), dtype=np.int32)
        for k, (ext, cell, name) in enumerate(nodes):
            ids[k] = (self.prims.setdefault(ext, len(self.prims)), cell, self.names.setdefault(name, len(self.names)))
        g = net_args['genotype']
        meta = {k: (v.item() if hasattr(v, 'item') else v) for k, v in net_args.items()
                if k not in ('genotype', 'is_imagenet_input', 'num_classes')}
        meta['genotype'] = dict(normal=[list(p) for p in g.normal], normal_concat=list(g.normal_concat),
                                reduce=[list(p) for p in g.reduce], reduce_concat=list(g.reduce_concat))
        meta['num_nodes'] = int(len(nodes))
        meta['num_params'] = num_params or {'cifar10': 0, 'imagenet': 0}
        recs.append((meta, np.asarray(adj, dtype=np.int16), ids))
        return len(recs) - 1

    def save(self, nets_dir, file_split):
        os.makedirs(nets_dir, exist_ok=True)
        base = os.path.join(nets_dir, split_file(file_split))
        vocab = dict(primitives_ext={str(i): n for n, i in self.prims.items()},
                     unique_op_names={str(i): n for n, i in self.names.items()})
        meta = {s: dict(nets=[r[0] for r in recs], meta=vocab) for s, recs in self.splits.items()}
        with open(base + '_meta.json', 'w') as fh:
            json.dump(meta, fh)
        if h5py is not None:
            with h5py.File(base + '.hdf5', 'w') as h:
                for s, recs in self.splits.items():
                    for i, (_, adj, ids) in enumerate(recs):
                        g = h.create_group('%s/%d' % (s, i))
                        g.create_dataset('adj', data=adj)
                        g.create_dataset('nodes', data=ids)
            return base + '.hdf5'
        arrays = {}
        for s, recs in self.splits.items():
            for i, (_, adj, ids) in enumerate(recs):
                arrays['%s/%d/adj' % (s, i)] = adj
                arrays['%s/%d/nodes' % (s, i)] = ids
        np.savez_compressed(base + '.npz', **arrays)
        return base + '.npz'


# ---------------------------------------------------------------------------------------------------------------------
# graph repairs + node features (deepnets1m.py:155-279)
# ---------------------------------------------------------------------------------------------------------------------
def recompute_virtual_edges(A, virtual_edges):
    """deepnets1m.py:271-279: drop the stored virtual edges and recompute the shortest-path lengths (2 .. cutoff) over the
    direct edges (breadth-first search; the reference walks networkx's all-pairs lengths)."""
    if virtual_edges > 1:
        from .graph_build import _virtual_edges
        A[A > 1] = 0
        A = _virtual_edges(A, int(virtual_edges))
    return A


def init_graph(A, nodes, net_args, primitives_ext, op_names_net, virtual_edges=50, dense=True, debug=False):
    """``DeepNets1MDDP._init_graph`` (deepnets1m.py:155-269).  A: (N, N) stored adjacency, nodes: (N, 3) ids."""
    prim_dict = {op[:4]: i for i, op in enumerate(PRIMITIVES_DEEPNETS1M)}                    # deepnets1m.py:56-58
    assert len(prim_dict) == len(PRIMITIVES_DEEPNETS1M)
    A = np.array(A, dtype=np.int64)
    layers = net_args['n_cells']
    g = net_args['genotype']
    is_vit = sum(n[0] == 'msa' for n in list(g.normal) + list(g.reduce)) > 0
    N = A.shape[0]
    assert N == len(nodes), (N, len(nodes))
    recompute_ve = False
    if net_args['stem_type'] == 1 and not is_vit:
        # deepnets1m.py:167-191: the last node of stem0 feeds [stem1, cells.0.preproc0] and, in the stored graphs, also the
        # second cell, which the network wires to stem1: move that edge
        if net_args['norm'] is not None:
            stem0, stem1 = 4, 6
            if debug:
                assert op_names_net[nodes[stem0][2]] == 'stem0.4.weight' and op_names_net[nodes[stem1][2]] == 'stem1.2.weight'
        else:
            stem0, stem1 = 2, 3
        stem0_out = np.nonzero(A[stem0, :] == 1)[0]
        stem1_out = np.nonzero(A[stem1, :] == 1)[0]
        if len(stem1_out) == 1 and len(stem0_out) > 1:
            if stem0_out[-1] - stem0_out[-2] > 1:
                A[stem0, stem0_out[-1]] = 0
                A[stem1, stem0_out[-1]] = 1
                recompute_ve = True
    # deepnets1m.py:193-199: only concat / sum / cse nodes may have several producers; the others keep their first one
    for i in np.nonzero((A == 1).sum(0) > 1)[0]:
        if primitives_ext[nodes[i][0]] not in ('concat', 'sum', 'cse'):
            incoming = np.nonzero(A[:, i] == 1)[0]
            A[incoming[1:], i] = 0
            recompute_ve = True
    if recompute_ve:
        A = recompute_virtual_edges(A, virtual_edges)

    node_feat = torch.empty(N, 1, dtype=torch.long)
    node_info = [[] for _ in range(layers)]
    param_shapes = []
    for node_ind, node in enumerate(nodes):
        name = primitives_ext[node[0]]
        name_op_net = op_names_net[node[2]]
        cell_ind = int(node[1])
        sz = None
        if not name_op_net.startswith('classifier'):
            # (deepnets1m.py:217, the expression as written there: str.find is -1 -- true -- when the text is absent)
            if (name_op_net.find('.to_qkv') or name_op_net.find('.to_out')) and name_op_net.find('attn.') < 0:
                name_op_net = name_op_net.replace('to_qkv', 'attn.to_qkv').replace('to_out', 'attn.to_out')
            if len(name_op_net) == 0:
                name_op_net = 'input'
            elif name_op_net.endswith('to_out.0.'):
                name_op_net += 'weight'
            else:
                parts = name_op_net.split('.')
                for i, s in enumerate(parts):
                    if s == '_ops' and i + 2 < len(parts) and parts[i + 2] != 'op' and parts[i + 2].lstrip('-').isdigit():
                        parts.insert(i + 2, 'op')
                        name_op_net = '.'.join(parts)
                        break
            name_op_net = 'cells.%d.%s' % (cell_ind, name_op_net)
            stem_p, pos_enc_p = name_op_net.find('stem'), name_op_net.find('pos_enc')
            if stem_p >= 0:
                name_op_net = name_op_net[stem_p:]
            elif pos_enc_p >= 0:
                name_op_net = name_op_net[pos_enc_p:]
            elif name.find('pool') >= 0:
                sz = (1, 1, 3, 3)
        if name.startswith('conv_'):
            if name == 'conv_1x1':
                sz = (16, 3, 1, 1)
            name = 'conv'
        elif name.find('conv_') > 0 or name.find('pool_') > 0:
            name = name[:len(name) - 4]
        elif name == 'fc-b':
            name = 'bias'
        param_shapes.append(sz)
        node_feat[node_ind] = prim_dict[name[:4]]
        if name.find('conv') >= 0 or name.find('pool') >= 0 or name in ('bias', 'bn', 'ln', 'pos_enc'):
            node_info[cell_ind].append((node_ind, name_op_net, name, sz, node_ind == len(nodes) - 2,
                                        node_ind == len(nodes) - 1))
    A = torch.as_tensor(A, dtype=torch.long)
    A[A > virtual_edges] = 0
    graph = Graph(node_feat=node_feat, node_info=node_info, A=A, dense=dense, net_args=net_args)
    graph._param_shapes = param_shapes
    return graph


# ---------------------------------------------------------------------------------------------------------------------
# per-item network arguments (deepnets1m.py:93-142)
# ---------------------------------------------------------------------------------------------------------------------
def rand_choice(x, n=None):
    """ppuda.utils.rand_choice: a uniformly drawn element of the first n entries of a tensor (torch's global generator)."""
    return x[torch.randint(len(x) if n is None else min(n, len(x)), (1,))]


NUM_CH = torch.arange(32, 128 + 1, 16)              # ppuda DeepNets1M defaults: num_ch=(32, 128), fc_dim=(64, 512)
FC_DIM = torch.arange(64, 512 + 1, 64)


def item_net_args(args, genotype, is_train, large_images, wider_nets, split, num_ch=NUM_CH, fc_dim=FC_DIM):
    """Network arguments of one stored architecture (deepnets1m.py:95-142): in training the width C and the classifier
    width are re-drawn per visit from ranges that shrink with the network's size (so that it fits the memory budget), and
    small ImageNet networks may get stride 2 (`wider_nets`)."""
    args = dict(args)
    n_cells = args['n_cells']
    args['imagenet_stride'] = 4
    if is_train:
        is_conv_dense = sum(n[0] in ('conv_5x5', 'conv_7x7') for n in list(genotype.normal) + list(genotype.reduce)) > 0
        num_params = args['num_params']['imagenet' if large_images and not wider_nets else 'cifar10'] / 10 ** 6
        if wider_nets and large_images and args['glob_avg'] and args['stem_type'] == 0 and args['stem_pool'] and \
                not (num_params > 0.2 or n_cells > 8 or is_conv_dense):
            args['imagenet_stride'] = int(np.random.choice([2, 4]))
        fc = rand_choice(fc_dim, 4)
        if num_params > (2.0 if wider_nets else 0.8) or not args['glob_avg'] or is_conv_dense or \
                n_cells > (14 if wider_nets else 12):
            C = num_ch.min()
        elif num_params > 0.4 or n_cells > 10:
            C = rand_choice(num_ch, 4 if wider_nets else 2)
        elif num_params > 0.2 or n_cells > 8:
            C = rand_choice(num_ch, 5 if wider_nets else 3)
        else:
            C = rand_choice(num_ch)
            if C <= 64:
                fc = rand_choice(fc_dim)
        args['C'], args['fc_dim'] = int(C.item()), int(fc.item())
    net_args = {'genotype': genotype}
    for key in ('norm', 'ks', 'preproc', 'glob_avg', 'stem_pool', 'C_mult', 'n_cells', 'fc_layers', 'C', 'fc_dim',
                'stem_type', 'imagenet_stride'):
        net_args[key] = args[key] * (2 if large_images else 4) if (key == 'C' and split == 'wide') else args[key]
    return net_args
