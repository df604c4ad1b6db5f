"""
Host-side graph containers: the input format of the hot path.

Mirrors the interface of /root/reference/ghn3/graph.py:38-352 for the dense (Graphormer) layout:
``Graph(node_feat=, node_info=, A=, dense=True)`` and ``GraphBatch(graphs, dense=True)`` with
``to_device / on_device / to_dense / to_sparse / __len__ / __getitem__ / __iter__`` and the fields
``node_feat (B,N,1) int64, edges (B,N,N) int64, mask (B,N,1) bool, n_nodes (B,) int64, node_info, net_args,
nets``.  ``Graph(model)`` builds the graph of an arbitrary ``nn.Module`` (graph.py:392-908: autograd walk, pruning,
virtual edges) through ``ghn3_amd.graph_build``.
"""

import numpy as np
import torch


def _tensors_to_numpy(state):
    """Pickle support: CPU tensors leave a process as numpy arrays.  torch registers shared-memory reducers for tensors with
    the multiprocessing pickler -- every tensor of a result then travels as a shm segment + a file descriptor passed over
    a socket, several milliseconds of syscalls and interpreter-lock time per batch in the process that feeds the GPU
    (measured: the host side of a fresh-architecture step 2 ms -> 9 ms while loader workers returned GraphBatch objects)."""
    def conv(v):
        if isinstance(v, torch.Tensor) and not v.is_cuda:
            return _NpTensor(v.numpy())
        if isinstance(v, list) and v and all(isinstance(e, torch.Tensor) and not e.is_cuda for e in v):
            return [_NpTensor(e.numpy()) for e in v]
        return v
    return {k: conv(v) for k, v in state.items()}


def _numpy_to_tensors(state):
    def conv(v):
        if isinstance(v, _NpTensor):
            return torch.from_numpy(v.a)
        if isinstance(v, list) and v and all(isinstance(e, _NpTensor) for e in v):
            return [torch.from_numpy(e.a) for e in v]
        return v
    return {k: conv(v) for k, v in state.items()}


class _NpTensor:
    __slots__ = ('a',)

    def __init__(self, a):
        self.a = a

    def __getstate__(self):
        return self.a

    def __setstate__(self, a):
        self.a = a


class Graph:
    r"""
    Container for a computational graph of a neural network (explicit-arrays constructor of graph.py:292-352).

    :param node_feat: (N,1) int64 primitive ids (order of ppuda PRIMITIVES_DEEPNETS1M)
    :param node_info: per cell, list of (node_ind, param_name, primitive_name, shape, is_last_weight, is_last_bias)
    :param A: (N,N) int64 shortest-path adjacency (0 = no edge, values <= ve_cutoff)
    """

    def __init__(self, model=None, node_feat=None, node_info=None, A=None, edges=None, net_args=None, net_idx=None,
                 ve_cutoff=50, dense=True, **kwargs):
        if model is not None:
            # automatic construction from the module's autograd graph (graph.py:392-908) -> graph_build.py
            assert node_feat is None, 'either model or other arguments must be specified'
            from .graph_build import build_graph, attach_layered_modules
            built = build_graph(model, ve_cutoff=ve_cutoff, reduce_graph=kwargs.get('reduce_graph', True),
                                fix_weight_edges=kwargs.get('fix_weight_edges', True),
                                fix_softmax_edges=kwargs.get('fix_softmax_edges', True),
                                list_all_nodes=kwargs.get('list_all_nodes', False),
                                verbose=kwargs.get('verbose', False), trace=kwargs.get('trace', 'meta'))
            self.model = model
            self.n_nodes = len(built['node_feat'])
            self.node_feat, self.node_info, self._Adj = built['node_feat'], built['node_info'], built['A']
            self._nodes, self._param_shapes = built['nodes'], built['param_shapes']
            self.expected_input_sz, self.n_cells = built['expected_input_sz'], built['n_cells']
            attach_layered_modules(model)
            self.net_args, self.net_idx = net_args, net_idx
            return
        assert dense, 'only the dense (Graphormer / GHN-3) layout is supported'
        assert node_feat is not None and A is not None and node_info is not None
        self.model = None
        self.n_nodes = len(node_feat)
        self.node_feat = torch.as_tensor(node_feat, dtype=torch.long).view(-1, 1)
        self.node_info = node_info
        self._Adj = torch.as_tensor(A, dtype=torch.long)
        assert self._Adj.shape == (self.n_nodes, self.n_nodes), self._Adj.shape
        self.net_args = net_args
        self.net_idx = net_idx

    def __getstate__(self):
        return _tensors_to_numpy(self.__dict__)

    def __setstate__(self, state):
        self.__dict__.update(_numpy_to_tensors(state))


class GraphBatch:
    r"""Container for a batch of Graph objects (graph.py:38-279, dense path)."""

    def __init__(self, graphs, dense=True):
        assert dense, 'only dense=True (GHN-3) is supported'
        self.n_nodes, self.node_feat, self.node_info, self.edges, self.net_args, self.net_inds = [], [], [], [], [], []
        self.graphs = graphs
        self.dense = dense
        self.mask = []
        self.max_edge = None
        if graphs is not None:
            if not isinstance(graphs, (list, tuple)):
                graphs = [graphs]
            for graph in graphs:
                self.append(graph)

    def append(self, graph):
        self.n_nodes.append(len(graph.node_feat))
        self.node_feat.append(graph.node_feat)
        self.edges.append(graph._Adj)
        self.node_info.append(graph.node_info)
        self.net_args.append(graph.net_args)
        self.net_inds.append(graph.net_idx)
        if hasattr(graph, 'net'):
            if not hasattr(self, 'nets'):
                self.nets = []
            self.nets.append(graph.net)

    def _cat(self):
        """graph.py:243-269: pad + stack on the host."""
        if isinstance(self.node_feat, torch.Tensor):
            return
        n = [int(v) for v in self.n_nodes]
        B, m = len(n), max(n)
        # numpy on the host: torch's CPU ops pay tens of milliseconds of thread-pool latency per call on many-core
        # boxes (zeros / max of a 256 x 256 int64 tensor: 20-60 ms each), which made a fresh batch cost 150 ms
        node_feat = np.zeros((B, m, 1), dtype=np.int64)
        edges = np.zeros((B, m, m), dtype=np.int64)
        mask = np.zeros((B, m, 1), dtype=bool)
        for b in range(B):
            node_feat[b, :n[b]] = np.asarray(self.node_feat[b]).reshape(n[b], 1)
            edges[b, :n[b], :n[b]] = np.asarray(self.edges[b])
            mask[b, :n[b]] = True
        self.max_edge = int(edges.max()) if edges.size else 0
        self._n_nodes_host = n
        self._node_type_host = np.concatenate([node_feat[b, :n[b], 0] for b in range(B)]).astype(np.int32)
        self.n_nodes = torch.from_numpy(np.asarray(n, dtype=np.int64))
        self.node_feat, self.edges, self.mask = (torch.from_numpy(node_feat), torch.from_numpy(edges),
                                                 torch.from_numpy(mask))

    def to_device(self, device):
        if isinstance(device, (tuple, list)):
            device = device[0]
        self._cat()
        if torch.device(device).type == 'cuda':
            from .nn import pinned_ring               # reusable pinned staging: asynchronous for the host
            ring = pinned_ring(device)

            def move(t):
                return ring.upload(t, device)
        else:
            def move(t):
                return t.to(device, non_blocking=True)
        self.n_nodes, self.node_feat = move(self.n_nodes), move(self.node_feat)
        self.edges, self.mask = move(self.edges), move(self.mask)
        return self

    def on_device(self, device):
        if isinstance(device, (tuple, list)):
            device = device[0]
        return isinstance(self.n_nodes, torch.Tensor) and isinstance(self.node_feat, torch.Tensor) and \
            self.node_feat.device == torch.device(device)

    def host_n_nodes(self):
        """Node counts as Python ints without a device sync (the reference syncs at nn.py:615, graph.py:175)."""
        if not hasattr(self, '_n_nodes_host'):
            self._cat()
        return self._n_nodes_host

    def precompile(self, config, training=True, predict_class_layers=True, reduce_graph=False):
        """Host half of GHN3.compile() for this batch, run where the batch is built (a loader worker): the Program is plain
        numpy, needs no GPU, and pickles together with the batch and its networks.  `config` = GHN3.program_config() of the
        model that will consume the batch; GHN3.compile() takes the attached program when the model's configuration and the
        call's flags are the ones given here, and builds its own otherwise.  (With reduce_graph=True the matched shape tables
        of the networks are consumed here, as the reference's _map_net_params does.)"""
        from .program import Program
        self._cat()
        args = dict(config, training=bool(training), predict_class_layers=bool(predict_class_layers),
                    reduce_graph=bool(reduce_graph))
        cfg = args.pop('cfg')
        self.program = Program(cfg, self.node_info, self.host_n_nodes(), self._node_type_host, self.max_edge, self.nets,
                               **args).strip()
        self.program_args = dict(args, cfg=cfg)
        return self

    def take_program(self, nets, **args):
        """The precompiled Program of precompile() when it was built for exactly these networks and arguments (and not used
        before: a plan owns its program), else None."""
        prog, have = getattr(self, 'program', None), getattr(self, 'program_args', None)
        same_nets = hasattr(self, 'nets') and len(nets) == len(self.nets) and all(a is b for a, b in zip(nets, self.nets))
        if prog is None or have != args or not same_nets:
            if prog is not None and have['reduce_graph']:
                differs = ', '.join(k for k in args if have.get(k) != args[k]) or 'other networks'
                raise ValueError('this batch was precompiled with reduce_graph=True for other arguments (%s): its networks '
                                 'no longer carry the shape tables a new compile needs' % differs)
            return None
        self.program = None
        return prog

    def to_dense(self, x=None):
        if x is None:
            x = self.node_feat
        n = self.host_n_nodes()
        B, M, C = len(n), max(n), x.shape[-1]
        out = torch.zeros(B, M, C, device=x.device, dtype=x.dtype)
        offset = [0]
        for b in range(B):
            out[b, :n[b]] = x[offset[-1]: offset[-1] + n[b]]
            offset.append(offset[-1] + n[b])
        return out, offset

    def to_sparse(self, x):
        n = self.host_n_nodes()
        return torch.cat([x[b, :n[b]] for b in range(len(n))])

    def __getitem__(self, idx):
        return self.graphs[idx]

    def __len__(self):
        return len(self.n_nodes)

    def __iter__(self):
        for graph in self.graphs:
            yield graph

    def __getstate__(self):
        return _tensors_to_numpy(self.__dict__)

    def __setstate__(self, state):
        self.__dict__.update(_numpy_to_tensors(state))

