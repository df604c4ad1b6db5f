"""
Computational graph of an ``nn.Module`` for GHN-3, host side only.

Produces what ``ghn3.Graph(model)`` of the reference produces (/root/reference/ghn3/graph.py:392-908): one forward
pass on a random input, a walk over the autograd graph, pruning of the nodes GHN-3 has no primitive for, the
weight / softmax edge repairs, an input node, a topological order, virtual edges (shortest-path lengths up to
``ve_cutoff``) and the node features / ``node_info`` consumed by ``GHN3.forward``.  The graph heuristics of the
reference are order dependent (neighbour positions in the node list, Kahn generations of its topological sort), so
every stage here keeps the reference's node order; the stages themselves are written over plain index arrays
(no networkx: BFS, path counting and the generation-wise topological order are a few lines each).

Stages (reference lines):  _trace 392-501 · _prune 666-769 · _repair_weight_edges 503-548 · _repair_softmax_edges
550-572 · input node + order 607-626 · model-specific fixes 628-646 · _virtual_edges 771-812 · _features 814-906.
Third-party classes (torchvision, ppuda) are recognised by class name along the MRO: neither package is needed.
"""

import copy
import os
import threading

import numpy as np
import torch
import torch.nn as nn

from .bookkeeping import PRIMITIVES_DEEPNETS1M, named_layered_modules

_NAMED_LAYER_CLASSES = {'LayerNorm2d': 'ln', 'PosEnc': 'pos_enc', 'Encoder': 'pos_enc'}
_OP_PRIMITIVES = {'input': 'input', 'Mean': 'glob_avg', 'AdaptiveAvgPool2D': 'glob_avg',
                  'MaxPool2DWithIndices': 'max_pool', 'AvgPool2D': 'avg_pool', 'Softmax': 'msa', 'Mul': 'cse',
                  'Add': 'sum', 'Cat': 'concat', 'skip_connect': 'sum'}


def _has_base(obj, *names):
    """isinstance by class name along the MRO (torchvision / ppuda / transformers classes without importing them)."""
    return obj is not None and any(c.__name__ in names for c in type(obj).__mro__)


def _conv_like(module):
    return isinstance(module, (nn.Conv2d, nn.Linear, nn.MultiheadAttention)) or _has_base(module, 'Conv1D')


def _layer_primitive(module, param_name):
    """Primitive name of a parameter node (graph.py:1098-1128)."""
    if _conv_like(module):
        if 'bias' in param_name:
            return 'bias'
        if isinstance(module, nn.Conv2d) and module.groups > 1:
            return 'dil_conv' if min(module.dilation) > 1 else 'sep_conv'
        return 'conv'
    if isinstance(module, nn.BatchNorm2d):
        return 'bn'
    if isinstance(module, nn.LayerNorm):
        return 'ln'
    if isinstance(module, nn.Embedding):
        return 'pos_enc'
    for cls_name, prim in _NAMED_LAYER_CLASSES.items():
        if _has_base(module, cls_name):
            return prim
    return None


def _op_prefix(name):
    k = name.find('Backward')
    return name if k < 0 else name[:k]


class _Node:
    __slots__ = ('name', 'module', 'size', 'ksize')

    def __init__(self, name, module=None, size=None, ksize=None):
        self.name, self.module, self.size, self.ksize = name, module, size, ksize


# ---------------------------------------------------------------------------------------------------------------
# stage 1: autograd walk
# ---------------------------------------------------------------------------------------------------------------
def _owners(model):
    """id(parameter) -> (qualified name, owning module); the first qualified name of a tensor is kept per name,
    the last owner wins per tensor (tied weights), as the reference's dict construction does."""
    named = {}
    for mod_name, mod in model.named_modules():
        for p_name, p in mod.named_parameters(recurse=False):
            if p is not None:
                named.setdefault(mod_name + '.' + p_name, (p, mod))
    return {id(p): (key, mod) for key, (p, mod) in named.items()}


def _trace(outputs, owners):
    """Nodes in creation order of a depth-first walk from the outputs and the directed edges between them.
    A function with parameter inputs is represented by those parameters (weight first); any other function by
    itself.  Edges point from producer to consumer, except towards a bias (weight -> bias)."""
    nodes, slot = [], {}            # slot: autograd object -> index into nodes
    link = {}                       # autograd object -> (node index or None, name) as seen by its consumers
    first = {}                      # function -> index of its first node
    entered = []

    def enter(fn):
        fname = type(fn).__name__
        if fname.endswith('Backward'):                    # a structure-only stand-in of the meta trace: named like the native node
            fname += '0'
        last = None
        if 'AccumulateGrad' not in fname:
            leaves = []
            for child, _ in fn.next_functions:
                if child is not None and hasattr(child, 'variable'):
                    pname, mod = owners[id(child.variable)]
                    leaves.append((child, pname, mod, tuple(child.variable.shape), None))
            if not leaves:
                ks = getattr(fn, '_saved_kernel_size', None)
                leaves.append((fn, fname, None, None, None if ks is None else tuple(int(v) for v in ks)))
            for obj, pname, mod, size, ks in leaves:
                if obj not in slot:
                    slot[obj] = len(nodes)
                    nodes.append(_Node(pname, mod, size, ks))
                last = slot[obj]
                first.setdefault(fn, last)
                link[obj] = (last, pname)
        link[fn] = (last, fname)
        entered.append(fn)

    stack = [fn for fn in reversed(outputs)]
    while stack:
        fn = stack.pop()
        if fn in link:
            continue
        enter(fn)
        for child, _ in reversed(fn.next_functions):
            if child is not None and child not in link:
                stack.append(child)
    edges = set()
    for fn in entered:
        src = first.get(fn)
        for child, _ in fn.next_functions:
            if child is None:
                continue
            idx, cname = link[child]
            if idx is not None and idx != src and src is not None:
                edges.add((src, idx) if 'bias' in cname else (idx, src))
    A = np.zeros((len(nodes), len(nodes)), dtype=np.int64)
    for a, b in edges:
        A[a, b] = 1
    return nodes, A


# ---------------------------------------------------------------------------------------------------------------
# stage 2: pruning (graph.py:666-769)
# ---------------------------------------------------------------------------------------------------------------
def _supported(node):
    mod = node.module
    if mod is not None and 'norm' in type(mod).__name__.lower() and _op_prefix(node.name).endswith('.bias'):
        return False                        # biases of normalisation layers are predicted but are not graph nodes
    return mod is not None and _layer_primitive(mod, node.name) is not None


def _prune(nodes, A, drop=None):
    """Removes the nodes whose name contains one of the `drop` strings (connecting their producers to their
    consumers), with the reference's special cases for squeeze-excitation products, pooling means and single-input
    sums / concatenations.  drop=None: everything GHN-3 has no primitive for."""
    if drop is None:
        unsupported = {n.name for n in nodes if not _supported(n) and _op_prefix(n.name) not in _OP_PRIMITIVES}
        drop = ['Mul'] + sorted(unsupported) + ['Mean', 'Add', 'Cat']
    has_cse = any('sigmoid' in n.name.lower() or 'swish' in n.name.lower() for n in nodes)
    n_in = [int(np.count_nonzero(A[:, i])) for i in range(len(nodes))]
    for pattern in drop:
        keep_idx = []
        for i, node in enumerate(nodes):
            keep = True
            if pattern in node.name:
                try:
                    near = {j: nodes[i + j].name.lower() for j in (-1, -2, -3, 1)}     # (negative indices wrap)
                    head = any(near[j].startswith(('classifier', 'fc', 'head')) for j in (-1, -2))
                except IndexError:
                    near, head = None, True
                if node.name.startswith('Mean'):
                    if has_cse:
                        keep = head
                elif node.name.startswith('Mul'):
                    keep = has_cse and not head and (near[-2].startswith(('hard', 'sigmoid')) or
                                                     near[-3].startswith(('relu', 'mean')) or
                                                     near[1].startswith(('hard', 'sigmoid', 'relu')))
                elif node.name.startswith(('Cat', 'Add')):
                    keep = n_in[i] > 1
                else:
                    keep = False
                if not keep:
                    outs, ins = np.nonzero(A[i, :])[0], np.nonzero(A[:, i])[0]
                    for n1 in outs:
                        for n2 in ins:
                            if n1 != n2:
                                A[n2, n1] = 1
            if keep:
                keep_idx.append(i)
        if len(keep_idx) < len(nodes):
            A = A[:, keep_idx][keep_idx, :]
            nodes = [nodes[i] for i in keep_idx]
            n_in = [n_in[i] for i in keep_idx]
    return nodes, A


# ---------------------------------------------------------------------------------------------------------------
# stage 3: edge repairs (graph.py:503-572)
# ---------------------------------------------------------------------------------------------------------------
def _repair_weight_edges(nodes, A):
    """A weight that autograd left without producer (weight -> bias of the same layer) is moved in front of its bias:
    producer -> weight -> bias -> consumers."""
    for i in range(len(nodes)):
        if A[:, i].sum() > 0 or 'weight' not in nodes[i].name:
            continue
        weight = nodes[i]                                # (stays the reference point after a swap below)
        for nb in np.nonzero(A[i, :])[0]:
            same_layer = weight.module == nodes[nb].module
            qkv = np.count_nonzero(A[:, i]) == 0 and 'softmax' in nodes[nb].name.lower()
            if not (same_layer or qkv):
                continue
            n_out = np.count_nonzero(A[i, :])
            others = np.setdiff1d(np.nonzero(A[:, nb])[0], i)
            if len(others) == 0:
                continue
            nodes[i], nodes[nb] = nodes[nb], nodes[i]
            A[i, nb], A[nb, i] = 0, 1
            if n_out == 1:
                rest = np.setdiff1d(np.nonzero(A[nb, :])[0], i)
                if len(rest) == 0:
                    continue
                A[nb, rest] = 0
                A[i, rest] = 1
    return nodes, A


def _path_counts_to(A, target):
    """Number of directed paths (capped at 2) from every node to `target` in the DAG given by A != 0."""
    n = len(A)
    succ = [np.nonzero(A[i, :])[0] for i in range(n)]
    memo = {target: 1}
    for s in range(n):
        if s in memo:
            continue
        stack, open_ = [s], {s}
        while stack:
            v = stack[-1]
            pending = [int(w) for w in succ[v] if int(w) not in memo and int(w) not in open_]
            if pending:
                stack.extend(pending)
                open_.update(pending)
                continue
            stack.pop()
            if v not in memo:
                memo[v] = min(2, sum(memo.get(int(w), 0) for w in succ[v]))      # (a back edge counts as no path)
    return memo


def _repair_softmax_edges(nodes, A):
    """Attention: the node after a softmax keeps the softmax as its only producer where the reference says so."""
    snap = A.copy()                                   # path counts refer to the graph before this repair
    for i, node in enumerate(nodes):
        if 'softmax' not in node.name.lower():
            continue
        for nb in np.nonzero(A[i, :])[0]:
            counts = None
            for j in np.setdiff1d(np.nonzero(A[:, nb])[0], i):
                if counts is None:
                    counts = _path_counts_to(snap, int(nb))
                n_paths = counts.get(int(j), 0)
                if n_paths > 1 or A[i, j] == 0:
                    A[j, nb] = 0
                if n_paths == 1 and A[i, j] == 0:
                    A[j, i] = 1
    return A


def _repair_swin_edges(nodes, A):
    """graph.py:577-601 (torchvision SwinTransformer only; restated, not exercised by the fixtures)."""
    for i, node in enumerate(nodes):
        low = node.name.lower()
        if low.endswith('norm.weight'):
            for nb in np.nonzero(A[i, :])[0]:
                if nodes[nb].name.endswith('norm1.weight') or 'Add' in nodes[nb].name:
                    A[i, nb] = 0
                    target = node.name.replace('norm', 'reduction')
                    for j, other in enumerate(nodes):
                        if target in other.name:
                            A[i, j] = 1
                            break
        elif low.endswith('attn.proj.bias'):
            for nb in np.nonzero(A[i, :])[0]:
                if nodes[nb].name.endswith('reduction.weight'):
                    A[i, nb] = 0
                    for nb2 in np.nonzero(A[nb, :])[0]:
                        if nodes[nb2].name.startswith('AddBackward'):
                            A[i, nb2] = 1
    return A


# ---------------------------------------------------------------------------------------------------------------
# stage 4: order, virtual edges
# ---------------------------------------------------------------------------------------------------------------
def _generation_order(A):
    """Topological order by generations (all nodes whose producers are placed, in discovery order) -- the order
    networkx.topological_sort yields for DiGraph(A)."""
    n = len(A)
    indeg = [int(np.count_nonzero(A[:, i])) for i in range(n)]
    succ = [np.nonzero(A[i, :])[0] for i in range(n)]
    gen = [i for i in range(n) if indeg[i] == 0]
    order = []
    while gen:
        nxt = []
        for v in gen:
            order.append(v)
            for w in succ[v]:
                indeg[w] -= 1
                if indeg[w] == 0:
                    nxt.append(int(w))
        gen = nxt
    if len(order) != n:
        raise ValueError('the computational graph has a cycle')
    return order


def _virtual_edges(A, cutoff):
    """A[i, j] = shortest-path length from i to j (2 .. cutoff) where there is no direct edge (graph.py:803-810)."""
    n = len(A)
    try:                                                   # breadth-first search in C (a 256-node graph: 1 ms instead of 10)
        from scipy.sparse import csr_matrix
        from scipy.sparse.csgraph import dijkstra
        d = dijkstra(csr_matrix((A == 1).astype(np.int8)), directed=True, unweighted=True, limit=cutoff)
        fill = np.isfinite(d) & (d > 0) & (A == 0)
        A[fill] = d[fill].astype(A.dtype)
        return A
    except ImportError:                                    # pragma: no cover
        pass
    succ = [np.nonzero(A[i, :] == 1)[0] for i in range(n)]
    for s in range(n):
        dist = {s: 0}
        frontier, d = [s], 0
        while frontier and d < cutoff:
            d += 1
            nxt = []
            for v in frontier:
                for w in succ[v]:
                    if int(w) not in dist:
                        dist[int(w)] = d
                        nxt.append(int(w))
            frontier = nxt
        for t, d in dist.items():
            if d > 0 and A[s, t] == 0:
                A[s, t] = d
    return A


def _cell_index(param_name, n_cells):
    """Cell of a node by its name (ppuda get_cell_ind, restated from the published package): 'cells.<k>.' -> k, the
    classifier and the auxiliary head -> the last cell, stems and positional encodings -> 0 -- the same split
    ``named_layered_modules`` makes on the network side; None keeps the previous node's cell."""
    if n_cells <= 1:
        return 0
    k = param_name.find('cells.')
    if k >= 0:
        digits = param_name[k + 6:].split('.')[0]
        if digits.isdigit():
            return int(digits)
    if param_name.startswith(('classifier', 'auxiliary')):
        return n_cells - 1
    if param_name.startswith(('stem', 'pos_enc')):
        return 0
    return None


# ---------------------------------------------------------------------------------------------------------------
class _MetaAffine(torch.autograd.Function):
    """Structure-only stand-in for an affine normalisation on the meta device: one autograd function whose inputs are
    (x, weight, bias) in the order of the native op, which is all the walk of _trace looks at for a function with parameter
    inputs.  (torch's own meta path of batch_norm / layer_norm is a Python decomposition: ~4 ms per layer, 0.2 s for ResNet-50.)"""

    @staticmethod
    def forward(ctx, x, w, b):
        return torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)   # (empty_like is a Python reference on meta)

    @staticmethod
    def backward(ctx, g):                                   # pragma: no cover  (the traced graph is never differentiated)
        raise RuntimeError('structure-only trace')


def _meta_unary(op_name):
    """Structure-only stand-in for a unary activation on the meta device: an autograd function `<op_name>` whose node is named
    `<op_name>Backward` like the native one (`ReluBackward0` ...; the walk and the pruning rules only look at the part in front of
    'Backward').  torch's meta path of these ops is a Python reference implementation: 0.5-3 ms per call (relu 0.5-1.3, gelu 1.1,
    hardtanh / relu6 2.0, hardsigmoid 2.4, hardswish 3.2) -- 49 ReLUs are 60 % of Graph(ResNet-50)."""
    def forward(ctx, x):
        return torch.empty_strided(x.shape, x.stride(), dtype=x.dtype, device=x.device)   # (empty_like is a Python reference on meta)

    def backward(ctx, g):                                   # pragma: no cover  (the traced graph is never differentiated)
        raise RuntimeError('structure-only trace')
    return type(op_name, (torch.autograd.Function,), {'forward': staticmethod(forward), 'backward': staticmethod(backward)})


_META_UNARY = {'relu': _meta_unary('Relu'), 'gelu': _meta_unary('Gelu'), 'hardswish': _meta_unary('Hardswish'),
               'hardtanh': _meta_unary('Hardtanh'), 'relu6': _meta_unary('Hardtanh'), 'hardsigmoid': _meta_unary('Hardsigmoid'),
               'leaky_relu': _meta_unary('LeakyRelu'), 'elu': _meta_unary('Elu')}


def _meta_add_cls():
    """x + y of two tensors on the meta device (torch's meta path of aten::add is a Python reference: 0.2-0.35 ms per call);
    the node is named AddBackward like the native AddBackward0 (the class must be CREATED under the name 'Add': autograd names
    the node after it), inputs in the same order."""
    def forward(ctx, x, y):
        shape = x.shape if x.shape == y.shape else torch.broadcast_shapes(x.shape, y.shape)
        return torch.empty(shape, dtype=torch.result_type(x, y), device=x.device)

    def backward(ctx, g):                                   # pragma: no cover
        raise RuntimeError('structure-only trace')
    return type('Add', (torch.autograd.Function,), {'forward': staticmethod(forward), 'backward': staticmethod(backward)})


_MetaAdd = _meta_add_cls()


class _cheap_meta_norms:
    """Context of the meta trace: torch.nn.functional / torch.Tensor entry points whose meta path is a Python reference are
    replaced by the structure-only stand-ins above; every replacement hands anything that is not a meta tensor carrying the
    trace straight to the original.  Process-wide patches, so: one lock, a depth counter, originals saved by the FIRST entry and
    restored by the LAST exit (two threads that trace at the same time -- loader threads do -- would otherwise save each other's
    wrappers as 'originals')."""
    _lock = threading.RLock()
    _depth = 0
    _saved = None

    @classmethod
    def _patch(cls):
        F, T = torch.nn.functional, torch.Tensor
        saved = {'bn': F.batch_norm, 'ln': F.layer_norm,
                 'add': (T.__add__, T.__radd__, T.__iadd__, T.add, T.add_, torch.add),
                 'unary': {k: getattr(F, k) for k in _META_UNARY}}
        add0 = saved['add']

        def meta_pair(a, b, kw):
            return (not kw and isinstance(a, T) and isinstance(b, T) and a.device.type == 'meta' and b.device.type == 'meta' and
                    (a.requires_grad or b.requires_grad) and a.is_floating_point() and b.is_floating_point())

        def mk(orig):
            def add(a, b, *args, **kw):
                if not args and meta_pair(a, b, kw):
                    return _MetaAdd.apply(a, b)
                return orig(a, b, *args, **kw)
            return add
        T.__add__, T.__iadd__, T.add, T.add_ = mk(add0[0]), mk(add0[2]), mk(add0[3]), mk(add0[4])
        T.__radd__ = lambda a, b, _o=add0[1]: _MetaAdd.apply(b, a) if meta_pair(b, a, None) else _o(a, b)
        torch.add = mk(add0[5])
        for k, fn_cls in _META_UNARY.items():
            def unary(input, *a, _f0=saved['unary'][k], _cls=fn_cls, **kw):
                if isinstance(input, torch.Tensor) and input.device.type == 'meta' and input.requires_grad:
                    return _cls.apply(input)
                return _f0(input, *a, **kw)
            setattr(F, k, unary)
        bn0, ln0 = saved['bn'], saved['ln']

        def batch_norm(input, running_mean, running_var, weight=None, bias=None, *a, **k):
            if input.device.type == 'meta' and weight is not None and bias is not None:
                return _MetaAffine.apply(input, weight, bias)
            return bn0(input, running_mean, running_var, weight, bias, *a, **k)

        def layer_norm(input, normalized_shape, weight=None, bias=None, *a, **k):
            if input.device.type == 'meta' and weight is not None and bias is not None:
                return _MetaAffine.apply(input, weight, bias)
            return ln0(input, normalized_shape, weight, bias, *a, **k)
        F.batch_norm, F.layer_norm = batch_norm, layer_norm
        return saved

    @classmethod
    def _unpatch(cls, saved):
        F, T = torch.nn.functional, torch.Tensor
        F.batch_norm, F.layer_norm = saved['bn'], saved['ln']
        for k, f0 in saved['unary'].items():
            setattr(F, k, f0)
        T.__add__, T.__radd__, T.__iadd__, T.add, T.add_, torch.add = saved['add']

    def __enter__(self):
        c = _cheap_meta_norms
        with c._lock:
            if c._depth == 0:
                c._saved = c._patch()
            c._depth += 1
        return self

    def __exit__(self, *exc):
        c = _cheap_meta_norms
        with c._lock:
            c._depth -= 1
            if c._depth == 0:
                c._unpatch(c._saved)
                c._saved = None
        return False


def _traced_outputs(model, in_sz, trace='meta'):
    """(outputs carrying the autograd graph, id(parameter) -> (name, module)) of one forward pass (graph.py:420-436).

    The reference pushes a random batch through the model on its own device: for ResNet-50 on a CPU that pass IS the cost of
    Graph(model) (0.25 s of 0.26; 74 such graphs in eval_ghn.py against 4 ms of prediction each).  Only the STRUCTURE of the
    autograd graph is used, so the pass runs on torch's meta device instead -- shapes and grad_fn nodes, no arithmetic, no
    memory, whatever device the model lives on: parameters and buffers are replaced by meta twins through
    torch.func.functional_call, and the owner table is keyed by the twins.  (Side effect not reproduced: the reference's pass
    also nudges BatchNorm running statistics with its random batch.)  Anything the meta device cannot run -- an op without a
    meta kernel, data-dependent control flow, a model that is not an nn.Module -- falls back to the real pass;
    trace='real' / GHN3_GRAPH_TRACE=real forces it."""
    mode = os.environ.get('GHN3_GRAPH_TRACE', trace)
    if mode == 'meta' and isinstance(model, torch.nn.Module) and not hasattr(model, 'get_var'):
        try:
            twin = {}
            def meta(t):
                if id(t) not in twin:
                    twin[id(t)] = torch.empty_like(t, device='meta').requires_grad_(t.requires_grad)
                return twin[id(t)]
            state = {n: meta(p) for n, p in model.named_parameters(remove_duplicate=False)}
            state.update({n: meta(b) for n, b in model.named_buffers(remove_duplicate=False)})
            with torch.enable_grad(), _cheap_meta_norms():
                out = torch.func.functional_call(model, state, (torch.empty(2, *in_sz, device='meta'),))
            owners = {id(twin[pid]): v for pid, v in _owners(model).items() if pid in twin}
            return out, owners
        except Exception:                                  # (NotImplementedError for a missing meta kernel, device mix-ups, ...)
            pass
    device = next(model.parameters()).device
    with torch.enable_grad():
        out = model.get_var() if hasattr(model, 'get_var') else model(torch.randn(2, *in_sz, device=device))
    return out, _owners(model)


def build_graph(model, ve_cutoff=50, reduce_graph=True, fix_weight_edges=True, fix_softmax_edges=True,
                list_all_nodes=False, verbose=False, trace='meta'):
    """Returns dict(node_feat (N,1) int64, node_info, A (N,N) int64, nodes, param_shapes, expected_input_sz, n_cells)."""
    sz = getattr(model, 'expected_input_sz', 299 if _has_base(model, 'Inception3') else 224)
    in_sz = tuple(sz) if isinstance(sz, (tuple, list)) else (3, sz, sz)
    n_cells = getattr(model, '_n_cells', 1)
    out, owners = _traced_outputs(model, in_sz, trace)
    if isinstance(out, dict):
        out = list(out.values())
    if not isinstance(out, (tuple, list)):
        out = [out]
    nodes, A = _trace([v.grad_fn for v in out if v is not None], owners)
    del out
    if reduce_graph:
        nodes, A = _prune(nodes, A)
    if fix_weight_edges:
        nodes, A = _repair_weight_edges(nodes, A)
    if fix_softmax_edges:
        A = _repair_softmax_edges(nodes, A)
    if verbose and np.trace(A) > 0:
        print('WARNING: diagonal elements of the adjacency matrix should be zero', np.trace(A))
    if _has_base(model, 'SwinTransformer'):
        A = _repair_swin_edges(nodes, A)
    if reduce_graph:
        nodes, A = _prune(nodes, A, drop=['Add', 'Cat'])
    # input node: feeds every weight that has no producer
    A = np.pad(A, ((0, 1), (0, 1)))
    nodes.append(_Node('input'))
    for i in np.nonzero(A.sum(0) == 0)[0]:
        if 'weight' in nodes[i].name:
            A[-1, i] = 1
    np.fill_diagonal(A, 0)
    try:
        order = _generation_order(A)
        nodes = [nodes[i] for i in order]
        A = A[order, :][:, order]
    except ValueError as e:                       # e.g. tied weights: the reference keeps the unsorted order too
        print('WARNING: topological sort failed:', e)
    if _has_base(model, 'VisionTransformer', 'Network'):
        # the positional encoding is followed by an explicit sum node (DeepNets-1M convention, graph.py:630-638)
        i = 0
        while i < len(nodes):
            if _has_base(nodes[i].module, 'PosEnc', 'Encoder'):
                nodes.insert(i + 1, _Node('AddBackward0'))
                A = np.insert(np.insert(A, i, 0, axis=0), i, 0, axis=1)
                A[i, i + 1] = 1
            i += 1
    elif _has_base(model, 'SqueezeNet'):
        assert nodes[-1].name.startswith('MeanBackward') and nodes[-3].name.startswith('classifier'), nodes[-3].name
        nodes.insert(len(nodes) - 3, copy.copy(nodes[-1]))
        del nodes[-1]
    assert np.trace(A) == 0, 'no loops should be in the graph'
    if ve_cutoff > 1:
        A = _virtual_edges(A, ve_cutoff)
    return _features(nodes, A, n_cells, list_all_nodes, in_sz, verbose)


def _features(nodes, A, n_cells, list_all_nodes, in_sz, verbose):
    prim_id = {p: i for i, p in enumerate(PRIMITIVES_DEEPNETS1M)}
    n = len(nodes)
    node_feat = torch.empty(n, 1, dtype=torch.long)
    node_info = [[] for _ in range(n_cells)]
    shapes = []
    cell, n_glob = 0, 0
    for k, node in enumerate(nodes):
        pname = node.name
        c = _cell_index(pname, n_cells)
        if c is not None:
            cell = c
        for marker in ('stem', 'pos_enc'):
            pos = pname.find(marker)
            if pos >= 0:
                pname = pname[pos:]
                break
        if node.module is not None:
            parts = pname.split('.')
            for j, s_ in enumerate(parts):                  # DeepNets-1M names: '_ops.<k>.<m>' -> '_ops.<k>.op.<m>'
                if s_ == '_ops' and j + 2 < len(parts) and parts[j + 2] != 'op' and parts[j + 2].isdigit():
                    parts.insert(j + 2, 'op')
                    pname = '.'.join(parts)
                    break
            prim = _layer_primitive(node.module, pname)
            if prim is None:
                raise KeyError('no GHN-3 primitive for %s (%s)' % (type(node.module).__name__, pname))
        else:
            prim = _OP_PRIMITIVES.get(_op_prefix(pname), 'sum')
            n_glob += int(prim == 'glob_avg')
            if n_cells > 1 and pname.startswith(('MaxPool', 'AvgPool')):
                pname = 'cells.%d.' % cell + prim
        sz = None
        if node.size is not None:
            sz = tuple(node.size)
        elif node.module is None and 'pool' in prim and node.name != 'input':
            sz = (1, 1) + (tuple(node.ksize) if node.ksize is not None else (3, 3))
        if sz is not None:
            if len(sz) == 3 and sz[0] == 1 and min(sz[1:]) > 1:           # (1, 197, 768) -> (1, 768, 14, 14)
                s_ = int(np.floor(sz[1] ** 0.5))
                sz = (1, sz[2], s_, s_)
            elif len(sz) == 4 and k == n - 2 and max(sz[2:]) == 1:
                sz = sz[:2]
        shapes.append(sz)
        if prim not in prim_id:
            raise KeyError('op/layer %s is not one of %s' % (prim, PRIMITIVES_DEEPNETS1M))
        node_feat[k] = prim_id[prim]
        if node.module is not None or 'pool' in prim or list_all_nodes:
            node_info[cell].append([k, pname if node.module is not None else prim, prim, sz,
                                    k == n - 2 and '.weight' in pname, k == n - 1 and '.bias' in pname])
    if n_glob != 1 and verbose:
        print('WARNING: n_glob_avg should be 1 in most architectures, but is %d in this architecture.' % n_glob)
    return dict(node_feat=node_feat, node_info=node_info, A=torch.as_tensor(A, dtype=torch.long), nodes=nodes,
                param_shapes=shapes, expected_input_sz=in_sz, n_cells=n_cells)


def attach_layered_modules(model):
    """graph.py:331-333: cache the parameter enumeration GHN3.forward matches node_info against."""
    if not hasattr(model, '_layered_modules'):
        model.__dict__['_layered_modules'] = named_layered_modules(model)
