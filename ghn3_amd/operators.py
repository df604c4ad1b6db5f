"""
Operator-level API of the reference on the HIP path (SURVEY 8(b) "internal operator API"):

    TransformerLayer.forward(x, edges, mask) -> (x, edges, mask) | x        /root/reference/ghn3/graphormer.py:208-248
    ConvDecoder3.forward(x, max_shape, class_pred)                            /root/reference/ghn3/nn.py:735-762

`GHN3.forward` never calls these (it runs ONE compiled op program for the whole model); they exist so that code written
against the reference's modules -- ``ghn.gnn[l](x, edges, mask)``, ``ghn.decoder(x, max_shape, class_pred)`` -- runs
on the same kernels through the same C ABI: each call compiles a small op program (the ops of that layer / decoder
only, exact-fp32 path), runs it with ``ghn3_run`` and returns torch tensors.  Like the reference's modules they are
differentiable: when autograd is recording, the call goes through a torch.autograd.Function whose backward is a second
small op program built from the same backward ops GHN3's backward program uses (dgrad / wgrad GEMMs, LayerNorm and
attention backward, the edge-bias histogram, the embedding scatter), so ``loss.backward()`` fills ``.grad`` of the
layer's / decoder's parameters and of the inputs.
"""

import numpy as np
import torch

from . import _lib as L
from .program import Program, param_names, round_up


class MiniProgram(Program):
    """Op-program builder without a graph batch: the bookkeeping of Program (buffer references, workspace layout,
    index blob, GEMM problem table) for a hand-written op sequence."""

    def __init__(self, ghn):
        self.cfg = dict(hid=ghn.hid, heads=ghn.heads, layers=ghn.layers, num_classes=ghn.num_classes,
                        max_shape=ghn.max_shape)
        self.SIDE = 0
        self.C, self.H, self.Lyr, self.K = ghn.hid, ghn.heads, ghn.layers, ghn.num_classes
        self.max_shape = tuple(ghn.max_shape)
        self.S = self.max_shape[2]
        self.layernorm = ghn.layernorm
        self.names = param_names(self.Lyr, ghn.layernorm)
        self.P = len(self.names)
        self.slot = {n: i for i, n in enumerate(self.names)}
        self.decoder_ctype = None
        self._ops, self._probs, self._ln = [], [], {}
        self.tag_flops = {}
        self._ws, self._ws_names = 0, {}
        self._idx_chunks, self._idx_size = [], 0
        self.x3 = False

    def finish(self):
        self.ops = self._finish_ops()
        self.problems = self._pack_problems()
        self.ws_bytes = round_up(self._ws + 1024, 256)
        self.idx_blob = np.zeros(max(self._idx_size, 16), dtype=np.uint8)
        for off, raw in self._idx_chunks:
            self.idx_blob[off:off + len(raw)] = raw
        self.n_bufs = 2 * self.P + self.X_COUNT
        return self


def _run(ghn, mini, inputs, edges=None, gflat=None):
    """inputs: {workspace name: tensor} copied into the zeroed workspace before the run.  Returns the workspace.
    gflat: flat gradient buffer the program's gradient references (Program.gref) point into."""
    dev = ghn.device
    if dev.type != 'cuda':
        raise L.Ghn3Error('ghn3_amd operators run on an MI355X only (no CPU path)')
    ws = torch.zeros(mini.ws_bytes, dtype=torch.uint8, device=dev)
    for name, t in inputs.items():
        off = mini._ws_names[name]
        raw = t.contiguous().view(-1).view(torch.uint8)
        ws[off:off + raw.numel()].copy_(raw)
    idx = torch.from_numpy(mini.idx_blob).to(dev)
    bufs = np.zeros(mini.n_bufs, dtype=np.uint64)
    bufs[:mini.P] = (ghn._flat.data_ptr() + 4 * ghn._offs).astype(np.uint64)
    if gflat is not None:
        bufs[mini.P:2 * mini.P] = (gflat.data_ptr() + 4 * ghn._offs).astype(np.uint64)
    bufs[mini.xbuf(mini.X_WS)] = ws.data_ptr()
    bufs[mini.xbuf(mini.X_IDX)] = idx.data_ptr()
    if edges is not None:
        bufs[mini.xbuf(mini.X_EDGES)] = edges.data_ptr()
    ghn._ctx().run(mini.ops, mini.problems, bufs, torch.cuda.current_stream().cuda_stream)
    return ws


def _read(ws, mini, name, shape, dtype=torch.float32):
    off = mini._ws_names[name]
    n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    return ws[off:off + n].view(dtype).view(*shape).clone()


def _prefix_node_counts(mask, B, N):
    """graphormer.py:211-248 takes an arbitrary (B, N, N) mask; GraphBatch only ever produces m_i & m_j of a node PREFIX
    (graph.py:172-181), which is what the kernels implement (n_nodes per graph)."""
    if mask is None:
        return [N] * B
    m = mask.reshape(B, N, -1)
    diag = torch.diagonal(m[:, :, :N], dim1=1, dim2=2) if m.shape[-1] == N else m[:, :, 0]
    n = diag.sum(1).tolist()
    ar = torch.arange(N, device=mask.device)
    if not bool((diag == (ar[None, :] < torch.as_tensor(n, device=mask.device)[:, None])).all()):
        raise NotImplementedError('attention masks other than a node prefix per graph (GraphBatch.mask)')
    return [int(v) for v in n]


def transformer_layer_forward(ghn, l, x, edges=None, mask=None, return_edges=True):
    """One Graphormer layer (graphormer.py:208-248) of `ghn` on the HIP kernels.  Layer 0 takes the integer shortest-path
    matrix `edges` (B, N, N); later layers take the edge bias (B, N, N, H) layer 0 returned.  Differentiable w.r.t. x, the
    edge bias (layers > 0) and the layer's parameters when autograd is recording."""
    needs_grad = torch.is_grad_enabled() and (x.requires_grad or (torch.is_tensor(edges) and edges.is_floating_point()
                                                                  and edges.requires_grad) or
                                              any(p_.requires_grad for p_ in _layer_params(ghn, l)))
    if needs_grad:
        e_in = edges if (l > 0 and edges is not None) else None
        dummy = x.new_zeros(())
        outs = _LayerFunction.apply(ghn, l, edges, mask, x, e_in if e_in is not None else dummy, *_layer_params(ghn, l))
        y, e_out = outs[0], outs[1]
        if not return_edges:
            return y
        return y, (e_out if l == 0 else edges), mask
    return _layer_forward_impl(ghn, l, x, edges, mask, return_edges)[0]


def _layer_param_names(ghn, l):
    pre = 'gnn.%d.' % l
    return [n for n in param_names(ghn.layers, ghn.layernorm) if n.startswith(pre)]


def _layer_params(ghn, l):
    named = dict(ghn.named_parameters())
    return [named[n] for n in _layer_param_names(ghn, l)]


def _layer_forward_impl(ghn, l, x, edges=None, mask=None, return_edges=True, save=False):
    """Returns (result, saved): result as transformer_layer_forward; saved = tensors the backward program needs."""
    sz = x.shape
    if x.dim() == 2:
        x = x.unsqueeze(0)
    assert x.dim() == 3, x.shape
    B, N, C = x.shape
    H = ghn.heads
    assert C == ghn.hid
    rows = B * N
    layer0 = l == 0
    n_nodes = _prefix_node_counts(mask, B, N)
    p = MiniProgram(ghn)
    pre = 'gnn.%d.' % l
    r_nn = p.idx(np.asarray(n_nodes, dtype=np.int32))
    x_in = p.wsf('x_in', rows * C)
    bias = p.wsf('bias', B * H * N * N)
    ins = {}
    dev = ghn.device
    edges_dev = None
    if layer0:
        assert edges is not None and edges.dim() == 3, 'layer 0 takes the dense (B, N, N) shortest-path matrix'
        edges_dev = edges.to(dev, torch.int64).contiguous()
        V = int(edges_dev.max().item()) + 1
        ldT = round_up(H, 4)
        deg_in = (p.xbuf(p.X_WS), p.ws('deg_in', 4 * rows))
        deg_out = (p.xbuf(p.X_WS), p.ws('deg_out', 4 * rows))
        dist0 = (p.xbuf(p.X_WS), p.ws('dist0', 4 * rows))
        pair = (p.xbuf(p.X_WS), p.ws('pair', 4 * rows * N))
        p.op(L.OP_GRAPH_PROLOGUE, refs=((p.xbuf(p.X_EDGES), 0), deg_in, deg_out, dist0, pair), ints=(B, N, V))
        # x += E_in[deg_in] + E_out[deg_out] + E_dist[A[0, :]]; x *= mask  (graphormer.py:229-235) with the node-embedding
        # kernel: its "type table" is the input x itself (type id = dense row), its shape tables a zero row
        x_raw = p.wsf('x_raw', rows * C)
        zero = p.wsf('zero_row', C)
        node_off = np.cumsum([0] + n_nodes[:-1]).astype(np.int32)
        types = np.concatenate([np.arange(b * N, b * N + n_nodes[b]) for b in range(B)]).astype(np.int32)
        shape_idx = np.zeros(4 * len(types), dtype=np.int32)
        p.op(L.OP_EMBED_NODES,
             refs=(x_in, p.idx(types), p.idx(shape_idx), r_nn, p.idx(node_off), x_raw, zero, zero,
                   p.pref('gnn.0.centrality_embed_in.weight'), p.pref('gnn.0.centrality_embed_out.weight'),
                   p.pref('gnn.0.input_dist_embed.weight'), deg_in, deg_out, dist0), ints=(B, N, C))
        ins['x_raw'] = x.to(dev, torch.float32)
        # edge bias (graphormer.py:115-117), factorised over the distinct (fw, bw) pairs
        E = 'gnn.0.attn.edge_embed.embed.weight'
        W0e, b0e = 'gnn.0.attn.proj_e.0.weight', 'gnn.0.attn.proj_e.0.bias'
        W2e, b2e = 'gnn.0.attn.proj_e.2.weight', 'gnn.0.attn.proj_e.2.bias'
        Pfw, Pbw = p.wsf('Pfw', V * C), p.wsf('Pbw', V * C)
        hid = p.wsf('hid', V * V * C)
        T = p.wsf('T', V * V * ldT)
        p0 = p.gemm(p.pref(E, 2 * C), p.pref(W0e, 0), Pfw, V, C, C, C, 2 * C, C)
        p.gemm(p.pref(E, 2 * C), p.pref(W0e, C), Pbw, V, C, C, C, 2 * C, C, bias=p.pref(b0e))
        p.gemm_op(p0)
        p.op(L.OP_EDGE_HIDDEN, refs=(hid, Pfw, Pbw), ints=(V, C))
        p0 = p.gemm(hid, p.pref(W2e), T, V * V, H, C, C, C, ldT, bias=p.pref(b2e))
        p.gemm_op(p0)
        p.op(L.OP_BIAS_GATHER, refs=(bias, T, pair), ints=(B, N, H))
    else:
        ins['x_in'] = x.to(dev, torch.float32)
        if edges is not None:
            assert edges.dim() == 4 and edges.shape[-1] == H, 'layers > 0 take the (B, N, N, H) edge bias of layer 0'
            ins['bias'] = edges.to(dev, torch.float32).permute(0, 3, 1, 2)
    h1, qkv, o = p.wsf('h1', rows * C), p.wsf('qkv', rows * 3 * C), p.wsf('o', rows * C)
    xmid, h2 = p.wsf('xmid', rows * C), p.wsf('h2', rows * C)
    f, x_out = p.wsf('f', rows * 4 * C), p.wsf('x_out', rows * C)
    m1, r1 = p.wsf('m1', rows), p.wsf('r1', rows)
    m2, r2 = p.wsf('m2', rows), p.wsf('r2', rows)
    Pm = p.wsf('P', B * H * N * N) if save else p.NONE
    z = p.wsf('z', rows * 4 * C) if save else None
    p.op(L.OP_LAYERNORM_FWD, refs=(h1, x_in, p.pref(pre + 'ln1.weight'), p.pref(pre + 'ln1.bias'), m1, r1),
         ints=(rows, C), floats=(1e-5,))
    p.gemm_op(p.gemm(h1, p.pref(pre + 'attn.to_qkv.weight'), qkv, rows, 3 * C, C, C, C, 3 * C))
    p.op(L.OP_ATTN_FWD, refs=(o, qkv, bias, Pm, r_nn), ints=(B, N, C, H))
    p.gemm_op(p.gemm(o, p.pref(pre + 'attn.to_out.0.weight'), xmid, rows, C, C, C, C, C,
                     bias=p.pref(pre + 'attn.to_out.0.bias'), residual=x_in))
    p.op(L.OP_LAYERNORM_FWD, refs=(h2, xmid, p.pref(pre + 'ln2.weight'), p.pref(pre + 'ln2.bias'), m2, r2),
         ints=(rows, C), floats=(1e-5,))
    p.gemm_op(p.gemm(h2, p.pref(pre + 'ff.net.0.weight'), f, rows, 4 * C, C, C, C, 4 * C,
                     bias=p.pref(pre + 'ff.net.0.bias'), act=L.ACT_GELU, aux_out=z))
    p.gemm_op(p.gemm(f, p.pref(pre + 'ff.net.3.weight'), x_out, rows, C, 4 * C, 4 * C, 4 * C, C,
                     bias=p.pref(pre + 'ff.net.3.bias'), residual=xmid))
    p.finish()
    ws = _run(ghn, p, ins, edges=edges_dev)
    y = _read(ws, p, 'x_out', (B, N, C))
    if len(sz) == 2:
        y = y[0]
    saved = None
    if save:
        saved = dict(B=B, N=N, n_nodes=n_nodes, squeeze=len(sz) == 2)
        for name, shape in (('x_in', (rows, C)), ('h1', (rows, C)), ('qkv', (rows, 3 * C)), ('P', (B * H * N * N,)),
                            ('o', (rows, C)), ('xmid', (rows, C)), ('h2', (rows, C)), ('z', (rows, 4 * C)),
                            ('f', (rows, 4 * C)), ('m1', (rows,)), ('r1', (rows,)), ('m2', (rows,)), ('r2', (rows,)),
                            ('bias', (B * H * N * N,))):
            saved[name] = _read(ws, p, name, shape)
        if layer0:
            saved['V'] = V
            saved['hid'] = _read(ws, p, 'hid', (V * V * C,))
            for name in ('deg_in', 'deg_out', 'dist0'):
                saved[name] = _read(ws, p, name, (rows,), torch.int32)
            saved['pair'] = _read(ws, p, 'pair', (rows * N,), torch.int32)
    e_out = _read(ws, p, 'bias', (B, H, N, N)).permute(0, 2, 3, 1) if layer0 else edges
    if not return_edges:
        return (y, saved) if save else (y, None)
    return ((y, e_out, mask), saved)


class _LayerFunction(torch.autograd.Function):
    """autograd of one Graphormer layer (graphormer.py:208-248).  Inputs after the non-tensor arguments: x, the edge bias
    (layers > 0; a dummy scalar for layer 0) and the layer's parameters in Program.names order."""

    @staticmethod
    def forward(ctx, ghn, l, edges, mask, x, e_in, *params):
        (y, e_out, _), saved = _layer_forward_impl(ghn, l, x.detach(), edges.detach() if torch.is_tensor(edges) else edges,
                                                   mask, True, save=True)
        ctx.ghn, ctx.l, ctx.saved = ghn, l, saved
        ctx.has_e = l > 0 and torch.is_tensor(edges)
        ctx.x_shape = tuple(x.shape)
        if l > 0:
            e_out = x.new_zeros(())                              # (the wrapper hands the caller's own tensor back)
        return y, e_out

    @staticmethod
    def backward(ctx, gy, ge):
        ghn, l, sv = ctx.ghn, ctx.l, ctx.saved
        B, N, n_nodes = sv['B'], sv['N'], sv['n_nodes']
        C, H = ghn.hid, ghn.heads
        rows = B * N
        layer0 = l == 0
        pre = 'gnn.%d.' % l
        p = MiniProgram(ghn)
        r_nn = p.idx(np.asarray(n_nodes, dtype=np.int32))
        ins = {}
        for name in ('x_in', 'h1', 'qkv', 'P', 'o', 'xmid', 'h2', 'z', 'f', 'm1', 'r1', 'm2', 'r2'):
            p.wsf(name, sv[name].numel())
            ins[name] = sv[name]
        g_cur = p.wsf('g_cur', rows * C)
        ins['g_cur'] = gy.reshape(rows, C).to(torch.float32)
        dBias = p.wsf('dBias', B * H * N * N)
        if layer0 and ge is not None and ge.dim() == 4:          # gradient the later layers sent into the returned bias
            ins['dBias'] = ge.permute(0, 3, 1, 2).to(torch.float32)
        dz, dhA, dhB = p.wsf('dz', rows * 4 * C), p.wsf('dhA', rows * C), p.wsf('dhB', rows * C)
        g_mid, g_in, do = p.wsf('g_mid', rows * C), p.wsf('g_in', rows * C), p.wsf('do', rows * C)
        dqkv = p.wsf('dqkv', rows * 3 * C)
        W3, W1f, Wo, Wq = pre + 'ff.net.3.weight', pre + 'ff.net.0.weight', pre + 'attn.to_out.0.weight', \
            pre + 'attn.to_qkv.weight'
        w = p.wref
        # dependent chain (the non-x3 branch of Program._build_backward)
        p.gemm_op(p.gemm(g_cur, p.pref(W3), dz, rows, 4 * C, C, C, 4 * C, 4 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                         dact=L.DACT_GELU, aux_in=w('z')))
        p.gemm_op(p.gemm(dz, p.pref(W1f), dhA, rows, C, 4 * C, 4 * C, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL))
        p.op(L.OP_LAYERNORM_BWD, refs=(g_mid, dhA, w('xmid'), p.pref(pre + 'ln2.weight'), w('m2'), w('r2'), g_cur, p.NONE),
             ints=(rows, C))
        p.gemm_op(p.gemm(g_mid, p.pref(Wo), do, rows, C, C, C, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL))
        p.op(L.OP_ATTN_BWD, refs=(dqkv, do, w('qkv'), w('P'), w('o'), p.NONE, dBias, r_nn), ints=(B, N, C, H))
        p.gemm_op(p.gemm(dqkv, p.pref(Wq), dhB, rows, C, 3 * C, 3 * C, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL))
        p.op(L.OP_LAYERNORM_BWD, refs=(g_in, dhB, w('x_in'), p.pref(pre + 'ln1.weight'), w('m1'), w('r1'), g_mid, p.NONE),
             ints=(rows, C))
        # parameter gradients
        p.op(L.OP_LN_PARAM_GRAD, refs=(p.gref(pre + 'ln2.weight'), p.gref(pre + 'ln2.bias'), dhA, w('xmid'), w('m2'),
                                       w('r2')), ints=(rows, C, 1))
        p0 = p.gemm(g_cur, w('f'), p.gref(W3), C, 4 * C, rows, C, 4 * C, 4 * C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                    accum=True, dbias=p.gref(pre + 'ff.net.3.bias'))
        p.gemm(dz, w('h2'), p.gref(W1f), 4 * C, C, rows, 4 * C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL, accum=True,
               dbias=p.gref(pre + 'ff.net.0.bias'))
        p.gemm(g_mid, w('o'), p.gref(Wo), C, C, rows, C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL, accum=True,
               dbias=p.gref(pre + 'attn.to_out.0.bias'))
        p.gemm(dqkv, w('h1'), p.gref(Wq), 3 * C, C, rows, 3 * C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL, accum=True)
        p.gemm_op(p0)
        p.op(L.OP_LN_PARAM_GRAD, refs=(p.gref(pre + 'ln1.weight'), p.gref(pre + 'ln1.bias'), dhB, w('x_in'), w('m1'),
                                       w('r1')), ints=(rows, C, 1))
        dx_name = 'g_in'
        if layer0:
            V = sv['V']
            ldT = round_up(H, 4)
            # x = (x_raw + E_in[deg_in] + E_out[deg_out] + E_dist[dist0]) * mask: scatter g_in back (the "type table" of the
            # embedding op was the input itself, so its table gradient is d x_raw, one row per dense node row)
            for name in ('deg_in', 'deg_out', 'dist0', 'pair'):
                p.ws(name, 4 * sv[name].numel())
                ins[name] = sv[name]
            i32 = lambda nm: (p.xbuf(p.X_WS), p._ws_names[nm])
            node_off = np.cumsum([0] + n_nodes[:-1]).astype(np.int32)
            types = np.concatenate([np.arange(b * N, b * N + n_nodes[b]) for b in range(B)]).astype(np.int32)
            shape_idx = np.zeros(4 * len(types), dtype=np.int32)
            dx_raw = p.wsf('dx_raw', rows * C)
            dzero, dzero2 = p.wsf('dzero', C), p.wsf('dzero2', C)      # (gradients of the zero shape rows: discarded)
            p.op(L.OP_EMBED_BWD,
                 refs=(g_in, p.idx(types), p.idx(shape_idx), r_nn, p.idx(node_off), dx_raw, dzero, dzero2,
                       p.gref('gnn.0.centrality_embed_in.weight'), p.gref('gnn.0.centrality_embed_out.weight'),
                       p.gref('gnn.0.input_dist_embed.weight'), i32('deg_in'), i32('deg_out'), i32('dist0')),
                 ints=(B, N, C, rows, 1, 1))
            dx_name = 'dx_raw'
            # edge bias: histogram over the (fw, bw) pairs -> table MLP backward (graphormer.py:115-117)
            E = 'gnn.0.attn.edge_embed.embed.weight'
            W0e, b0e = 'gnn.0.attn.proj_e.0.weight', 'gnn.0.attn.proj_e.0.bias'
            W2e, b2e = 'gnn.0.attn.proj_e.2.weight', 'gnn.0.attn.proj_e.2.bias'
            p.wsf('hid', sv['hid'].numel())
            ins['hid'] = sv['hid']
            dT, dhid = p.wsf('dT', V * V * ldT), p.wsf('dhid', V * V * C)
            dPfw, dPbw = p.wsf('dPfw', V * C), p.wsf('dPbw', V * C)
            hist = (p.xbuf(p.X_WS), p.ws('dT_fix', 8 * V * V * H + 64))
            p.op(L.OP_BIAS_HIST, refs=(dT, dBias, i32('pair'), hist), ints=(B, N, H, V))
            p0 = p.gemm(dT, w('hid'), p.gref(W2e), H, C, V * V, ldT, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL, accum=True,
                        dbias=p.gref(b2e))
            p.gemm(dT, p.pref(W2e), dhid, V * V, C, H, ldT, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL)
            p.gemm_op(p0)
            p.op(L.OP_EDGE_HIDDEN_BWD, refs=(dPfw, dPbw, dhid, w('hid')), ints=(V, C))
            p0 = p.gemm(dPfw, p.pref(E, 2 * C), p.gref(W0e, 0), C, C, V, C, C, 2 * C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                        accum=True)
            p.gemm(dPbw, p.pref(E, 2 * C), p.gref(W0e, C), C, C, V, C, C, 2 * C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                   accum=True, dbias=p.gref(b0e))
            p.gemm(dPfw, p.pref(W0e, 0), p.gref(E, 2 * C), V, C, C, C, 2 * C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                   accum=True)
            p.gemm_op(p0)
            p.gemm_op(p.gemm(dPbw, p.pref(W0e, C), p.gref(E, 2 * C), V, C, C, C, 2 * C, C, a_mode=L.MODE_ROW,
                             b_mode=L.MODE_COL, accum=True))
        p.finish()
        gflat = torch.zeros(int(ghn._flat_numel), dtype=torch.float32, device=ghn.device)
        ws = _run(ghn, p, ins, gflat=gflat)
        dx = _read(ws, p, dx_name, (B, N, C)).reshape(ctx.x_shape)
        de = None
        if ctx.has_e and ctx.needs_input_grad[5]:
            de = _read(ws, p, 'dBias', (B, H, N, N)).permute(0, 2, 3, 1).contiguous()
        names = _layer_param_names(ghn, l)
        slot = {n: k for k, n in enumerate(param_names(ghn.layers, ghn.layernorm))}
        named = dict(ghn.named_parameters())
        grads = []
        for n in names:
            o_, prm = int(ghn._offs[slot[n]]), named[n]
            grads.append(gflat[o_:o_ + prm.numel()].view(prm.shape).clone())
        return (None, None, None, None, dx, de) + tuple(grads)



_DECODER_PARAMS = ('decoder.fc.0.weight', 'decoder.fc.0.bias', 'decoder.conv.0.weight', 'decoder.conv.0.bias',
                   'decoder.conv.2.weight', 'decoder.conv.2.bias', 'decoder.class_layer_predictor.1.weight',
                   'decoder.class_layer_predictor.1.bias')


def conv_decoder3_forward(ghn, x, max_shape=(1, 1, 1, 1), class_pred=False):
    """ConvDecoder3.forward (nn.py:735-762) on the HIP GEMMs: fc over the centre-cropped positions only, conv.0, conv.2
    restricted to the consumed rows (o' < max_shape[0], i' < max_shape[1]); with class_pred the classifier head on the
    centre position.  x: (n, C) node embeddings.  Returns (n, o, i, h, w) or (n, num_classes, i).  Differentiable w.r.t.
    x and the decoder's parameters when autograd is recording."""
    named = dict(ghn.named_parameters())
    params = [named[k] for k in _DECODER_PARAMS]
    if torch.is_grad_enabled() and (x.requires_grad or any(p_.requires_grad for p_ in params)):
        return _DecoderFunction.apply(ghn, tuple(int(v) for v in max_shape), bool(class_pred), x, *params)
    return _decoder_forward_impl(ghn, x, max_shape, class_pred)[0]


class _DecoderFunction(torch.autograd.Function):
    """autograd of ConvDecoder3.forward (nn.py:735-762): the decoder backward ops of Program._build_backward on the
    exact-fp32 GEMMs (dgrad with the ReLU masks of the saved activations, wgrad + fused bias gradients, the classifier
    head backward, the per-position fc backward)."""

    @staticmethod
    def forward(ctx, ghn, max_shape, class_pred, x, *params):
        out, sv = _decoder_forward_impl(ghn, x.detach(), max_shape, class_pred, save=True)
        ctx.ghn, ctx.sv = ghn, sv
        return out

    @staticmethod
    def backward(ctx, g):
        ghn, sv = ctx.ghn, ctx.sv
        n, hw, M, o, i, i_ld, ld, pos, class_pred = (sv[k] for k in ('n', 'hw', 'M', 'o', 'i', 'i_ld', 'ld', 'pos', 'class_pred'))
        C, ms, K = ghn.hid, ghn.max_shape, ghn.num_classes
        S2 = ms[2] * ms[3]
        p = MiniProgram(ghn)
        w = p.wref
        ins = {}
        for name in ('xe', 't', 'u', 'tiles'):
            p.wsf(name, sv[name].numel())
            ins[name] = sv[name]
        d_tiles = p.wsf('d_tiles', M * ld)
        Wfc, bfc = 'decoder.fc.0.weight', 'decoder.fc.0.bias'
        W0, b0 = 'decoder.conv.0.weight', 'decoder.conv.0.bias'
        W2, b2 = 'decoder.conv.2.weight', 'decoder.conv.2.bias'
        Wc, bc = 'decoder.class_layer_predictor.1.weight', 'decoder.class_layer_predictor.1.bias'
        g = g.to(torch.float32)
        if class_pred:
            ldK = round_up(K, 4)
            centre = sv['centre']
            d_cls = p.wsf('d_cls', n * i_ld * ldK)
            dc = torch.zeros(n, i_ld, ldK, device=g.device)
            dc[:, :i, :K] = g.permute(0, 2, 1)                     # (n, K, i) -> (n, i, K)
            ins['d_cls'] = dc
            # d relu(tile)[o'][i'] = sum_k dout[i'][k] Wcls[k][o'], masked by tile > 0; then the head's own gradients
            p0 = len(p._probs)
            for node in range(n):
                off = (node * hw + centre) * ld
                p.gemm(p.pref(Wc), (d_cls[0], d_cls[1] + 4 * node * i_ld * ldK), (d_tiles[0], d_tiles[1] + 4 * off),
                       ms[0], i, K, ms[0], ldK, i_ld, a_mode=L.MODE_COL, b_mode=L.MODE_ROW, dact=L.DACT_RELU,
                       aux_in=(w('tiles')[0], w('tiles')[1] + 4 * off))
            p.gemm_op(p0)
            for node in range(n):
                off = (node * hw + centre) * ld
                p.gemm_op(p.gemm((d_cls[0], d_cls[1] + 4 * node * i_ld * ldK), (w('tiles')[0], w('tiles')[1] + 4 * off),
                                 p.gref(Wc), K, ms[0], i, ldK, i_ld, ms[0], a_mode=L.MODE_COL, b_mode=L.MODE_ROW,
                                 accum=True, dbias=p.gref(bc)))
        else:
            dt2 = torch.zeros(n, hw, o, i_ld, device=g.device)
            dt2[:, :, :, :i] = g.reshape(n, o, i, hw).permute(0, 3, 1, 2)
            dt = torch.zeros(n, hw, ld, device=g.device)
            dt[:, :, :o * i_ld] = dt2.reshape(n, hw, o * i_ld)
            ins['d_tiles'] = dt
        d_u, d_t = p.wsf('d_u', M * 8 * C), p.wsf('d_t', M * 4 * C)
        d_xe = p.wsf('d_xe', n * C)
        # conv.2: d_u = (d_tiles . W2sub) * (u > 0);  dW2 rows (o', i') += d_tiles^T u, bias gradient fused
        p.gemm_op(p.gemm(d_tiles, p.pref(W2), d_u, M, 8 * C, o * i_ld, ld, 8 * C, 8 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                         b_qs=(i_ld, ms[1]), dact=L.DACT_RELU, aux_in=w('u')))
        p.gemm_op(p.gemm(d_tiles, w('u'), p.gref(W2), o * i_ld, 8 * C, M, ld, 8 * C, 8 * C, a_mode=L.MODE_COL,
                         b_mode=L.MODE_COL, c_qs=(i_ld, ms[1]), accum=True, dbias=p.gref(b2)))
        # conv.0
        p.gemm_op(p.gemm(d_u, p.pref(W0), d_t, M, 4 * C, 8 * C, 8 * C, 4 * C, 4 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                         dact=L.DACT_RELU, aux_in=w('t')))
        p.gemm_op(p.gemm(d_u, w('t'), p.gref(W0), 8 * C, 4 * C, M, 8 * C, 4 * C, 4 * C, a_mode=L.MODE_COL,
                         b_mode=L.MODE_COL, accum=True, dbias=p.gref(b0)))
        # fc, per used grid position (row subset ch * 256 + q of Wfc): d_xe += d_t[rows of q] Wfc_q ; dWfc_q += d_t^T xe
        for k, q in enumerate(pos):
            rws = p.idx((np.arange(n, dtype=np.int32) * hw + k).astype(np.int32))
            p.gemm_op(p.gemm(d_t, p.pref(Wfc, q * C), d_xe, n, C, 4 * C, 4 * C, S2 * C, C, a_mode=L.MODE_ROW,
                             b_mode=L.MODE_COL, a_gather=rws, accum=True))
            p.gemm_op(p.gemm(d_t, w('xe'), p.gref(Wfc, q * C), 4 * C, C, n, 4 * C, C, S2 * C, a_mode=L.MODE_COL,
                             b_mode=L.MODE_COL, a_gather=rws, accum=True, dbias=p.gref(bfc, q), dbias_stride=S2))
        p.finish()
        gflat = torch.zeros(int(ghn._flat_numel), dtype=torch.float32, device=ghn.device)
        ws = _run(ghn, p, ins, gflat=gflat)
        dx = _read(ws, p, 'd_xe', (n, C))
        slot = {nm: k for k, nm in enumerate(param_names(ghn.layers, ghn.layernorm))}
        named = dict(ghn.named_parameters())
        grads = []
        for nm in _DECODER_PARAMS:
            o_, prm = int(ghn._offs[slot[nm]]), named[nm]
            grads.append(gflat[o_:o_ + prm.numel()].view(prm.shape).clone())
        return (None, None, None, dx) + tuple(grads)


def _decoder_forward_impl(ghn, x, max_shape=(1, 1, 1, 1), class_pred=False, save=False):
    n, C = x.shape
    assert C == ghn.hid
    ms = ghn.max_shape
    S, K = ms[2], ghn.num_classes
    S2 = S * S
    kh, kw = int(max_shape[2]), int(max_shape[3])
    if min(kh, kw) > S:
        raise NotImplementedError('kernels larger than the %dx%d decoder grid are resized inside GHN3.forward' % (S, S))
    o = ms[0] if class_pred else min(int(max_shape[0]), ms[0])
    i = min(int(max_shape[1]), ms[1])
    h, w = min(kh, S), min(kw, S)
    half = S // 2
    y0, x0 = max(0, half - h // 2), max(0, half - w // 2)
    pos = [(y0 + a) * S + (x0 + b) for a in range(h) for b in range(w)]
    hw = len(pos)
    M = n * hw
    i_ld = round_up(i, 4)
    p = MiniProgram(ghn)
    xe = p.wsf('xe', n * C)
    t, u = p.wsf('t', M * 4 * C), p.wsf('u', M * 8 * C)
    ld = round_up(o * i_ld, 4)
    tiles = p.wsf('tiles', M * ld)
    Wfc, bfc = 'decoder.fc.0.weight', 'decoder.fc.0.bias'
    W0, b0 = 'decoder.conv.0.weight', 'decoder.conv.0.bias'
    W2, b2 = 'decoder.conv.2.weight', 'decoder.conv.2.bias'
    p0 = len(p._probs)
    for k, q in enumerate(pos):                                  # fc: one row subset of Wfc per used grid position
        rws = (np.arange(n, dtype=np.int32) * hw + k).astype(np.int32)
        p.gemm(xe, p.pref(Wfc, q * C), t, n, 4 * C, C, C, S2 * C, 4 * C, bias=p.pref(bfc, q), bias_stride=S2,
               act=L.ACT_RELU, c_gather=p.idx(rws))
    p.gemm_op(p0)
    p.gemm_op(p.gemm(t, p.pref(W0), u, M, 8 * C, 4 * C, 4 * C, 4 * C, 8 * C, bias=p.pref(b0), act=L.ACT_RELU))
    p.gemm_op(p.gemm(u, p.pref(W2), tiles, M, o * i_ld, 8 * C, 8 * C, 8 * C, ld, b_qs=(i_ld, ms[1]), bias=p.pref(b2),
                     bias_q=i_ld, bias_s=ms[1], act=L.ACT_RELU if class_pred else L.ACT_NONE))
    if class_pred:
        assert h == w, ('require squared weights at this point', (h, w))
        ldK = round_up(K, 4)
        centre = (h // 2) * w + (w // 2)
        cls = p.wsf('cls', n * i_ld * ldK)
        p0 = len(p._probs)
        for node in range(n):                                    # out[i'][k] = sum_o' relu(tile[o'][i']) Wcls[k][o'] + b
            p.gemm((tiles[0], tiles[1] + 4 * (node * hw + centre) * ld),
                   p.pref('decoder.class_layer_predictor.1.weight'), (cls[0], cls[1] + 4 * node * i_ld * ldK),
                   i, K, ms[0], i_ld, ms[0], ldK, a_mode=L.MODE_COL,
                   bias=p.pref('decoder.class_layer_predictor.1.bias'))
        p.gemm_op(p0)
    p.finish()
    ws = _run(ghn, p, {'xe': x.to(ghn.device, torch.float32)})
    saved = None
    if save:
        saved = dict(n=n, hw=hw, M=M, o=o, i=i, i_ld=i_ld, ld=ld, pos=pos, class_pred=class_pred,
                     centre=(h // 2) * w + (w // 2))
        for name, cnt in (('xe', n * C), ('t', M * 4 * C), ('u', M * 8 * C), ('tiles', M * ld)):
            saved[name] = _read(ws, p, name, (cnt,))
    if class_pred:
        out = _read(ws, p, 'cls', (n, i_ld, round_up(K, 4)))[:, :i, :K]
        return out.permute(0, 2, 1).contiguous(), saved           # (n, num_classes, in)
    tl = _read(ws, p, 'tiles', (n, hw, ld))[:, :, :o * i_ld].reshape(n, hw, o, i_ld)[:, :, :, :i]
    return tl.permute(0, 2, 3, 1).reshape(n, o, i, h, w), saved
