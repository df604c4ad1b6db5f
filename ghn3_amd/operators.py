"""
Operator-level API of the reference on the HIP path (SURVEY 8(b) "internal operator API"):

    TransformerLayer.forward(x, edges, mask) -> (x, edges, mask) | x        /root/reference/ghn3/graphormer.py:208-248
    ConvDecoder3.forward(x, max_shape, class_pred)                            /root/reference/ghn3/nn.py:735-762

`GHN3.forward` never calls these (it runs ONE compiled op program for the whole model); they exist so that code written
against the reference's modules -- ``ghn.gnn[l](x, edges, mask)``, ``ghn.decoder(x, max_shape, class_pred)`` -- runs
on the same kernels through the same C ABI: each call compiles a small op program (the ops of that layer / decoder
only, exact-fp32 path), runs it with ``ghn3_run`` and returns torch tensors.  Forward only: gradients of the GHN flow
through GHN3.forward's backward program, not through these calls.
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


def _run(ghn, mini, inputs, edges=None):
    """inputs: {workspace name: tensor} copied into the zeroed workspace before the run.  Returns the workspace."""
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
    matrix `edges` (B, N, N); later layers take the edge bias (B, N, N, H) layer 0 returned."""
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
    p.op(L.OP_LAYERNORM_FWD, refs=(h1, x_in, p.pref(pre + 'ln1.weight'), p.pref(pre + 'ln1.bias'), m1, r1),
         ints=(rows, C), floats=(1e-5,))
    p.gemm_op(p.gemm(h1, p.pref(pre + 'attn.to_qkv.weight'), qkv, rows, 3 * C, C, C, C, 3 * C))
    p.op(L.OP_ATTN_FWD, refs=(o, qkv, bias, p.NONE, r_nn), ints=(B, N, C, H))
    p.gemm_op(p.gemm(o, p.pref(pre + 'attn.to_out.0.weight'), xmid, rows, C, C, C, C, C,
                     bias=p.pref(pre + 'attn.to_out.0.bias'), residual=x_in))
    p.op(L.OP_LAYERNORM_FWD, refs=(h2, xmid, p.pref(pre + 'ln2.weight'), p.pref(pre + 'ln2.bias'), m1, r1),
         ints=(rows, C), floats=(1e-5,))
    p.gemm_op(p.gemm(h2, p.pref(pre + 'ff.net.0.weight'), f, rows, 4 * C, C, C, C, 4 * C,
                     bias=p.pref(pre + 'ff.net.0.bias'), act=L.ACT_GELU))
    p.gemm_op(p.gemm(f, p.pref(pre + 'ff.net.3.weight'), x_out, rows, C, 4 * C, 4 * C, 4 * C, C,
                     bias=p.pref(pre + 'ff.net.3.bias'), residual=xmid))
    p.finish()
    ws = _run(ghn, p, ins, edges=edges_dev)
    y = _read(ws, p, 'x_out', (B, N, C))
    if len(sz) == 2:
        y = y[0]
    if not return_edges:
        return y
    e_out = _read(ws, p, 'bias', (B, H, N, N)).permute(0, 2, 3, 1) if layer0 else edges
    return y, e_out, mask


def conv_decoder3_forward(ghn, x, max_shape=(1, 1, 1, 1), class_pred=False):
    """ConvDecoder3.forward (nn.py:735-762) on the HIP GEMMs: fc over the centre-cropped positions only, conv.0, conv.2
    restricted to the consumed rows (o' < max_shape[0], i' < max_shape[1]); with class_pred the classifier head on the
    centre position.  x: (n, C) node embeddings.  Returns (n, o, i, h, w) or (n, num_classes, i)."""
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
    if class_pred:
        out = _read(ws, p, 'cls', (n, i_ld, round_up(K, 4)))[:, :i, :K]
        return out.permute(0, 2, 1).contiguous()                  # (n, num_classes, in)
    tl = _read(ws, p, 'tiles', (n, hw, ld))[:, :, :o * i_ld].reshape(n, hw, o, i_ld)[:, :, :, :i]
    return tl.permute(0, 2, 3, 1).reshape(n, o, i, h, w)
