"""
GHN-3 model on MI355X: host-side mirror of /root/reference/ghn3/nn.py (``from_pretrained``, ``GHN3``) whose
forward/backward run as hand-written HIP kernels through libghn3_hip.so.

Same constructor arguments, ``forward`` signature, state-dict key layout and assignment semantics as the
reference (nn.py:31-125, 128-351, 508-552), so it drops into the train_ghn_ddp.py / eval_ghn.py call
sequences.  The module holds parameters only; no torch op touches them on the hot path.  There is no CPU
path: calling ``forward`` without a HIP device raises.
"""

import os
import math
import copy
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import bookkeeping as bk
from .graph import Graph, GraphBatch
from .program import Program, param_names


def log(*args, **kwargs):
    rank = int(os.environ.get('RANK', '0'))
    if rank == 0:
        print(*args, **kwargs)


# ---------------------------------------------------------------------------------------------------
# the reference's module tree: parameter containers (state-dict names) whose forward() runs on the HIP kernels
# ---------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    pass


class TransformerLayer(nn.Module):
    """Graphormer layer (graphormer.py:144-248): ``forward(x, edges, mask) -> (x, edges, mask)`` if return_edges else x.
    GHN3.forward never calls it (the whole model is one compiled op program); the call compiles and runs the ops of
    this layer alone through the same C ABI (ghn3_amd/operators.py), forward only."""

    def __init__(self, dim, heads, layer0, mlp_ratio=4, return_edges=True):
        super().__init__()
        self.dim, self.num_heads, self.edge_dim, self.return_edges = dim, heads, (2 if layer0 else 0), return_edges
        self.ln1 = nn.LayerNorm(dim)
        self.attn = _Holder()
        self.attn.to_qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.attn.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Identity())
        if layer0:
            self.attn.edge_embed = _Holder()
            self.attn.edge_embed.embed = nn.Embedding(257, dim)             # graphormer.py:95-96
            self.attn.proj_e = nn.Sequential(nn.Linear(2 * dim, dim), nn.ReLU(), nn.Linear(dim, heads))
        self.ln2 = nn.LayerNorm(dim)
        self.ff = _Holder()
        self.ff.net = nn.Sequential(nn.Linear(dim, mlp_ratio * dim), nn.GELU(), nn.Identity(),
                                    nn.Linear(mlp_ratio * dim, dim), nn.Identity())
        self.max_degree, self.max_input_dist = 100, 1000                      # graphormer.py:196-197
        self._index = 0

    def forward(self, x, edges=None, mask=None):
        from .operators import transformer_layer_forward
        ghn = self.__dict__.get('_owner')                # (set by GHN3 past nn.Module.__setattr__: not a sub-module)
        if ghn is None:
            raise L.Ghn3Error('TransformerLayer.forward needs the GHN3 that owns the layer (flat parameter buffer)')
        return transformer_layer_forward(ghn, self._index, x, edges, mask, return_edges=self.return_edges)


def _graphormer_layer(dim, heads, layer0, mlp_ratio=4, return_edges=True):
    return TransformerLayer(dim, heads, layer0, mlp_ratio, return_edges)


class ConvDecoder3(nn.Module):
    """nn.py:716-762: fc -> centre crop -> conv.0 -> conv.2 (-> class_layer_predictor).  ``forward(x, max_shape,
    class_pred)`` runs the decoder GEMMs of ghn3_amd/operators.py (forward only; inside GHN3.forward the decoders are
    part of the compiled op program)."""

    def __init__(self, in_features, hid, out_shape, num_classes):
        super().__init__()
        self.out_shape = out_shape
        self.num_classes = num_classes
        s2 = int(out_shape[2] * out_shape[3])
        self.fc = nn.Sequential(nn.Linear(in_features, hid[0] * s2), nn.ReLU())
        self.conv = nn.Sequential(nn.Linear(hid[0], hid[1]), nn.ReLU(),
                                  nn.Linear(hid[1], int(out_shape[0] * out_shape[1])), nn.Identity())
        self.class_layer_predictor = nn.Sequential(nn.ReLU(), nn.Linear(out_shape[0], num_classes))

    def forward(self, x, max_shape=(1, 1, 1, 1), class_pred=False):
        from .operators import conv_decoder3_forward
        ghn = self.__dict__.get('_owner')
        if ghn is None:
            raise L.Ghn3Error('ConvDecoder3.forward needs the GHN3 that owns the decoder (flat parameter buffer)')
        return conv_decoder3_forward(ghn, x, max_shape, class_pred)


class SequentialMultipleInOut(nn.Sequential):
    """Sequence of Graphormer layers threading (x, edges, mask) (nn.py:765-780)."""

    def forward(self, input, *args):
        for module in self:
            input = module(*input) if isinstance(input, tuple) else module(input, *args)
        return input


# ---------------------------------------------------------------------------------------------------
class _PinnedRing:
    """Reusable page-locked staging slots for host -> device uploads.  `upload(array, device)` copies the array into a
    slot and issues an asynchronous copy on the current stream; a slot is reused once its copy has completed.  (Pinning a
    fresh buffer per upload -- Tensor.pin_memory() -- costs a hipHostMalloc each time: milliseconds; a pageable source
    makes the copy synchronous.)  ONE arena allocated at first use, two size classes: small tensors (graph fields) and
    index blobs (a few MB); anything larger than a slot falls back to a synchronous copy."""
    SMALL, BIG = 1 << 16, 12 << 20

    def __init__(self, small_slots=32, big_slots=6):
        self.layout = [(self.SMALL, small_slots), (self.BIG, big_slots)]
        self.arena = None
        self.next = [0, 0]
        self.events = [[None] * small_slots, [None] * big_slots]

    def upload(self, array, device):
        if isinstance(array, torch.Tensor):
            if array.is_cuda:
                return array.to(device, non_blocking=True)
            array = array.numpy()
        # (numpy for the host-side copy: torch's CPU ops wake a thread pool -- tens of milliseconds on many-core hosts)
        host = np.ascontiguousarray(array)
        flat = host.reshape(-1).view(np.uint8)
        if flat.size > self.BIG or os.environ.get('GHN3_PINNED_UPLOAD', '1') == '0':
            return torch.from_numpy(host).to(device)
        if self.arena is None:
            self.arena = torch.empty(sum(sz * n for sz, n in self.layout), dtype=torch.uint8).pin_memory()
        c = 0 if flat.size <= self.SMALL else 1
        k = self.next[c]
        self.next[c] = (k + 1) % self.layout[c][1]
        if self.events[c][k] is not None:
            self.events[c][k].synchronize()
        base = (0 if c == 0 else self.layout[0][0] * self.layout[0][1]) + k * self.layout[c][0]
        stage = self.arena[base:base + flat.size]
        np.copyto(stage.numpy(), flat)
        out = stage.to(device, non_blocking=True).view(torch.from_numpy(host[:0].reshape(-1)).dtype).view(host.shape)
        ev = self.events[c][k] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events[c][k] = ev
        return out


_RINGS = {}


def pinned_ring(device):
    key = torch.device(device).index or 0
    if key not in _RINGS:
        _RINGS[key] = _PinnedRing()
    return _RINGS[key]


class _Plan:
    """A compiled batch: Program + device-resident index blob + workspace."""

    def __init__(self, ghn, program, edges, nets):
        self.program = program
        self.edges = edges
        self.nets = nets
        dev = ghn.device
        # (reusable pinned staging + asynchronous copy in stream order: a pageable .to(device) blocks the calling thread for
        # the copy AND for the work queued on the stream before it -- 0.8 ms each, five per plan, in the loop that feeds
        # the GPU.  A separate upload stream was measured and dropped: buffers allocated on it come from another pool of
        # the caching allocator, and the 2.3 GB workspace of every plan then costs a hipMalloc.)
        self.idx = pinned_ring(dev).upload(program.idx_blob, dev)
        # The padding of the 16-bit operand copies (Program.ws16 regions: 26 % of the 2.4 GB at ghn3xlm16 / 256 nodes) must read
        # as finite zeros: those regions are zero-filled once per plan; every other region is written before it is read (or
        # zeroed by an op of the programs) and starts uninitialised -- a NEW architecture every step paid 0.4 ms of GPU time
        # for the full fill.  GHN3_WS_ZERO_ALL=1: the full fill; GHN3_WS_POISON=1 (tests): NaN bytes everywhere else.
        regions = getattr(program, 'ws_zero', None)
        self.ws_dout_ready = regions is None or os.environ.get('GHN3_WS_ZERO_ALL', '0') == '1'
        self.zero_fill_bytes = int(program.ws_bytes + program.scal_bytes)     # (what this plan zero-fills: see bench.py)
        if self.ws_dout_ready:
            self.ws = torch.zeros(program.ws_bytes, dtype=torch.uint8, device=dev)
        else:
            self.ws = torch.empty(program.ws_bytes, dtype=torch.uint8, device=dev)
            if os.environ.get('GHN3_WS_POISON', '0') == '1' or getattr(ghn, 'ws_poison', False):
                self.ws.fill_(0xff)
            merged = []
            for off, n in sorted(regions):
                if merged and off <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], off + n)
                else:
                    merged.append([off, off + n])
            for a, b in merged:
                self.ws[a:min(b, program.ws_bytes)].zero_()
            self.zero_fill_bytes = int(sum(min(b, program.ws_bytes) - a for a, b in merged) + program.scal_bytes)
        self.scal = torch.zeros(program.scal_bytes, dtype=torch.uint8, device=dev)
        self.bufs = np.zeros(program.n_bufs, dtype=np.uint64)
        self.sizes = [p['numel'] for p in program.predicted]
        self.tok = None
        self.out = None


class _GHN3Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ghn, plan, *params):
        out = ghn._run_forward(plan)
        ctx.ghn, ctx.plan = ghn, plan
        return out

    @staticmethod
    def backward(ctx, dout):
        # (the placeholder _ParamNormLoss returned -- recognised by IDENTITY of its one-float storage, not by its strides: a
        # genuine broadcast gradient such as flat.sum().backward() has the same layout -- means the norm term is the only
        # upstream gradient and the tile backward forms it from the predicted values: nothing to read)
        plan = ctx.plan
        norm_g, plan.norm_g = getattr(plan, 'norm_g', None), None
        ph, plan.norm_placeholder = getattr(plan, 'norm_placeholder', None), None
        if dout is not None and ph is not None and dout.dim() == 1 and dout.stride(0) == 0 and \
                dout.data_ptr() == ph.data_ptr():
            dout = None
        grads = ctx.ghn._run_backward(plan, None if dout is None else dout.contiguous(), reducer=ctx.ghn.grad_reducer,
                                      norm_g=norm_g)
        # The plan references the target modules (to assign into), the modules hold the predicted tensors and those
        # reference this node: dropping the plan here breaks the cycle, so a step's buffers (4 GB of workspace + 2.6 GB
        # of gradients at ghn3xlm16) are freed by reference counting instead of waiting for Python's cycle collector.
        ctx.plan = ctx.ghn = None
        return (None, None) + tuple(grads)


class _ParamNormLoss(torch.autograd.Function):
    """sum_t ||p_t||_F over the predicted tensors of a plan (the reference's predparam_wd term, trainer.py:97-98,
    288-294), fused into the tile kernels (round 4): the tile forward left every work block's sum of squares, so the norms
    are two tiny launches (GHN3_OP_PARAM_NORM_FIN) and the gradient  g p / ||p||  is formed inside the tile backward from
    the predicted values -- no pass over the 346 MB output in either direction (rounds 1-3: GHN3_OP_PARAM_NORM_FWD / BWD,
    0.24 ms per step).  Plans compiled without a backward program keep the streaming ops."""

    @staticmethod
    def forward(ctx, flat, ghn, plan):
        prog = plan.program
        ghn._fill_bufs(plan, out=flat)
        stream = torch.cuda.current_stream().cuda_stream
        ctx.fused = prog.training and getattr(prog, 'tile_bwd_op', None) is not None
        if ctx.fused:
            ghn._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, stream)
            ctx.b_ops = None
        else:
            f_ops, ctx.b_ops = prog.norm_ops(1.0)
            ghn._ctx().run(f_ops, prog.problems, plan.bufs, stream)
        ctx.ghn, ctx.plan = ghn, plan
        ctx.save_for_backward(flat)
        return plan.scal[:4].view(torch.float32)[0].clone()

    @staticmethod
    def backward(ctx, g):
        (flat,) = ctx.saved_tensors
        ghn, plan = ctx.ghn, ctx.plan
        if ctx.fused:
            # hand the (device-side) weight of the term to the GHN's backward and return a stride-0 zero gradient: the
            # autograd engine still runs _GHN3Function.backward, which recognises the placeholder
            plan.norm_g = g.detach().to(torch.float32).reshape(1).contiguous()
            plan.norm_placeholder = flat.new_zeros(1)
            ctx.ghn = ctx.plan = None
            return plan.norm_placeholder.expand(flat.numel()), None, None
        dflat = torch.empty_like(flat)
        ghn._fill_bufs(plan, out=flat, dout=dflat)
        ghn._ctx().run(ctx.b_ops, plan.program.problems, plan.bufs, torch.cuda.current_stream().cuda_stream)
        dflat.mul_(g)                          # (the upstream gradient lives on the device: no host read)
        ctx.ghn = ctx.plan = ctx.b_ops = None
        return dflat, None, None


class GHN3(nn.Module):
    r"""
    Transformer-based Graph HyperNetwork (GHN-3); see the reference class at nn.py:128.

    Extra keyword arguments (not in the reference):
      index_mode   'reference' (default; reproduces quirk Q1 of SURVEY 3.2 for B > 1) or 'correct'
      compute      MFMA operand type of the decoder GEMMs (fc / W0 / W2, forward and backward):
                   'f32' (default: exact fp32 MFMA), 'f16' or 'bf16' (fp32 accumulate).
                   The Graphormer and the small heads always run exact fp32.
      compute_bwd  operand type of the decoder backward GEMMs in 16-bit mode (default: same as `compute`; f16
                   gradient copies are scaled by a power of two derived from their running max, so the ~1e-6
                   upstream gradients keep 11 significant bits; 'bf16' trades precision for range without scaling)
      side_stream  True (default): weight gradients, LayerNorm parameter gradients and operand copies overlap with
                   the dependent chain of the program on a second HIP stream
      direct16     True (default): in 16-bit mode the W2 GEMMs read 16-bit operand copies (GHN3_OP_CAST16) through the
                   LDS-DMA kernel; False: fp32 operands converted while staged
      graphormer_x3  None (default): in the 16-bit modes the Graphormer linears multiply split-bf16 operands on the 16-bit
                   matrix cores (hi.hi + hi.lo + lo.hi, ~1e-5 relative; GHN3_GEMM_X3); False: exact fp32 everywhere
    """

    def __init__(self, max_shape, num_classes, hid, heads=8, layers=3, is_ghn2=False, pretrained=False, **kwargs):
        super().__init__()
        if is_ghn2:
            raise NotImplementedError('GHN-2 (GatedGNN) checkpoints are outside the GHN-3 hot path')
        kwargs.pop('act_layer', None)
        hypernet = kwargs.pop('hypernet', 'gatedgnn')
        decoder = kwargs.pop('decoder', 'conv')
        assert decoder == 'conv', decoder
        self.weight_norm = kwargs.pop('weight_norm', False)
        self.ve = kwargs.pop('ve', False)
        self.layernorm = kwargs.pop('layernorm', False)
        self.debug_level = kwargs.pop('debug_level', 0)
        self.index_mode = kwargs.pop('index_mode', 'reference')
        self.compute = kwargs.pop('compute', 'f32')
        self.compute_bwd = kwargs.pop('compute_bwd', None)
        self.direct16 = kwargs.pop('direct16', True)
        self.graphormer_x3 = kwargs.pop('graphormer_x3', None)
        self.side_stream = kwargs.pop('side_stream', True)
        assert not kwargs, 'unknown arguments %s' % list(kwargs)
        assert len(max_shape) == 4, max_shape
        self.max_shape = tuple(int(v) for v in max_shape)
        self.num_classes = num_classes
        self.hid, self.heads, self.layers = hid, heads, layers
        self._is_ghn2 = False

        if self.layernorm:
            self.ln = nn.LayerNorm(hid)
        self.embed = nn.Embedding(len(bk.PRIMITIVES_DEEPNETS1M), hid)
        vocab = bk.ShapeVocab(num_classes, self.max_shape)
        self.shape_enc = _Holder()
        self.shape_enc.embed_spatial = nn.Embedding(vocab.n_sp + 1, hid // 4)
        self.shape_enc.embed_channel = nn.Embedding(vocab.n_ch + 1, hid // 4)
        self.gnn = SequentialMultipleInOut(*[_graphormer_layer(hid, heads, layer0=(l == 0),
                                                               return_edges=(l < layers - 1))      # nn.py:154
                                             for l in range(layers)])
        for l, layer in enumerate(self.gnn):
            layer._index = l
            object.__setattr__(layer, '_owner', self)    # (plain attribute: the operator-level forwards need the model)
        self.gnn[0].centrality_embed_in = nn.Embedding(101, hid)
        self.gnn[0].centrality_embed_out = nn.Embedding(101, hid)
        self.gnn[0].input_dist_embed = nn.Embedding(1001, hid)
        self.decoder = ConvDecoder3(hid, (hid * 4, hid * 8), self.max_shape, num_classes)
        object.__setattr__(self.decoder, '_owner', self)
        max_ch = max(self.max_shape[:2])
        self.decoder_1d = _Holder()
        self.decoder_1d.fc = nn.Sequential(nn.Linear(hid, hid * 2), nn.ReLU(), nn.Linear(hid * 2, 2 * max_ch),
                                           nn.Identity())
        self.bias_class = nn.Sequential(nn.ReLU(), nn.Linear(max_ch, num_classes))
        # nn.py:167-170: small init of the last decoder layers, transformer-style embedding init
        for m in (self.decoder_1d.fc[-2], self.decoder.conv[-2], self.decoder.class_layer_predictor[-1]):
            m.weight.data /= 5.0
            m.bias.data *= 0
        for m in self.modules():
            if isinstance(m, nn.Embedding):
                nn.init.trunc_normal_(m.weight.data, std=m.weight.shape[1] ** (-0.5))
        self._names = param_names(layers, self.layernorm)
        self.last_plan = self._last_flat = None
        # data parallel: set to a ddp_utils.FlatGradReducer and loss.backward() all-reduces the flat gradient buffer
        # itself, overlapped with the Graphormer backward (instead of DistributedDataParallel's bucket copies)
        self.grad_reducer = None
        self._flat = None
        self._flatten()

    # ------------------------------------------------------------------ parameter storage
    def _slot_params(self):
        cached = self.__dict__.get('_slot_cache')
        if cached is None:
            named = dict(self.named_parameters())
            cached = [named[n] for n in self._names]
            self.__dict__['_slot_cache'] = cached         # (the Parameter objects: stable until modules are replaced)
        return cached

    def _flatten(self):
        """All parameters become views of one flat fp32 buffer (slot order, 64-float aligned)."""
        self.__dict__['_slot_cache'] = None
        ps = self._slot_params()
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64
        dev = ps[0].device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)      # (alignment gaps stay zero: AdamW walks them)
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).to(torch.float32))
            p.data = flat[o:o + p.numel()].view(p.shape)
        self._flat, self._offs, self._flat_numel = flat, np.asarray(offs, dtype=np.int64), total
        # split sizes of a flat buffer into (parameter, alignment gap) pieces: the per-parameter views in one call
        self._split_sizes = []
        for k, p in enumerate(ps):
            nxt = offs[k + 1] if k + 1 < len(offs) else total
            self._split_sizes += [p.numel(), nxt - offs[k] - p.numel()]
        self._plans = {}
        self._shadow = None                       # 16-bit copies of the decoder weights (Program.shadow_layout)
        self._shadowed = None
        self._shadow_state = None                 # (parameter version, has the transposed copies) they were cast from
        self._shadow_w2_state = None              # the same for the W2 copies alone, when an optimizer step wrote them
        self._param_epoch = getattr(self, '_param_epoch', 0) + 1

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._flatten()                                   # (the copy's parameters become views of ITS flat buffer)
        return new

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    def load_state_dict(self, state_dict, strict=True):
        sd = dict(state_dict)
        # HF checkpoints keep the three layer-0 embeddings at the top level (nn.py:87,174-184)
        for k in ('centrality_embed_in', 'centrality_embed_out', 'input_dist_embed'):
            if k + '.weight' in sd:
                sd['gnn.0.' + k + '.weight'] = sd.pop(k + '.weight')
        return super().load_state_dict(sd, strict=strict)

    def params_changed(self):
        """Call after writing parameters behind torch's back (through ``p.data`` / raw pointers): the 16-bit shadow
        copies of the decoder weights are re-cast by the next forward.  In-place torch ops on the parameters
        (optimizers, ``load_state_dict``, ``.to()``) and ``FusedAdamW.step`` are tracked automatically."""
        self._param_epoch += 1

    def _shadow_version(self):
        w2, w0 = self.decoder.conv[2].weight, self.decoder.conv[0].weight
        if getattr(self, '_shadowed', None) is None:
            self._shadowed = [w2, w0, self.decoder.fc[0].weight] + [p for n, p in self.named_parameters()
                                         if n.startswith('gnn.') and n.endswith('.weight') and p.dim() == 2
                                         and ('to_qkv' in n or 'to_out' in n or 'ff.net' in n)]
        return (self._param_epoch, sum(p._version for p in self._shadowed), w2.data_ptr())

    def _refresh_shadows(self, plan, stream):
        """Replays Program.shadow_ops (fp32 -> 16-bit copies of W2, W2^T, W0^T) when the decoder weights changed since
        the copies were written; otherwise every forward / backward reuses them."""
        prog = plan.program
        if not prog.uses_shadow:
            return
        if self._shadow is None:
            nbytes = prog.shadow_layout(prog.C, prog.max_shape, prog.Lyr)['nbytes']
            self._shadow = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        plan.bufs[prog.xbuf(prog.X_SHADOW)] = self._shadow.data_ptr()
        ver = self._shadow_version()
        types = (prog.decoder_ctype, prog.decoder_bwd_ctype, prog.x3, prog.uses_op16)
        st = self._shadow_state
        if st is not None and st[0] == ver and st[2] == types and (st[1] or not prog.training):
            return
        # the W2 copies may already be current: a fused optimizer step wrote them while it updated W2 (FusedAdamW.step with
        # a plan; GHN3_OP_ADAMW_CAST16) -- then only the other copies (W0^T, the Graphormer linears) are re-cast
        w2 = getattr(self, '_shadow_w2_state', None)
        w2_current = w2 is not None and w2 == (ver, bool(prog.training), types)
        self._ctx().run(prog.shadow_ops_rest if w2_current else prog.shadow_ops, prog.problems, plan.bufs, stream)
        self._shadow_state = (ver, bool(prog.training), types)

    def fix_embed_layers(self):
        return  # the embeddings already live under gnn.0 (nn.py:174-184)

    def is_dense(self):
        return True

    @property
    def device(self):
        return self.embed.weight.device

    # ------------------------------------------------------------------ compile / run
    def compile(self, nets, graphs, predict_class_layers=True, reduce_graph=False, training=None):
        """Host bookkeeping for one batch (nn.py:242 _map_net_params + all shape logic) -> _Plan."""
        dev = self.device
        if dev.type != 'cuda':
            raise L.Ghn3Error('GHN3 runs on an MI355X only: move the model with .to("cuda") (no CPU path; the '
                              'CPU oracle under oracle/ is test infrastructure)')
        if isinstance(graphs, Graph) or isinstance(graphs, (list, tuple)):
            graphs = GraphBatch([graphs] if isinstance(graphs, Graph) else list(graphs), dense=True)
        if not graphs.on_device(dev):
            graphs.to_device(dev)
        assert graphs.dense, 'GraphBatch must be created with dense=True for GHN-3'
        training = self.training if training is None else training
        args = dict(self.program_config(), training=bool(training), predict_class_layers=bool(predict_class_layers),
                    reduce_graph=bool(reduce_graph))
        prog = graphs.take_program(nets, **args)             # (built by a loader worker: GraphBatch.precompile)
        if prog is None:
            cfg = args.pop('cfg')
            prog = Program(cfg, graphs.node_info, graphs.host_n_nodes(), graphs._node_type_host, graphs.max_edge, nets, **args)
        return self.plan(prog, graphs, nets)

    def plan(self, prog, graphs, nets):
        """Device half of compile(): wraps a host-compiled Program (which a loader worker may have built in another
        process -- it is plain numpy and pickles together with its graphs and networks) into a runnable plan: index
        blob upload, workspace, graph tensors on the device."""
        if not graphs.on_device(self.device):
            graphs.to_device(self.device)
        plan = _Plan(self, prog, graphs.edges, nets)
        plan.graphs = graphs
        return plan

    def program_config(self):
        """Keyword arguments for ghn3_amd.program.Program that reproduce what compile() would build (loader workers)."""
        return dict(cfg=dict(hid=self.hid, heads=self.heads, layers=self.layers, num_classes=self.num_classes,
                             max_shape=self.max_shape),
                    index_mode=self.index_mode, layernorm=self.layernorm, weight_norm=self.weight_norm,
                    decoder_ctype=L.COMPUTE_TYPES[self.compute],
                    decoder_bwd_ctype=L.COMPUTE_TYPES[self.compute_bwd] if self.compute_bwd else None,
                    direct16=self.direct16, side_stream=self.side_stream, graphormer_x3=self.graphormer_x3)

    def _fill_bufs(self, plan, out=None, dout=None, gflat=None):
        prog = plan.program
        P = prog.P
        b = plan.bufs
        base = self._flat.data_ptr()
        b[:P] = (base + 4 * self._offs).astype(np.uint64)
        if gflat is not None:
            b[P:2 * P] = (gflat.data_ptr() + 4 * self._offs).astype(np.uint64)
            b[prog.xbuf(prog.X_GRADFLAT)] = gflat.data_ptr()
        b[prog.xbuf(prog.X_WS)] = plan.ws.data_ptr()
        b[prog.xbuf(prog.X_IDX)] = plan.idx.data_ptr()
        b[prog.xbuf(prog.X_EDGES)] = plan.edges.data_ptr()
        b[prog.xbuf(prog.X_SCAL)] = plan.scal.data_ptr()
        if plan.tok is not None:
            b[prog.xbuf(prog.X_TOK)] = plan.tok.data_ptr()
        if out is not None:
            b[prog.xbuf(prog.X_OUT)] = out.data_ptr()
        if dout is not None:
            b[prog.xbuf(prog.X_DOUT)] = dout.data_ptr()

    def _ctx(self):
        ctx = L.context(self.device.index or 0)
        ctx.set_compute_type('f32')          # everything but the decoder GEMMs (Program.decoder_ctype) is fp32
        return ctx

    def _run_forward(self, plan):
        prog = plan.program
        dev = self.device
        out = torch.empty(prog.out_numel, dtype=torch.float32, device=dev)
        # Q3: random class-token rows of positional encodings (nn.py:446)
        plan.tok = torch.normal(0.0, 0.02, (prog.tok_floats,), device=dev)
        self._fill_bufs(plan, out=out)
        stream = torch.cuda.current_stream().cuda_stream
        self._refresh_shadows(plan, stream)
        self._ctx().run(prog.fwd_ops, prog.problems, plan.bufs, stream)
        plan.out = out.detach()                   # (an alias without autograd history: see _GHN3Function.backward)
        return out

    def decoder_grad_range(self, prog):
        """[begin, end) of the decoder parameters in the flat gradient buffer (floats)."""
        lo, hi = prog.decoder_slots
        return int(self._offs[lo]), int(self._offs[hi]) if hi < len(self._offs) else int(self._flat_numel)

    def _run_backward(self, plan, dout, reducer=None, norm_g=None):
        """dout: upstream gradient of the flat predicted buffer (or None); norm_g: device float, weight of the fused
        predicted-parameter-norm loss  norm_g * sum_t ||p_t||_F  whose gradient the tile backward forms itself (needs the
        norms of Program.norm_fin_ops in the plan's scalar buffer).
        reducer (data parallel, ddp_utils.FlatGradReducer): the backward program runs in parts (Program.bwd_parts);
        the all-reduce of the W2 gradient (69 % of the bytes at ghn3xlm16) starts as soon as the side stream has produced
        it, the rest of the decoder (24 %) follows, both overlap with the Graphormer backward; the remaining gradients
        are reduced at the end."""
        prog = plan.program
        if len(prog.bwd_ops) == 0:
            raise L.Ghn3Error('this plan was compiled without a backward program (training=False)')
        if dout is None and norm_g is None:
            raise L.Ghn3Error('backward without an upstream gradient')
        gflat = torch.empty(self._flat_numel, dtype=torch.float32, device=self.device)
        # (the direct 16-bit tile route -- the fused norm term alone, and compiled in: not with GHN3_TILE_D16=0 / bf16 backward
        # operands -- never reads the fp32 tile gradient; every other route does)
        direct = dout is None and norm_g is not None and bool(getattr(prog, 'tile_bwd_h16', 0))
        if not direct and not getattr(plan, 'ws_dout_ready', False):
            # (regions only the fp32 tile-gradient route reads before it has written every byte of them: see _Plan)
            for off, n in getattr(prog, 'ws_zero_dout', ()):
                plan.ws[off:min(off + n, prog.ws_bytes)].zero_()
            plan.ws_dout_ready = True
        self._fill_bufs(plan, out=plan.out, dout=dout, gflat=gflat)
        self._patch_grad_memsets(prog)
        kt = getattr(prog, 'tile_bwd_op', None)
        if kt is not None:                               # which upstream terms the tile backward reads in this step
            r = prog.bwd_ops[kt]['r']
            for slot, (buf, off) in prog.tile_bwd_refs.items():
                on = (dout is not None) if slot == 0 else (norm_g is not None)
                r[slot]['buf'], r[slot]['off'] = (buf, off) if on else (-1, 0)
            if norm_g is not None:
                plan._norm_g = norm_g                    # (kept alive until the kernels ran)
                plan.bufs[prog.xbuf(prog.X_NORMG)] = norm_g.data_ptr()
            # direct 16-bit tiles (Program._build_backward): only the norm term has an a-priori bound of the tile gradient
            patched = [kt] + prog.set_tile_route(direct=dout is None and norm_g is not None)
        elif norm_g is not None:
            raise L.Ghn3Error('the fused norm loss needs a plan with predicted tensors')
        stream = torch.cuda.current_stream().cuda_stream
        # GHN3_BWD_PARTS=1 (experiment, round 6): a single process runs the data-parallel ORDER too -- tile backward -> operand
        # copies -> the persistent weight gradient on the run's own stream and every CU -> dgrad -> rest of the decoder ->
        # Graphormer backward.  Measured alternating on one box (profiles/r06s_*): 6.12 ms per step against 5.90 for the default
        # side-stream schedule at one graph, 10.16 / 9.82 at two, 16.27 / 15.96 at four: slower everywhere, off.
        use_parts = reducer is not None or (len(getattr(prog, 'bwd_parts', ())) > 1 and os.environ.get('GHN3_BWD_PARTS', '0') == '1')
        if not use_parts:
            self._ctx().run(prog.bwd_ops, prog.problems, plan.bufs, stream)
        else:
            ctx = self._ctx()
            k = prog.memset_grad_op                      # (the memset placeholders live in the first part)
            prog.bwd_parts[0][0][k:k + 2] = prog.bwd_ops[k:k + 2]
            if kt is not None:
                for k_ in patched:                       # (all in front of the W2 weight gradient, i.e. in part 1)
                    prog.bwd_parts[0][0][prog.ddp_index.get(k_, k_)] = prog.bwd_ops[k_]
            if reducer is not None:
                reducer.begin()
            n_off = len(self._offs)
            for ops, slots in prog.bwd_parts:
                ctx.run(ops, prog.problems, plan.bufs, stream)
                if reducer is None:
                    continue
                for s_lo, s_hi in slots:                 # gradients complete once the side stream has drained
                    if s_hi > s_lo:
                        reducer.start(gflat, int(self._offs[s_lo]),
                                      int(self._offs[s_hi]) if s_hi < n_off else int(self._flat_numel),
                                      wait_for=ctx.side_wait)
            if reducer is not None:
                reducer.finish(gflat)
        plan.gflat = gflat
        pieces = gflat.split_with_sizes(self._split_sizes)
        return [pieces[2 * k].view(p.shape) for k, p in enumerate(self._slot_params())]

    def _patch_grad_memsets(self, prog):
        """Zero the flat gradient buffer except tensors the backward program fully overwrites."""
        k = prog.memset_grad_op
        total = 4 * self._flat_numel
        ops = prog.bwd_ops
        if prog.grad_no_memset:
            name = prog.grad_no_memset[0]
            slot = prog.slot[name]
            beg = 4 * int(self._offs[slot])
            end = 4 * int(self._offs[slot + 1]) if slot + 1 < len(self._offs) else total
            ops[k]['r'][0]['off'], ops[k]['i'][0] = 0, beg
            ops[k + 1]['r'][0]['off'], ops[k + 1]['i'][0] = end, total - end
        else:
            ops[k]['r'][0]['off'], ops[k]['i'][0] = 0, total
            ops[k + 1]['r'][0]['off'], ops[k + 1]['i'][0] = 0, 0

    def embeddings(self, plan):
        """Node embeddings after the last Graphormer layer + LayerNorm, (B*N_max, C) (nn.py:259-263)."""
        prog = plan.program
        off = prog._ws_names['xe']
        n = prog.B * prog.N * prog.C
        return plan.ws[off:off + 4 * n].view(torch.float32).view(prog.B * prog.N, prog.C)

    # ------------------------------------------------------------------ reference API
    def forward(self, nets_torch, graphs=None, return_embeddings=False, predict_class_layers=True,
                bn_track_running_stats=True, keep_grads=False, reduce_graph=False):
        r"""Predict parameters for a list of >=1 networks (signature of nn.py:186-193)."""
        is_lst = isinstance(nets_torch, (list, tuple))
        if not is_lst:
            nets_torch = [nets_torch]
        if graphs is None:                                   # nn.py:217-219: graphs built on the fly
            graphs = [Graph(net, ve_cutoff=50 if self.ve else 1) for net in nets_torch]
        keep = self.training if keep_grads is None else keep_grads
        debug_info = self._init_debug_info(nets_torch)
        plan = self.compile(nets_torch, graphs, predict_class_layers=predict_class_layers, reduce_graph=reduce_graph,
                            training=bool(keep and torch.is_grad_enabled()))
        if keep and torch.is_grad_enabled():
            flat = _GHN3Function.apply(self, plan, *self._slot_params())
        else:
            with torch.no_grad():
                flat = self._run_forward(plan)
        self._last_flat = flat                    # (autograd-connected in training: predicted_param_norm; kept on
        self.assign(plan, flat, keep_grads=keep)  #  the model, not on the plan: the plan must not own autograd tensors)
        if bn_track_running_stats is None:
            bn_track_running_stats = self.training
        if not bn_track_running_stats:
            def bn_set_train(module):
                if isinstance(module, nn.BatchNorm2d):
                    module.track_running_stats = False
                    module.training = True
            for net in nets_torch:
                if isinstance(net, nn.Module):
                    net.apply(bn_set_train)
        self.last_plan = plan
        self._print_debug_info(nets_torch, plan, debug_info)
        out = nets_torch if is_lst else nets_torch[0]
        return (out, self.embeddings(plan)) if return_embeddings else out

    # ------------------------------------------------------------------ debug_level self-checks (nn.py:354-420)
    @staticmethod
    def _count_params(net):
        """Number of parameter elements of a target network: nn.Module parameters, or -- for the light networks whose
        `weight` / `bias` are shape lists until the GHN assigns tensors (light_ops.py) -- the shapes / tensors found on the
        modules (ppuda's `capacity(net)[1]` counts the same elements)."""
        if hasattr(net, 'named_parameters'):
            total = 0
            for _, p in net.named_parameters():
                total += int(p.numel()) if torch.is_tensor(p) else int(np.prod(p)) if p is not None and len(p) else 0
            if total:
                return total
        if hasattr(net, 'num_params'):
            return int(net.num_params())
        return 0

    def _init_debug_info(self, nets_torch):
        if not self.debug_level:
            return None
        import time
        n_params = sum(self._count_params(net) for net in nets_torch)
        if self.device.type == 'cuda':
            torch.cuda.synchronize()
        return {'n_params': n_params, 'start_time': time.time()}

    def _print_debug_info(self, nets_torch, plan, info):
        """The reference's end-of-forward report (nn.py:373-420): number of predicted tensors / parameters, the
        MATCHED! / ERROR! comparison with the networks' own parameter count (an end-to-end assertion that every parameter
        received a prediction), the prediction time, and -- debug_level > 2 -- per-tensor statistics."""
        if not self.debug_level or info is None:
            return
        import time
        from .utils import log
        if self.device.type == 'cuda':
            torch.cuda.synchronize()
        preds = plan.program.predicted
        n_tensors_pred, n_params_pred = len(preds), int(sum(p['numel'] for p in preds))
        has_none = [sum(n[0] == 'none' for n in net.genotype.normal + net.genotype.reduce) > 0
                    for net in nets_torch if hasattr(net, 'genotype')]
        n_recompute = sum(self._count_params(net) for net in nets_torch)
        matched = info['n_params'] == n_params_pred
        log('number of parameter tensors predicted using GHN: {}, total parameters predicted: {} ({}), '
            'time to predict (on {}): {:.4f} sec'.format(
                n_tensors_pred, n_params_pred,
                'MATCHED!' if matched else 'ERROR! NOT MATCHED WITH {} ACTUAL PARAMS (HAS_NONE={}, N_PARAMS={})'.format(
                    info['n_params'], has_none, n_recompute),
                str(self.device).upper(), time.time() - info['start_time']))
        self.last_debug_info = {'n_tensors_pred': n_tensors_pred, 'n_params_pred': n_params_pred,
                                'n_params': info['n_params'], 'matched': matched}
        if self.training and len(nets_torch) > 1:
            assert matched or (any(has_none) and n_recompute == n_params_pred), 'not all params predicted!'
        if self.debug_level > 2:
            for net_id, net in enumerate(nets_torch):
                log('\npredicted parameter stats for net %d:' % net_id)
                named = net.named_parameters() if isinstance(net, nn.Module) else []
                for n, p in named:
                    log('{:30s} ({:30s}): min={:.3f} \t max={:.3f} \t mean={:.3f} \t std={:.3f} \t norm={:.3f}'.format(
                        n[:30], str(tuple(p.shape))[:30], p.min().item(), p.max().item(), p.mean().item(),
                        p.std().item() if p.numel() > 1 else 0.0, torch.norm(p).item()))

    def predicted_param_norm(self, plan=None):
        """Differentiable sum of the Frobenius norms of all tensors predicted by the last forward (or `plan`): the
        regulariser the reference's trainer adds with weight predparam_wd (trainer.py:97-98,288-294), computed by two
        streaming kernels on the flat output buffer.  (Positional-encoding tensors include their random class-token
        row, as in the reference.)"""
        plan = self.last_plan if plan is None else plan
        flat = self._last_flat if plan is self.last_plan else None
        if flat is None:
            raise L.Ghn3Error('predicted_param_norm: run the GHN forward first (only the last forward is kept)')
        return _ParamNormLoss.apply(flat, self, plan)

    def assign(self, plan, flat, keep_grads):
        """nn.py:508-552 _set_params for every predicted tensor (views of the flat output buffer)."""
        preds = plan.program.predicted
        pieces = None
        if flat.requires_grad:
            # ONE autograd node for all views: slicing tensor by tensor would make the backward of every slice
            # allocate and add a full-size zero buffer (250 x 346 MB at ghn3xlm16 -- 70 ms per step)
            sizes, pos = [], 0
            for p in preds:
                sizes += [p['offset'] - pos, p['numel']]
                pos = p['offset'] + p['numel']
            sizes.append(flat.numel() - pos)
            pieces = flat.split_with_sizes(sizes)
        own = []                                             # eval: (parameter, view of the flat buffer) pairs, copied in one batch
        for k, p in enumerate(preds):
            t = (pieces[2 * k + 1] if pieces is not None else flat[p['offset']:p['offset'] + p['numel']]).view(p['shape'])
            m, key = p['module'], p['attr']
            target = getattr(m, key, None)
            if keep_grads or not isinstance(target, nn.Parameter):
                if isinstance(target, (list, tuple)) or not isinstance(m, nn.Module):
                    setattr(m, key, t)
                else:
                    m.__dict__[key] = t
                    m._parameters[key] = t
            else:
                # eval (nn.py:545-548): `target_param.data = tensor.clone()` -- no memory shared with the flat
                # batch buffer, so a saved / pickled network carries its own parameters only
                own.append((target, t))
        if own:
            # (one multi-tensor copy instead of a clone kernel per tensor: 161 launches for a ResNet-50, ~1 ms of the 5.7 ms that
            # ghn(model, graph) takes end to end)
            srcs = [t for _, t in own]
            dsts = [torch.empty_like(t) for t in srcs]
            torch._foreach_copy_(dsts, srcs)
            for (target, _), d in zip(own, dsts):
                target.data = d


# What each architecture argument can be read from in a bare state dict (checkpoints without a 'config' entry,
# nn.py:62-95).  (key suffix, required substring, config field, value from the tensor)
_CONFIG_EVIDENCE = (
    ('', 'class_layer_predictor', 'num_classes', lambda t: int(t.shape[0])),
    ('embed.weight', '', 'hid', lambda t: int(t.shape[-1])),
    ('decoder.conv.2.weight', '', 'max_ch', lambda t: int(math.isqrt(int(t.shape[0])))),
    ('shape_enc.embed_spatial.weight', '', 'spatial', lambda t: 11 if int(t.shape[0]) == 9 else 16),
    ('ln.weight', '', 'layernorm', lambda t: True),
)


def infer_config(state_dict, num_classes=10, layers=0, hid=32, layernorm=False, max_shape=64):
    """GHN-3 constructor arguments from the tensors of a state dict: the width from the node embedding, the class count
    from the classifier head of the decoder, the channel limit from the W2 row count (max_ch ** 2 rows), the spatial
    limit from the shape-encoder vocabulary (9 entries <-> 11 x 11, otherwise 16 x 16; without that table 16 for
    ImageNet-sized heads), one layer per 'gnn.<l>.ln1.weight'.  Arguments are the fall-backs for what the state dict
    does not show."""
    found = {'num_classes': num_classes, 'hid': hid, 'layernorm': layernorm, 'max_ch': max_shape}
    depth = layers
    for name, t in state_dict.items():
        if '.ln1.weight' in name and 'gnn.' in name:
            depth += 1
        for suffix, needle, field, read in _CONFIG_EVIDENCE:
            if name.endswith(suffix) and needle in name:
                found[field] = read(t)
    spatial = found.get('spatial', 16 if found['num_classes'] >= 1000 else 11)
    limit = found['max_ch']
    return {'hid': found['hid'], 'max_shape': limit if isinstance(limit, tuple) else (limit, limit, spatial, spatial),
            'num_classes': found['num_classes'], 'heads': 16 if found['hid'] > 64 else 8, 'layers': depth,
            'weight_norm': True, 've': True, 'layernorm': found['layernorm']}


def from_pretrained(ghn3_name='ghn3xlm16.pt', **kwargs):
    """
    Loads a GHN-3 checkpoint (nn.py:31-125).  Local files are tried first (the reference's local fallback is
    unreachable offline); otherwise the HuggingFace hub is queried like the reference does.
    Returns the model in training mode, on CPU.
    """
    assert ghn3_name is not None, 'GHN ckpt must be specified'
    ghn_config = None
    if os.path.exists(ghn3_name):
        state_dict = torch.load(ghn3_name, map_location='cpu')
        if isinstance(state_dict, dict) and 'state_dict' in state_dict:
            ghn_config = state_dict.get('config', None)
            state_dict = state_dict['state_dict']
    else:
        import joblib
        from huggingface_hub import hf_hub_download
        state_dict = joblib.load(hf_hub_download(repo_id='SamsungSAILMontreal/ghn3', filename=ghn3_name))
    if any(k.find('gnn.gru.') >= 0 for k in state_dict):
        raise NotImplementedError('GHN-2 checkpoints are not supported')
    extra = {k: kwargs.pop(k) for k in ('index_mode', 'compute', 'compute_bwd', 'direct16', 'side_stream', 'debug_level',
                                        'graphormer_x3')
             if k in kwargs}
    if ghn_config is None:
        defaults = {k: kwargs.pop(k) for k in ('num_classes', 'layers', 'hid', 'layernorm', 'max_shape') if k in kwargs}
        ghn_config = infer_config(state_dict, **defaults)
    else:
        ghn_config = {k: v for k, v in ghn_config.items() if k not in ('is_ghn2', 'pretrained')}
    ghn = GHN3(**ghn_config, **extra, **kwargs)
    ghn.load_state_dict(state_dict)
    return ghn


# ---------------------------------------------------------------------------------------------------
# Known-answer harness for the released checkpoints (nn.py:783-861)
# ---------------------------------------------------------------------------------------------------
_RESULT_KEYS = {'ghn3xlm16.pt': 'ghn3', 'ghn3tm8.pt': 'ghn3-t', 'ghn2.pt': 'ghn2', 'randinit': 'randinit'}
_RESULTS_MD5 = 'c9ffc3b9222e872af316eb1cb1ee1c08'


def get_metadata(ghn3_name='ghn3xlm16.pt', arch=None, attr=None, path=None):
    """
    Per-architecture known answers of the released GHNs (total norm of the predicted parameters, accuracies) from
    the reference's ``ghn3_results.json`` (one JSON object per line, md5-checked).  ``path`` / $GHN3_RESULTS_JSON name
    a local copy (the repository keeps one under tests/golden/); otherwise the file is fetched from the HuggingFace
    hub like the reference does.  Returns None when the file or the GHN name is unknown.
    """
    import hashlib
    import json
    key = None
    if ghn3_name is not None:
        key = _RESULT_KEYS.get(ghn3_name)
        if key is None:
            log('WARNING: meta data not unavailable for %s' % ghn3_name)
            return None
    path = path or os.environ.get('GHN3_RESULTS_JSON')
    if path is None:
        try:
            from huggingface_hub import hf_hub_download
            path = hf_hub_download(repo_id='SamsungSAILMontreal/ghn3', filename='ghn3_results.json')
        except Exception as e:
            print('Error: ', e)
            return None
    with open(path, 'rb') as f:
        raw = f.read()
    digest = hashlib.md5(raw).hexdigest()
    assert digest == _RESULTS_MD5, 'corrupted %s: md5sum=%s' % (path, digest)
    table = {}
    for line in raw.decode().splitlines():
        if line.strip():
            table.update(json.loads(line))
    if key is None:
        return table
    out = {}
    for a, row in table.items():
        out[a] = {k.split('-')[-1]: float(v) for k, v in row.items()
                  if k.startswith(key) and not (key == 'ghn3' and k.startswith('ghn3-t'))}
    if arch is not None:
        out = out[arch]
        return out[attr] if attr is not None else out
    if attr is not None:
        return {a: out[a][attr] for a in out}
    return out


def norm_check(model, arch='resnet50', ghn3_name='ghn3xlm16.pt', path=None, expected=None):
    """Total norm of the model's parameters against the released known answer (tolerance 1e-2, nn.py:795).
    `expected` overrides the table lookup (a known answer for weights other than the released checkpoints, e.g. the
    seeded fixtures under tests/golden/).  Returns (total_norm, expected or None, passed or None)."""
    total_norm = torch.norm(torch.stack([p.norm() for p in model.parameters()]), 2).item()
    norm = expected if expected is not None else get_metadata(ghn3_name, arch=arch, attr='paramnorm', path=path)
    ok = None if not norm else abs(norm - total_norm) < 1e-2
    log('Predicted params total norm={:.4f} ({})'.format(
        total_norm, 'no norm check available' if ok is None else
        ('check passed!' if ok else 'ERROR: norm check not matched with %.2f' % norm)))
    return total_norm, norm, ok
