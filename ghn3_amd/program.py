"""
Compiles one batch of graphs + target networks into the op programs libghn3_hip.so executes
(include/ghn3_hip.h): a forward program (GHN3.forward, /root/reference/ghn3/nn.py:247-328) and, in
training, the matching backward program (what torch autograd would run for the same lines).

Everything here is host-side shape bookkeeping (numpy); no tensor math.  The compiled object holds
  * ops / GEMM problem tables (numpy structured arrays mirroring the C structs)
  * the workspace layout (byte offsets into one device buffer) and its size
  * an index blob (int32/int64 arrays + tile descriptors) uploaded once per batch
  * the list of predicted tensors (module, attribute, shape, offset into the flat output)
"""

import math
import os

import numpy as np

from . import _lib as L
from . import bookkeeping as bk

ALIGN = 256


def global_param_names():
    return ['ln.weight', 'ln.bias', 'embed.weight', 'shape_enc.embed_spatial.weight',
            'shape_enc.embed_channel.weight', 'gnn.0.centrality_embed_in.weight',
            'gnn.0.centrality_embed_out.weight', 'gnn.0.input_dist_embed.weight',
            'gnn.0.attn.edge_embed.embed.weight', 'gnn.0.attn.proj_e.0.weight', 'gnn.0.attn.proj_e.0.bias',
            'gnn.0.attn.proj_e.2.weight', 'gnn.0.attn.proj_e.2.bias',
            'decoder.fc.0.weight', 'decoder.fc.0.bias', 'decoder.conv.0.weight', 'decoder.conv.0.bias',
            'decoder.conv.2.weight', 'decoder.conv.2.bias', 'decoder.class_layer_predictor.1.weight',
            'decoder.class_layer_predictor.1.bias', 'decoder_1d.fc.0.weight', 'decoder_1d.fc.0.bias',
            'decoder_1d.fc.2.weight', 'decoder_1d.fc.2.bias', 'bias_class.1.weight', 'bias_class.1.bias']


LAYER_PARAM_NAMES = ['ln1.weight', 'ln1.bias', 'attn.to_qkv.weight', 'attn.to_out.0.weight', 'attn.to_out.0.bias',
                     'ln2.weight', 'ln2.bias', 'ff.net.0.weight', 'ff.net.0.bias', 'ff.net.3.weight',
                     'ff.net.3.bias']


def param_names(layers, layernorm=True):
    names = global_param_names()
    if not layernorm:                      # GHN(layernorm=False) has no final LayerNorm (ppuda GHN.__init__)
        names = [n for n in names if not n.startswith('ln.')]
    for l in range(layers):
        names += ['gnn.%d.%s' % (l, n) for n in LAYER_PARAM_NAMES]
    return names


def round_up(x, m):
    return (x + m - 1) // m * m


class Program:
    """See module docstring.  cfg: dict(hid, heads, layers, num_classes, max_shape)."""

    # extra buffer slots after the 2*P parameter / gradient pointers
    X_WS, X_IDX, X_EDGES, X_OUT, X_DOUT, X_TOK, X_SCAL, X_GRADFLAT, X_SHADOW, X_NORMG, X_COUNT = range(11)

    def __init__(self, cfg, node_infos, n_nodes, node_types, max_edge, nets, index_mode='reference',
                 training=True, predict_class_layers=True, reduce_graph=False, layernorm=True, weight_norm=True,
                 decoder_ctype=None, decoder_bwd_ctype=None, direct16=True, side_stream=True, graphormer_x3=None):
        self.cfg = cfg
        # side_stream: weight-gradient GEMMs, LayerNorm parameter gradients and operand copies (everything off the
        # dependent chain of the program) carry GHN3_OPFLAG_SIDE and overlap with the chain on a second stream;
        # the temporaries they read are then per-layer buffers instead of reused ones.
        self.SIDE = L.OPFLAG_SIDE if side_stream else 0
        # fuse_ln (off by default): the LayerNorms of the Graphormer layers run as row prologues of the GEMMs that
        # consume them (forward: LN1 -> to_qkv, LN2 -> ff.net.0; backward: LN2' -> to_out dgrad, LN1' -> the next
        # layer's ff.net.3 dgrad) instead of ~95 separate launches.  Measured a LOSS at ghn3xlm16: the prologue's
        # registers (123 VGPRs against 52-60) halve the occupancy of the 1024-thread workgroups, a fused GEMM takes 34 us
        # against 13 + 6 us for the separate kernels (11.2 -> 12.2 ms per step); kept for narrower models / as a record.
        self.fuse_ln = os.environ.get('GHN3_FUSE_LN', '0') != '0'
        self.split_k2 = os.environ.get('GHN3_SPLIT_K2', '1') != '0'       # see split_small()
        self.split_k2_min = int(os.environ.get('GHN3_SPLIT_K2_MIN', '768'))  # (lowered by the CPU tests)
        # MFMA operand type of the decoder GEMMs (fc / W0 / W2, forward and backward): None = context default.
        # The Graphormer, the edge MLP and the small heads always multiply in exact fp32.
        self.decoder_ctype = decoder_ctype
        # Backward operand type of the W2 trio / D2 in the 16-bit pipeline.  The upstream gradients (d_tiles, d_u) are
        # ~1e-6 in magnitude -- f16 subnormals -- so f16 backward copies are scaled by a power of two derived from the
        # running max |x| of their fp32 source (GHN3_CAST_SCALED; the GEMM epilogue divides it out again): 2^-11
        # operand rounding instead of bf16's 2^-8, gradients within 3e-4 of the oracle instead of 1.3e-3.
        if decoder_bwd_ctype is None:
            decoder_bwd_ctype = decoder_ctype
        self.decoder_bwd_ctype = decoder_bwd_ctype
        self.bwd_scaled = decoder_bwd_ctype == L.CT_F16
        # direct16: the W2 GEMMs (98 % of the decoder flops) read 16-bit operand COPIES written once per step by
        # GHN3_OP_CAST16 (k-contiguous, zero padded, LDS-DMA friendly) instead of converting fp32 while staging.
        self.direct16 = bool(direct16) and decoder_ctype in (L.CT_F16, L.CT_BF16) and (8 * int(cfg['hid'])) % 64 == 0
        # Exact ReLU masks: the first two decoder linears (fc, conv.0 -- 2.3 % of the decoder flops) multiply in exact
        # fp32 even in the 16-bit modes.  Their outputs t, u carry the ReLU masks of the backward: a 3e-4 forward error
        # flips the mask of ~3e-4 of the elements (those with |u| below the error), and every flipped element is an
        # O(1) error of d_u / d_t there -- measured 1-5 % per decoder row against the fp32 path, 6e-3 on the gradient
        # of decoder.fc.0.weight at ghn3xlm16.  Likewise the fc backward (d_t ~ 1e-4, f16 subnormal range) stays fp32.
        self.d12_fwd_ctype = {'f32': L.CT_F32, 'f16': None}[os.environ.get('GHN3_D12_FWD', 'f32')]
        # 8-phase kernel (tile code 28) with per-family row tiles for the W2 forward and dgrad (round 3)
        self.use_p8 = os.environ.get('GHN3_P8', '1') != '0'
        self.d1_bwd_ctype = {'f32': L.CT_F32, 'f16': None}[os.environ.get('GHN3_D1_BWD', 'f32')]
        self.C = C = int(cfg['hid'])
        # x3: the Graphormer linears (to_qkv, to_out, ff.net.0, ff.net.3; forward and dgrad) as split-bf16 products on the
        # 16-bit matrix cores (GHN3_GEMM_X3: hi.hi + hi.lo + lo.hi, ~1e-5 relative) against persistent bf16 hi / lo copies
        # of the weights, instead of the exact-fp32 matrix instruction.  Default: on in the 16-bit modes when the width
        # suits the kernel's K slices (C a multiple of 64, <= 384: every released GHN-3); off in the exact 'f32' mode.
        if graphormer_x3 is None:
            graphormer_x3 = decoder_ctype in (L.CT_F16, L.CT_BF16) and os.environ.get('GHN3_X3', '1') != '0'
        self.x3 = bool(graphormer_x3) and C % 64 == 0 and 64 <= C <= 384
        # x3s (round 4): the split-bf16 linears on the STAGED kernels (gemm_x3d.hip, tile codes 44 / 45): fragment-major weight
        # copies loaded straight into registers, whole-K activation rows in LDS, K split over the waves of a workgroup -- no
        # partial planes -- and the LayerNorms as row prologues of the GEMMs that consume them: 5 dependent launches per
        # layer forward / backward instead of 7.  Needs a kernel for K = C, 3C and 4C (every released width).
        self.x3s = self.x3 and C in (64, 128, 256, 384) and os.environ.get('GHN3_X3S', '1') != '0'
        # x3_exact_last: the LAST n Graphormer layers keep the exact-fp32 matrix instruction (experiment of round 5: does the
        # ~1e-5 deviation of the node embeddings -- which flips knife-edge ReLU masks of the decoders -- come from the end of
        # the chain?  It does not: docs/EXPERIMENTS.md)
        self.x3_exact_last = int(os.environ.get('GHN3_X3_EXACT_LAST', '0')) if self.x3 else 0
        # x3f16 (round 5): the FORWARD linears of the staged kernels multiply F16 pieces (11 + 11 bits of mantissa, fp32-grade)
        # instead of bf16 pieces (8 + 8): same matrix instruction rate, the straight weight copies hold f16 pieces of
        # W * 2^6 (GHN3_CAST_SPLIT_F16) and alpha carries 2^-6.  Forward operands are O(1) (LayerNorm / attention / GELU
        # outputs); the backward's gradient operands (~1e-7) keep bf16 pieces.  Measured over 500 random batches against the
        # fp32 mode: batches with a parameter gradient above 1e-3 (a ReLU mask of the decoders on a knife edge, flipped by
        # the deviation of the node embeddings) 4.8 % -> see docs/EXPERIMENTS.md
        self.x3f16 = self.x3s and os.environ.get('GHN3_X3_F16', '1') != '0'
        self.H = int(cfg['heads'])
        self.Lyr = int(cfg['layers'])
        self.K = int(cfg['num_classes'])
        self.max_shape = tuple(int(v) for v in cfg['max_shape'])
        self.S = self.max_shape[2]                    # decoder spatial size (16)
        assert self.max_shape[2] == self.max_shape[3]
        self.layernorm = layernorm
        self.weight_norm = weight_norm          # False: predicted tensors are assigned un-normalised (nn.py:543-546)
        self.training = training
        self.index_mode = index_mode
        self.names = param_names(self.Lyr, layernorm)
        self.P = len(self.names)
        self.slot = {n: i for i, n in enumerate(self.names)}
        self.B = len(n_nodes)
        self.n_nodes = [int(v) for v in n_nodes]
        self.N = max(self.n_nodes)
        self.V = int(max_edge) + 1
        if self.V + 1 > 257:
            raise ValueError('shortest-path length %d exceeds the 257-row edge embedding (graphormer.py:95-96)'
                             % max_edge)
        if self.N > 4096:
            raise ValueError('graphs with more than 4096 nodes are not supported (got %d)' % self.N)
        self._ops, self._probs, self._ln = [], [], {}
        self.tag_flops = {}
        self._ws = 0
        self._ws_names = {}
        self._idx_chunks, self._idx_size = [], 0

        self.param_groups, self.params_map = bk.map_net_params(node_infos, self.n_nodes, nets, self.max_shape,
                                                               reduce_graph=reduce_graph)
        vocab = bk.ShapeVocab(self.K, self.max_shape)
        self.vocab_rows = (vocab.n_ch + 1, vocab.n_sp + 1)          # rows of the channel / spatial embedding tables
        total_nodes = sum(self.n_nodes)
        self.shape_idx = vocab.indices(total_nodes, self.params_map, predict_class_layers)
        self.node_types = np.asarray(node_types, dtype=np.int32)
        assert len(self.node_types) == total_nodes
        self.predict_class_layers = predict_class_layers

        self._layout_decoder()
        # 16-bit shadows of the decoder weights (a separate, plan-independent program: GHN3 runs it only when the
        # parameters changed since the shadows were written)
        self._cast_w2()
        if self._ops:
            self.op(L.OP_DETACH)
        self.shadow_ops = self._finish_ops()
        self.shadow_ops_rest = self.shadow_ops
        if getattr(self, 'shadow_w2', None) is not None:
            self.shadow_ops_rest = self.shadow_ops.copy()
            assert int(self.shadow_ops_rest[self.shadow_w2['op']]['kind']) == L.OP_CAST16
            self.shadow_ops_rest[self.shadow_w2['op']]['kind'] = L.OP_NOP
        self._build_forward()
        self.fwd_ops = self._finish_ops()
        self.n_fwd_problems = len(self._probs)
        if training:
            self._build_backward()
            # direct 16-bit tiles: ops of the direct route (off unless GHN3._run_backward turns them on) and of the fp32 route
            on, off = getattr(self, '_d16_on', []), getattr(self, '_d16_off', [])
            self.d16_on_idx = [k for k, o in enumerate(self._ops) if any(o is t for t in on)]
            self.d16_off_idx = [k for k, o in enumerate(self._ops) if any(o is t for t in off)]
            assert all(k < self.bwd_cut_w2 and (self.wgrad_op_range is None or k < self.wgrad_op_range[0])
                       for k in self.d16_on_idx + self.d16_off_idx)
            self.bwd_ops = self._finish_ops()
            self.d16_kinds = {k: int(self.bwd_ops[k]['kind']) for k in self.d16_on_idx + self.d16_off_idx}
            if not hasattr(self, 'tile_bwd_h16'):
                self.tile_bwd_h16 = 0
            for k in self.d16_on_idx:
                self.bwd_ops[k]['kind'] = L.OP_NOP               # (compiled state = the fp32 route)
            detach = np.zeros(1, dtype=L.OP_DT)
            detach['kind'] = L.OP_DETACH
            detach['r']['buf'][:] = -1
            self.bwd_ops_a = np.concatenate([self.bwd_ops[:self.bwd_split], detach])
            self.bwd_ops_b = self.bwd_ops[self.bwd_split:]
            # flat-gradient slots of the decoder parameters (contiguous in the parameter order)
            self.decoder_slots = (self.slot['decoder.fc.0.weight'], self.slot['bias_class.1.bias'] + 1)
            # Data-parallel schedule: the backward program is run in parts; after part k the gradients of `slots`
            # (half-open slot ranges) are complete once the side stream has drained, and their all-reduce may start
            # while the next parts execute.  Part 1 ends behind the W2 weight gradient (69 % of all gradient bytes at
            # ghn3xlm16), part 2 behind the rest of the decoder (24 %), the Graphormer backward is the last part.
            # The schedule (number of parts, slot ranges = collective sizes) depends on the GHN's parameter layout ONLY,
            # never on the rank's graph: every rank compiles a different architecture each step and mismatched
            # collective sequences would hang RCCL.  A batch without any 2-D / 4-D weight node gets an empty first part
            # (just the gradient memsets, which always live in part 1).
            w2 = self.slot['decoder.conv.2.weight']
            lo, hi = self.decoder_slots
            cut1 = min(max(self.bwd_cut_w2, self.memset_grad_op + 2), self.bwd_split)
            cuts = [(cut1, [(w2, w2 + 1)]), (self.bwd_split, [(lo, w2), (w2 + 1, hi)])]
            self.bwd_parts, pos = [], 0
            self.ddp_index = {}                          # op index in bwd_ops -> index inside part 1, where they differ
            wg_first = self.wgrad_op_range is not None and getattr(self, '_i_dgrad', None) is not None and \
                os.environ.get('GHN3_DDP_WGRAD_FIRST', '1') != '0'
            self.ddp_wgrad_first = bool(wg_first)
            if wg_first:
                # Round 5: the W2 weight gradient FIRST.  dW2 is 69 % of the gradient bytes and needs nothing but the tile
                # gradient and the forward's u: issued right behind the tile backward and its operand copies -- on the
                # run's own stream, every CU -- its exchange overlaps the W2 dgrad, the conv.0 / fc backward AND the whole
                # Graphormer backward (~2.5 ms at ghn3xlm16) instead of the Graphormer backward alone (~1.1 ms).
                o = self.bwd_ops
                a, b = self.wgrad_op_range
                dg0, l0, l1 = self._i_dgrad, self._i_late0, self._i_late1
                assert self.memset_grad_op + 2 <= dg0 <= l0 <= l1 <= a <= b <= self.bwd_split
                # (the operand copies run on the side stream: MARK + WAIT on slot 1 -- not a full join, which would also
                # consume the pending mark of the 1-D decoder backward that part 3 waits for)
                join = np.zeros(2, dtype=L.OP_DT)
                join['kind'] = L.OP_JOIN
                join['r']['buf'][:] = -1
                join[0]['i'][:2] = (1, 1)
                join[1]['i'][:2] = (2, 1)
                wg = o[a:b].copy()
                for q in range(len(wg)):                 # the persistent band launch: chain's stream, no grid cap
                    if int(wg[q]['kind']) == L.OP_GEMM and int(wg[q]['i'][2]) == 29:
                        wg[q]['flags'] &= ~L.OPFLAG_SIDE
                        wg[q]['i'][3] = 0
                part1 = np.concatenate([o[:dg0], o[l0:l1], join, wg, detach])
                part2 = np.concatenate([o[dg0:l0], o[l1:a], o[b:self.bwd_split], detach])
                self.ddp_index = {k: dg0 + (k - l0) for k in range(l0, l1)}
                self.bwd_parts = [(part1, cuts[0][1]), (part2, cuts[1][1])]
                pos = self.bwd_split
            else:
                for end, slots in cuts:
                    self.bwd_parts.append((np.concatenate([self.bwd_ops[pos:end], detach]), slots))
                    pos = end
            self.bwd_parts.append((self.bwd_ops[pos:], []))
            # Single-process order (bwd_ops; the parts above keep the early order so that a data-parallel run can start the
            # exchange of dW2 -- 69 % of the gradient bytes -- as soon as possible): the W2 weight gradient is issued
            # BEHIND the rest of the decoder backward.  The conv.0 / fc dgrads, the plane sums and the node-row gather are on
            # the dependent chain and no longer share the chip with it (fc dgrad 0.39 -> 0.1 ms); the weight gradient then
            # runs beside the Graphormer backward only, which is long enough to hide it (GHN3_WGRAD_LATE=0: early order).
            # Round 5b, GHN3_WGRAD_ORDER=first (default when the weight gradient runs on the side stream): issued right behind the
            # tile backward and its operand copies, on 128 workgroups -- half of every XCD.  The lowest-priority stream gets its CUs
            # when the W2 dgrad has handed out its last tiles (forcing the launch beside the dgrad gains nothing: two MFMA-bound
            # kernels share the chip's work), so it runs beside the conv.0 / fc backward and only the first third of the Graphormer
            # backward, which is the part of the step that suffers beside it (tools/contention_probe: a streaming kernel on the
            # other CUs turns the chain's L2 hits into 350 ns misses and costs 8-15 % of the clock).  5.85-5.86 against 5.91-5.93
            # ms (one graph), 9.65-9.83 against 10.03-10.12 (two), 15.76-15.80 against 16.28-16.35 (four); 112 / 144 / 160
            # workgroups are all slower than 128 (profiles/r05s_*, r05v_*, r05w_*).
            order = self._wgrad_order_env()
            if order == 'first' and not getattr(self, 'wgrad_cap', 0):
                order = 'late'                           # (on the chain's own stream it stays in front of the Graphormer backward)
            self.wgrad_order = order if (self.wgrad_op_range is not None and self.SIDE) else 'early'
            if self.wgrad_order == 'late':
                a, b = self.wgrad_op_range
                o = self.bwd_ops
                self.bwd_ops = np.concatenate([o[:a], o[b:self.bwd_split], o[a:b], o[self.bwd_split:]])
            elif self.wgrad_order == 'first' and getattr(self, '_i_dgrad', None) is not None:
                a, b = self.wgrad_op_range               # (a: the zero-fill of its sum-of-squares slots, then the launches)
                dg0, l0, l1 = self._i_dgrad, self._i_late0, self._i_late1
                o = self.bwd_ops
                perm = list(range(dg0)) + list(range(l0, l1)) + list(range(a, b)) + list(range(dg0, l0)) + \
                    list(range(l1, a)) + list(range(b, len(o)))
                assert sorted(perm) == list(range(len(o)))
                inv = {old: new for new, old in enumerate(perm)}
                self.bwd_ops = o[np.asarray(perm, dtype=np.int64)]
                # (op indices recorded for the route switch of the tile gradient / the data-parallel parts follow the ops)
                if hasattr(self, 'd16_kinds'):
                    self.d16_kinds = {inv[k]: v for k, v in self.d16_kinds.items()}
                    self.d16_on_idx = [inv[k] for k in self.d16_on_idx]
                    self.d16_off_idx = [inv[k] for k in self.d16_off_idx]
                self.ddp_index = {inv[k]: v for k, v in self.ddp_index.items()}
        else:
            self.bwd_ops = np.zeros(0, dtype=L.OP_DT)
        self.problems = self._pack_problems()
        # scalar buffer (X_SCAL): loss [0:256) | per-tensor norms | per-(chunk, tensor) partial sums of the norm pass
        self.scal_bytes = 256 + round_up(4 * max(self.n_seg, 1), 64) + \
            4 * (max(self.n_seg, 1) + (self.out_numel + 8191) // 8192 + 1) + 64
        self.ws_bytes = round_up(self._ws + 1024, ALIGN)
        self.idx_blob = np.zeros(max(self._idx_size, 16), dtype=np.uint8)
        for off, raw in self._idx_chunks:
            self.idx_blob[off:off + len(raw)] = raw
        self.n_bufs = 2 * self.P + self.X_COUNT

    def strip(self):
        """Drops the builder-only state (problem tuples, index chunks, group dictionaries): what remains is what a plan
        needs at run time.  A loader worker calls it before sending the program to the training process (the pickle
        shrinks from 4.5 MB / ~60k objects to the packed arrays)."""
        for name in ('_probs', '_ops', '_ln', '_idx_chunks', 'conv_groups', 'gemm_groups', 'param_groups', 'params_map',
                     'wgrad_bands', 'd1', 'row_src', 'row_pos', 'oned_src', 'oned_index', 'oned_plain', 'oned_clsb',
                     'shape_idx', 'node_types', 'shadow_lay', '_tile_desc_arr', '_d16_on', '_d16_off'):
            if hasattr(self, name):
                setattr(self, name, None)
        for p_ in self.predicted:
            p_.pop('tok', None)
        return self

    # ------------------------------------------------------------------ small helpers
    def xbuf(self, which):
        return 2 * self.P + which

    def pref(self, name, off_floats=0):
        return (self.slot[name], 4 * off_floats)

    def gref(self, name, off_floats=0):
        return (self.P + self.slot[name], 4 * off_floats)

    def ws(self, name, nbytes):
        if name in self._ws_names:
            return self._ws_names[name]
        off = self._ws
        self._ws = round_up(self._ws + max(int(nbytes), 4), ALIGN)
        self._ws_names[name] = off
        return off

    def wsf(self, name, nfloats):
        """workspace region of nfloats floats (plus slack for vector over-reads); returns a ref."""
        return (self.xbuf(self.X_WS), self.ws(name, 4 * (int(nfloats) + 64)))

    def wref(self, name, off_floats=0):
        return (self.xbuf(self.X_WS), self._ws_names[name] + 4 * off_floats)

    def ws16(self, name, n_halfs):
        """workspace region of 16-bit elements, written by GHN3_OP_CAST16 and -- the family matrices `dth` on the direct route --
        by GHN3_OP_TILE_BWD; neither writes the k / row padding the GEMMs read, so the region is recorded in `ws_zero` and
        zero-filled once per plan (_Plan); returns the offset in 16-bit elements from the ws base."""
        known = name in self._ws_names
        off = self.ws(name, 2 * (int(n_halfs) + 128))
        if not known:
            self.__dict__.setdefault('ws_zero', []).append((off, round_up(2 * (int(n_halfs) + 128), ALIGN)))
        return off // 2

    def href(self, off_halfs):
        return (self.xbuf(self.X_WS), 2 * int(off_halfs))

    def cast16(self, src_base, items, dbias=None, flags=0, grid_cap=0, amax=None, dst_base=None):
        """One GHN3_OP_CAST16 over `items` = dicts(src_off [floats from src_base], rows, cols, ld_src,
        straight=(off_halfs, ld, ctype) | None, transposed=(off_halfs, ld, ctype) | None, colsum=(q, s) | None)."""
        descs, blocks = self.pack_cast_descs(items, dbias is not None, amax is not None, dst_base is None)
        if blocks:
            self.op(L.OP_CAST16, refs=(src_base, dst_base or (self.xbuf(self.X_WS), 0), self.idx(descs),
                                       dbias if dbias is not None else self.NONE,
                                       amax if amax is not None else self.NONE), ints=(len(items), blocks, grid_cap),
                    flags=flags)

    @staticmethod
    def pack_cast_descs(items, dbias=False, amax=False, ws_dst=True):
        """ghn3_cast_desc table of `items` (see cast16) and its number of 64 x 64 work tiles."""
        descs = np.zeros(len(items), dtype=L.CAST_DT)
        blocks = 0
        for D, it in zip(descs, items):
            D['src_off'], D['rows'], D['cols'], D['ld_src'] = it['src_off'], it['rows'], it['cols'], it['ld_src']
            dflags = 0
            if it.get('straight'):
                off, ld, ct = it['straight']
                assert ld % 8 == 0 and ld >= round_up(it['cols'], 64) and off % 8 == 0
                D['dst_off'], D['ld_dst'] = off, ld
                dflags |= L.CAST_STRAIGHT | (L.CAST_STRAIGHT_BF16 if ct == L.CT_BF16 else 0)
            if it.get('transposed'):
                off, ld, ct = it['transposed']
                assert ld % 8 == 0 and (ld >= round_up(it['rows'], 64) or it.get('tight')) and off % 8 == 0
                D['dstT_off'], D['ld_dstT'] = off, ld
                dflags |= L.CAST_TRANSPOSED | (L.CAST_TRANSPOSED_BF16 if ct == L.CT_BF16 else 0)
            if it.get('colsum') is not None:
                assert dbias
                D['bias_q'], D['bias_s'] = it['colsum'][:2]
                D['bias_off'] = it['colsum'][2] if len(it['colsum']) > 2 else 0
                dflags |= L.CAST_COLSUM
            if it.get('colsum_parts') is not None:
                # deterministic column sums: one slot row per 64-row tile of the descriptor (dbias = the slot buffer)
                assert dbias
                D['part_off'] = it['colsum_parts']
                dflags |= L.CAST_COLSUM | L.CAST_COLSUM_PARTS
            if it.get('src_map'):
                D['src_q'], D['src_s'] = it['src_map']
                assert D['src_q'] % 4 == 0 and D['src_s'] % 4 == 0 and it['cols'] % D['src_q'] == 0
            if it.get('tight'):
                dflags |= L.CAST_TIGHT
            if it.get('split'):
                dflags |= L.CAST_SPLIT
                D['lo_off'] = it['split']
            if it.get('frag'):
                assert it.get('split') and it['rows'] % 32 == 0 and it['cols'] % 32 == 0
                dflags |= L.CAST_FRAG
            if it.get('f16s'):
                assert it.get('split') and it.get('straight')
                dflags |= L.CAST_SPLIT_F16
            if it.get('scaled'):
                assert amax
                dflags |= L.CAST_SCALED
            if it.get('src16'):
                # (source = a 16-bit matrix of the destination buffer that already carries the scale, see GHN3_CAST_SRC16)
                assert not it.get('straight') and not it.get('split') and ws_dst
                dflags |= L.CAST_SRC16
            assert it['ld_src'] % 4 == 0 and it['src_off'] % 4 == 0
            D['flags'] = dflags
            D['block_start'] = blocks
            blocks += ((it['rows'] + 63) // 64) * ((it['cols'] + 63) // 64)
        return descs, blocks

    def idx(self, arr):
        raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        off = round_up(self._idx_size, 16)
        self._idx_chunks.append((off, raw))
        self._idx_size = off + len(raw)
        return (self.xbuf(self.X_IDX), off)

    NONE = (-1, 0)

    _ABLATE = tuple(int(v) for v in os.environ.get('GHN3_ABLATE_OPS', '').split(',') if v)    # (timing experiments only)

    def op(self, kind, refs=(), ints=(), floats=(), flags=0):
        if kind in self._ABLATE:
            return
        self._ops.append((kind, flags, tuple(ints), tuple(floats), tuple(refs)))

    def _finish_ops(self):
        n = len(self._ops)
        out = np.zeros(n, dtype=L.OP_DT)
        if n:
            out['r']['buf'][:] = -1
            out['kind'] = [o[0] for o in self._ops]
            out['flags'] = [o[1] for o in self._ops]
            # ragged per-op fields scattered with one fancy assignment each
            ir, ic, iv, fr, fc, fv, rr, rc, rb, ro = ([] for _ in range(10))
            for k, (_, _, ints, floats, refs) in enumerate(self._ops):
                for j, v in enumerate(ints):
                    ir.append(k); ic.append(j); iv.append(int(v))
                for j, v in enumerate(floats):
                    fr.append(k); fc.append(j); fv.append(v)
                for j, (b_, o_) in enumerate(refs):
                    rr.append(k); rc.append(j); rb.append(b_); ro.append(o_)
            if ir:
                out['i'][ir, ic] = iv
            if fr:
                out['f'][fr, fc] = fv
            if rr:
                out['r']['buf'][rr, rc] = rb
                out['r']['off'][rr, rc] = ro
        self._ops = []
        return out

    def gemm(self, A, B, C, M, N, K, lda, ldb, ldc, a_mode=L.MODE_ROW, b_mode=L.MODE_ROW, bias=None, bias_q=0,
             bias_s=0, bias_stride=1, act=L.ACT_NONE, dact=L.DACT_NONE, aux_in=None, aux_out=None, residual=None,
             a_gather=None, b_gather=None, c_gather=None, a_qs=(0, 0), b_qs=(0, 0), c_qs=(0, 0), accum=False,
             alpha=1.0, dbias=None, dbias_stride=1, ksplit=1, op16=False, b_kmap=(0, 0), lim=None, lim_kind=0, alpha_amax=None,
             ln=None, x3=None, xcd=None, mtiles=None, sumsq=None, x3f16=False):
        # sumsq = ref of the GHN3_GEMM_SUMSQ slot table (tile code 29)
        # mtiles = (ref of int32 triples {m0, mi, extent}, count): row-tile table of the 8-phase kernel (tile code 28)
        # ln = (kind, [refs p0..p5 or None], eps): LayerNorm row prologue of A (ghn3_gemm_problem::ln_kind)
        # dbias: fused bias gradient of a wgrad problem (GHN3_GEMM_BIASGRAD): dbias[cmap(m)*stride] += sum_k A(m,k)
        if dbias is not None:
            assert bias is None and a_mode == L.MODE_COL
            bias, bias_stride = dbias, dbias_stride
        N_ = self.NONE
        flags = (L.GEMM_ACCUM if accum else 0) | (L.GEMM_BIASGRAD if dbias is not None else 0) | \
            (L.GEMM_OP16 if op16 else 0) | (L.GEMM_X3 if x3 is not None else 0) | (L.GEMM_SUMSQ if sumsq is not None else 0) | \
            (L.GEMM_X3F16 if x3f16 else 0)
        if x3f16:
            assert x3 is not None
            alpha = alpha * 2.0 ** -L.X3F16_WSHIFT
        if sumsq is not None:
            assert aux_out is None
            aux_out = sumsq
        # one plain tuple per problem (field order of _PROBLEM_REFS + _PROBLEM_INTS + the tail); the structured array
        # is packed column by column in _pack_problems -- filling a numpy record per call cost 15 us per problem
        self._probs.append((
            A or N_, B or N_, C or N_, bias or N_, residual or N_, aux_in or N_, aux_out or N_, a_gather or N_,
            b_gather or N_, c_gather or N_, lim or N_, alpha_amax or N_, (x3[0] if x3 is not None else N_),
            (mtiles[0] if mtiles is not None else N_),
            M, N, K, lda, ldb, ldc, a_mode, b_mode, a_qs[0], a_qs[1], b_qs[0], b_qs[1], c_qs[0], c_qs[1], bias_q, bias_s,
            bias_stride, act, dact, flags, b_kmap[0], b_kmap[1],
            (x3[1] if x3 is not None else 0), alpha, ksplit, (lim_kind if (lim is not None or mtiles is not None) else 0),
            (0 if xcd is None else 1 + int(xcd) % 8), (mtiles[1] if mtiles is not None else 0)))
        if ln is not None:
            self._ln[len(self._probs) - 1] = ln
        return len(self._probs) - 1

    def split_small(self, M, N, K):
        """K split of a lone 32 x 32-tile GEMM over two workgroup sets.  A [256 x 384] output is 96 tiles on 256 CUs and
        its K loop is bound by the fp32 MFMA rate of those 96 CUs (4 waves per SIMD take turns on the matrix core: ~1 us
        per 128-wide K chunk); two K halves as two problems of the same launch use twice the CUs.  The second half
        goes to a plane that the consuming LayerNorm kernel adds (and writes back), so no extra launch and a fixed
        summation order."""
        return self.split_k2 and not self.fuse_ln and K >= self.split_k2_min and \
            ((M + 31) // 32) * ((N + 31) // 32) <= 128

    def gemm_k2(self, A, B, C, plane, M, N, K, lda, ldb, ldc, b_mode, **epilogue):
        """ROW-mode A.  Problem 1: columns [0, k0) with the epilogue -> C; problem 2: [k0, K) -> plane [M][N]."""
        k0 = round_up((K + 1) // 2, 128 if K >= 512 else 4)
        p0 = self.gemm(A, B, C, M, N, k0, lda, ldb, ldc, a_mode=L.MODE_ROW, b_mode=b_mode, **epilogue)
        b_off = k0 if b_mode == L.MODE_ROW else k0 * ldb
        self.gemm((A[0], A[1] + 4 * k0), (B[0], B[1] + 4 * b_off), plane, M, N, K - k0, lda, ldb, N,
                  a_mode=L.MODE_ROW, b_mode=b_mode)
        return p0

    _PROBLEM_REFS = ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather', 'lim',
                     'alpha_amax', 'B2', 'mtiles')
    _PROBLEM_INTS = ('M', 'N', 'K', 'lda', 'ldb', 'ldc', 'a_mode', 'b_mode', 'a_q', 'a_s', 'b_q', 'b_s', 'c_q', 'c_s',
                     'bias_q', 'bias_s', 'bias_stride', 'act', 'dact', 'flags', 'b_kq', 'b_ks', 'x3_slice')

    def _pack_problems(self):
        n = len(self._probs)
        arr = np.zeros(n, dtype=L.PROBLEM_DT)
        if n == 0:
            return arr
        cols = list(zip(*self._probs))
        nr = len(self._PROBLEM_REFS)
        for k, name in enumerate(self._PROBLEM_REFS):
            ref = np.asarray(cols[k], dtype=np.int64).reshape(n, 2)
            arr[name]['buf'], arr[name]['off'] = ref[:, 0], ref[:, 1]
        for k, name in enumerate(self._PROBLEM_INTS):
            arr[name] = np.asarray(cols[nr + k], dtype=np.int64)
        k = nr + len(self._PROBLEM_INTS)
        arr['alpha'], arr['ksplit'], arr['lim_kind'], arr['xcd_pin'] = cols[k], cols[k + 1], cols[k + 2], cols[k + 3]
        arr['n_mtiles'] = cols[k + 4]
        arr['ln_p']['buf'] = -1
        for q, (kind, refs, eps) in self._ln.items():
            arr['ln_kind'][q], arr['ln_eps'][q] = kind, eps
            for e, ref in enumerate(refs):
                if ref is not None:
                    arr['ln_p']['buf'][q, e], arr['ln_p']['off'][q, e] = ref
        return arr

    # timing tags (ghn3_profile_enable mode 2): the decoder kernels that dominate the step
    TAG_D3_FWD, TAG_D3_DGRAD, TAG_D3_WGRAD, TAG_D2_FWD, TAG_D1_FWD, TAG_TILE_FWD, TAG_TILE_BWD, TAG_D2_BWD, \
        TAG_D1_BWD = range(1, 10)
    TAG_NAMES = {1: 'w2_fwd', 2: 'w2_dgrad', 3: 'w2_wgrad', 4: 'w0_fwd', 5: 'fc_fwd', 6: 'tile_fwd', 7: 'tile_bwd',
                 8: 'w0_bwd', 9: 'fc_bwd'}

    def gemm_op(self, first, count=None, tile=0, ctype=None, tag=0, side=False, flops=None, grid_cap=0):
        if count is None:
            count = len(self._probs) - first
        if count > 0:
            if tag and ctype is None:
                ctype = self.decoder_ctype
            flags = 0 if ctype is None else 1 + ctype
            if side:
                flags |= self.SIDE
            if tag:
                flags |= L.OPFLAG_TIMED | (tag << 16)
                fl = flops if flops is not None else \
                    sum(2.0 * int(p[14]) * int(p[15]) * int(p[16]) for p in self._probs[first:first + count])
                self.tag_flops[tag] = self.tag_flops.get(tag, 0.0) + fl
            self.op(L.OP_GEMM, ints=(first, count, tile, grid_cap), flags=flags)

    @staticmethod
    def _row_parts(rows):
        """Split a stacked row range into a 128-multiple part and a short remainder (<= 64 rows) so that the
        128x128 tiles waste no MFMA rows on the tail (533 rows -> 512 + 21 instead of 5 x 128 = 640)."""
        main = rows // 128 * 128
        rem = rows - main
        if main >= 128 and 0 < rem <= 64:
            return [(0, main), (main, rem)]
        return [(0, rows)]

    # cost of one k-tile step of a (64 mi) x 256 tile of the 8-phase kernel, mi = 3, 4, 5 (tools/gemm_lab.hip, relative)
    P8_COST = {3: 1.72, 4: 2.0, 5: 2.42}
    FC_DGRAD_T = os.environ.get('GHN3_FC_DGRAD_T', '0') != '0'
    P8_MI = tuple(int(v) for v in os.environ.get('GHN3_P8_MI', '3,4,5').split(','))

    @classmethod
    def row_tiles(cls, ext):
        """Row tiles of the 8-phase kernel (tile code 28) for stacked rows whose extents `ext` (one per row: valid columns of
        the forward = reduction length of the dgrad) do not increase: tiles of 192 / 256 / 320 rows, each paying for the
        extent of its FIRST row over all its rows.  Exact minimum of sum(cost[mi] * extent) by dynamic programming over
        64-row positions (533 full-width rows followed by 235 rows of a third of the width: 320 + 256 + 192 instead of
        three full-width 256-row tiles).  Returns an int32 array of {m0, mi, extent} triples."""
        ext = np.asarray(ext, dtype=np.int64)
        R = len(ext)
        if cls.P8_FINE:
            # round 6: heights of 32 n rows, n in 6 .. 10 (224 and 288 are new), positions in steps of 32 rows
            n = (R + 31) // 32
            best = [0.0] * (n + 11)
            choice = [0] * (n + 11)
            for k in range(n - 1, -1, -1):
                e = float(ext[k * 32]) + 1e-3
                best[k], choice[k] = min((cls.P8_COSTN[h] * e + best[min(k + h, n)], h) for h in cls.P8_HEIGHTS)
            out, k = [], 0
            while k < n:
                out.append((k * 32, choice[k], int(ext[k * 32])))
                k += choice[k]
            return np.asarray(out, dtype=np.int32).reshape(-1, 3)
        n = (R + 63) // 64
        best = [0.0] * (n + 6)
        choice = [0] * (n + 6)
        for k in range(n - 1, -1, -1):
            e = float(ext[k * 64]) + 1e-3           # (the epsilon: fewer tiles among equals)
            best[k], choice[k] = min((cls.P8_COST[mi] * e + best[min(k + mi, n)], mi) for mi in cls.P8_MI)
        out, k = [], 0
        while k < n:
            out.append((k * 64, 2 * choice[k], int(ext[k * 64])))      # (height code = rows / 32)
            k += choice[k]
        return np.asarray(out, dtype=np.int32).reshape(-1, 3)

    # Round 6: EQUAL row tiles per W2 panel.  Height codes 6 .. 10 = 32 n rows (192 / 224 / 256 / 288 / 320; 7 and 9 are new in
    # gemm_p8.hip), relative cost of one k-tile step interpolated between the measured 3 / 4 / 5 (tools/gemm_lab.hip).
    P8_COSTN = {2: 1.5, 4: 1.55, 6: 1.72, 7: 1.86, 8: 2.0, 9: 2.21, 10: 2.42}   # (2 / 4: bound by the W2 stream, not the matrix cores)
    P8_HEIGHTS = (2, 4, 6, 7, 8, 9, 10)                                    # row-tile heights / 32 the kernel has
    P8_BALANCED = os.environ.get('GHN3_P8_BALANCED', '1') != '0'            # forward: dense column ranges with equal tiles
    P8_BALANCED_DGRAD = os.environ.get('GHN3_P8_BALANCED_DGRAD', '0') != '0'    # dgrad: equal tiles per K chunk
    P8_FINE = os.environ.get('GHN3_P8_FINE', '1') != '0'                   # row_tiles on 32-row positions, heights 192 .. 320

    @classmethod
    def p8_cost(cls, code):
        return cls.P8_COSTN[int(code)]

    @classmethod
    def balanced_tiles(cls, rows):
        """(T, n): T EQUAL row tiles of 32 n rows (n in 6 .. 10) covering `rows` rows at the least cost T * cost[n]; ties go to
        fewer tiles.  The row tiles of one streamed W2 panel then run at the same pace on the CUs of one XCD and share the
        panel through its L2 for their whole length (533 rows: 2 x 288 instead of 256 + 320, whose faster tile runs a fifth of
        the panel ahead by the end and makes both fetch it: 5.0 GB counted for 2.9 GB consumed in round 5)."""
        best = None
        for n in cls.P8_HEIGHTS:
            t = max(1, -(-int(rows) // (32 * n)))
            c = t * cls.P8_COSTN[n]
            if best is None or c < best[0] - 1e-9 or (abs(c - best[0]) <= 1e-9 and t < best[1]):
                best = (c, t, n)
        return best[1], best[2]

    @staticmethod
    def alive_rows(gg, o_lo):
        """Rows of a stacked family that consume the W2 rows o' >= o_lo (o_r > o_lo): a prefix, the rows are sorted by decreasing o."""
        return sum(sb['rows'] for sb in gg['subs'] if sb['o'] > o_lo)

    @classmethod
    def column_ranges(cls, gg, min_cols=1024):
        """Cuts a family's forward GEMM at the distinct output widths of its members: [(o_lo, o_hi, alive rows)] with o' in
        [o_lo, o_hi) consumed by exactly the first `alive` rows.  Every range is a DENSE problem with its own equal row tiles.
        Ranges narrower than min_cols columns are merged into the next wider one (its extra rows compute don't-care columns)."""
        bounds = sorted({sb['o'] for sb in gg['subs']})
        out, o_lo = [], 0
        for o_hi in bounds:
            out.append([o_lo, o_hi, cls.alive_rows(gg, o_lo)])
            o_lo = o_hi
        k = 0
        while k < len(out) - 1:
            if (out[k][1] - out[k][0]) * gg['i_ld'] < min_cols:
                out[k + 1][0] = out[k][0]
                out[k + 1][2] = out[k][2]
                del out[k]
            else:
                k += 1
        return [tuple(r) for r in out]

    @classmethod
    def range_tiles(cls, alive, total_rows, ext_of_row, k0=0, kc=None):
        """Row-tile table {m0, n, extent} of one dense range / K chunk: T equal tiles over the `alive` rows, extent of a tile =
        ext_of_row(first row) shifted into the chunk [k0, k0 + kc) (kc None: no clipping); the rows behind them (dead in this
        chunk: K = 0, their plane rows are written as zeros) in 320-row tiles."""
        out = []
        m0 = 0
        if alive > 0:
            t, n = cls.balanced_tiles(alive)
            for _ in range(t):
                if m0 >= total_rows:
                    break
                e = int(ext_of_row(m0))
                out.append((m0, n, e - k0 if kc is None else int(np.clip(e - k0, 0, kc))))
                m0 += 32 * n
        while m0 < total_rows:
            e = int(ext_of_row(m0))
            out.append((m0, 10, e - k0 if kc is None else int(np.clip(e - k0, 0, kc))))
            m0 += 320
        return np.asarray(out, dtype=np.int32).reshape(-1, 3)

    def _dgrad_sub_split(self, g16, n_cols, n_cu=32):
        """Split of every XCD's K chunk of the W2 dgrad into parts (see _build_backward).  Candidates are scored by a
        longest-first list schedule of ONE XCD's work items (families x row tiles x 256-column tiles x parts) on its CUs;
        chunk 0 (all extents still alive) is the fullest and is the one simulated."""
        nt = (n_cols + 255) // 256
        def makespan(sub):
            items = []
            for g in g16:
                if not g.get('p8'):
                    continue
                oc = (g['o'] + 7) // 8
                if self.P8_BALANCED_DGRAD:                 # chunk 0: every row alive, equal tiles
                    t_, n_ = self.balanced_tiles(g['rows'])
                    tiles_ = [(self.P8_COSTN[n_], min(int(g['ext'][min(q * 32 * n_, g['rows'] - 1)]), oc * g['i_ld']))
                              for q in range(t_) if q * 32 * n_ < g['rows']]
                else:
                    tiles_ = [(self.p8_cost(mi), min(int(ext), oc * g['i_ld'])) for (m0, mi, ext) in g['mtiles']]
                for cost_, ext_ in tiles_:
                    steps = ext_ / 64.0
                    for f in sub:
                        items += [cost_ * steps * f / sum(sub)] * nt
            items.sort(reverse=True)
            cu = [0.0] * n_cu
            for it in items:
                k = cu.index(min(cu))
                cu[k] += it + 4.0                      # (+ prologue / epilogue of a tile, in the same relative units)
            return max(cu) if items else 0.0
        best, best_sub = None, (1,)
        for sub in ((1,), (3, 1), (1, 1), (2, 1, 1), (1, 1, 1, 1)):
            score = makespan(sub) + 14.0 * 8 * (len(sub) - 1)     # (a partial plane: ~7 us written + read again)
            if best is None or score < best - 1e-9:
                best, best_sub = score, sub
        return best_sub

    # ------------------------------------------------------------------ decoder layout (host bookkeeping)
    def _src_row(self, ind):
        """xe row read for sparse-flat node index `ind` (quirk Q1: reference reads dense-flat row `ind`)."""
        if self.index_mode == 'reference':
            return ind
        b, off = 0, 0
        while ind >= off + self.n_nodes[b]:
            off += self.n_nodes[b]
            b += 1
        return b * self.N + (ind - off)

    def _layout_decoder(self):
        C, S = self.C, self.S
        ms = self.max_shape
        half = S // 2
        self.conv_groups, self.oned_plain, self.oned_clsb = [], [], []
        row = 0
        for key, inds in self.param_groups.items():
            if len(inds) == 0:
                continue
            if len(key) == 4 or (len(key) == 2 and key[1] > 0):
                if len(key) == 4:
                    kh, kw = int(key[2]), int(key[3])
                    o_g, i_g = min(int(key[0]), ms[0]), min(int(key[1]), ms[1])
                    # input widths that are not a multiple of 8 (the 3 -> 4 channels of a stem) are padded to 8 in the
                    # 16-bit pipeline: the group then runs with the families (forward, dgrad, wgrad bands) instead of as
                    # three lone launches of the fp32-operand kernels; the extra tile columns (real W2 rows i' < 8) are
                    # computed but never consumed, their gradient is written as zeros by the tile backward
                    i_ld = i_g
                    if self.direct16 and i_g % 8 and round_up(i_g, 8) <= ms[1] and \
                            os.environ.get('GHN3_PAD_I8', '1') != '0':
                        i_ld = round_up(i_g, 8)
                    g = dict(kind='conv', key=key, inds=list(inds), o=o_g, i=i_g, i_ld=i_ld, kh=kh, kw=kw,
                             cols=o_g * i_ld)
                    if min(kh, kw) > S:
                        # nn.py:751-753: the whole 16x16 grid is decoded and bilinearly resized to (kh, kw); the
                        # reference asserts a single node per such group.  Decoded at the grid size here; the resize
                        # is a constant (kh*kw x S*S) matrix applied by one GEMM (_resize_matrix).
                        if len(inds) != 1:
                            raise ValueError('kernels larger than the %dx%d decoder grid need one node per group '
                                             '(nn.py:752), got %d for key %s' % (S, S, len(inds), str(key)))
                        g['resize'] = (kh, kw)
                        kh = kw = g['kh'] = g['kw'] = S
                    elif max(kh, kw) > S:
                        raise NotImplementedError('kernel %s exceeds the %dx%d decoder grid along one axis only '
                                                  '(the reference fails its shape assert, nn.py:550)' % (str(key), S, S))
                else:
                    kh = kw = 1
                    i_true = min(int(key[1]), ms[1])
                    i_ld = round_up(i_true, 4)
                    g = dict(kind='cls', key=key, inds=list(inds), o=ms[0], i=i_true, i_ld=i_ld, kh=1, kw=1,
                             cols=ms[0] * i_ld)
                g['hw'] = kh * kw
                y0, x0 = max(0, half - kh // 2), max(0, half - kw // 2)
                g['pos'] = np.asarray([(y0 + y) * S + (x0 + x) for y in range(kh) for x in range(kw)],
                                      dtype=np.int32)
                # leading dimension of the tile buffer: +64 floats so that rows are never a power-of-two apart
                # (the wgrad reads this buffer k-strided; 2^n strides alias onto the same HBM channels)
                g['ld'] = round_up(g['cols'], 4) + 64
                g['rows'] = len(inds) * g['hw']
                self.conv_groups.append(g)
            else:
                if len(key) == 2 and key[1] < 0:
                    self.oned_clsb.append((key, list(inds)))
                else:
                    self.oned_plain.append((key, list(inds)))
        # The W2 GEMM of a group depends only on (o, i): parameter groups that differ just in the kernel size
        # (1x1, 3x3, 5x5, 7x7 ...) are stacked along M into one GEMM problem ("gemm group").  This removes most
        # of the tile-quantisation waste of the reference's per-key grouping (many groups have < 32 rows).
        # In the 16-bit pipeline ("families") groups are stacked even further: all conv groups with the same input
        # width i (i % 8 == 0) form ONE gemm group, rows sorted by decreasing o.  Row r then needs the W2 rows
        # o' < o_r only -- a ragged extent that the kernel receives as one limit per 128 rows (forward: columns,
        # dgrad: reduction length, ghn3_gemm_problem::lim) -- and every W2 tile is streamed from HBM once per
        # family instead of once per (o, i) group.
        # The classifier-weight rows (nn.py:755-758: tile over ALL C output rows, then relu -> class_layer_predictor) are
        # family members like any convolution row with o = C (one row alone cost a whole 128-row tile sweep over W2:
        # 0.19 ms of the forward at ghn3xlm16, again in the dgrad); their ReLU is applied afterwards by GHN3_OP_RELU_FIX.
        fam = lambda g_: self.direct16 and g_['i_ld'] % 8 == 0 and \
            (g_['kind'] == 'conv' or os.environ.get('GHN3_CLS_FAMILY', '1') != '0')
        # (groups the 16-bit pipeline can take -- i % 8 == 0 -- come first: their decoder rows are contiguous, which the
        # plane-wise dgrad reduction relies on)
        cap16 = lambda g_: self.direct16 and g_['i_ld'] % 8 == 0
        order = sorted(range(len(self.conv_groups)),
                       key=lambda k: (not cap16(self.conv_groups[k]),
                                      self.conv_groups[k]['kind'] == 'cls' and not fam(self.conv_groups[k]),
                                      (0, -self.conv_groups[k]['i_ld'], -self.conv_groups[k]['o'])
                                      if fam(self.conv_groups[k]) else
                                      (1, -self.conv_groups[k]['cols'], self.conv_groups[k]['o']),
                                      self.conv_groups[k]['i_ld'], k))
        self.conv_groups = [self.conv_groups[k] for k in order]
        self.gemm_groups = []
        for g in self.conv_groups:
            g['row0'] = row
            row += g['rows']
            last = self.gemm_groups[-1] if self.gemm_groups else None
            same = last is not None and last['i_ld'] == g['i_ld'] and \
                ((fam(g) and last['family']) or
                 (g['kind'] == 'conv' and last['kind'] == 'conv' and last['o'] == g['o']))
            if same:
                last['rows'] += g['rows']
                last['members'].append(g)
            else:
                self.gemm_groups.append(dict(kind=g['kind'], o=g['o'], i=g['i'], i_ld=g['i_ld'], cols=g['cols'],
                                             ld=g['ld'], row0=g['row0'], rows=g['rows'], members=[g],
                                             family=fam(g)))
        for gg in self.gemm_groups:
            for m in gg['members']:
                m['ld'] = gg['ld']            # every member lives in the family's tile matrix (widest extent)
            # sub-ranges of equal o (members are sorted by decreasing o) and the per-128-row extents
            subs = []
            for m in gg['members']:
                if subs and subs[-1]['o'] == m['o']:
                    subs[-1]['rows'] += m['rows']
                else:
                    subs.append(dict(o=m['o'], row0=m['row0'], rows=m['rows'], cols=m['o'] * gg['i_ld']))
            gg['subs'] = subs
            ext = np.zeros(gg['rows'], dtype=np.int32)
            for sb in subs:
                ext[sb['row0'] - gg['row0']: sb['row0'] - gg['row0'] + sb['rows']] = sb['cols']
            gg['lim128'] = np.asarray([ext[t0:t0 + 128].max() for t0 in range(0, gg['rows'], 128)], dtype=np.int32)
            gg['ragged'] = len(subs) > 1
            gg['ext'] = ext
            # row tiles of the 8-phase kernel (families with at least ~a tile of rows; the rest runs on 128 x 128 tiles)
            # (round 6: with 64- / 128-row tiles an INFERENCE forward streams every family's W2 rows through the LDS-DMA ring --
            # 230 decoder rows of a ResNet-50 are four families of 32 .. 109 rows --; training keeps the small families on the
            # 128 x 128 kernel unless GHN3_P8_MIN_ROWS says otherwise: their dgrad K chunks are too short for XCD-pinned tiles)
            min_rows = int(os.environ.get('GHN3_P8_MIN_ROWS', '160' if self.training else '8'))
            gg['p8'] = self.use_p8 and self.direct16 and gg['i_ld'] % 8 == 0 and gg['rows'] >= min_rows
            if gg['p8']:
                gg['mtiles'] = self.row_tiles(ext)
        self.M = row
        # per-row arrays
        self.row_src = np.zeros(self.M, dtype=np.int32)
        self.row_pos = np.zeros(self.M, dtype=np.int32)
        for g in self.conv_groups:
            for n_idx, ind in enumerate(g['inds']):
                r = g['row0'] + n_idx * g['hw']
                self.row_src[r:r + g['hw']] = self._src_row(ind)
                self.row_pos[r:r + g['hw']] = g['pos']
        # 1-D nodes: plain first, then last-bias (cls-b) nodes
        self.oned_index = {}                # sparse-flat node index -> row in the 1-D buffers
        rows = []
        for _, inds in self.oned_plain:
            for ind in inds:
                self.oned_index[ind] = len(rows)
                rows.append(self._src_row(ind))
        self.n1_plain = len(rows)
        for _, inds in self.oned_clsb:
            for ind in inds:
                self.oned_index[ind] = len(rows)
                rows.append(self._src_row(ind))
        self.n1 = len(rows)
        self.n1_clsb = self.n1 - self.n1_plain
        self.oned_src = np.asarray(rows, dtype=np.int32)

    X3_WEIGHTS = (('attn.to_qkv.weight', 3, 1), ('attn.to_out.0.weight', 1, 1), ('ff.net.0.weight', 4, 1),
                  ('ff.net.3.weight', 1, 4))          # (name, rows / C, cols / C)

    @staticmethod
    def shadow_layout(C, max_shape, layers=0):
        """Offsets (16-bit elements) of the persistent 16-bit copies of GHN weights inside the model-owned shadow buffer
        (X_SHADOW) and its size in bytes.  Decoder: W2 [C^2][8C] (forward B operand), W2^T [8C][C^2 + 64] (dgrad B
        operand), W0^T [4C][8C] (D2 dgrad B operand).  Graphormer (GHN3_GEMM_X3), per layer and linear weight W [r][c]:
        one block [hi straight r x c | hi transposed c x r] and the identical block of the lo halves `lo` elements behind it
        (straight = forward B operand, transposed = dgrad B operand).  Depends on the model only, never on the batch."""
        n_w2 = int(max_shape[0]) * int(max_shape[1])
        w2hT_ld = round_up(n_w2, 64) + 64
        w2h = 0
        w2hT = round_up(w2h + n_w2 * 8 * C + 128, 128)
        w0hT = round_up(w2hT + 8 * C * w2hT_ld + 128, 128)
        # bf16 hi / lo copies of decoder.fc.0.weight, transposed per grid position ([position p][C][4C], k-contiguous), for the
        # split-bf16 fc dgrad (see _cast_w2)
        # (off by default since the weight gradient is issued behind the decoder backward: 6.97 ms per step either way, and
        # the copies are 604 MB at ghn3xlm16; GHN3_FC_DGRAD_T=1 enables them)
        S2 = int(max_shape[2]) * int(max_shape[3])
        wfcT = round_up(w0hT + 4 * C * 8 * C + 128, 128)
        pos = round_up(wfcT + (2 * S2 * C * 4 * C if Program.FC_DGRAD_T else 0) + 128, 128)
        lay = dict(w2h=w2h, w2hT=w2hT, w2hT_ld=w2hT_ld, w0hT=w0hT, wfcT=wfcT, x3={})
        for l in range(layers):
            for name, r, c in Program.X3_WEIGHTS:
                n = r * C * c * C
                lay['x3']['gnn.%d.%s' % (l, name)] = dict(hi=pos, hiT=pos + n, lo=2 * n, rows=r * C, cols=c * C)
                pos += 4 * n
        lay['nbytes'] = 2 * round_up(pos + 128, 128)
        return lay

    def sref(self, off_halfs):
        return (self.xbuf(self.X_SHADOW), 2 * int(off_halfs))

    def _cast_w2(self):
        """16-bit operand copies of decoder.conv.2.weight: W2 [C^2][8C] (forward B operand) and its transpose
        [8C][C^2 + 64] (dgrad B operand, backward type); one pass over W2 writes both.  They depend on nothing but
        the parameters: the ops form their own program (Program.shadow_ops) that writes a persistent buffer owned by
        the model; GHN3 replays it only after the parameters changed (optimizer step, load_state_dict), on the side
        stream under the Graphormer, and every forward in between reuses the copies (was 1.65 ms / 3.6 GB per step)."""
        C, ms = self.C, self.max_shape
        for g in self.gemm_groups:
            g['op16'] = self.direct16 and g['i_ld'] % 8 == 0
        op16 = self.uses_op16 = any(g['op16'] for g in self.gemm_groups)
        self.uses_shadow = op16 or self.x3
        if not self.uses_shadow:
            return
        fct, bct = self.decoder_ctype, self.decoder_bwd_ctype
        n_w2 = ms[0] * ms[1]
        lay = self.shadow_lay = self.shadow_layout(C, ms, self.Lyr if self.x3 else 0)
        shadow = (self.xbuf(self.X_SHADOW), 0)
        if op16:
            self.w2h = lay['w2h']
            item = dict(src_off=0, rows=n_w2, cols=8 * C, ld_src=8 * C, straight=(self.w2h, 8 * C, fct))
            if self.training:
                self.w2hT_ld = lay['w2hT_ld']
                self.w2hT = lay['w2hT']
                item['transposed'] = (self.w2hT, self.w2hT_ld, bct)
            # (the op a fused optimizer step makes unnecessary: GHN3_OP_ADAMW_CAST16 writes the same copies while it updates
            # W2 -- FusedAdamW.step(plan=...), GHN3._refresh_shadows then replays shadow_ops_rest)
            self.shadow_w2 = dict(op=len(self._ops), item=item, name='decoder.conv.2.weight')
            self.cast16(self.pref('decoder.conv.2.weight'), [item], flags=self.SIDE,
                        grid_cap=int(os.environ.get('GHN3_W2CAST_CAP', '512')) if self.SIDE else 0, dst_base=shadow)
            if self.training and (4 * C) % 64 == 0:
                # decoder.conv.0.weight^T [4C][8C] (backward type): B operand of the D2 dgrad
                self.w0hT = lay['w0hT']
                self.cast16(self.pref('decoder.conv.0.weight'),
                            [dict(src_off=0, rows=8 * C, cols=4 * C, ld_src=4 * C, transposed=(self.w0hT, 8 * C, bct))],
                            flags=self.SIDE, dst_base=shadow)
        if self.training and self.FC_DGRAD_T and (4 * C) % 64 == 0 and C % 4 == 0:
            # decoder.fc.0.weight transposed per grid position, as bf16 hi / lo copies (GHN3_GEMM_X3 B operand):
            # wfcT[p][n][ch] = Wfc[ch * 256 + p][n].  The fc dgrad reduces over ch; on the exact-fp32 small-problem kernel it
            # was ~50 problems (one per used grid position) of a few dozen rows, 12 column tiles each walking K = 4C in
            # latency-bound 128-wide chunks: 0.39 ms on the critical path (4 TFLOP/s).  Split-bf16 products against these
            # copies run the same problems with the whole K slice in LDS.  Written on the side stream when the weights
            # changed, used by the backward only.
            S2 = self.S * self.S
            self.wfcT = lay['wfcT']
            self.wfcT_lo = S2 * C * 4 * C
            items = [dict(src_off=p_ * C, rows=4 * C, cols=C, ld_src=S2 * C,
                          transposed=(self.wfcT + p_ * C * 4 * C, 4 * C, L.CT_BF16), split=self.wfcT_lo) for p_ in range(S2)]
            self.cast16(self.pref('decoder.fc.0.weight'), items, flags=self.SIDE, dst_base=shadow)
        if self.x3:
            # bf16 hi / lo copies of the Graphormer linears, straight (forward) and transposed (dgrad): ONE launch for all
            # layers (the source base of a cast op is one parameter; every weight is addressed from the first layer's first
            # one -- the flat parameter buffer is contiguous) -- on the main stream: layer 0 needs them at once.
            # (One launch per layer until round 4: 24 dependent launches of ~14 us each in front of every forward of a
            # TRAINING step, where the weights change every step; GHN3_X3_CAST_PER_LAYER=1 restores that.)
            per_layer = os.environ.get('GHN3_X3_CAST_PER_LAYER', '0') != '0'
            base0 = 'gnn.0.%s' % self.X3_WEIGHTS[0][0]
            items = []
            for l in range(self.Lyr):
                base = 'gnn.%d.%s' % (l, self.X3_WEIGHTS[0][0]) if per_layer else base0
                for name, r, c in self.X3_WEIGHTS:
                    e = lay['x3']['gnn.%d.%s' % (l, name)]
                    it = dict(src_off=self.param_gap(base, 'gnn.%d.%s' % (l, name)), rows=e['rows'], cols=e['cols'],
                              ld_src=e['cols'], straight=(e['hi'], e['cols'], L.CT_BF16), split=e['lo'], frag=self.x3s,
                              f16s=self.x3f16)
                    if self.training:
                        it['transposed'] = (e['hiT'], e['rows'], L.CT_BF16)
                    items.append(it)
                if per_layer:
                    self.cast16(self.pref(base), items, dst_base=shadow)
                    items = []
            if items:
                self.cast16(self.pref(base0), items, dst_base=shadow)

    def param_gap(self, a, b):
        """floats from the start of parameter `a` to the start of parameter `b` in the flat parameter buffer (slot
        order, every slot padded to 64 floats: GHN3._flatten)."""
        sa, sb = self.slot[a], self.slot[b]
        assert sa <= sb
        # running sums from slot sa, extended on demand (the one-launch cast of all Graphormer layers asks for the gap from the
        # first layer to every later weight: summing afresh each time was 25 of the 41 ms a ghn3xlm16 program took to compile)
        cache = self.__dict__.setdefault('_gap_cache', {})
        pre = cache.setdefault(sa, [0])
        while len(pre) <= sb - sa:
            pre.append(pre[-1] + round_up(self.param_numel(self.names[sa + len(pre) - 1]), 64))
        return pre[sb - sa]

    def param_numel(self, name):
        C, H, K, ms = self.C, self.H, self.K, self.max_shape
        tail = name.split('.', 2)[2] if name.startswith('gnn.') else name
        table = {'ln1.weight': C, 'ln1.bias': C, 'ln2.weight': C, 'ln2.bias': C, 'attn.to_qkv.weight': 3 * C * C,
                 'attn.to_out.0.weight': C * C, 'attn.to_out.0.bias': C, 'ff.net.0.weight': 4 * C * C,
                 'ff.net.0.bias': 4 * C, 'ff.net.3.weight': 4 * C * C, 'ff.net.3.bias': C}
        return table[tail]

    # ---- split-bf16 Graphormer linears (GHN3_GEMM_X3) ----------------------------------------------------------
    @staticmethod
    def x3_split(M, N, K, may_split):
        """(tile code, K splits over workgroups, K slice per staging round) of a [M x N x K] Graphormer linear.
        The kernel is bound by what one CU can pull through L2 -> LDS (~17 B/clk): the bytes per workgroup are
        4 K_slice (BM + BN), so narrow outputs (N <= 512: to_out, ff.net.3, the dgrads of to_qkv / ff.net.0) take 32 x 32
        tiles and -- where the consumer is a LayerNorm op that can sum partial planes -- K slices of 192 (then 128, 64)
        run by different workgroup sets; wide outputs (to_qkv, ff.net.0, ff.net.3 dgrad) took 32 x 64 tiles over the whole
        K = C until round 2 (see below).  Measured at ghn3xlm16 (tools/diag/x3_bench.py, us per dependent launch, exact-fp32 kernel in brackets):
        to_qkv 6.4 (9.9), to_out 4.4 (6.5), ff.net.0 6.4 (10.3), ff.net.3 7.3 (10.0 with its two K halves)."""
        nk = K // 64
        whole = max(d for d in (6, 4, 3, 2, 1) if nk % d == 0)         # k-tiles per slice without a split
        plan = os.environ.get('GHN3_X3_PLAN', 'A')
        if N > 512:
            # 32 x 32 tiles walking K in slices of 192: 49 KB of LDS per workgroup, so that several of the 288-384 workgroups
            # share a CU -- a workgroup's fetch rate is ~5 B/clk per WAVE (20 B/clk for a lone 4-wave workgroup with its
            # 147 KB at 32 x 64 / whole K; tools/fetch_rate_probe.hip).  Step 8.35 -> 8.20 ms; GHN3_X3_PLAN=W: the former plan
            if plan != 'W' and K % 192 == 0:
                return 42, 1, 192
            return 40, 1, 64 * whole
        if plan == 'B':                                                  # 32 x 64 tiles, slices of up to 384: few planes
            return 40, (nk // whole if may_split else 1), 64 * whole
        if plan == 'C':                                                  # 64 x 64 tiles, slices of 192 / 128
            for d in (3, 2, 4, 1):
                if nk % d == 0 and (nk // d <= 8 and may_split or nk == d):
                    return 41, nk // d, 64 * d
            return 40, (nk // whole if may_split else 1), 64 * whole
        if not may_split:
            return 42, 1, 64 * whole
        for d in (3, 2, 4, 6, 1):
            if nk % d == 0 and nk // d <= 8:
                return 42, nk // d, 64 * d
        return 42, 1, 64 * whole

    def x3_linear(self, A, wname, transposed, Cref, M, N, K, lda, ldc, split=False, planes=None, **epi):
        """C = A W^T (nn.Linear forward, W [N][K]) or, transposed, C = A W (dgrad, W [K][N]) with split-bf16 operands.
        With split, K is cut into `ks` slices run by different workgroups: slice 0 carries the epilogue into C, slice
        j > 0 goes to plane j - 1 of the workspace buffer `planes` ([ks - 1][M][N], summed by the consuming LayerNorm).
        Returns (ks - 1, planes ref or None) and launches the problems."""
        e = self.shadow_lay['x3'][wname]
        hi = e['hiT'] if transposed else e['hi']
        tile, ks, slice_ = self.x3_split(M, N, K, split)
        pref = None
        if ks > 1:
            pref = self.wsf(planes, (ks - 1) * M * N)
        p0 = len(self._probs)
        kc = K // ks
        for j in range(ks):
            Cj = Cref if j == 0 else (pref[0], pref[1] + 4 * (j - 1) * M * N)
            self.gemm((A[0], A[1] + 4 * j * kc), self.sref(hi + j * kc), Cj, M, N, kc, lda, K, ldc if j == 0 else N,
                      x3=(self.sref(hi + e['lo'] + j * kc), slice_), **(epi if j == 0 else {}))
        self.gemm_op(p0, tile=tile)
        return ks - 1, pref

    def x3s_linear(self, A, wname, transposed, Cref, M, N, K, lda, ldc, ln=None, **epi):
        """C = f(A) W^T (nn.Linear forward, W [N][K]) or, transposed, C = f(A) W (dgrad, W [K][N]) on the staged split-bf16
        kernels (gemm_x3d.hip): 32 x 48 tiles (tile code 44) for the wide outputs, 16 x 32 tiles with the reduction split
        over the waves (45) for N <= 512; f = the LayerNorm forward / backward row prologue `ln` = (kind, refs) or nothing."""
        e = self.shadow_lay['x3'][wname]
        hi = e['hiT'] if transposed else e['hi']
        p0 = self.gemm(A, self.sref(hi), Cref, M, N, K, lda, K, ldc, x3=(self.sref(hi + e['lo']), K),
                       ln=None if ln is None else (ln[0], ln[1], 1e-5), x3f16=self.x3f16 and not transposed, **epi)
        self.gemm_op(p0, tile=45 if N <= 512 else 44)

    # ------------------------------------------------------------------ forward
    def _build_forward(self):
        C, H, B, N, V, K = self.C, self.H, self.B, self.N, self.V, self.K
        rows = B * N
        F = 4
        ldT = round_up(H, 4)
        ldK = round_up(K, 4)
        self.ldT, self.ldK = ldT, ldK
        train = self.training

        node_off = np.cumsum([0] + self.n_nodes[:-1]).astype(np.int32)
        r_types = self.idx(self.node_types)
        r_shape = self.idx(self.shape_idx.astype(np.int32))
        r_nn = self.idx(np.asarray(self.n_nodes, dtype=np.int32))
        r_noff = self.idx(node_off)
        self.r_nn = r_nn

        deg_in = (self.xbuf(self.X_WS), self.ws('deg_in', 4 * rows))
        deg_out = (self.xbuf(self.X_WS), self.ws('deg_out', 4 * rows))
        dist0 = (self.xbuf(self.X_WS), self.ws('dist0', 4 * rows))
        pair = (self.xbuf(self.X_WS), self.ws('pair', 4 * rows * N))
        self.r_graph = (r_types, r_shape, r_nn, r_noff, deg_in, deg_out, dist0, pair)

        self.op(L.OP_GRAPH_PROLOGUE, refs=((self.xbuf(self.X_EDGES), 0), deg_in, deg_out, dist0, pair),
                ints=(B, N, V))
        x0 = self.wsf('x0', rows * C)
        self.op(L.OP_EMBED_NODES,
                refs=(x0, r_types, r_shape, r_nn, r_noff, self.pref('embed.weight'),
                      self.pref('shape_enc.embed_channel.weight'), self.pref('shape_enc.embed_spatial.weight'),
                      self.pref('gnn.0.centrality_embed_in.weight'), self.pref('gnn.0.centrality_embed_out.weight'),
                      self.pref('gnn.0.input_dist_embed.weight'), deg_in, deg_out, dist0),
                ints=(B, N, C))

        # ---- edge bias table (layer 0), graphormer.py:115-117 factorised --------------------------
        Pfw, Pbw = self.wsf('Pfw', V * C), self.wsf('Pbw', V * C)
        hid = self.wsf('hid', V * V * C)
        T = self.wsf('T', V * V * ldT)
        bias = self.wsf('bias', B * H * N * N)
        E = 'gnn.0.attn.edge_embed.embed.weight'
        W0e, b0e = 'gnn.0.attn.proj_e.0.weight', 'gnn.0.attn.proj_e.0.bias'
        W2e, b2e = 'gnn.0.attn.proj_e.2.weight', 'gnn.0.attn.proj_e.2.bias'
        p0 = self.gemm(self.pref(E, 2 * C), self.pref(W0e, 0), Pfw, V, C, C, C, 2 * C, C)
        self.gemm(self.pref(E, 2 * C), self.pref(W0e, C), Pbw, V, C, C, C, 2 * C, C, bias=self.pref(b0e))
        self.gemm_op(p0)
        self.op(L.OP_EDGE_HIDDEN, refs=(hid, Pfw, Pbw), ints=(V, C))
        p0 = self.gemm(hid, self.pref(W2e), T, V * V, H, C, C, C, ldT, bias=self.pref(b2e))
        self.gemm_op(p0)
        self.op(L.OP_BIAS_GATHER, refs=(bias, T, pair), ints=(B, N, H))

        # ---- Graphormer layers --------------------------------------------------------------------
        x_in = x0
        x_plane = None             # second K half of the previous layer's ff.net.3 (split_small)
        for l in range(self.Lyr):
            pre = 'gnn.%d.' % l
            sfx = '_%d' % l
            h1 = self.wsf('h1' + sfx, rows * C)
            m1, r1 = self.wsf('m1' + sfx, rows), self.wsf('r1' + sfx, rows)
            qkv = self.wsf('qkv' + sfx, rows * 3 * C)
            Pm = self.wsf('P' + sfx, B * H * N * N) if train else None
            o = self.wsf('o' + sfx, rows * C)
            xmid = self.wsf('xmid' + sfx, rows * C)
            h2 = self.wsf('h2' + sfx, rows * C)
            m2, r2 = self.wsf('m2' + sfx, rows), self.wsf('r2' + sfx, rows)
            z = self.wsf('z' + sfx, rows * 4 * C)
            f = self.wsf('f' + sfx, rows * 4 * C)
            x_out = self.wsf('x%d' % (l + 1), rows * C)
            x3_here = l < self.Lyr - self.x3_exact_last
            if self.x3s and x3_here:
                # staged split-bf16 linears: LayerNorm 1 / 2 are row prologues of to_qkv / ff.net.0 (their by-products --
                # normalised rows, mean, rstd -- are written by the column-tile-0 workgroups when the backward needs them),
                # the narrow linears split K inside the workgroup: five dependent launches per layer
                self.x3s_linear(x_in, pre + 'attn.to_qkv.weight', False, qkv, rows, 3 * C, C, C, 3 * C,
                                ln=(1, [self.pref(pre + 'ln1.weight'), self.pref(pre + 'ln1.bias'),
                                        m1 if train else None, r1 if train else None, h1 if train else None]))
                self.op(L.OP_ATTN_FWD, refs=(o, qkv, bias, Pm if Pm is not None else self.NONE, r_nn),
                        ints=(B, N, C, H))
                self.x3s_linear(o, pre + 'attn.to_out.0.weight', False, xmid, rows, C, C, C, C,
                                bias=self.pref(pre + 'attn.to_out.0.bias'), residual=x_in)
                self.x3s_linear(xmid, pre + 'ff.net.0.weight', False, f, rows, 4 * C, C, C, 4 * C,
                                ln=(1, [self.pref(pre + 'ln2.weight'), self.pref(pre + 'ln2.bias'),
                                        m2 if train else None, r2 if train else None, h2 if train else None]),
                                bias=self.pref(pre + 'ff.net.0.bias'), act=L.ACT_GELU, aux_out=z if train else None)
                self.x3s_linear(f, pre + 'ff.net.3.weight', False, x_out, rows, C, 4 * C, 4 * C, C,
                                bias=self.pref(pre + 'ff.net.3.bias'), residual=xmid)
                x_in = x_out
                continue
            if self.x3 and x3_here:
                # split-bf16 linears (GHN3_GEMM_X3); K splits of the linear-epilogue GEMMs go to partial planes that the
                # next LayerNorm op sums (and writes back) -- x_plane: (number of planes, ref) of the previous ff.net.3
                self.op(L.OP_LAYERNORM_FWD, refs=(h1, x_in, self.pref(pre + 'ln1.weight'), self.pref(pre + 'ln1.bias'),
                                                  m1, r1, x_plane[1] if x_plane else self.NONE),
                        ints=(rows, C, x_plane[0] if x_plane else 0, rows * C), floats=(1e-5,))
                x_plane = None
                self.x3_linear(h1, pre + 'attn.to_qkv.weight', False, qkv, rows, 3 * C, C, C, 3 * C)
                self.op(L.OP_ATTN_FWD, refs=(o, qkv, bias, Pm if Pm is not None else self.NONE, r_nn),
                        ints=(B, N, C, H))
                np_, pl = self.x3_linear(o, pre + 'attn.to_out.0.weight', False, xmid, rows, C, C, C, C, split=True,
                                         planes='xmid_plane', bias=self.pref(pre + 'attn.to_out.0.bias'), residual=x_in)
                self.op(L.OP_LAYERNORM_FWD, refs=(h2, xmid, self.pref(pre + 'ln2.weight'), self.pref(pre + 'ln2.bias'),
                                                  m2, r2, pl if np_ else self.NONE), ints=(rows, C, np_, rows * C),
                        floats=(1e-5,))
                self.x3_linear(h2, pre + 'ff.net.0.weight', False, f, rows, 4 * C, C, C, 4 * C,
                               bias=self.pref(pre + 'ff.net.0.bias'), act=L.ACT_GELU, aux_out=z if train else None)
                np_, pl = self.x3_linear(f, pre + 'ff.net.3.weight', False, x_out, rows, C, 4 * C, 4 * C, C,
                                         split=bool(self.layernorm or l + 1 < self.Lyr), planes='x_plane',
                                         bias=self.pref(pre + 'ff.net.3.bias'), residual=xmid)
                x_plane = (np_, pl) if np_ else None
                x_in = x_out
                continue
            if self.fuse_ln:
                p0 = self.gemm(x_in, self.pref(pre + 'attn.to_qkv.weight'), qkv, rows, 3 * C, C, C, C, 3 * C,
                               ln=(1, [self.pref(pre + 'ln1.weight'), self.pref(pre + 'ln1.bias'),
                                       m1 if train else None, r1 if train else None, h1 if train else None], 1e-5))
            else:
                self.op(L.OP_LAYERNORM_FWD, refs=(h1, x_in, self.pref(pre + 'ln1.weight'), self.pref(pre + 'ln1.bias'),
                                                  m1, r1, x_plane or self.NONE), ints=(rows, C), floats=(1e-5,))
                x_plane = None
                p0 = self.gemm(h1, self.pref(pre + 'attn.to_qkv.weight'), qkv, rows, 3 * C, C, C, C, 3 * C)
            self.gemm_op(p0)
            self.op(L.OP_ATTN_FWD, refs=(o, qkv, bias, Pm if Pm is not None else self.NONE, r_nn),
                    ints=(B, N, C, H))
            p0 = self.gemm(o, self.pref(pre + 'attn.to_out.0.weight'), xmid, rows, C, C, C, C, C,
                           bias=self.pref(pre + 'attn.to_out.0.bias'), residual=x_in)
            self.gemm_op(p0)
            if self.fuse_ln:
                p0 = self.gemm(xmid, self.pref(pre + 'ff.net.0.weight'), f, rows, 4 * C, C, C, C, 4 * C,
                               bias=self.pref(pre + 'ff.net.0.bias'), act=L.ACT_GELU, aux_out=z if train else None,
                               ln=(1, [self.pref(pre + 'ln2.weight'), self.pref(pre + 'ln2.bias'),
                                       m2 if train else None, r2 if train else None, h2 if train else None], 1e-5))
            else:
                self.op(L.OP_LAYERNORM_FWD, refs=(h2, xmid, self.pref(pre + 'ln2.weight'), self.pref(pre + 'ln2.bias'),
                                                  m2, r2), ints=(rows, C), floats=(1e-5,))
                p0 = self.gemm(h2, self.pref(pre + 'ff.net.0.weight'), f, rows, 4 * C, C, C, C, 4 * C,
                               bias=self.pref(pre + 'ff.net.0.bias'), act=L.ACT_GELU, aux_out=z if train else None)
            self.gemm_op(p0)
            if self.split_small(rows, C, 4 * C) and (self.layernorm or l + 1 < self.Lyr) and not self.x3:
                # x_out = xmid + f W3^T + b3 in two K halves; the next LayerNorm adds the second one
                x_plane = self.wsf('x_plane', rows * C)
                p0 = self.gemm_k2(f, self.pref(pre + 'ff.net.3.weight'), x_out, x_plane, rows, C, 4 * C, 4 * C, 4 * C, C,
                                  L.MODE_ROW, bias=self.pref(pre + 'ff.net.3.bias'), residual=xmid)
            else:
                p0 = self.gemm(f, self.pref(pre + 'ff.net.3.weight'), x_out, rows, C, 4 * C, 4 * C, 4 * C, C,
                               bias=self.pref(pre + 'ff.net.3.bias'), residual=xmid)
            self.gemm_op(p0)
            x_in = x_out
        xe = self.wsf('xe', rows * C)
        mf, rf = self.wsf('mf', rows), self.wsf('rf', rows)
        if self.layernorm:
            if isinstance(x_plane, tuple) and isinstance(x_plane[0], int) and self.x3:
                self.op(L.OP_LAYERNORM_FWD, refs=(xe, x_in, self.pref('ln.weight'), self.pref('ln.bias'), mf, rf,
                                                  x_plane[1]), ints=(rows, C, x_plane[0], rows * C), floats=(1e-5,))
            else:
                self.op(L.OP_LAYERNORM_FWD, refs=(xe, x_in, self.pref('ln.weight'), self.pref('ln.bias'), mf, rf,
                                                  x_plane or self.NONE), ints=(rows, C), floats=(1e-5,))
        else:
            self._ws_names['xe'] = self._ws_names['x%d' % self.Lyr]
            xe = x_in
        self.r_xe = xe

        # ---- decoders -------------------------------------------------------------------------------
        self._build_decoder_forward(xe)

    def _build_decoder_forward(self, xe):
        C, K, S = self.C, self.K, self.S
        S2 = S * S
        M, ldK = self.M, self.ldK
        ms = self.max_shape
        Wfc, bfc = 'decoder.fc.0.weight', 'decoder.fc.0.bias'
        W0, b0 = 'decoder.conv.0.weight', 'decoder.conv.0.bias'
        W2, b2 = 'decoder.conv.2.weight', 'decoder.conv.2.bias'
        Wc, bc = 'decoder.class_layer_predictor.1.weight', 'decoder.class_layer_predictor.1.bias'
        def decoder_1d(side):
            # 1-D decoder (nn.py:286-295)
            if self.n1 <= 0:
                return
            W1, b1 = 'decoder_1d.fc.0.weight', 'decoder_1d.fc.0.bias'
            W2d, b2d = 'decoder_1d.fc.2.weight', 'decoder_1d.fc.2.bias'
            Wb, bb = 'bias_class.1.weight', 'bias_class.1.bias'
            mc = max(self.max_shape[:2])
            self.mc = mc
            r_src1 = self.idx(self.oned_src)
            self.r_src1 = r_src1
            h1d = self.wsf('h1d', self.n1 * 2 * C)
            w1d = self.wsf('w1d', self.n1 * 2 * mc)
            p0 = self.gemm(xe, self.pref(W1), h1d, self.n1, 2 * C, C, C, C, 2 * C, bias=self.pref(b1),
                           act=L.ACT_RELU, a_gather=r_src1)
            self.gemm_op(p0, side=side)
            p0 = len(self._probs)
            if self.n1_plain:
                self.gemm(h1d, self.pref(W2d), w1d, self.n1_plain, 2 * mc, 2 * C, 2 * C, 2 * C, 2 * mc,
                          bias=self.pref(b2d))
            if self.n1_clsb:
                self.gemm(self.wref('h1d', self.n1_plain * 2 * C), self.pref(W2d),
                          self.wref('w1d', self.n1_plain * 2 * mc), self.n1_clsb, 2 * mc, 2 * C, 2 * C, 2 * C, 2 * mc,
                          bias=self.pref(b2d), act=L.ACT_RELU)
            self.gemm_op(p0, side=side)
            if self.n1_clsb:
                cbout = self.wsf('cbout', self.n1_clsb * ldK)
                p0 = self.gemm(self.wref('w1d', self.n1_plain * 2 * mc + mc), self.pref(Wb), cbout, self.n1_clsb, K,
                               mc, 2 * mc, mc, ldK, bias=self.pref(bb))
                self.gemm_op(p0, side=side)

        # The 1-D decoder (bn / bias / norm nodes) depends on the node embeddings only: with two streams it runs on the side
        # stream beside the W2 GEMMs instead of behind them (~35 us of small launches off the forward chain)
        # (measured: in a forward-only run the cross-stream hand-off costs 0.13 ms -- more than the 35 us it hides -- so the
        # forward keeps the 1-D decoder on the chain unless GHN3_EARLY_1D_FWD=1; the backward, whose side stream is busy
        # anyway, moves it: see _build_backward)
        early_1d = bool(self.SIDE) and M > 0 and os.environ.get('GHN3_EARLY_1D_FWD', '0') == '1'
        if self.SIDE:
            # An overlapped optimizer step (FusedAdamW.step(overlap=True)) may still be updating the decoder parameters on
            # the side stream: the Graphormer above needed none of them, everything below does (no-op when nothing is pending)
            self.op(L.OP_JOIN)
        if early_1d:
            decoder_1d(True)
        if M > 0:
            t = self.wsf('t', M * 4 * C)
            u = self.wsf('u', M * 8 * C)
            # D1: fc, one problem per used position of the 16x16 grid (row subset of Wfc)
            self.d1 = []
            p0 = len(self._probs)
            for p in np.unique(self.row_pos):
                rws = np.nonzero(self.row_pos == p)[0].astype(np.int32)
                r_rows = self.idx(rws)
                r_src = self.idx(self.row_src[rws])
                self.d1.append((int(p), len(rws), r_rows, r_src))
                self.gemm(xe, self.pref(Wfc, int(p) * C), t, len(rws), 4 * C, C, C, S2 * C, 4 * C,
                          bias=self.pref(bfc, int(p)), bias_stride=S2, act=L.ACT_RELU, a_gather=r_src,
                          c_gather=r_rows)
            self.gemm_op(p0, tag=self.TAG_D1_FWD, ctype=self.d12_fwd_ctype)
            # D2.  GHN3_D2_FIX (default in the 16-bit modes): the 16-bit product + GHN3_OP_RELU_FIX -- the ~0.4 % of the
            # elements the 16-bit GEMM leaves within 5e-3 rms of zero are recomputed in fp32 before the ReLU, so the ReLU
            # mask of the backward is as exact as with an fp32 GEMM (0.07 + 0.04 ms instead of 0.25 ms for the fp32 GEMM)
            d2_fix = self.decoder_ctype in (L.CT_F16, L.CT_BF16) and os.environ.get('GHN3_D2_FIX', '1') != '0' and \
                self.d12_fwd_ctype == L.CT_F32
            p0 = self.gemm(t, self.pref(W0), u, M, 8 * C, 4 * C, 4 * C, 4 * C, 8 * C, bias=self.pref(b0),
                           act=L.ACT_NONE if d2_fix else L.ACT_RELU)
            self.gemm_op(p0, tag=self.TAG_D2_FWD, ctype=self.decoder_ctype if d2_fix else self.d12_fwd_ctype)
            if d2_fix:
                self.op(L.OP_RELU_FIX, refs=(u, t, self.pref(W0), self.pref(b0)),
                        ints=(M, 8 * C, 8 * C, 4 * C, 0, 0),
                        floats=(float(os.environ.get('GHN3_RELU_FIX_TAU', '5e-3')),))
            # D3: only the W2 rows (o' < o, i' < i) each group consumes; all groups in one launch
            p0 = len(self._probs)
            tiles_floats = 0
            for gg in self.gemm_groups:
                gg['tile_off'] = tiles_floats
                for g in gg['members']:
                    g['tile_off'] = tiles_floats + (g['row0'] - gg['row0']) * gg['ld']
                tiles_floats += round_up(gg['rows'] * gg['ld'], 64)
            # kernels larger than the decoder grid: the resized (kh*kw, o*i) matrix lives behind the grid-sized ones
            # in the same buffer (and, in the backward, in d_tiles), so that tile descriptors address it like any tile
            for g in self.conv_groups:
                if 'resize' in g:
                    g['rs_off'] = tiles_floats
                    tiles_floats += round_up(g['resize'][0] * g['resize'][1] * g['ld'], 64)
            self.tiles_floats = tiles_floats
            tiles = self.wsf('tiles', tiles_floats)
            use16 = any(g['op16'] for g in self.gemm_groups)
            if use16:
                fct = self.decoder_ctype
                self.op(L.OP_JOIN)                       # the W2 copies were cast on the side stream (_cast_w2)
                self.uh = self.ws16('uh', M * 8 * C)
                self.cast16((self.xbuf(self.X_WS), 0),
                            [dict(src_off=u[1] // 4, rows=M, cols=8 * C, ld_src=8 * C, straight=(self.uh, 8 * C, fct))])
            fl = 0.0
            for g in self.gemm_groups:
                fl += sum(2.0 * sb['rows'] * sb['cols'] * 8 * C for sb in g['subs'])   # algorithmic: own extents only
                if g['op16'] and g['p8'] and self.P8_BALANCED:
                    # Round 6: one DENSE problem per column range of the family (cut at its members' output widths), each with
                    # EQUAL row tiles (balanced_tiles): the row tiles of a W2 panel run at the same pace and share it in L2
                    for (o_lo, o_hi, alive) in self.column_ranges(g):
                        n_lo, ncols = o_lo * g['i_ld'], (o_hi - o_lo) * g['i_ld']
                        mt = self.range_tiles(alive, alive, lambda r_: ncols)
                        self.gemm(self.href(self.uh + g['row0'] * 8 * C), self.sref(self.w2h + o_lo * ms[1] * 8 * C),
                                  self.wref('tiles', g['tile_off'] + n_lo),
                                  alive, ncols, 8 * C, 8 * C, 8 * C, g['ld'], b_qs=(g['i_ld'], ms[1]),
                                  bias=self.pref(b2, o_lo * ms[1]), bias_q=g['i_ld'], bias_s=ms[1], op16=True,
                                  lim=None, lim_kind=1, mtiles=(self.idx(mt), len(mt)))
                    continue
                if g['op16']:
                    # one problem per family: all its rows share every W2 tile; ragged column extents via `lim`
                    self.gemm(self.href(self.uh + g['row0'] * 8 * C), self.sref(self.w2h),
                              self.wref('tiles', g['tile_off']),
                              g['rows'], g['cols'], 8 * C, 8 * C, 8 * C, g['ld'], b_qs=(g['i_ld'], ms[1]),
                              bias=self.pref(b2), bias_q=g['i_ld'], bias_s=ms[1], op16=True,
                              lim=self.idx(g['lim128']) if g['ragged'] else None, lim_kind=1,
                              mtiles=(self.idx(g['mtiles']), len(g['mtiles'])) if g['p8'] else None)
                    continue
                for (r0, nr) in self._row_parts(g['rows']):
                    self.gemm((u[0], u[1] + 4 * (g['row0'] + r0) * 8 * C), self.pref(W2),
                              self.wref('tiles', g['tile_off'] + r0 * g['ld']),
                              nr, g['cols'], 8 * C, 8 * C, 8 * C, g['ld'], b_qs=(g['i_ld'], ms[1]),
                              bias=self.pref(b2), bias_q=g['i_ld'], bias_s=ms[1],
                              act=L.ACT_RELU if g['kind'] == 'cls' else L.ACT_NONE)
            self.gemm_op(p0, tag=self.TAG_D3_FWD, flops=fl, tile=28 if any(g.get('p8') for g in self.gemm_groups) else 0)
            # classifier rows of the 16-bit pipeline: ReLU afterwards, elements the f16 product left within 5e-3 rms of
            # zero recomputed in fp32 (their sign is the ReLU mask of the backward)
            for gg in self.gemm_groups:
                if not gg['op16']:
                    continue
                for g in gg['members']:
                    if g['kind'] == 'cls':
                        self.op(L.OP_RELU_FIX,
                                refs=(self.wref('tiles', g['tile_off']), (u[0], u[1] + 4 * g['row0'] * 8 * C),
                                      self.pref(W2), self.pref(b2)),
                                ints=(g['rows'], g['cols'], g['ld'], 8 * C, g['i_ld'], ms[1]),
                                floats=(float(os.environ.get('GHN3_RELU_FIX_TAU', '5e-3')),))
            p0 = len(self._probs)
            for g in self.conv_groups:
                if 'resize' in g:
                    kh, kw = g['resize']
                    g['r_wb'] = self.idx(self._resize_matrix(S, S, kh, kw))
                    # resized[p_out][c] = sum_p Wb[p_out][p] tiles[p][c]     (F.interpolate(..., mode='bilinear'))
                    self.gemm(g['r_wb'], self.wref('tiles', g['tile_off']), self.wref('tiles', g['rs_off']),
                              kh * kw, g['cols'], S2, S2, g['ld'], g['ld'], b_mode=L.MODE_COL)
            self.gemm_op(p0)
            # classifier head (nn.py:755-758): out[i'][k] = sum_o' relu(tile[o'][i']) Wcls[k][o'] + bcls[k]
            n_cls_rows = sum(g['rows'] * g['i_ld'] for g in self.conv_groups if g['kind'] == 'cls')
            if n_cls_rows:
                clsout = self.wsf('clsout', n_cls_rows * ldK)
                p0 = len(self._probs)
                off = 0
                for g in self.conv_groups:
                    if g['kind'] != 'cls':
                        continue
                    g['cls_off'] = []
                    for n_idx in range(len(g['inds'])):
                        g['cls_off'].append(off)
                        self.gemm(self.wref('tiles', g['tile_off'] + n_idx * g['ld']), self.pref(Wc),
                                  self.wref('clsout', off), g['i'], K, ms[0], g['i_ld'], ms[0], ldK,
                                  a_mode=L.MODE_COL, bias=self.pref(bc))
                        off += g['i_ld'] * ldK
                self.gemm_op(p0)
        if not early_1d:
            decoder_1d(False)
        elif self.n1 > 0:
            self.op(L.OP_JOIN)                          # the side-stream 1-D decoder is complete before the tile kernels read it
        self._build_tile_descriptors()

    @staticmethod
    def _resize_matrix(hi, wi, ho, wo):
        """(ho*wo, hi*wi) fp32 matrix of torch's bilinear resize (align_corners=False), the branch of nn.py:751-753:
        source coordinate = (dst + 0.5) * (in / out) - 0.5 clamped at 0, neighbours i0 and min(i0 + 1, in - 1)."""
        def axis(n_in, n_out):
            m = np.zeros((n_out, n_in), dtype=np.float32)
            scale = np.float32(n_in) / np.float32(n_out)
            for d in range(n_out):
                src = max(np.float32(scale * np.float32(d + 0.5) - np.float32(0.5)), np.float32(0))
                i0 = min(int(src), n_in - 1)
                i1 = i0 + (1 if i0 < n_in - 1 else 0)
                l1 = np.float32(src - np.float32(i0))
                m[d, i0] += np.float32(1) - l1
                m[d, i1] += l1
            return m
        wy, wx = axis(hi, ho), axis(wi, wo)
        return np.ascontiguousarray(np.einsum('ab,cd->acbd', wy, wx).reshape(ho * wo, hi * wi).astype(np.float32))

    # ------------------------------------------------------------------ tile / normalise descriptors
    def _build_tile_descriptors(self):
        C, K, ldK = self.C, self.K, self.ldK
        mc = max(self.max_shape[:2])
        descs, predicted = [], []
        out_off = 0
        tok_off = 0
        group_of = {}
        for g in self.conv_groups:
            for n_idx, ind in enumerate(g['inds']):
                group_of[ind] = (g, n_idx)
        clsb_row = {}
        for ind, r in self.oned_index.items():
            if r >= self.n1_plain:
                clsb_row[ind] = r - self.n1_plain

        desc_seg = []                                   # descriptor -> predicted tensor (the fused norm loss)

        def add(dst_off, src_buf, src_off, T, E, Sd, R, mode, scale):
            for k in range(4):
                assert 1 <= E[k] <= T[k] and E[k] <= R[k], (T, E, R)
            descs.append((dst_off, src_off, tuple(Sd), tuple(T), tuple(E), tuple(R), src_buf, mode, scale))
            desc_seg.append(len(predicted) - 1)

        for key, inds in self.param_groups.items():
            if len(inds) == 0:
                continue
            is_cls = (len(key) == 2 and key[1] != 0)
            if is_cls and not self.predict_class_layers:
                continue
            for ind in inds:
                matched, _, w_ind = self.params_map[ind]
                if w_ind is None:
                    continue
                m, sz, is_w = matched['module'], tuple(matched['sz']), matched['is_w']
                for it in range(2 if (len(sz) == 1 and is_w) else 1):
                    w_flag = bool(is_w) and not it
                    attr = bk.target_attr(m, w_flag)
                    tgt = getattr(m, attr, None)
                    if it == 1 and tgt is None:
                        continue                     # norm layer without bias
                    tile_t = sz                       # _tile_params target (nn.py:325); norm-layer bias shares it
                    # nn.py:526-528: a 2-D tile assigned to a 4-D (O,I,1,1) parameter is unsqueezed
                    t_assign = bk._sz(tgt) if tgt is not None else tile_t
                    mode, scale = bk.norm_rule(tile_t, w_flag) if self.weight_norm else (0, 1.0)
                    numel = int(math.prod(tile_t))
                    dst = out_off
                    out_off = round_up(out_off + numel, 16)
                    predicted.append(dict(node=ind, module=m, attr=attr, shape=tuple(t_assign),
                                          tile_shape=tuple(tile_t), offset=dst, numel=numel, is_w=w_flag))
                    t = tile_t
                    if len(key) == 4:
                        g, n_idx = group_of[ind]
                        base = g['tile_off'] + n_idx * g['hw'] * g['ld']
                        kh, kw, ld = g['kh'], g['kw'], g['ld']
                        if 'resize' in g:
                            base, (kh, kw) = g['rs_off'], g['resize']
                        if len(t) == 4:
                            if (t[2], t[3]) != (kh, kw):
                                raise NotImplementedError('target kernel %s vs group key %s' % (str(t), str(key)))
                            add(dst, 0, base, (t[0], t[1], kh, kw), (min(t[0], g['o']), min(t[1], g['i']), kh, kw),
                                (g['i_ld'], 1, kw * ld, ld), (g['o'], g['i_ld'], kh, kw), mode, scale)
                        elif len(t) == 2:
                            cy, cx = kh // 2, kw // 2
                            add(dst, 0, base + (cy * kw + cx) * ld, (t[0], t[1], 1, 1),
                                (min(t[0], g['o']), min(t[1], g['i']), 1, 1), (g['i_ld'], 1, 0, 0),
                                (g['o'], g['i_ld'], 1, 1), mode, scale)
                            if kh * kw != 1:
                                raise NotImplementedError('2-D target from a %dx%d tile' % (kh, kw))
                        elif len(t) == 3:
                            # positional encoding (nn.py:442-446): rows 1.. from the tile, row 0 random (Q3)
                            hw = kh * kw
                            L1, D = t[1], t[2]
                            if t[0] != 1 or L1 - 1 > hw or g['o'] != 1:
                                raise NotImplementedError('3-D target %s from key %s' % (str(t), str(key)))
                            i_e = min(D, g['i'])
                            add(dst + D, 0, base, (1, L1 - 1, D, 1), (1, L1 - 1, i_e, 1), (0, ld, 1, 0),
                                (1, hw, g['i_ld'], 1), mode, scale)
                            add(dst, 4, tok_off, (1, 1, D, 1), (1, 1, i_e, 1), (0, 0, 1, 0), (1, 1, i_e, 1), mode,
                                scale)
                            predicted[-1]['tok'] = (tok_off, i_e)
                            tok_off += round_up(i_e, 4)
                        else:
                            raise NotImplementedError('1-D target from a 4-D tile')
                    elif len(key) == 2 and key[1] > 0:
                        g, n_idx = group_of[ind]
                        base = g['cls_off'][n_idx]
                        if len(t) == 4 and (t[2], t[3]) != (1, 1):
                            raise NotImplementedError('classifier target %s' % str(t))
                        add(dst, 1, base, (t[0], t[1], 1, 1), (min(t[0], K), min(t[1], g['i']), 1, 1),
                            (1, ldK, 0, 0), (K, g['i'], 1, 1), mode, scale)
                    elif len(key) == 2 and key[1] < 0:
                        r = clsb_row[ind]
                        assert len(t) == 1
                        add(dst, 3, r * ldK, (1, t[0], 1, 1), (1, min(t[0], K), 1, 1), (0, 1, 0, 0), (1, K, 1, 1),
                            mode, scale)
                    else:
                        r = self.oned_index[ind]
                        if len(t) == 1:
                            half = (1 - int(bool(is_w)) + it)
                            add(dst, 2, r * 2 * mc + half * mc, (1, t[0], 1, 1), (1, min(t[0], mc), 1, 1),
                                (0, 1, 0, 0), (1, mc, 1, 1), mode, scale)
                        elif len(t) == 3:
                            if t[1] > 1 and t[2] > 1:
                                raise NotImplementedError('3-D target %s from the 1-D decoder' % str(t))
                            add(dst, 2, r * 2 * mc, (t[0], t[1], t[2], 1), (min(t[0], 2 * mc), 1, 1, 1),
                                (1, 0, 0, 0), (2 * mc, 1, 1, 1), mode, scale)
                        else:
                            raise NotImplementedError('target %s from the 1-D decoder' % str(t))
        self.predicted = predicted
        self.out_numel = max(out_off, 16)
        self.tok_floats = max(tok_off, 4)
        self.n_desc = nd = len(descs)
        CH = 2048
        self.tile_lds = [0, 0]
        # descriptor table and work-block tables, built column-wise (one numpy call per field, not per descriptor)
        desc_arr = np.zeros(max(nd, 1), dtype=L.TILE_DT)
        fb = bb = np.zeros((0, 2), dtype=np.int64)
        if nd:
            cols = list(zip(*descs))
            desc_arr['dst_off'], desc_arr['src_off'] = cols[0], cols[1]
            S = np.asarray(cols[2], dtype=np.int64).reshape(nd, 4)
            T = np.asarray(cols[3], dtype=np.int64).reshape(nd, 4)
            E = np.asarray(cols[4], dtype=np.int64).reshape(nd, 4)
            R = np.asarray(cols[5], dtype=np.int64).reshape(nd, 4)
            desc_arr['S'], desc_arr['T'], desc_arr['E'], desc_arr['R'] = S, T, E, R
            desc_arr['src_buf'], desc_arr['mode'], desc_arr['scale'] = cols[6], cols[7], cols[8]
            mode = np.asarray(cols[7], dtype=np.int64)
            hw = T[:, 2] * T[:, 3]
            # convolution kernels with kh * kw > 1: LDS-transposed row blocks (see tile_fwd_kernel), one per o and
            # chunk of `ich` input channels (<= 16 KB of LDS, so that many blocks share a CU)
            row_ok = (mode == 0) & (hw > 1) & (S[:, 1] == 1) & (E[:, 2] == T[:, 2]) & (T[:, 2] == R[:, 2]) & \
                (E[:, 3] == T[:, 3]) & (T[:, 3] == R[:, 3]) & (hw <= 1024)
            # (a multiple of 16 channels unless the tensor has fewer: the row blocks move 16 bytes per lane when the chunk is
            # a multiple of 4 floats)
            ich = np.maximum(16, np.minimum(np.maximum(T[:, 1], R[:, 1]), (4096 // np.maximum(hw, 1)) // 16 * 16))
            desc_arr['_pad'] = np.where(row_ok, ich, 0)
            if row_ok.any():
                lds = int((4 * ich * hw)[row_ok].max())
                self.tile_lds = [lds, lds]
            ids = np.arange(nd, dtype=np.int64)

            def tables(n0, n1, numel):
                """(descriptor id or ~id, start / packed (o, first i)) rows: element blocks of CH elements, row blocks
                of one o and `ich` input channels; descriptors in order, blocks of a descriptor in order."""
                n_i = (n1 + ich - 1) // ich                                   # i chunks per o (row blocks)
                cnt = np.where(row_ok, n0 * n_i, (numel + CH - 1) // CH)
                total = int(cnt.sum())
                first = np.cumsum(cnt) - cnt
                d_of = np.repeat(ids, cnt)
                j = np.arange(total, dtype=np.int64) - np.repeat(first, cnt)     # block index inside its descriptor
                rb = row_ok[d_of]
                ni_b = np.maximum(n_i[d_of], 1)
                word = np.where(rb, (j // ni_b) | (((j % ni_b) * ich[d_of]) << 24), j * CH)
                return np.stack([np.where(rb, ~d_of, d_of), word], axis=1), first

            fb, fwd_first = tables(T[:, 0], T[:, 1], T.prod(axis=1))
            bb, _ = tables(R[:, 0], R[:, 1], R.prod(axis=1))
        dseg = np.asarray(desc_seg if nd else [0], dtype=np.int32)
        raw = np.concatenate([desc_arr.view(np.uint8).reshape(-1), fb.view(np.uint8).reshape(-1),
                              bb.view(np.uint8).reshape(-1), dseg.view(np.uint8).reshape(-1)])
        self.r_desc = self.idx(raw)
        self.fwd_blk = (len(fb), desc_arr.nbytes)
        self.bwd_blk = (len(bb), desc_arr.nbytes + fb.nbytes)
        self.desc_seg_off = desc_arr.nbytes + fb.nbytes + bb.nbytes
        # first forward work block of every predicted tensor (+ the total): the blocks of a tensor are consecutive
        seg_blk = np.zeros(len(predicted) + 1, dtype=np.int32)
        if nd:
            first_desc = np.full(len(predicted), nd, dtype=np.int64)
            np.minimum.at(first_desc, dseg, np.arange(nd))
            seg_blk[:-1] = fwd_first[first_desc]
        seg_blk[-1] = len(fb)
        self.r_seg_blk = self.idx(seg_blk)
        seg = np.asarray([[p['offset'], p['offset'] + p['numel']] for p in predicted], dtype=np.int64).reshape(-1, 2)
        self.r_seg = self.idx(seg if len(seg) else np.zeros((1, 2), dtype=np.int64))
        self.n_seg = len(predicted)
        # first segment that ends beyond the start of every 8192-float chunk of the flat buffer (param-norm passes)
        ends = seg[:, 1] if len(seg) else np.zeros(1, dtype=np.int64)
        starts = np.arange(0, max(self.out_numel, 1), 8192, dtype=np.int64)
        self.r_seg_first = self.idx(np.searchsorted(ends, starts, side='right').astype(np.int32))
        srcs = self._tile_sources(False)
        if self.n_desc:
            # (training: every work block leaves the sum of squares of what it wrote -- the predicted-parameter norm loss
            # then needs no pass over the output, see norm_fin_ops)
            parts = self.wsf('sq_parts', self.fwd_blk[0]) if self.training else self.NONE
            # (+ per-block ingredients of the tile backward's a-priori output bound, see _build_backward: direct 16-bit tiles)
            bparts = self.wsf('b_parts', self.fwd_blk[0]) if self.training else self.NONE
            self._tile_desc_arr = desc_arr
            self.op(L.OP_TILE_FWD, refs=[(self.xbuf(self.X_OUT), 0)] + srcs + [self.r_desc, parts, bparts],
                    ints=(self.n_desc, self.fwd_blk[0], self.fwd_blk[1], self.tile_lds[0]),
                    flags=L.OPFLAG_TIMED | (self.TAG_TILE_FWD << 16))

    def _tile_sources(self, grad):
        pre = 'd_' if grad else ''
        out = []
        for name in ('tiles', 'clsout', 'w1d', 'cbout'):
            nm = pre + name
            out.append(self.wref(nm) if nm in self._ws_names else self.NONE)
        if grad:
            out.append(self.wsf('d_tok', self.tok_floats))
        else:
            out.append((self.xbuf(self.X_TOK), 0))
        out.append(self.NONE)
        return out

    # ------------------------------------------------------------------ loss used by bench / tests
    def norm_ops(self, upstream=1.0):
        """(forward ops, backward ops) of sum_t ||p_t||_F over all predicted tensors (trainer.py:288-294)."""
        scal = self.xbuf(self.X_SCAL)
        f_ops, b_ops = [], []
        saved = self._ops
        self._ops = []
        self.op(L.OP_MEMSET0, refs=((scal, 0),), ints=(4,))
        self.op(L.OP_PARAM_NORM_FWD, refs=((scal, 0), (self.xbuf(self.X_OUT), 0), self.r_seg, (scal, 256),
                                           self.r_seg_first, (scal, 256 + round_up(4 * max(self.n_seg, 1), 64))),
                ints=(self.n_seg, self.out_numel))
        f_ops = self._finish_ops()
        self.op(L.OP_PARAM_NORM_BWD, refs=((self.xbuf(self.X_DOUT), 0), (self.xbuf(self.X_OUT), 0), self.r_seg,
                                           (scal, 256), self.r_seg_first), ints=(self.n_seg, self.out_numel),
                floats=(upstream,))
        b_ops = self._finish_ops()
        self._ops = saved
        return f_ops, b_ops

    def norm_fin_ops(self):
        """Ops of the FUSED sum_t ||p_t||_F (trainer.py:288-294): the per-tensor norms from the block sums the tile forward
        left in the workspace (no pass over the flat output); loss -> X_SCAL[0], norms -> X_SCAL[256 ...].  The gradient of
        the term is formed inside GHN3_OP_TILE_BWD (GHN3._run_backward(norm_g=...))."""
        scal = self.xbuf(self.X_SCAL)
        saved, self._ops = self._ops, []
        if self.n_desc and 'sq_parts' in self._ws_names:
            # (r4..r6: the bound of the tile gradient for GHN3_OP_TILE_BWD's direct 16-bit output, left in front of the norms)
            self.op(L.OP_PARAM_NORM_FIN, refs=((scal, 0), (scal, 256), self.wref('sq_parts'), self.r_seg_blk,
                                               self.wref('b_parts'), (scal, 256 + round_up(4 * max(self.n_seg, 1), 64)),
                                               (scal, 252)),
                    ints=(self.n_seg,))
        else:
            self.op(L.OP_MEMSET0, refs=((scal, 0),), ints=(4,))
        ops = self._finish_ops()
        self._ops = saved
        return ops

    def _layer_side_ops(self, layers, rows):
        """Weight gradients (one grouped GEMM launch) and LayerNorm parameter gradients of the given Graphormer layers
        (autograd of graphormer.py:208-248), on the side stream."""
        C = self.C
        p0 = len(self._probs)
        for d in layers:
            pre = d['pre']
            W3, W1f, Wo, Wq = pre + 'ff.net.3.weight', pre + 'ff.net.0.weight', pre + 'attn.to_out.0.weight', \
                pre + 'attn.to_qkv.weight'
            self.gemm(d['g_cur'], d['f'], self.gref(W3), C, 4 * C, rows, C, 4 * C, 4 * C, a_mode=L.MODE_COL,
                      b_mode=L.MODE_COL, accum=True, dbias=self.gref(pre + 'ff.net.3.bias'))
            self.gemm(d['dz'], d['h2'], self.gref(W1f), 4 * C, C, rows, 4 * C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                      accum=True, dbias=self.gref(pre + 'ff.net.0.bias'))
            self.gemm(d['g_mid'], d['o'], self.gref(Wo), C, C, rows, C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                      accum=True, dbias=self.gref(pre + 'attn.to_out.0.bias'))
            self.gemm(d['dqkv'], d['h1'], self.gref(Wq), 3 * C, C, rows, 3 * C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                      accum=True)
        # the split-bf16 weight-gradient kernel on 64 x 64 tiles (48); exact fp32 tiles (64) outside the 16-bit modes; the
        # small-problem kernel for narrow models.  (49 = the same kernel on 128 x 128 tiles, half the operand bytes per flop:
        # measured slower for the grouped launch, 6.61-6.63 against 6.54-6.56 ms per step -- fewer, fatter workgroups.)
        dflt = ('48' if self.x3 else '64') if C >= 256 else '0'
        self.gemm_op(p0, side=True, tile=int(os.environ.get('GHN3_LAYER_WGRAD_TILE', dflt)),
                     grid_cap=int(os.environ.get('GHN3_LAYER_WGRAD_CAP', '0')) if self.SIDE else 0)
        if len(layers) == 1:
            d = layers[0]
            pre = d['pre']
            self.op(L.OP_LN_PARAM_GRAD, refs=(self.gref(pre + 'ln2.weight'), self.gref(pre + 'ln2.bias'), d['dhA'], d['xmid'],
                                              d['m2'], d['r2']), ints=(rows, C, 1), flags=self.SIDE)
            self.op(L.OP_LN_PARAM_GRAD, refs=(self.gref(pre + 'ln1.weight'), self.gref(pre + 'ln1.bias'), d['dhB'], d['x_in'],
                                              d['m1'], d['r1']), ints=(rows, C, 1), flags=self.SIDE)
            return
        # batched: offsets (floats) from the first layer parameter's gradient / from the workspace base
        base = 'gnn.0.' + LAYER_PARAM_NAMES[0]
        ws = self.xbuf(self.X_WS)
        tab = []
        for d in layers:
            pre = d['pre']
            for (gn, bn, dy, x, mu, rs) in ((pre + 'ln2.weight', pre + 'ln2.bias', d['dhA'], d['xmid'], d['m2'], d['r2']),
                                           (pre + 'ln1.weight', pre + 'ln1.bias', d['dhB'], d['x_in'], d['m1'], d['r1'])):
                assert all(r_[0] == ws and r_[1] % 4 == 0 for r_ in (dy, x, mu, rs))
                tab.append((self.param_gap(base, gn), self.param_gap(base, bn), dy[1] // 4, x[1] // 4, mu[1] // 4, rs[1] // 4))
        self.op(L.OP_LN_PARAM_GRAD_BATCH, refs=(self.gref(base), (ws, 0), self.idx(np.asarray(tab, dtype=np.int64))),
                ints=(len(tab), rows, C), flags=self.SIDE)

    # ------------------------------------------------------------------ backward
    def _colsum(self, out, X, M, N, ld, q=0, s=0, stride=1, gather=None):
        self.op(L.OP_COLSUM, refs=(out, X, gather if gather is not None else self.NONE),
                ints=(M, N, ld, q, s, stride, 1))

    def set_tile_route(self, direct):
        """Switches the compiled backward between the fp32 route of the tile gradient (tile backward -> fp32 d_tiles ->
        cast passes with the measured maximum) and the direct 16-bit route (see _build_backward; only valid when the fused
        norm loss is the ONLY upstream term of the step).  Returns the indices of the ops it touched (all lie in front of
        the W2 weight gradient, i.e. in the first part of bwd_parts at the same index)."""
        direct = bool(direct and getattr(self, 'tile_bwd_h16', 0))
        if self.tile_bwd_op is None or not hasattr(self, 'd16_kinds'):
            return []
        self.bwd_ops[self.tile_bwd_op]['i'][5] = self.tile_bwd_h16 if direct else 0
        touched = []
        for idx, on in ((self.d16_on_idx, direct), (self.d16_off_idx, not direct)):
            for k in idx:
                self.bwd_ops[k]['kind'] = self.d16_kinds[k] if on else L.OP_NOP
                touched.append(k)
        return touched

    @staticmethod
    def _wgrad_order_env():
        """GHN3_WGRAD_ORDER: where the side-stream W2 weight gradient is issued in the single-process backward -- 'first'
        (default; behind the tile backward and its operand copies, beside the W2 dgrad and the rest of the decoder backward),
        'late' (rounds 3-5a: behind the decoder backward, beside the Graphormer backward only), 'early' (behind the dgrad)."""
        return os.environ.get('GHN3_WGRAD_ORDER', 'first' if os.environ.get('GHN3_WGRAD_LATE', '1') != '0' else 'early')

    def _wgrad_schedule(self, flops, rows):
        """Workgroups of the side-stream W2 weight gradient (a multiple of 8: a worker's tiles keep their XCD), or 0 = on the
        chain's stream in front of the Graphormer backward.  See the call site for the model and its measurements."""
        env_main, env_cap = os.environ.get('GHN3_WGRAD_MAIN'), os.environ.get('GHN3_WGRAD_CAP')
        if not self.SIDE or env_main == '1':
            return 0
        if env_cap is not None:
            return max(8, int(env_cap) // 8 * 8)
        # the persistent kernel alone: its k loop at ~1150 TF + the stores of dW2 at ~4.3 TB/s (0.97 / 1.52 / 2.63 ms measured
        # for 1 / 2 / 4 graphs of 256 nodes at ghn3xlm16; profiles/r05b_ab_wgrad_schedule_b1_b2_b4.txt)
        w_ms = flops / 1150e12 * 1e3 + 4.0 * self.max_shape[0] * self.max_shape[1] * 8 * self.C / 4.3e12 * 1e3
        ch_ms = self.Lyr * (0.030 + 0.014 * (rows / 256.0) * (self.C / 384.0) ** 2)      # the backward chain alone
        if env_main != '0' and w_ms < 0.75 * ch_ms:
            return 0
        if self._wgrad_order_env() == 'first':
            return 128                                   # half of every XCD (see the call site)
        c = int(round(256.0 * w_ms / (1.75 * ch_ms) / 8.0)) * 8
        if c >= 232:
            return 0 if env_main != '0' else 224
        return max(64, c)

    def _build_h16_table(self, g16, direct_member):
        """Per tile descriptor {h, rel0, ld32 | ld16 << 32} of GHN3_OP_TILE_BWD's direct 16-bit output: where the fp32
        region of the descriptor lies inside its family's tile matrix and where that matrix's 16-bit copy `dth` starts
        (relative to d_tiles, in 16-bit elements).  h = INT64_MIN: the descriptor keeps its fp32 output."""
        D = self._tile_desc_arr
        nd = self.n_desc
        tab = np.zeros((nd, 3), dtype=np.int64)
        tab[:, 0] = np.iinfo(np.int64).min
        members = []
        for g in g16:
            for m_ in g['members']:
                if direct_member(m_):
                    members.append((m_['tile_off'], m_['tile_off'] + m_['rows'] * g['ld'], g))
        members.sort(key=lambda t: t[0])
        if members and nd:
            lo = np.asarray([t[0] for t in members], dtype=np.int64)
            hi = np.asarray([t[1] for t in members], dtype=np.int64)
            src = D['src_off'].astype(np.int64)
            k = np.searchsorted(lo, src, side='right') - 1
            ok = (D['src_buf'] == 0) & (k >= 0)
            ok &= src < hi[np.maximum(k, 0)]
            reach = src + ((D['R'].astype(np.int64) - 1) * D['S'].astype(np.int64)).sum(axis=1)
            d_tiles_h = self._ws_names['d_tiles'] // 2
            for d_ in np.nonzero(ok)[0]:
                g = members[int(k[d_])][2]
                assert reach[d_] < hi[int(k[d_])], 'tile descriptor straddles a family member'
                assert D['mode'][d_] == 0
                rel0 = int(src[d_]) - g['tile_off']
                assert 0 <= rel0 and g['rows'] * g['ld'] < 2 ** 32
                tab[d_] = (g['dth'] - d_tiles_h, rel0, g['ld'] | (g['dth_ld'] << 32))
        r_tab = self.idx(tab)
        assert r_tab[0] == self.r_desc[0] and r_tab[1] > self.r_desc[1]
        self.tile_bwd_h16 = r_tab[1] - self.r_desc[1]        # ints[5] of the tile backward when the direct route is on

    def _build_backward(self):
        C, H, B, N, V, K = self.C, self.H, self.B, self.N, self.V, self.K
        rows = B * N
        ldT, ldK = self.ldT, self.ldK
        ms = self.max_shape
        S2 = self.S * self.S
        M, n1 = self.M, self.n1
        (r_types, r_shape, r_nn, r_noff, deg_in, deg_out, dist0, pair) = self.r_graph
        xe = self.r_xe
        # gradients of all GHN parameters start at zero; every parameter-gradient op accumulates
        # two memset ranges over the flat gradient buffer, patched at run time (offset, bytes): everything except
        # the parameters listed in grad_no_memset (tensors the backward fully overwrites)
        # (0.8 GB at ghn3xlm16.  Measured on the side stream beside the tile backward: no gain -- both are HBM-bound, the
        # tile backward slows down by what the memsets take)
        self.op(L.OP_MEMSET0, refs=((self.xbuf(self.X_GRADFLAT), 0),), ints=(-1,))
        self.op(L.OP_MEMSET0, refs=((self.xbuf(self.X_GRADFLAT), 0),), ints=(-1,))
        self.memset_grad_op = 0
        self.tile_bwd_op = None
        self.grad_no_memset = []
        self.bwd_cut_w2 = 0
        late_ops = []
        self.wgrad_op_range = None
        self.grad_sumsq = None      # set when the W2 weight gradient leaves its own sum of squares (see the band problems)

        d_rows = self.wsf('d_xrows', (M + n1) * C)
        # ---- tile backward -------------------------------------------------------------------------
        # (zero-fills of this program point are collected and emitted as ONE fill per run of neighbouring workspace regions -- the
        # regions are bump-allocated here, one behind the other, separated by their slack and alignment padding only: four
        # 5 us launches on the critical stream become one)
        zeros = []
        if M > 0:
            self.wsf('d_tiles', self.tiles_floats)
            # (fp32 tile gradient: its row padding is read -- and multiplied by zeros -- on the upstream-gradient route, so it
            # must hold finite values: zero-filled by GHN3._run_backward in front of a plan's first backward on that route)
            self.ws_zero_dout = [(self._ws_names['d_tiles'], 4 * (int(self.tiles_floats) + 64))]
            if 'clsout' in self._ws_names:
                n_cls = sum(g['rows'] * g['i_ld'] for g in self.conv_groups if g['kind'] == 'cls')
                self.wsf('d_clsout', n_cls * ldK)
                zeros.append((self.wref('d_clsout'), 4 * n_cls * ldK))
        if n1 > 0:
            mc = self.mc
            self.wsf('d_w1d', n1 * 2 * mc)
            zeros.append((self.wref('d_w1d'), 4 * n1 * 2 * mc))
            if self.n1_clsb:
                self.wsf('d_cbout', self.n1_clsb * ldK)
                zeros.append((self.wref('d_cbout'), 4 * self.n1_clsb * ldK))
        # running max |x| of d_tiles ([0], written by TILE_BWD) and d_u ([4], written by DACT): the power-of-two
        # scales of the f16 gradient copies
        self.r_amax = self.wsf('amax', 16)
        scaled = self.bwd_scaled and self.direct16 and M > 0
        amax_t = self.r_amax if scaled else None
        amax_u = (self.r_amax[0], self.r_amax[1] + 16) if scaled else None
        if scaled:
            zeros.append((self.r_amax, 64))
        run = None
        for (buf, off), n in sorted(zeros, key=lambda it: (it[0][0], it[0][1])) + [((None, 0), 0)]:
            if run is not None and buf == run[0] and 0 <= off - (run[1] + run[2]) <= 1024:
                run[2] = off + n - run[1]                   # (the gap: slack / padding of the previous region)
                continue
            if run is not None:
                self.op(L.OP_MEMSET0, refs=((run[0], run[1]),), ints=(run[2],))
            run = [buf, off, n]
        if self.n_desc:
            grads = self._tile_sources(True)
            if scaled:
                grads[5] = amax_t
            # r0 = upstream gradient of the predicted tensors and / or the fused norm loss (r6 = its device-side weight g,
            # r14 = per-tensor norms, r15 = the predicted values): GHN3._run_backward disables what a step does not use
            # (compiled with the norm term off: r6 / r14 / r15 absent; tile_bwd_refs = what GHN3._run_backward patches in)
            srcs = self._tile_sources(False)
            self.tile_bwd_op = len(self._ops)
            self.tile_bwd_refs = {0: (self.xbuf(self.X_DOUT), 0), 6: (self.xbuf(self.X_NORMG), 0),
                                  14: (self.xbuf(self.X_SCAL), 256), 15: (self.xbuf(self.X_OUT), 0)}
            self.op(L.OP_TILE_BWD, refs=[(self.xbuf(self.X_DOUT), 0)] + srcs + [self.r_desc] + grads +
                    [self.NONE, self.NONE],
                    ints=(self.n_desc, self.bwd_blk[0], self.bwd_blk[1], self.tile_lds[1], self.desc_seg_off, 0,
                          1 if self.decoder_bwd_ctype == L.CT_BF16 else 0),
                    flags=L.OPFLAG_TIMED | (self.TAG_TILE_BWD << 16))

        def decoder_1d_bwd(side):
            if n1 <= 0:
                return
            mc = self.mc
            W1, b1 = 'decoder_1d.fc.0.weight', 'decoder_1d.fc.0.bias'
            W2d, b2d = 'decoder_1d.fc.2.weight', 'decoder_1d.fc.2.bias'
            Wb, bb = 'bias_class.1.weight', 'bias_class.1.bias'
            h1d, w1d, d_w1d = self.wref('h1d'), self.wref('w1d'), self.wref('d_w1d')
            d_h1d = self.wsf('d_h1d', n1 * 2 * C)
            if self.n1_clsb:
                cb0 = self.n1_plain
                d_cb = self.wref('d_cbout')
                w_cb = self.wref('w1d', cb0 * 2 * mc + mc)
                p0 = self.gemm(d_cb, self.pref(Wb), self.wref('d_w1d', cb0 * 2 * mc + mc), self.n1_clsb, mc, K, ldK,
                               mc, 2 * mc, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, dact=L.DACT_RELU, aux_in=w_cb)
                self.gemm_op(p0, side=side)
                p0 = self.gemm(d_cb, w_cb, self.gref(Wb), K, mc, self.n1_clsb, ldK, 2 * mc, mc, a_mode=L.MODE_COL,
                               b_mode=L.MODE_COL, accum=True, dbias=self.gref(bb))
                self.gemm_op(p0, side=True)
            p0 = self.gemm(d_w1d, self.pref(W2d), d_h1d, n1, 2 * C, 2 * mc, 2 * mc, 2 * C, 2 * C, a_mode=L.MODE_ROW,
                           b_mode=L.MODE_COL, dact=L.DACT_RELU, aux_in=h1d)
            self.gemm_op(p0, side=side)
            p0 = self.gemm(d_w1d, h1d, self.gref(W2d), 2 * mc, 2 * C, n1, 2 * mc, 2 * C, 2 * C, a_mode=L.MODE_COL,
                           b_mode=L.MODE_COL, accum=True, dbias=self.gref(b2d))
            self.gemm_op(p0, side=True)
            p0 = self.gemm(d_h1d, self.pref(W1), (d_rows[0], d_rows[1] + 4 * M * C), n1, C, 2 * C, 2 * C, C, C,
                           a_mode=L.MODE_ROW, b_mode=L.MODE_COL)
            self.gemm_op(p0, side=side)
            p0 = self.gemm(d_h1d, xe, self.gref(W1), 2 * C, C, n1, 2 * C, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                           b_gather=self.r_src1, accum=True, dbias=self.gref(b1))
            self.gemm_op(p0, side=True)

        # The 1-D decoder backward needs the tile backward only: on the side stream beside the W2 dgrad instead of behind the
        # decoder.fc backward on the chain; joined in front of the node-row gather that adds its rows
        early_1d = bool(self.SIDE) and M > 0 and n1 > 0 and os.environ.get('GHN3_EARLY_1D', '1') != '0'
        if early_1d:
            decoder_1d_bwd(True)
            self.op(L.OP_JOIN, ints=(1, 0))              # mark: the 1-D backward (waited for in front of the node-row gather)
        Wfc, bfc = 'decoder.fc.0.weight', 'decoder.fc.0.bias'
        W0, b0 = 'decoder.conv.0.weight', 'decoder.conv.0.bias'
        W2, b2 = 'decoder.conv.2.weight', 'decoder.conv.2.bias'
        Wc, bc = 'decoder.class_layer_predictor.1.weight', 'decoder.class_layer_predictor.1.bias'
        if M > 0:
            # kernels larger than the decoder grid: d_tiles[p][c] = sum_pout Wb[pout][p] d_resized[pout][c]
            p0 = len(self._probs)
            for g in self.conv_groups:
                if 'resize' in g:
                    kh, kw = g['resize']
                    self.gemm(g['r_wb'], self.wref('d_tiles', g['rs_off']), self.wref('d_tiles', g['tile_off']),
                              S2, g['cols'], kh * kw, S2, g['ld'], g['ld'], a_mode=L.MODE_COL, b_mode=L.MODE_COL)
            self.gemm_op(p0)
            t, u = self.wref('t'), self.wref('u')
            d_t = self.wsf('d_t', M * 4 * C)
            d_u = self.wsf('d_u', M * 8 * C)
            # classifier head backward
            for g in self.conv_groups:
                if g['kind'] != 'cls':
                    continue
                # the head dgrad writes columns i' < i only; the pad columns (i <= i' < i_ld) must read as zero
                self.op(L.OP_MEMSET0, refs=(self.wref('d_tiles', g['tile_off']),), ints=(4 * g['rows'] * g['ld'],))
                p0 = len(self._probs)
                for n_idx in range(len(g['inds'])):
                    tile_n = self.wref('tiles', g['tile_off'] + n_idx * g['ld'])
                    dtile_n = self.wref('d_tiles', g['tile_off'] + n_idx * g['ld'])
                    dout_n = self.wref('d_clsout', g['cls_off'][n_idx])
                    # d relu(tile)[o'][i'] = sum_k dout[i'][k] Wcls[k][o'], masked by tile > 0
                    self.gemm(self.pref(Wc), dout_n, dtile_n, ms[0], g['i'], K, ms[0], ldK, g['i_ld'],
                              a_mode=L.MODE_COL, b_mode=L.MODE_ROW, dact=L.DACT_RELU, aux_in=tile_n)
                self.gemm_op(p0)
                for n_idx in range(len(g['inds'])):
                    tile_n = self.wref('tiles', g['tile_off'] + n_idx * g['ld'])
                    dout_n = self.wref('d_clsout', g['cls_off'][n_idx])
                    p0 = self.gemm(dout_n, tile_n, self.gref(Wc), K, ms[0], g['i'], ldK, g['i_ld'], ms[0],
                                   a_mode=L.MODE_COL, b_mode=L.MODE_ROW, accum=True, dbias=self.gref(bc))
                    self.gemm_op(p0, side=True)
            # D3 backward: d_u = (d_tiles . W2sub) * (u > 0)   -- all groups, one launch.  The reduction runs over
            # the o*i columns of a group (up to C^2 = 147456) while M x N is only rows x 8C, so the K range is
            # split into chunks (partial sums added atomically into the zeroed d_u) to fill the 256 CUs; the ReLU
            # mask is applied afterwards in place.
            bct = self.decoder_bwd_ctype
            g16 = [g for g in self.gemm_groups if g['op16']]
            # planes: the K splits of the 16-bit dgrad write separate partial planes (plane 0 = d_u) that the DACT pass
            # sums in a fixed order -- no atomics, no memset of d_u, deterministic gradients
            planes = bool(g16) and os.environ.get('GHN3_DGRAD_PLANES', '1') != '0'
            # exact-fp32 mode (no 16-bit groups): the K chunks write partial planes as well (round 3; they used to add
            # atomically into a zeroed d_u, the one non-deterministic reduction left on the path)
            planes32 = (not g16) and os.environ.get('GHN3_DGRAD_PLANES', '1') != '0'
            if not planes and not planes32:
                self.op(L.OP_MEMSET0, refs=(d_u,), ints=(4 * M * 8 * C,))
            if g16:
                # 16-bit copies of the backward operands: per group d_tiles straight (dgrad A operand); for the wgrad
                # transposed copies of d_tiles (A) and u (B) + the column sums of d_tiles = the conv.2 bias gradient.
                # wgrad bands.  dW2 row (o', i') sums over every decoder row r with o_r > o' and i_r > i' (nn.py:749-750,
                # 760: a row only consumes the W2 rows o' < o_r, i' < i_r).  The input-channel axis is cut at the distinct
                # group widths into BANDS [i_lo, i_hi): for i' in a band the contributing rows are exactly those with
                # i_r >= i_hi, whatever their family.  Per band these rows are concatenated along k in order of
                # decreasing o_r (transposed copy dthT_b [(o', i' - i_lo)][k], u copy uhT_b [8C][k]), so the W2 rows
                # with o' in [o_lo, o_hi) need exactly a k PREFIX: one GEMM problem per (band, o range), all in ONE
                # launch, and every dW2 row is written exactly once -- no accumulation into dW2 between families (was
                # 2 GB read + 2 GB written per step at ghn3xlm16) and no memset of it when the bands cover it.
                # Direct 16-bit tiles (round 4, GHN3_TILE_D16): with the fused norm loss alone the tile backward writes the
                # scaled 16-bit copy `dth` itself (scale from an a-priori bound, GHN3_OP_TILE_BWD) -- no fp32 d_tiles for
                # the rows it produces, no straight cast pass on the critical path, and the transposed wgrad operands are
                # re-laid out from the 16-bit copy.  Rows the tile backward does not produce (classifier-weight rows: head
                # dgrad; resized kernels: a GEMM over d_tiles) keep the fp32 route.  Both op sets are compiled; GHN3.
                # _run_backward switches per step (an upstream gradient `dout` has no bound: fp32 + measured maximum).
                d16 = bool(scaled and self.tile_bwd_op is not None and os.environ.get('GHN3_TILE_D16', '1') != '0')
                direct_member = lambda m_: m_['kind'] == 'conv' and 'resize' not in m_
                items, side_items, items_d16, side_d16, u_items = [], [], [], [], []
                rowsets, bias_sets, n_parts = [], [], 0
                for g in g16:
                    g['dth_ld'] = round_up(g['cols'], 64) + 64        # (+64: rows never a power of two apart)
                    g['dth'] = self.ws16('dth%d' % g['row0'], g['rows'] * g['dth_ld'])
                    # (rows of a family with a smaller extent keep zeros beyond it: d_tiles is never written there)
                    items.append(dict(src_off=self._ws_names['d_tiles'] // 4 + g['tile_off'], rows=g['rows'],
                                      cols=g['cols'], ld_src=g['ld'], straight=(g['dth'], g['dth_ld'], bct),
                                      scaled=scaled))
                    for m_ in g['members']:
                        if not direct_member(m_):
                            r_ = m_['row0'] - g['row0']
                            items_d16.append(dict(src_off=self._ws_names['d_tiles'] // 4 + g['tile_off'] + r_ * g['ld'],
                                                  rows=m_['rows'], cols=g['cols'], ld_src=g['ld'],
                                                  straight=(g['dth'] + r_ * g['dth_ld'], g['dth_ld'], bct), scaled=scaled))
                    for sb in g['subs']:
                        rowsets.append(dict(o=sb['o'], i=g['i_ld'], row0=sb['row0'], rows=sb['rows'], g=g))
                self.wgrad_bands = []
                i_lo = 0
                for i_hi in sorted({rs['i'] for rs in rowsets}):
                    members = sorted((rs for rs in rowsets if rs['i'] >= i_hi), key=lambda rs: -rs['o'])
                    bw = i_hi - i_lo
                    k_off, mem = 0, []
                    for rs in members:
                        mem.append(dict(rs, k_off=k_off))
                        k_off += round_up(rs['rows'], 8)
                    ktot = round_up(k_off, 64)
                    band = dict(i_lo=i_lo, bw=bw, members=mem, ktot=ktot, o_max=mem[0]['o'])
                    band['dthT'] = self.ws16('dthT_b%d' % i_lo, band['o_max'] * bw * ktot)
                    band['uhT'] = self.ws16('uhT_b%d' % i_lo, 8 * C * ktot)
                    self.wgrad_bands.append(band)
                    for m_ in mem:
                        g = m_['g']
                        side_items.append(dict(src_off=self._ws_names['d_tiles'] // 4 + g['tile_off'] +
                                               (m_['row0'] - g['row0']) * g['ld'] + i_lo, rows=m_['rows'],
                                               cols=m_['o'] * bw, ld_src=g['ld'], src_map=(bw, m_['i']),
                                               transposed=(band['dthT'] + m_['k_off'], ktot, bct),
                                               scaled=scaled, tight=True, colsum_parts=n_parts))
                        # (direct 16-bit tiles: the same band copy from the family's 16-bit matrix -- every row of it is
                        # complete once the straight cast of the fp32-route rows has run)
                        side_d16.append(dict(src_off=g['dth'] + (m_['row0'] - g['row0']) * g['dth_ld'] + i_lo, rows=m_['rows'],
                                             cols=m_['o'] * bw, ld_src=g['dth_ld'], src_map=(bw, m_['i']),
                                             transposed=(band['dthT'] + m_['k_off'], ktot, bct),
                                             scaled=scaled, tight=True, colsum_parts=n_parts, src16=True))
                        bias_sets.append((n_parts, (m_['rows'] + 63) // 64, m_['o'], bw, m_['o'] * bw, i_lo))
                        n_parts += ((m_['rows'] + 63) // 64) * m_['o'] * bw
                        u_items.append(dict(src_off=u[1] // 4 + m_['row0'] * 8 * C, rows=m_['rows'], cols=8 * C,
                                            ld_src=8 * C, transposed=(band['uhT'] + m_['k_off'], ktot, bct),
                                            tight=True))
                    i_lo = i_hi
                self._d16_on, self._d16_off = [], []

                def marked(into, fn):
                    n0 = len(self._ops)
                    fn()
                    if d16:
                        into.extend(self._ops[n0:])

                # the dgrad operand on the critical path; the wgrad operands (and the bias gradient) beside it
                marked(self._d16_off, lambda: self.cast16((self.xbuf(self.X_WS), 0), items, amax=amax_t))
                if d16:
                    marked(self._d16_on, lambda: self.cast16((self.xbuf(self.X_WS), 0), items_d16, amax=amax_t))
                    self._build_h16_table(g16, direct_member)
                # decoder.conv.2.bias gradient = column sums of d_tiles: every 64-row tile of the transposed band copies
                # leaves its partial sums in a slot (no atomics), GHN3_OP_ROWSET_COLSUM adds the slots of a bias entry
                # (all row tiles of all row sets with o_r > o', i_r > i') in a fixed order: deterministic
                parts = self.wsf('b2_parts', n_parts)
                # The operand copies of the weight gradient are issued BEHIND the dgrad: the side stream starts them when
                # the dgrad is done instead of beside it (GHN3_WGRAD_PREP_LATE=0).  Same step time, but the dgrad -- on the
                # critical path -- no longer shares the fabric with a 0.8 GB copy: 1.12 -> 1.03 ms.
                prep_late = os.environ.get('GHN3_WGRAD_PREP_LATE', '1') == '1'
                main_ops = self._ops
                if prep_late:
                    self._ops = []
                marked(self._d16_off, lambda: self.cast16((self.xbuf(self.X_WS), 0), side_items + u_items, dbias=parts,
                                                          flags=self.SIDE, amax=amax_t))
                if d16:
                    marked(self._d16_on, lambda: self.cast16((self.xbuf(self.X_WS), 0), side_d16 + u_items, dbias=parts,
                                                             flags=self.SIDE, amax=amax_t))
                sets = np.zeros(len(bias_sets), dtype=L.ROWSET_DT)
                for k_, (off, n_rt, o_, bw_, ld_, i0_) in enumerate(bias_sets):
                    sets[k_] = (off, n_rt, o_, bw_, ld_, i0_, 0)
                self.op(L.OP_ROWSET_COLSUM, refs=(self.gref(b2), parts, self.idx(sets)),
                        ints=(len(sets), ms[0], ms[1]), flags=self.SIDE)
                late_ops, self._ops = (self._ops, main_ops) if prep_late else ([], main_ops)
            p0 = len(self._probs)
            fl = 0.0
            use_rect = bool(g16) and all(g['op16'] for g in self.gemm_groups if g['rows'] >= 512) and \
                any(g['op16'] and g['rows'] >= 512 for g in self.gemm_groups) and \
                os.environ.get('GHN3_DGRAD_RECT', '1') != '0'
            def splits(g):
                """K splits of a group's dgrad: until the launch offers ~512 tiles of 128 x 128 (two workgroups per CU;
                measured best among 512 .. 3072: more splits only add traffic and tile-count quantisation); families with
                >= 512 rows run on 256 x 128 three-stage tiles (tile code 20, one workgroup per CU) and are split to ~1.75
                workgroups per CU (measured plateau 5 <= ks <= 8 for 768 rows: 1.1-1.2 ms against 1.45-1.55 ms)."""
                tiles = ((g['rows'] + 127) // 128) * ((8 * C + 127) // 128)
                tgt = 512 if g['op16'] else 2048       # (the fp32-operand kernel likes ~2048 tiles: 8.2 vs 9.8 ms)
                ks = int(max(2, min(64, (tgt + tiles - 1) // tiles, g['cols'] // 1024)))
                if g['op16'] and g['rows'] >= 512 and use_rect:
                    t20 = ((g['rows'] + 255) // 256) * ((8 * C + 127) // 128)
                    ks = int(max(2, min(16, round(448.0 / t20), g['cols'] // 1024)))
                return ks

            n_planes, rows16 = 1, 0
            if planes32:
                n_planes = int(max(1, min(8, max(g['o'] for g in self.gemm_groups))))
                rows16 = M
                d_up = self.wsf('d_u_parts', max(n_planes - 1, 1) * M * 8 * C)
            if planes:
                # every 16-bit group writes the same number of planes (its own chunk count <= 8, K = 0 problems -- zeros
                # -- for the rest), so that one reduction pass serves all rows
                # GHN3_DGRAD_PIN (default on): eight K chunks per family, chunk j on XCD j -- the chunk's slices of d_tiles and
                # of W2^T are then fetched by one L2 instead of by all eight (4.1 GB of fetches per step without it)
                pin = os.environ.get('GHN3_DGRAD_PIN', '1') != '0'
                for g in g16:
                    g['nc'] = int(max(1, min(8, g['o']))) if pin else int(max(1, min(8, splits(g), g['o'])))
                nc_max = max(g['nc'] for g in g16)
                # Sub-chunks (round 3).  With one K chunk per XCD a family of two or three row tiles is 24 .. 36 output tiles on
                # the XCD's 32 CUs: a quarter of the CUs idles, or a second round starts for a handful of tiles.  Each chunk is
                # therefore cut again into parts `sub` (e.g. 3/4 + 1/4: the eight CUs the long parts leave free take three
                # quarter parts each); every part writes its own partial plane.  The split is chosen by simulating the
                # longest-first dispatch of one XCD's tiles (a plane costs ~7 us of extra traffic).
                sub = self._dgrad_sub_split(g16, 8 * C) if (pin and nc_max == 8 and self.use_p8) else (1,)
                forced = os.environ.get('GHN3_DGRAD_SUB')
                if forced:
                    sub = tuple(int(v) for v in forced.split(','))
                self.dgrad_sub = sub
                n_planes = nc_max * len(sub)
                rows16 = sum(g['rows'] for g in g16)
                assert all(g['row0'] < rows16 for g in g16), 'op16 groups first'
                d_up = self.wsf('d_u_parts', max(n_planes - 1, 1) * M * 8 * C)
            for g in self.gemm_groups:
                fl += sum(2.0 * sb['rows'] * sb['cols'] * 8 * C for sb in g['subs'])
                if planes and g['op16']:
                    # chunk j covers the W2 rows o' in [j * oc, (j + 1) * oc): k = o' * i + i' is contiguous in the copy
                    # of d_tiles and a multiple of the k-map period i, so A and B just start further in
                    # (the kernel consumes whole 64-wide k tiles: a chunk is a multiple of 64 so that the next chunk's
                    # data is never read as padding)
                    unit = 64 // math.gcd(g['i_ld'], 64)
                    sub = self.dgrad_sub
                    # chunk boundaries in W2 rows o' (multiples of `unit`: every part is whole 64-wide k tiles).  A family's
                    # narrow members only reach the first chunks, so equal chunks would give the first XCDs more work than
                    # the last: the boundaries equalise the WORK (row tiles alive at o' x their cost) instead of the length.
                    if g['p8'] and g['nc'] == 8 and os.environ.get('GHN3_DGRAD_EQ', '1') != '0':
                        dens = np.zeros(g['o'])
                        if self.P8_BALANCED_DGRAD:              # work alive at o': its equal tiles
                            lo_ = 0
                            for hi_ in sorted({sb['o'] for sb in g['subs']}):
                                t_, n_ = self.balanced_tiles(self.alive_rows(g, lo_))
                                dens[lo_:hi_] = t_ * self.P8_COSTN[n_]
                                lo_ = hi_
                        else:
                            for (m0_, mi_, ext_) in g['mtiles']:
                                dens[:min(g['o'], int(ext_) // g['i_ld'])] += self.p8_cost(mi_)
                        cw = np.concatenate([[0.0], np.cumsum(dens)])
                        bounds = [0]
                        for j in range(1, g['nc']):
                            b = int(np.searchsorted(cw, cw[-1] * j / g['nc']))
                            bounds.append(min(g['o'], max(bounds[-1], round_up(b, unit))))
                        bounds.append(round_up(g['o'], unit))
                    else:
                        oc = round_up((g['o'] + g['nc'] - 1) // g['nc'], unit)
                        bounds = [min(j * oc, round_up(g['o'], unit)) for j in range(g['nc'])] + [round_up(g['o'], unit)]
                    for pl in range(n_planes):
                        j, sidx = pl // len(sub), pl % len(sub)
                        if j < g['nc']:
                            lo_, hi_ = bounds[j], bounds[j + 1]
                            cuts = [lo_ + min(hi_ - lo_, round_up(int(round((hi_ - lo_) * float(v) / sum(sub))), unit))
                                    for v in np.concatenate([[0], np.cumsum(sub)])]
                            cuts[-1] = hi_
                            o0, o1 = cuts[sidx], max(cuts[sidx], cuts[sidx + 1])
                        else:
                            o0 = o1 = round_up(g['o'], unit)
                        k0 = o0 * g['i_ld']
                        kc = max(0, min(g['cols'] - k0, (o1 - o0) * g['i_ld']))
                        dst = (d_u[0], d_u[1] + 4 * g['row0'] * 8 * C) if pl == 0 else \
                            (d_up[0], d_up[1] + 4 * ((pl - 1) * M + g['row0']) * 8 * C)
                        lim = None
                        if g['ragged'] or kc < g['cols']:
                            lim = self.idx(np.clip(g['lim128'] - k0, 0, kc).astype(np.int32))
                        mt = None
                        if g['p8'] and self.P8_BALANCED_DGRAD:
                            # equal tiles over the rows alive at the chunk's first W2 row (a prefix); rows dead in this chunk
                            # write the zeros of their plane rows (K = 0)
                            alive = self.alive_rows(g, min(o0, g['o'])) if kc > 0 else 0
                            mt = self.range_tiles(alive, g['rows'], lambda r_: g['ext'][r_], k0, kc)
                            mt = (self.idx(mt), len(mt))
                        elif g['p8']:
                            mt = g['mtiles'].copy()
                            mt[:, 2] = np.clip(mt[:, 2] - k0, 0, kc)
                            mt = (self.idx(mt), len(mt))
                        self.gemm(self.href(g['dth'] + min(k0, g['cols'])),
                                  self.sref(self.w2hT + min(o0, g['o']) * ms[1]), dst,
                                  g['rows'], 8 * C, kc, g['dth_ld'], self.w2hT_ld, 8 * C, op16=True,
                                  b_kmap=(g['i_ld'], ms[1]), lim=lim, lim_kind=2, alpha_amax=amax_t,
                                  xcd=j if (pin and nc_max == 8) else None, mtiles=mt)
                    continue
                if planes32:
                    # chunk j = W2 rows o' in [j oc, (j + 1) oc): a multiple of the row-map period i, so A and B just start
                    # further in; chunks beyond a group's own o range are K = 0 problems (zeros)
                    oc = round_up((g['o'] + n_planes - 1) // n_planes, 4)      # (k0 * 4 bytes stays 16-byte aligned)
                    for j in range(n_planes):
                        k0 = min(j * oc, round_up(g['o'], 4)) * g['i_ld']
                        kc = max(0, min(g['cols'] - k0, oc * g['i_ld']))
                        dst = (d_u[0], d_u[1] + 4 * g['row0'] * 8 * C) if j == 0 else \
                            (d_up[0], d_up[1] + 4 * ((j - 1) * M + g['row0']) * 8 * C)
                        self.gemm(self.wref('d_tiles', g['tile_off'] + k0), self.pref(W2, min(j * oc, round_up(g['o'], 4)) * ms[1] * 8 * C),
                                  dst, g['rows'], 8 * C, kc, g['ld'], 8 * C, 8 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                                  b_qs=(g['i_ld'], ms[1]))
                    continue
                if planes:
                    # (groups outside the 16-bit pipeline have i <= 4: a short reduction, one pass into plane 0)
                    self.gemm(self.wref('d_tiles', g['tile_off']), self.pref(W2),
                              (d_u[0], d_u[1] + 4 * g['row0'] * 8 * C),
                              g['rows'], 8 * C, g['cols'], g['ld'], 8 * C, 8 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                              b_qs=(g['i_ld'], ms[1]))
                    continue
                ks = splits(g)
                if g['op16']:
                    # one problem per family; the K loop of a row tile stops at the largest extent of its rows
                    self.gemm(self.href(g['dth']), self.sref(self.w2hT), (d_u[0], d_u[1] + 4 * g['row0'] * 8 * C),
                              g['rows'], 8 * C, g['cols'], g['dth_ld'], self.w2hT_ld, 8 * C, ksplit=ks, op16=True,
                              b_kmap=(g['i_ld'], ms[1]), lim=self.idx(g['lim128']) if g['ragged'] else None,
                              lim_kind=2, alpha_amax=amax_t)
                    continue
                self.gemm(self.wref('d_tiles', g['tile_off']), self.pref(W2),
                          (d_u[0], d_u[1] + 4 * g['row0'] * 8 * C),
                          g['rows'], 8 * C, g['cols'], g['ld'], 8 * C, 8 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                          b_qs=(g['i_ld'], ms[1]), ksplit=ks)
            p8_dgrad = planes and any(g.get('p8') for g in g16)
            self._i_dgrad = len(self._ops)               # (op indices the data-parallel order is cut at, see bwd_parts)
            self.gemm_op(p0, ctype=bct if g16 else None, tag=self.TAG_D3_DGRAD, flops=fl,
                         tile=28 if p8_dgrad else 20 if use_rect else 0)
            self._i_late0 = len(self._ops)
            self._ops.extend(late_ops)
            self._i_late1 = len(self._ops)
            if (planes or planes32) and n_planes > 1:
                self.op(L.OP_DACT, refs=(d_u, u, amax_u if amax_u is not None else self.NONE, d_up),
                        ints=(M, 8 * C, 8 * C, L.DACT_RELU, n_planes - 1, M * 8 * C, rows16))
            else:
                self.op(L.OP_DACT, refs=(d_u, u, amax_u if amax_u is not None else self.NONE),
                        ints=(M, 8 * C, 8 * C, L.DACT_RELU))
            # dW2 = d_tiles^T u.  16-bit bands first (one launch; every dW2 row they cover is written once, without
            # reading it), then the groups on the fp32-operand path accumulate; all on the side stream, in order.
            bands = getattr(self, 'wgrad_bands', []) if g16 else []
            if bands:
                covered = all(b_['o_max'] == ms[0] for b_ in bands) and sum(b_['bw'] for b_ in bands) == ms[1]
                if covered:
                    self.grad_no_memset.append(W2)      # every row of dW2 is written by the band problems
                p0 = len(self._probs)
                fl = 0.0
                # Sum of squares of dW2 from the weight-gradient kernel itself (GHN3_GEMM_SUMSQ: one slot per output tile and
                # wave), when the band problems are its only writers: FusedAdamW's clip_grad_norm_ then skips the 1.8 GB.
                # The table covers every tile id of the launch (the runtime numbers a problem's tiles XCD-blocked: at most
                # tiles + 8 (tiles_m + tiles_n + 1) ids per problem); ids without a tile keep the zero of the memset.
                wg_tile = int(os.environ.get('GHN3_WGRAD_TILE', '29'))
                wg_tn = 128 if wg_tile == 30 else 256     # (column width of the kernel's tiles: its tile ids index the slots)
                sq_on = covered and all(g_['op16'] for g_ in self.gemm_groups) and wg_tile in (29, 30) and \
                    os.environ.get('GHN3_WGRAD_SUMSQ', '1') != '0'
                sq_ids = 0
                n_before = len(self._ops)               # (the zero-fill of the sum-of-squares slots belongs to the weight gradient:
                if sq_on:                               #  it moves with it when the order of the backward changes)
                    for b_ in bands:
                        thr = sorted({m_['o'] for m_ in b_['members']}, reverse=True)
                        for j, o_hi in enumerate(thr):
                            o_lo = thr[j + 1] if j + 1 < len(thr) else 0
                            tm, tn = ((o_hi - o_lo) * b_['bw'] + 255) // 256, (8 * C + wg_tn - 1) // wg_tn
                            sq_ids += tm * tn + 8 * (tm + tn + 1)
                    sq_ref = self.wsf('dw2_sq', 8 * sq_ids)
                    self.op(L.OP_MEMSET0, refs=(sq_ref,), ints=(4 * 8 * sq_ids,))
                    self.grad_sumsq = dict(name=W2, ws_off=sq_ref[1], count=8 * sq_ids)
                for b_ in bands:
                    mem = b_['members']
                    fl += sum(2.0 * m_['rows'] * m_['o'] * b_['bw'] * 8 * C for m_ in mem)   # algorithmic
                    thr = sorted({m_['o'] for m_ in mem}, reverse=True)
                    for j, o_hi in enumerate(thr):
                        o_lo = thr[j + 1] if j + 1 < len(thr) else 0
                        kpre = sum(round_up(m_['rows'], 8) for m_ in mem if m_['o'] >= o_hi)     # a prefix (sorted by o)
                        # (columns of the last k tile beyond kpre belong to rows with o_r <= o_lo: zeros in these W2 rows)
                        self.gemm(self.href(b_['dthT'] + o_lo * b_['bw'] * b_['ktot']), self.href(b_['uhT']),
                                  self.gref(W2, (o_lo * ms[1] + b_['i_lo']) * 8 * C), (o_hi - o_lo) * b_['bw'], 8 * C,
                                  kpre, b_['ktot'], b_['ktot'], 8 * C, c_qs=(b_['bw'], ms[1]), op16=True,
                                  alpha_amax=amax_t, sumsq=sq_ref if sq_on else None)
                # tile 25 = the persistent output-heavy kernel (K = the family's rows: 12 k-tiles per 256 KB of output):
                # 1.44 -> 1.10 ms.  On the side stream it runs as 224 workgroups (GHN3_WGRAD_CAP), leaving 4 CUs per XCD
                # to the dependent chain on the main stream, which the faster kernel otherwise slows down by what it
                # gained (step 8.51 ms with the old kernel, 8.56 with tile 25 on every CU, 8.46 at 224, 8.38 at 192 --
                # but there the weight gradient is back at 1.49 ms).
                # Where the W2 weight gradient runs (round 5: the fastest step is the default; this paragraph is the model of
                # GHN3_WGRAD_ORDER=late -- for the default order 'first', 128 workgroups, see the end of __init__).  The persistent kernel takes
                # 128 KB of LDS and 256 VGPRs on every CU it sits on -- nothing co-resides with it -- so beside it the
                # Graphormer backward chain only gets the CUs the launch leaves free, and it runs ~1.75x slower there
                # whatever their number (memory-system contention: 1.12 -> 1.9-2.0 ms on 128 / 96 / 64 free CUs,
                # profiles/r05a_ab_wgrad_schedule.txt).  With W = the kernel's full-chip time and Ch = the chain's own time
                # the step pays max(W * 256 / c, 1.75 Ch) with c workgroups on the side stream against W + Ch serial: side
                # stream when W >= 0.75 Ch, with c = 256 W / (1.75 Ch) (ghn3xlm16, one 256-node graph: 136; measured
                # 5.82-5.84 ms at 128, 5.84-5.92 at 160, 5.93 at 192, 6.01-6.03 serial; two graphs: 160 -> 9.59-9.67 ms
                # against 9.86-10.09 serial; four graphs: 184 -> 15.8-16.0 against 16.2-16.3).  The kernel's own rate
                # (`roofline.frac`) is measured in a serialised pass (profile mode 3), where the cap is dropped.
                # GHN3_WGRAD_MAIN=1 / 0 and GHN3_WGRAD_CAP override.
                self.wgrad_cap = self._wgrad_schedule(fl, B * N)
                wg_main = self.wgrad_cap == 0
                self.gemm_op(p0, ctype=bct, tag=self.TAG_D3_WGRAD, side=not wg_main, flops=fl,
                             tile=wg_tile,
                             grid_cap=(self.wgrad_cap | (int(os.environ.get('GHN3_WGRAD_TPW', '0')) << 16))
                             if (self.SIDE and not wg_main) else 0)
                self.wgrad_op_range = (n_before, len(self._ops))
            fam_list = bands
            for gi, g in enumerate(self.gemm_groups):
                if g['op16']:
                    continue
                # (fp32-operand path: one launch per group; without 16-bit families the first full-size group
                # writes every dW2 row and needs neither the memset nor the accumulate)
                full = (not fam_list and gi == 0 and g['o'] == ms[0] and g['i_ld'] == ms[1] and g['kind'] == 'conv')
                if full:
                    self.grad_no_memset.append(W2)
                p0 = self.gemm(self.wref('d_tiles', g['tile_off']), (u[0], u[1] + 4 * g['row0'] * 8 * C),
                               self.gref(W2), g['cols'], 8 * C, g['rows'], g['ld'], 8 * C, 8 * C,
                               a_mode=L.MODE_COL, b_mode=L.MODE_COL, c_qs=(g['i_ld'], ms[1]),
                               accum=not full, dbias=self.gref(b2))
                # short reduction (K = rows of the group): 64x64 tiles beat 128x128 here (tools/diag/gemm_bench.py)
                self.gemm_op(p0, tile=64 if g['rows'] <= 1024 else 0, ctype=bct if g16 else None,
                             tag=self.TAG_D3_WGRAD, side=True)
            self.bwd_cut_w2 = len(self._ops)             # every op that writes dW2 has been issued
            if self.wgrad_op_range is not None:
                # (the late order moves ALL of them: the fp32-operand groups accumulate into what the band problems write)
                self.wgrad_op_range = (self.wgrad_op_range[0], self.bwd_cut_w2)
            # D2 backward
            if g16 and hasattr(self, 'w0hT'):
                # 16-bit operands: d_u (straight for the dgrad on the chain; transposed + column sums = bias
                # gradient for the wgrad beside it) and t^T
                Mp = round_up(M, 64)
                duh = self.ws16('duh', M * 8 * C)
                duhT = self.ws16('duhT', 8 * C * Mp)
                thT = self.ws16('thT', 4 * C * Mp)
                self.cast16((self.xbuf(self.X_WS), 0),
                            [dict(src_off=d_u[1] // 4, rows=M, cols=8 * C, ld_src=8 * C, straight=(duh, 8 * C, bct),
                                  scaled=scaled)], amax=amax_u)
                self.cast16((self.xbuf(self.X_WS), 0),
                            [dict(src_off=d_u[1] // 4, rows=M, cols=8 * C, ld_src=8 * C, transposed=(duhT, Mp, bct),
                                  scaled=scaled, colsum_parts=0),
                             dict(src_off=t[1] // 4, rows=M, cols=4 * C, ld_src=4 * C, transposed=(thT, Mp, bct))],
                            dbias=self.wsf('b0_parts', ((M + 63) // 64) * 8 * C), flags=self.SIDE, amax=amax_u)
                set0 = np.zeros(1, dtype=L.ROWSET_DT)
                set0[0] = (0, (M + 63) // 64, 1, 8 * C, 8 * C, 0, 0)
                self.op(L.OP_ROWSET_COLSUM, refs=(self.gref(b0), self.wref('b0_parts'), self.idx(set0)),
                        ints=(1, 1, 8 * C), flags=self.SIDE)
                p0 = self.gemm(self.href(duhT), self.href(thT), self.gref(W0), 8 * C, 4 * C, Mp, Mp, Mp, 4 * C,
                               accum=True, op16=True, alpha_amax=amax_u)
                self.gemm_op(p0, ctype=bct, tag=self.TAG_D2_BWD, side=True, flops=2.0 * 8 * C * 4 * C * M)
                # conv.0 dgrad: [M x 4C] output with K = 8C is 144 tiles of 128 x 128 at ghn3xlm16 -- half the CUs, 48 k-tiles
                # each (154 us, 87 TF).  K chunks as separate problems writing partial planes (summed + masked by a DACT
                # pass in plane order: deterministic) fill the chip.
                d2_tile = int(os.environ.get('GHN3_D2_DGRAD_TILE', '0'))
                ks = int(os.environ.get('GHN3_D2_DGRAD_KS', '0')) or \
                    max(1, min(4, 320 // max(1, ((M + 127) // 128) * ((4 * C + 127) // 128))))
                while ks > 1 and (8 * C) % (64 * ks):
                    ks -= 1
                if ks > 1:
                    kc = 8 * C // ks
                    planes = self.wsf('d_t_parts', (ks - 1) * M * 4 * C)
                    p0 = len(self._probs)
                    for j in range(ks):
                        dst = d_t if j == 0 else (planes[0], planes[1] + 4 * (j - 1) * M * 4 * C)
                        self.gemm(self.href(duh + j * kc), self.sref(self.w0hT + j * kc), dst, M, 4 * C, kc, 8 * C, 8 * C,
                                  4 * C, op16=True, alpha_amax=amax_u)
                    self.gemm_op(p0, ctype=bct, tag=self.TAG_D2_BWD, flops=2.0 * M * 4 * C * 8 * C, tile=d2_tile)
                    self.op(L.OP_DACT, refs=(d_t, t, self.NONE, planes),
                            ints=(M, 4 * C, 4 * C, L.DACT_RELU, ks - 1, M * 4 * C, M))
                else:
                    p0 = self.gemm(self.href(duh), self.sref(self.w0hT), d_t, M, 4 * C, 8 * C, 8 * C, 8 * C, 4 * C,
                                   dact=L.DACT_RELU, aux_in=t, op16=True, alpha_amax=amax_u)
                    self.gemm_op(p0, ctype=bct, tag=self.TAG_D2_BWD)
            else:
                p0 = self.gemm(d_u, t, self.gref(W0), 8 * C, 4 * C, M, 8 * C, 4 * C, 4 * C, a_mode=L.MODE_COL,
                               b_mode=L.MODE_COL, accum=True, dbias=self.gref(b0))
                self.gemm_op(p0, tag=self.TAG_D2_BWD, side=True)
                p0 = self.gemm(d_u, self.pref(W0), d_t, M, 4 * C, 8 * C, 8 * C, 4 * C, 4 * C, a_mode=L.MODE_ROW,
                               b_mode=L.MODE_COL, dact=L.DACT_RELU, aux_in=t)
                self.gemm_op(p0, tag=self.TAG_D2_BWD)
            # D1 backward (per used position)
            p0 = len(self._probs)
            for (p, cnt, r_rows, r_src) in self.d1:
                if hasattr(self, 'wfcT'):             # split-bf16 products against the transposed copies (see _cast_w2)
                    b_hi = self.wfcT + p * C * 4 * C
                    self.gemm(d_t, self.sref(b_hi), d_rows, cnt, C, 4 * C, 4 * C, 4 * C, C, a_gather=r_rows,
                              c_gather=r_rows, x3=(self.sref(b_hi + self.wfcT_lo), 384 if (4 * C) % 384 == 0 else 64))
                    continue
                self.gemm(d_t, self.pref(Wfc, p * C), d_rows, cnt, C, 4 * C, 4 * C, S2 * C, C, a_mode=L.MODE_ROW,
                          b_mode=L.MODE_COL, a_gather=r_rows, c_gather=r_rows)
            self.gemm_op(p0, tag=self.TAG_D1_BWD, ctype=self.d1_bwd_ctype,
                         tile=42 if hasattr(self, 'wfcT') else int(os.environ.get('GHN3_D1_DGRAD_TILE', '0')))
            p0 = len(self._probs)
            for (p, cnt, r_rows, r_src) in self.d1:
                self.gemm(d_t, xe, self.gref(Wfc, p * C), 4 * C, C, cnt, 4 * C, C, S2 * C, a_mode=L.MODE_COL,
                          b_mode=L.MODE_COL, a_gather=r_rows, b_gather=r_src, accum=True,
                          dbias=self.gref(bfc, p), dbias_stride=S2)
            self.gemm_op(p0, tag=self.TAG_D1_BWD, side=True, ctype=self.d1_bwd_ctype)
        if not early_1d:
            decoder_1d_bwd(False)
        # Everything above produces the decoder gradients (93 % of the gradient bytes at XL); a data-parallel caller
        # may run the program in two parts (bwd_ops[:bwd_split] + DETACH, then the rest) and start their all-reduce here.
        self.bwd_split = len(self._ops)
        # ---- d_xe[row] = sum of the decoder rows that read it (deterministic gather-sum) -----------------
        all_src = np.concatenate([self.row_src, self.oned_src]) if (M + n1) else np.zeros(0, dtype=np.int32)
        order = np.argsort(all_src, kind='stable').astype(np.int32)
        counts = np.bincount(all_src, minlength=rows)[:rows]
        seg_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        d_xe = self.wsf('d_xe', rows * C)
        if early_1d:
            self.op(L.OP_JOIN, ints=(2, 0))              # wait for the mark only: the W2 weight gradient keeps running
        self.op(L.OP_ROWSEG_SUM, refs=(d_xe, d_rows, self.idx(seg_ptr), self.idx(order)), ints=(rows, C, C, C, 0))

        # ---- final LayerNorm ----------------------------------------------------------------------------
        dxa = self.wsf('dxa', rows * C)
        do = self.wsf('do', rows * C)
        # (the 64-bit fixed-point histogram of GHN3_OP_BIAS_HIST + its max |dBias| slot live right behind dBias: one zero-fill
        # for both, and the slot is ready when the last attention backward -- layer 0 -- reports the maximum of what it wrote)
        nb_ = round_up(4 * B * H * N * N, 16)
        dBias = self.wsf('dBias', nb_ // 4 + 2 * V * V * H + 16)
        hist = (dBias[0], dBias[1] + nb_)
        hist_amax = (dBias[0], dBias[1] + nb_ + 8 * V * V * H)
        self.op(L.OP_MEMSET0, refs=(dBias,), ints=(nb_ + 8 * V * V * H + 16,))
        xL = self.wref('x%d' % self.Lyr)
        if self.layernorm:
            self.op(L.OP_LAYERNORM_BWD, refs=(dxa, d_xe, xL, self.pref('ln.weight'), self.wref('mf'), self.wref('rf'),
                                              self.NONE), ints=(rows, C))
            self.op(L.OP_LN_PARAM_GRAD, refs=(self.gref('ln.weight'), self.gref('ln.bias'), d_xe, xL,
                                              self.wref('mf'), self.wref('rf')), ints=(rows, C, 1), flags=self.SIDE)
            g_cur = dxa
        else:
            g_cur = d_xe
        # ---- Graphormer layers, reversed ----------------------------------------------------------------
        bias = self.wref('bias')
        pending_ln1 = None                  # (dy buffer, prologue refs) of the LayerNorm backward fused into the next GEMM
        # Side-stream ops of `side_group` consecutive layers are issued together: every main -> side hand-off is an event
        # record that stalls the main chain for 6-12 us, and the temporaries the side ops read are per layer anyway.
        side_group = max(1, int(os.environ.get('GHN3_SIDE_GROUP', '4'))) if self.SIDE else 1
        side_pending = []
        layer_side = []
        defer = bool(self.SIDE) and os.environ.get('GHN3_LAYER_WGRAD_DEFER', '1') != '0'
        defer_group = int(os.environ.get('GHN3_LAYER_WGRAD_GROUP', '0'))     # 0: one launch behind the chain
        side_done = 0
        for l in reversed(range(self.Lyr)):
            pre = 'gnn.%d.' % l
            sfx = '_%d' % l
            x_in = self.wref('x%d' % l)
            h1, qkv, Pm, o = self.wref('h1' + sfx), self.wref('qkv' + sfx), self.wref('P' + sfx), self.wref('o' + sfx)
            xmid, h2, z, f = self.wref('xmid' + sfx), self.wref('h2' + sfx), self.wref('z' + sfx), self.wref('f' + sfx)
            m1, r1, m2, r2 = self.wref('m1' + sfx), self.wref('r1' + sfx), self.wref('m2' + sfx), self.wref('r2' + sfx)
            W3, W1f, Wo, Wq = pre + 'ff.net.3.weight', pre + 'ff.net.0.weight', pre + 'attn.to_out.0.weight', \
                pre + 'attn.to_qkv.weight'
            # The four dgrad GEMMs, the two LayerNorm backward passes and the attention backward form the dependent
            # chain of the layer.  The four wgrad GEMMs (+ fused bias gradients, ONE grouped launch) and the two
            # LayerNorm parameter gradients are off that chain: they run on the side stream, which is why every
            # temporary they read (g_cur, dz, dhA, g_mid, dqkv, dhB) is a buffer of THIS layer -- the chain of the
            # next layer may start while they are still being read.
            lsfx = sfx if self.SIDE else ''
            dz = self.wsf('dz' + lsfx, rows * 4 * C)
            dhA, dhB = self.wsf('dhA' + lsfx, rows * C), self.wsf('dhB' + lsfx, rows * C)
            g_mid = self.wsf('gmid' + lsfx, rows * C)
            g_out = self.wsf(('gout' + sfx) if self.SIDE else 'gout%d' % (l & 1), rows * C)
            dqkv = self.wsf('dqkv' + lsfx, rows * 3 * C)
            x3_here = l < self.Lyr - self.x3_exact_last
            if self.x3s and x3_here:
                # staged split-bf16 dgrads against the transposed (fragment-major) weight copies.  The two LayerNorm
                # backward passes are row prologues: LN2' (+ the residual gradient g_cur) of the to_out dgrad, which writes
                # g_mid; LN1' (+ g_mid) of the ff.net.3 dgrad of the layer BELOW, which writes g_out = that layer's g_cur.
                self.x3s_linear(g_cur if pending_ln1 is None else pending_ln1[0], W3, True, dz, rows, 4 * C, C, C, 4 * C,
                                ln=None if pending_ln1 is None else (2, pending_ln1[1]), dact=L.DACT_GELU, aux_in=z)
                if pending_ln1 is not None:
                    g_cur = pending_ln1[1][5]           # (written by the prologue: this layer's upstream gradient)
                pending_ln1 = None
                self.x3s_linear(dz, W1f, True, dhA, rows, C, 4 * C, 4 * C, C)
                self.x3s_linear(dhA, Wo, True, do, rows, C, C, C, C,
                                ln=(2, [self.pref(pre + 'ln2.weight'), xmid, m2, r2, g_cur, g_mid]))
                self.op(L.OP_ATTN_BWD, refs=(dqkv, do, qkv, Pm, o, hist_amax if l == 0 else self.NONE, dBias, r_nn), ints=(B, N, C, H))
                self.x3s_linear(dqkv, Wq, True, dhB, rows, C, 3 * C, 3 * C, C)
                if l > 0:
                    pending_ln1 = (dhB, [self.pref(pre + 'ln1.weight'), x_in, m1, r1, g_mid, g_out])
                else:
                    self.op(L.OP_LAYERNORM_BWD, refs=(g_out, dhB, x_in, self.pref(pre + 'ln1.weight'), m1, r1, g_mid,
                                                      self.NONE), ints=(rows, C))
            elif self.x3 and x3_here:
                # split-bf16 dgrads against the transposed weight copies; K splits -> planes summed by the LayerNorm
                # backward ops (which write the sums back for the LayerNorm parameter gradients on the side stream)
                self.x3_linear(g_cur, W3, True, dz, rows, 4 * C, C, C, 4 * C, dact=L.DACT_GELU, aux_in=z)
                np_, pl = self.x3_linear(dz, W1f, True, dhA, rows, C, 4 * C, 4 * C, C, split=True, planes='dhA_plane')
                self.op(L.OP_LAYERNORM_BWD, refs=(g_mid, dhA, xmid, self.pref(pre + 'ln2.weight'), m2, r2, g_cur,
                                                  pl if np_ else self.NONE), ints=(rows, C, np_, rows * C))
                # (its output feeds the attention backward, which reads ONE matrix: no K split)
                self.x3_linear(g_mid, Wo, True, do, rows, C, C, C, C)
                self.op(L.OP_ATTN_BWD, refs=(dqkv, do, qkv, Pm, o, hist_amax if l == 0 else self.NONE, dBias, r_nn), ints=(B, N, C, H))
                np_, pl = self.x3_linear(dqkv, Wq, True, dhB, rows, C, 3 * C, 3 * C, C, split=True, planes='dhB_plane')
                self.op(L.OP_LAYERNORM_BWD, refs=(g_out, dhB, x_in, self.pref(pre + 'ln1.weight'), m1, r1, g_mid,
                                                  pl if np_ else self.NONE), ints=(rows, C, np_, rows * C))
            else:
                # FFN second linear: x_out = xmid + f W3^T + b3.  With fuse_ln the upstream gradient g_cur of every layer
                # but the last is LN1'(dhB) + g_mid of the layer above, produced by this GEMM's row prologue.
                p0 = self.gemm(g_cur if pending_ln1 is None else pending_ln1[0], self.pref(W3), dz, rows, 4 * C, C, C,
                               4 * C, 4 * C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL, dact=L.DACT_GELU, aux_in=z,
                               ln=None if pending_ln1 is None else (2, pending_ln1[1], 0.0))
                pending_ln1 = None
                self.gemm_op(p0)
                # FFN first linear
                dhA_p = None
                if self.split_small(rows, C, 4 * C):
                    dhA_p = self.wsf('dhA_plane', rows * C)
                    p0 = self.gemm_k2(dz, self.pref(W1f), dhA, dhA_p, rows, C, 4 * C, 4 * C, C, C, L.MODE_COL)
                else:
                    p0 = self.gemm(dz, self.pref(W1f), dhA, rows, C, 4 * C, 4 * C, C, C, a_mode=L.MODE_ROW,
                                   b_mode=L.MODE_COL)
                self.gemm_op(p0)
                # LN2 (+ residual branch gradient g_cur)
                if self.fuse_ln:
                    # attention output projection: xmid = x_in + o Wo^T + bo; its A operand g_mid = LN2'(dhA) + g_cur is
                    # computed (and written for the wgrad / the residual path) by the GEMM's row prologue
                    p0 = self.gemm(dhA, self.pref(Wo), do, rows, C, C, C, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL,
                                   ln=(2, [self.pref(pre + 'ln2.weight'), xmid, m2, r2, g_cur, g_mid], 0.0))
                else:
                    self.op(L.OP_LAYERNORM_BWD, refs=(g_mid, dhA, xmid, self.pref(pre + 'ln2.weight'), m2, r2, g_cur,
                                                      dhA_p or self.NONE), ints=(rows, C))
                    # attention output projection: xmid = x_in + o Wo^T + bo
                    p0 = self.gemm(g_mid, self.pref(Wo), do, rows, C, C, C, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL)
                self.gemm_op(p0)
                self.op(L.OP_ATTN_BWD, refs=(dqkv, do, qkv, Pm, o, hist_amax if l == 0 else self.NONE, dBias, r_nn), ints=(B, N, C, H))
                dhB_p = None
                if self.split_small(rows, C, 3 * C):
                    dhB_p = self.wsf('dhB_plane', rows * C)
                    p0 = self.gemm_k2(dqkv, self.pref(Wq), dhB, dhB_p, rows, C, 3 * C, 3 * C, C, C, L.MODE_COL)
                else:
                    p0 = self.gemm(dqkv, self.pref(Wq), dhB, rows, C, 3 * C, 3 * C, C, C, a_mode=L.MODE_ROW,
                                   b_mode=L.MODE_COL)
                self.gemm_op(p0)
                if self.fuse_ln and l > 0:
                    pending_ln1 = (dhB, [self.pref(pre + 'ln1.weight'), x_in, m1, r1, g_mid, g_out])
                else:
                    self.op(L.OP_LAYERNORM_BWD, refs=(g_out, dhB, x_in, self.pref(pre + 'ln1.weight'), m1, r1, g_mid,
                                                      dhB_p or self.NONE), ints=(rows, C))
            # Side-stream work of the layer: four weight gradients (+ fused bias gradients) and the two LayerNorm parameter
            # gradients.  Round 4 (`defer`, side stream only): ALL layers' weight gradients are ONE grouped launch and all
            # LayerNorm parameter gradients ONE batched launch behind the chain -- the 24 x 3 small side launches beside the
            # chain cost 0.30 ms of step time (hand-off stalls, CU slots) for 0.56 ms of kernel time; every temporary they read
            # is a per-layer buffer anyway.  Without a side stream (or GHN3_LAYER_WGRAD_DEFER=0) they follow their layer.
            layer_side.append(dict(pre=pre, g_cur=g_cur, f=f, dz=dz, h2=h2, g_mid=g_mid, o=o, dqkv=dqkv, h1=h1, dhA=dhA,
                                   xmid=xmid, m2=m2, r2=r2, dhB=dhB, x_in=x_in, m1=m1, r1=r1))
            if defer and defer_group and len(layer_side) - side_done >= defer_group and l > 0:
                self._layer_side_ops(layer_side[side_done:], rows)       # (a group of layers, beside the chain)
                side_done = len(layer_side)
            if not defer:
                main_ops, self._ops = self._ops, []
                self._layer_side_ops([layer_side[-1]], rows)
                if os.environ.get('GHN3_SKIP_LAYER_SIDE', '0') != '0':   # (timing experiment: wrong gradients)
                    self._ops = []
                side_pending, self._ops = side_pending + self._ops, main_ops
                if (self.Lyr - l) % side_group == 0 or l == 0:
                    self._ops.extend(side_pending)
                    side_pending = []
            g_cur = g_out                   # d x_l
        if defer and os.environ.get('GHN3_SKIP_LAYER_SIDE', '0') == '0' and side_done < len(layer_side):
            self._layer_side_ops(layer_side[side_done:], rows)
        # ---- node embeddings (side stream: beside the edge-bias backward below, which does not depend on it) --------
        self.op(L.OP_EMBED_BWD,
                refs=(g_cur, r_types, r_shape, r_nn, r_noff, self.gref('embed.weight'),
                      self.gref('shape_enc.embed_channel.weight'), self.gref('shape_enc.embed_spatial.weight'),
                      self.gref('gnn.0.centrality_embed_in.weight'), self.gref('gnn.0.centrality_embed_out.weight'),
                      self.gref('gnn.0.input_dist_embed.weight'), deg_in, deg_out, dist0),
                ints=(B, N, C, len(bk.PRIMITIVES_DEEPNETS1M), self.vocab_rows[0], self.vocab_rows[1]),
                flags=0 if os.environ.get('GHN3_EMBED_BWD_MAIN', '1') == '1' else self.SIDE)
        # ---- layer-0 edge bias: histogram -> table MLP backward ------------------------------------------
        E = 'gnn.0.attn.edge_embed.embed.weight'
        W0e, b0e = 'gnn.0.attn.proj_e.0.weight', 'gnn.0.attn.proj_e.0.bias'
        W2e, b2e = 'gnn.0.attn.proj_e.2.weight', 'gnn.0.attn.proj_e.2.bias'
        dT = self.wsf('dT', V * V * ldT)
        dhid = self.wsf('dhid', V * V * C)
        dPfw, dPbw = self.wsf('dPfw', V * C), self.wsf('dPbw', V * C)
        hid = self.wref('hid')
        self.op(L.OP_MEMSET0, refs=(dT,), ints=(4 * V * V * ldT,))
        self.op(L.OP_BIAS_HIST, refs=(dT, dBias, pair, hist), ints=(B, N, H, V, 1))
        p0 = self.gemm(dT, hid, self.gref(W2e), H, C, V * V, ldT, C, C, a_mode=L.MODE_COL, b_mode=L.MODE_COL,
                       accum=True, dbias=self.gref(b2e))
        self.gemm_op(p0, side=True)                      # (weight gradient: off the chain)
        p0 = self.gemm(dT, self.pref(W2e), dhid, V * V, C, H, ldT, C, C, a_mode=L.MODE_ROW, b_mode=L.MODE_COL)
        self.gemm_op(p0)
        self.op(L.OP_EDGE_HIDDEN_BWD, refs=(dPfw, dPbw, dhid, hid), ints=(V, C))
        p0 = self.gemm(dPfw, self.pref(E, 2 * C), self.gref(W0e, 0), C, C, V, C, C, 2 * C, a_mode=L.MODE_COL,
                       b_mode=L.MODE_COL, accum=True)
        self.gemm(dPbw, self.pref(E, 2 * C), self.gref(W0e, C), C, C, V, C, C, 2 * C, a_mode=L.MODE_COL,
                  b_mode=L.MODE_COL, accum=True, dbias=self.gref(b0e))
        self.gemm(dPfw, self.pref(W0e, 0), self.gref(E, 2 * C), V, C, C, C, 2 * C, C, a_mode=L.MODE_ROW,
                  b_mode=L.MODE_COL, accum=True)
        self.gemm_op(p0)
        p0 = self.gemm(dPbw, self.pref(W0e, C), self.gref(E, 2 * C), V, C, C, C, 2 * C, C, a_mode=L.MODE_ROW,
                       b_mode=L.MODE_COL, accum=True)
        self.gemm_op(p0)
