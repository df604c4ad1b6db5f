"""
Fused trainer step on the GHN's flat buffers.

Replaces, for a ghn3_amd.GHN3 whose parameters and gradients live in one flat fp32 buffer each, the reference's
    nn.utils.clip_grad_norm_(parameters, grad_clip)      /root/reference/ghn3/trainer.py:356-360
    optimizer.step()   (torch.optim.AdamW)               /root/reference/ghn3/trainer.py:165-175,379
by two kernels over the flat buffers (GHN3_OP_SUMSQ + GHN3_OP_ADAMW) -- no per-tensor launches.
"""

import os

import numpy as np
import torch

from . import _lib as L


def _dbits(x):
    return int(np.asarray([x], dtype=np.float64).view(np.int64)[0])


class FusedAdamW:
    """AdamW + gradient-norm clipping for a ghn3_amd.GHN3 (same update rule and defaults as torch.optim.AdamW)."""

    def __init__(self, ghn, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0,
                 nan_guard=True):
        self.ghn = ghn
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        # nan_guard: the squared gradient norm is always computed; when it is not finite (a NaN / inf anywhere in the
        # flat gradient -- after the data-parallel average every rank sees the same value) the kernel leaves parameters
        # and moments untouched: the reference trainer's skip-this-batch (trainer.py:240-257) without a host sync
        self.nan_guard = nan_guard
        flat = ghn._flat
        if not flat.is_cuda:
            raise L.Ghn3Error('FusedAdamW runs on an MI355X only (no CPU path)')
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scal = torch.zeros(16 + 8192, dtype=torch.float32, device=flat.device)   # [norm^2 ...| partial sums]
        self.steps = 0
        self._w2_descs = {}

    def _w2_fusion(self, plan):
        """(lo, hi, device descriptor table, work tiles, state key) when this step may write the 16-bit copies of
        decoder.conv.2.weight itself (GHN3_OP_ADAMW_CAST16), else None: the plan's program casts W2, the copies exist and were
        current for the parameters of the last forward."""
        ghn = self.ghn
        prog = plan.program if plan is not None else None
        w2 = getattr(prog, 'shadow_w2', None) if prog is not None else None
        if w2 is None or not prog.training or ghn._shadow is None or os.environ.get('GHN3_ADAMW_CAST', '1') == '0':
            return None
        # (the W2 copies must be current for the parameters as they are now -- written by the last forward's refresh or by
        # the previous fused step -- or a step the NaN guard skips would leave stale copies marked as current)
        ver = ghn._shadow_version()
        st, st2 = ghn._shadow_state, ghn._shadow_w2_state
        if not ((st is not None and st[0] == ver) or (st2 is not None and st2[0] == ver)):
            return None
        it = w2['item']
        if it['ld_src'] != it['cols'] or it['cols'] % 4:
            return None
        lo = int(ghn._offs[prog.slot[w2['name']]])
        key = (it['rows'], it['cols'], tuple(it.get('straight') or ()), tuple(it.get('transposed') or ()))
        if key not in self._w2_descs:
            descs, blocks = prog.pack_cast_descs([it])
            self._w2_descs[key] = (torch.from_numpy(descs.view(np.uint8).copy()).to(ghn._flat.device), blocks)
        d, blocks = self._w2_descs[key]
        types = (prog.decoder_ctype, prog.decoder_bwd_ctype, prog.x3, prog.uses_op16)
        return lo, lo + it['rows'] * it['cols'], d, blocks, types

    def wait(self):
        """Makes torch's current stream wait for an overlapped step's side-stream half (see step(overlap=True)): call before
        anything but a GHN3 forward reads the parameters or the moments (checkpoints, .cpu(), torch ops on them)."""
        if getattr(self, '_side_busy', False):
            self.ghn._ctx().side_wait(torch.cuda.current_stream().cuda_stream)
            self._side_busy = False

    def step(self, gflat, grad_scale=1.0, plan=None, local_grads=False, overlap=False):
        """gflat: the flat gradient buffer of the last backward (plan.gflat); grad_scale: the loss scale the gradients
        carry (AMP: they are divided by it inside the kernel, no separate unscale pass).  Returns the gradient norm
        (device scalar, like clip_grad_norm_) when clipping or the NaN guard is on; when it is not finite the kernel
        left parameters and moments untouched.
        plan: the plan of the step's forward / backward.  With it the update of decoder.conv.2.weight (69 % of the
        parameters at ghn3xlm16) also writes the weight's 16-bit operand copies (GHN3_OP_ADAMW_CAST16), so the next
        forward does not read the 1.8 GB again to re-cast them; same parameters bit for bit.
        local_grads: `gflat` is exactly what the plan's last backward wrote (NOT averaged over ranks afterwards).  The
        squared norm of the W2 gradient is then taken from the sums the weight-gradient kernel left per output tile
        (Program.grad_sumsq, GHN3_GEMM_SUMSQ) instead of from a pass over its 1.8 GB.
        overlap: the update of the DECODER parameters (93 % of them at ghn3xlm16: 3 ms of HBM-bound streaming) runs on the
        context's side stream and the call returns with it pending; the embeddings and the Graphormer (updated on the
        caller's stream first) are all the next forward needs for its first ~1 ms -- the latency-bound Graphormer chain --
        and its program joins the side stream in front of the decoders (Program._build_decoder_forward).  Same kernels, same
        order per element: parameters bit-identical to the serial step.  Until that forward (or wait()) nothing else may
        read the decoder parameters / moments; the gradient buffer is kept alive here until the next call."""
        ghn = self.ghn
        flat = ghn._flat
        assert gflat.numel() == flat.numel() and gflat.is_cuda
        # a previous overlapped step may still be reading `scal` and writing the decoder ranges on the side stream (two steps
        # without a GHN3 forward in between: gradient accumulation, a timing loop): order this step behind it
        self.wait()
        self.steps += 1
        n = flat.numel()
        clip = self.max_grad_norm and self.max_grad_norm > 0
        fuse = self._w2_fusion(plan)
        overlap = bool(overlap) and getattr(ghn, 'side_stream', True) and plan is not None and \
            hasattr(plan.program, 'decoder_slots') and os.environ.get('GHN3_ADAMW_OVERLAP', '1') != '0'
        ops = np.zeros(9, dtype=L.OP_DT)
        ops['r']['buf'][:] = -1
        bufs = [flat.data_ptr(), gflat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                self.scal.data_ptr(), self.scal.data_ptr() + 64, 0, 0]
        ops[0]['kind'] = L.OP_MEMSET0
        ops[0]['r']['buf'][0] = 4
        ops[0]['i'][0] = 4
        guard = clip or self.nan_guard
        ops[1]['kind'] = L.OP_SUMSQ if guard else L.OP_NOP
        ops[1]['r']['buf'][:3] = (4, 1, 5)         # (r2: scratch for the fixed-order sum of the workgroup partials)
        ops[1]['i'][0] = n
        gs = getattr(plan.program, 'grad_sumsq', None) if (plan is not None and local_grads and guard) else None
        if gs is not None and getattr(plan, 'gflat', None) is gflat and os.environ.get('GHN3_WGRAD_SUMSQ', '1') != '0':
            prog = plan.program
            k2 = prog.slot[gs['name']]
            lo2 = int(ghn._offs[k2])
            hi2 = int(ghn._offs[k2 + 1]) if k2 + 1 < len(ghn._offs) else n     # (the slot's padding holds zeros)
            bufs.append(int(plan.bufs[prog.xbuf(prog.X_WS)]) + gs['ws_off'])
            ops[1]['i'][1:4] = (lo2, hi2, gs['count'])
            ops[1]['r']['buf'][3] = len(bufs) - 1
        hyper = (self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                 1.0 - self.betas[0] ** self.steps, 1.0 - self.betas[1] ** self.steps)

        def adamw(op, kind, lo, count):
            op['kind'] = kind
            op['r']['buf'][:5] = (0, 1, 2, 3, 4 if guard else -1)
            op['r']['off'][:4] = 4 * lo
            op['i'][0] = count
            for k, h in enumerate(hyper):
                op['i'][1 + k] = _dbits(h)
            op['f'][0] = float(self.max_grad_norm or 0.0) if clip else 0.0
            op['f'][1] = 1.0 / float(grad_scale)

        if overlap:
            # [everything but the decoders] on the caller's stream, then the decoder ranges on the side stream, detached
            d_lo, d_hi = ghn.decoder_grad_range(plan.program)
            adamw(ops[2], L.OP_ADAMW, 0, d_lo)
            adamw(ops[3], L.OP_ADAMW, d_hi, n - d_hi)
            if fuse is None:
                adamw(ops[4], L.OP_ADAMW, d_lo, d_hi - d_lo)
            else:
                lo, hi, descs, blocks, types = fuse
                assert d_lo <= lo and hi <= d_hi
                bufs[6], bufs[7] = ghn._shadow.data_ptr(), descs.data_ptr()
                adamw(ops[4], L.OP_ADAMW, d_lo, lo - d_lo)
                adamw(ops[5], L.OP_ADAMW, hi, d_hi - hi)
                adamw(ops[6], L.OP_ADAMW_CAST16, lo, 1 + (blocks << 32))
                ops[6]['r']['buf'][5:7] = (6, 7)
            for k in (4, 5, 6):
                if int(ops[k]['kind']) != L.OP_NOP:
                    ops[k]['flags'] |= L.OPFLAG_SIDE
            # DETACH must END the run: the runtime ignores GHN3_OP_NOP padding behind it, but nothing else may follow
            # (round 6: it sat at ops[7] of 9 with a NOP behind it and the runtime re-armed the final join -- the
            # "overlapped" step of round 5 was serialised)
            ops[-1]['kind'] = L.OP_DETACH
            self._side_busy = True
            self._keep = gflat                   # (read by the side stream after this call returns)
        elif fuse is None:
            adamw(ops[2], L.OP_ADAMW, 0, n)
        else:
            lo, hi, descs, blocks, types = fuse
            bufs[6], bufs[7] = ghn._shadow.data_ptr(), descs.data_ptr()
            adamw(ops[2], L.OP_ADAMW, 0, lo)
            adamw(ops[3], L.OP_ADAMW, hi, n - hi)
            adamw(ops[4], L.OP_ADAMW_CAST16, lo, 1 + (blocks << 32))
            ops[4]['r']['buf'][5:7] = (6, 7)
        ghn._ctx().run(ops, np.zeros(0, dtype=L.PROBLEM_DT), np.asarray(bufs, dtype=np.uint64),
                       torch.cuda.current_stream().cuda_stream)
        ghn.params_changed()                     # (the kernel wrote the parameters through raw pointers)
        if fuse is not None:
            ghn._shadow_w2_state = (ghn._shadow_version(), True, fuse[4])
        return self.scal[0].sqrt() / float(grad_scale) if guard else None

    # ------------------------------------------------------------------ checkpoints (trainer.py:413-432)
    def state_dict(self):
        """State in the layout of ``torch.optim.AdamW.state_dict()`` over ``ghn.parameters()`` (the order the reference's
        trainer hands to its optimizer, trainer.py:165-175), so that ``{'state_dict', 'optimizer', 'epoch', 'step'}``
        checkpoints written by either trainer resume in the other.  The moment tensors are views of the flat buffers."""
        ghn = self.ghn
        self.wait()                               # (an overlapped step may still be writing the moments)
        slot_of = {id(p): k for k, p in enumerate(ghn._slot_params())}
        state, order = {}, []
        for i, p in enumerate(ghn.parameters()):
            k = slot_of[id(p)]
            o, n = int(ghn._offs[k]), p.numel()
            order.append(i)
            if self.steps > 0:
                state[i] = {'step': torch.tensor(float(self.steps)),
                            'exp_avg': self.exp_avg[o:o + n].view(p.shape),
                            'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape)}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.weight_decay,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'params': order}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts a ``torch.optim.AdamW`` (or FusedAdamW) state dict over ``ghn.parameters()``."""
        ghn = self.ghn
        self.wait()
        groups = sd['param_groups']
        assert len(groups) == 1, 'one parameter group expected (trainer.py:175)'
        g = groups[0]
        self.lr, self.betas, self.eps = float(g['lr']), tuple(g['betas']), float(g['eps'])
        self.weight_decay = float(g['weight_decay'])
        assert not g.get('amsgrad', False), 'amsgrad is not supported'
        params = list(ghn.parameters())
        assert len(g['params']) == len(params), (len(g['params']), len(params))
        slot_of = {id(p): k for k, p in enumerate(ghn._slot_params())}
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = 0
        for idx, p in zip(g['params'], params):
            st = sd['state'].get(idx)
            if st is None:
                continue
            k = slot_of[id(p)]
            o, n = int(ghn._offs[k]), p.numel()
            self.exp_avg[o:o + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps = max(steps, int(float(st['step'])))
        self.steps = steps


def adamw_reference_(p, g, m, v, sumsq, step, lr, betas, eps, weight_decay, max_norm, inv_scale):
    """torch restatement of one GHN3_OP_ADAMW launch on views of the flat buffers (host-side reference of the sharded step's
    CPU tests; the GPU path runs the HIP op): g / scale, clip coefficient from the GLOBAL squared norm `sumsq`, decoupled
    weight decay, bias-corrected moments -- torch.optim.AdamW's update.  A non-finite norm leaves everything untouched."""
    norm = float(sumsq) ** 0.5 * inv_scale
    if not np.isfinite(norm):
        return
    coef = inv_scale * (min(1.0, max_norm / (norm + 1e-6)) if max_norm > 0 else 1.0)
    gg = g * coef
    p.mul_(1.0 - lr * weight_decay)
    m.mul_(betas[0]).add_(gg, alpha=1.0 - betas[0])
    v.mul_(betas[1]).addcmul_(gg, gg, value=1.0 - betas[1])
    denom = (v / (1.0 - betas[1] ** step)).sqrt_().add_(eps)
    p.addcdiv_(m / (1.0 - betas[0] ** step), denom, value=-lr)


class ShardedAdamW:
    """
    The optimizer step split over the data-parallel ranks (ZeRO-1 shape; round 5).

    The replicated trainer all-reduces 2.62 GB of gradients (ghn3xlm16) and then EVERY rank streams the same 18 GB through
    AdamW (3.7 ms).  Here the gradient exchange stops after its reduce-scatter half (FlatGradReducer(gather=False)): a rank
    holds the mean gradient of 1 / W of every chunk of the flat buffer, updates exactly those parameters (clip coefficient
    from the all-reduced squared norm: one scalar) and all-gathers the updated PARAMETERS -- the same bytes on the wire as
    the all-gather half of the gradient exchange it replaces, AdamW's HBM traffic divided by W (0.5 ms at W = 8).
    Parameters after a step are bit-identical to the replicated path with the same exchange (every element is reduced once,
    on its owner, in both).  The 16-bit copies of decoder.conv.2.weight are re-cast by the next forward (the fused
    GHN3_OP_ADAMW_CAST16 works on whole 64 x 64 tiles of the matrix, not on arbitrary shards): 1.2 ms, against 3.2 ms saved
    at W = 8.

    `update(lo, hi, sumsq, step)`: applies AdamW to the range of the flat buffers; default = the HIP op on the GHN's buffers,
    tests on CPU pass a torch restatement (adamw_reference_).
    """

    def __init__(self, ghn=None, flat=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0,
                 update=None, local_sumsq=None):
        self.ghn = ghn
        self.flat = ghn._flat if flat is None else flat
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self._update, self._local_sumsq = update, local_sumsq
        self.scal = torch.zeros(16 + 8192, dtype=torch.float32, device=self.flat.device)

    def _hip_ops(self, recs, bufs):
        ops = np.zeros(len(recs), dtype=L.OP_DT)
        ops['r']['buf'][:] = -1
        for k, fill in enumerate(recs):
            fill(ops[k])
        self.ghn._ctx().run(ops, np.zeros(0, dtype=L.PROBLEM_DT), np.asarray(bufs, dtype=np.uint64),
                            torch.cuda.current_stream().cuda_stream)

    def step(self, gflat, reducer, grad_scale=1.0):
        """gflat: the flat gradient buffer after a backward whose exchange was `reducer` = FlatGradReducer(gather=False) --
        reducer.owned / .replicated say which ranges hold the mean gradient here.  Returns the global gradient norm."""
        import torch.distributed as dist
        flat = self.flat
        W = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        self.steps += 1
        own, rep = sorted(reducer.owned), sorted(reducer.replicated)
        # reducer.gather (the exchange also all-gathered the GRADIENTS): the replicated update -- every rank updates every
        # element, nothing to gather afterwards; the squared norm is still summed shard-wise (1 / W of the buffer per rank +
        # one scalar all-reduce), so both modes clip with the same bits
        update_all = bool(getattr(reducer, 'gather', False))
        ranges = [(0, flat.numel())] if update_all else sorted(own + rep)
        norm_ranges = sorted(own + (rep if rank == 0 else []))             # (every element counted on exactly one rank)
        hip = self._update is None
        inv_scale = 1.0 / float(grad_scale)
        if hip:
            bufs = [flat.data_ptr(), gflat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                    self.scal.data_ptr(), self.scal.data_ptr() + 64]

            def memset(op):
                op['kind'], op['r']['buf'][0], op['i'][0] = L.OP_MEMSET0, 4, 4

            def sumsq(lo, hi):
                def f(op):
                    op['kind'] = L.OP_SUMSQ
                    op['r']['buf'][:3] = (4, 1, 5)
                    op['r']['off'][1] = 4 * lo
                    op['i'][0] = hi - lo
                return f
            self._hip_ops([memset] + [sumsq(lo, hi) for lo, hi in norm_ranges], bufs)
            total = self.scal[:1]
        else:
            total = torch.zeros(1, dtype=torch.float32, device=flat.device)
            for lo, hi in norm_ranges:
                total += (self._local_sumsq or (lambda t: (t.double() ** 2).sum().float()))(gflat[lo:hi])
        if W > 1:
            dist.all_reduce(total, op=dist.ReduceOp.SUM)
        if hip:
            hyper = (self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                     1.0 - self.betas[0] ** self.steps, 1.0 - self.betas[1] ** self.steps)

            def adamw(lo, hi):
                def f(op):
                    op['kind'] = L.OP_ADAMW
                    op['r']['buf'][:5] = (0, 1, 2, 3, 4)
                    op['r']['off'][:4] = 4 * lo
                    op['i'][0] = hi - lo
                    for k, h in enumerate(hyper):
                        op['i'][1 + k] = _dbits(h)
                    op['f'][0] = float(self.max_grad_norm or 0.0)
                    op['f'][1] = inv_scale
                return f
            self._hip_ops([adamw(lo, hi) for lo, hi in ranges], bufs)
            self.ghn.params_changed()
        else:
            for lo, hi in ranges:
                self._update(flat[lo:hi], gflat[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], total, self.steps,
                             self.lr, self.betas, self.eps, self.weight_decay, float(self.max_grad_norm or 0.0), inv_scale)
        # all-gather of the updated parameters: the geometry of the gradient chunks (replicated tails are identical already)
        if (W > 1 or reducer.force) and not update_all:
            for (s, main), (lo, hi) in zip(reducer.chunks, reducer.owned):
                dist.all_gather_into_tensor(flat[s:s + main], flat[lo:hi].clone())
        return total.sqrt() * inv_scale


def save_checkpoint(path, ghn, optimizer, epoch, step, config=None):
    """Checkpoint in the reference trainer's format (trainer.py:413-426): {'state_dict', 'optimizer', 'epoch', 'step',
    **config}; `ghn3_amd.from_pretrained(path)` and the reference's resume code both read it."""
    ckpt = {'state_dict': {k: v.detach().cpu() for k, v in ghn.state_dict().items()},
            'optimizer': {'state': {i: {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in st.items()}
                                    for i, st in optimizer.state_dict()['state'].items()},
                          'param_groups': optimizer.state_dict()['param_groups']},
            'epoch': epoch, 'step': step}
    ckpt.update(config or {})
    torch.save(ckpt, path)
    return path
