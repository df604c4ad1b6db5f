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
            ops[7]['kind'] = L.OP_DETACH
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
