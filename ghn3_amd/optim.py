"""
Fused trainer step on the GHN's flat buffers.

Replaces, for a ghn3_amd.GHN3 whose parameters and gradients live in one flat fp32 buffer each, the reference's
    nn.utils.clip_grad_norm_(parameters, grad_clip)      /root/reference/ghn3/trainer.py:356-360
    optimizer.step()   (torch.optim.AdamW)               /root/reference/ghn3/trainer.py:165-175,379
by two kernels over the flat buffers (GHN3_OP_SUMSQ + GHN3_OP_ADAMW) -- no per-tensor launches.
"""

import numpy as np
import torch

from . import _lib as L


def _dbits(x):
    return int(np.asarray([x], dtype=np.float64).view(np.int64)[0])


class FusedAdamW:
    """AdamW + gradient-norm clipping for a ghn3_amd.GHN3 (same update rule and defaults as torch.optim.AdamW)."""

    def __init__(self, ghn, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=0.0):
        self.ghn = ghn
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        flat = ghn._flat
        if not flat.is_cuda:
            raise L.Ghn3Error('FusedAdamW runs on an MI355X only (no CPU path)')
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scal = torch.zeros(16, dtype=torch.float32, device=flat.device)
        self.steps = 0

    def step(self, gflat):
        """gflat: the flat gradient buffer of the last backward (plan.gflat).  Returns the gradient norm tensor
        (device scalar, like clip_grad_norm_) when clipping is on."""
        ghn = self.ghn
        flat = ghn._flat
        assert gflat.numel() == flat.numel() and gflat.is_cuda
        self.steps += 1
        n = flat.numel()
        clip = self.max_grad_norm and self.max_grad_norm > 0
        ops = np.zeros(3, dtype=L.OP_DT)
        ops['r']['buf'][:] = -1
        bufs = np.asarray([flat.data_ptr(), gflat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                           self.scal.data_ptr()], dtype=np.uint64)
        ops[0]['kind'] = L.OP_MEMSET0
        ops[0]['r']['buf'][0] = 4
        ops[0]['i'][0] = 4
        ops[1]['kind'] = L.OP_SUMSQ if clip else L.OP_NOP
        ops[1]['r']['buf'][:2] = (4, 1)
        ops[1]['i'][0] = n
        ops[2]['kind'] = L.OP_ADAMW
        ops[2]['r']['buf'][:5] = (0, 1, 2, 3, 4 if clip else -1)
        ops[2]['i'][0] = n
        hyper = (self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                 1.0 - self.betas[0] ** self.steps, 1.0 - self.betas[1] ** self.steps)
        for k, h in enumerate(hyper):
            ops[2]['i'][1 + k] = _dbits(h)
        ops[2]['f'][0] = float(self.max_grad_norm or 0.0)
        ghn._ctx().run(ops, np.zeros(0, dtype=L.PROBLEM_DT), bufs, torch.cuda.current_stream().cuda_stream)
        return self.scal[0].sqrt() if clip else None
