"""
Small training utilities with the names the reference exports from ghn3/utils.py (``log``, ``Logger``,
``print_grads``), so that scripts written against it import unchanged.  ``transforms_imagenet`` needs torchvision and is
outside the parameter-prediction path.
"""

import os
import time

import torch

from .ddp_utils import get_ddp_rank


def log(*args, **kwargs):
    """print on rank 0 only (utils.py:26-28)."""
    if get_ddp_rank() == 0:
        print(*args, **kwargs)


def _rss_gb():
    try:
        import psutil
        return psutil.Process(os.getpid()).memory_info().rss / 1e9
    except Exception:
        return float('nan')


class Logger:
    """Progress line per logging step: metrics, seconds per batch since construction, host / device memory
    (utils.py:31-51).  One device synchronisation per call, none in between."""

    def __init__(self, max_steps, start_step=0):
        self.max_steps, self.start_step = max_steps, start_step
        self.on_gpu = torch.cuda.is_available()
        if self.on_gpu:
            torch.cuda.synchronize()
        self.t0 = time.time()

    def __call__(self, step, metrics_dict):
        if self.on_gpu:
            torch.cuda.synchronize()
        per_batch = (time.time() - self.t0) / max(1, step + 1 - self.start_step)
        gpu = '%.2f' % (torch.cuda.memory_reserved() / 1e9) if self.on_gpu else 'nan'
        fields = '\t'.join('%s=%.4f' % (k, float(v)) for k, v in metrics_dict.items())
        log('batch=%04d/%04d \t %s \t %.4f (sec/batch), mem ram/gpu: %.2f/%s (G)'
            % (step, self.max_steps, fields, per_batch, _rss_gb(), gpu), flush=True)


def print_grads(model, verbose=True):
    """Gradient and parameter norms per tensor, sorted by gradient norm, plus the totals (utils.py:54-97).
    Returns (total gradient norm, total parameter norm)."""
    rows = []
    for name, p in model.named_parameters():
        if p.grad is not None:
            rows.append((name, tuple(p.shape), float(p.grad.detach().norm()), float(p.detach().norm())))
    rows.sort(key=lambda r: r[2])
    if verbose:
        print('\n ======== gradient and param norms (sorted by grads) ========')
        for i, (name, shape, g, w) in enumerate(rows):
            print('param #%03d: %35s: \t shape=%-20s, \t grad norm=%.3f, \t param norm=%.3f' % (i, name, str(shape), g, w))
    total_g = float(torch.tensor([r[2] for r in rows]).norm()) if rows else 0.0
    total_w = float(torch.tensor([r[3] for r in rows]).norm()) if rows else 0.0
    print('%d params with gradients, total grad norm=%.3f, total param norm=%.3f\n' % (len(rows), total_g, total_w))
    return total_g, total_w
