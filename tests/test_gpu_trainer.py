"""GPU tests of the Trainer (the GHN branch of trainer.py:238-440 on the flat-buffer optimizer) and of the counterpart
scripts' call sequences."""

import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import recipe
from util_parity import make_models

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _batch(names):
    import graph_nets
    from ghn3_amd import Graph, GraphBatch
    nets = [graph_nets.all_nets(graph_nets.local_bases())[n].to('cuda') for n in names]
    gb = GraphBatch([Graph(net, ve_cutoff=50) for net in nets], dense=True)
    gb.nets = nets                                           # (the loader attaches the target networks, deepnets1m.py:80)
    return gb


def test_trainer_update_log_save_resume(tmp_path):
    """Trainer.update / log / save / scheduler_step over three steps on image batches (cross-entropy of the predicted
    networks + the predicted-parameter regulariser), then resume from the checkpoint it wrote: same parameters, same
    optimizer state, next step index."""
    from ghn3_amd import Trainer
    cfg = dict(recipe.TINY_CFG)
    hip, _ = make_models(cfg, recipe.TINY_SEED)
    gen = torch.Generator().manual_seed(1)
    images = torch.randn(4, 3, 32, 32, generator=gen)
    targets = torch.tensor([1, 7, 3, 9])
    tr = Trainer(hip, 'adamw', {'lr': 1e-3, 'weight_decay': 1e-2}, 'cosine-warmup-steps2', n_batches=3, grad_clip=5,
                 device='cuda', log_interval=1, predparam_wd=3e-5, save_dir=str(tmp_path), epochs=4)
    before = hip._flat.detach().clone()
    losses = []
    for step in range(3):
        m = tr.update(images, targets, _batch(['resnet_tiny', 'mobile_se']))
        tr.log(step)
        tr.save(0, step, {'config': cfg}, save_freq=3)
        losses.append(m.avg()['loss'])
    assert all(np.isfinite(l) for l in losses) and tr.skipped_updates == 0
    assert not torch.equal(before, hip._flat)
    assert os.path.exists(tmp_path / 'checkpoint.pt')
    lr0 = tr.get_lr()
    tr.scheduler_step()
    assert tr.get_lr() != lr0
    # resume
    hip2, _ = make_models(cfg, recipe.TINY_SEED + 1)
    tr2 = Trainer(hip2, 'adamw', {'lr': 1e-3, 'weight_decay': 1e-2}, 'cosine-warmup-steps2', n_batches=3, grad_clip=5,
                  device='cuda', save_dir=str(tmp_path), epochs=4)
    assert torch.equal(hip2._flat, hip._flat)
    assert tr2._optimizer.steps == 3 and torch.equal(tr2._optimizer.exp_avg, tr._optimizer.exp_avg)
    assert (tr2.start_epoch, tr2.start_step) == (1, 0)       # the saved step was the last of its epoch


def test_trainer_skips_non_finite_steps():
    """A NaN loss (here: NaN images) must leave parameters and optimizer moments untouched -- the device-side guard that
    replaces the reference's NaN-loss all-gather / skip (trainer.py:240-257)."""
    from ghn3_amd import Trainer
    hip, _ = make_models(dict(recipe.TINY_CFG), recipe.TINY_SEED)
    tr = Trainer(hip, 'adamw', {'lr': 1e-3}, 'cosine', n_batches=10, grad_clip=5, device='cuda', epochs=2, log_interval=1)
    images = torch.randn(2, 3, 32, 32)
    targets = torch.tensor([1, 2])
    tr.update(images, targets, _batch(['resnet_tiny']))
    torch.cuda.synchronize()
    p, m = hip._flat.detach().clone(), tr._optimizer.exp_avg.clone()
    tr.update(images * float('nan'), targets, _batch(['resnet_tiny']))
    torch.cuda.synchronize()
    assert torch.equal(p, hip._flat) and torch.equal(m, tr._optimizer.exp_avg)
    first_loss = None
    tr.log(0)
    assert tr.skipped_updates == 1
    # the skipped step entered neither the sums nor the counts of the metrics, and did not advance the bias correction
    avg = tr.metrics.avg()
    assert np.isfinite(avg['loss']) and float(tr.metrics.cnt.item()) == 2.0          # (2 images x 1 network, one step)
    assert tr._optimizer.steps == 1
    tr.update(images, targets, _batch(['resnet_tiny']))      # and training goes on
    torch.cuda.synchronize()
    assert not torch.equal(p, hip._flat) and torch.isfinite(hip._flat).all()
    # a run of nothing but non-finite steps is an error, as in the reference (trainer.py:240-257 raises)
    tr.log(1)
    for _ in range(3):
        tr.update(images * float('nan'), targets, _batch(['resnet_tiny']))
    with pytest.raises(RuntimeError):
        tr.log(2)


def test_trainer_dynamic_loss_scale():
    """AMP: GradScaler's rule on the device-side overflow flags -- 65536 at start, halved after a skipped (overflow) step,
    doubled after amp_growth_interval clean steps, floored at amp_min_scale (trainer.py:346-379)."""
    from ghn3_amd import Trainer
    hip, _ = make_models(dict(recipe.TINY_CFG), recipe.TINY_SEED)
    tr = Trainer(hip, 'adamw', {'lr': 1e-4}, 'cosine', n_batches=100, grad_clip=5, device='cuda', epochs=2, amp=True,
                 amp_init_scale=4096, amp_min_scale=64, amp_growth_interval=4, amp_check_interval=2, log_interval=1000)
    import inspect
    assert inspect.signature(Trainer.__init__).parameters['amp_init_scale'].default == 65536     # GradScaler's default
    assert tr.loss_scale == 4096.0       # (a scale the tiny model's f16 gradients do not overflow at, so that only the
    #                                       injected overflow below is skipped)
    images = torch.randn(2, 3, 32, 32)
    targets = torch.tensor([1, 2])
    tr.update(images, targets, _batch(['resnet_tiny']))
    tr.update(images * float('inf'), targets, _batch(['resnet_tiny']))     # overflow -> skipped, scale halves at the check
    assert tr.loss_scale == 2048.0 and tr.skipped_updates == 1
    for _ in range(5):                   # (after an overflow every step is checked until a clean one, then every second)
        tr.update(images, targets, _batch(['resnet_tiny']))
    assert tr.loss_scale == 4096.0                                         # four clean steps seen by the host: doubled
    torch.cuda.synchronize()
    assert torch.isfinite(hip._flat).all()


def test_trainer_loss_scale_warm_down_from_the_default(tmp_path):
    """The normal fp16 warm-down (GradScaler: 65536 -> a scale the gradients fit, one halving per overflowing step) must not
    be mistaken for divergence: several consecutive overflow steps inside one check window are ONE back-off (they all ran at
    the same scale), after an overflow every step is checked, overflows at the floor scale are skipped steps, and the run only
    counts as diverged when the LOSS is not finite amp_check_interval steps in a row.  The scale travels in the checkpoint."""
    from ghn3_amd import Trainer
    hip, _ = make_models(dict(recipe.TINY_CFG), recipe.TINY_SEED)
    tr = Trainer(hip, 'adamw', {'lr': 1e-4}, 'cosine', n_batches=100, grad_clip=5, device='cuda', epochs=2, amp=True,
                 log_interval=10, save_dir=str(tmp_path))          # defaults: 65536 at start, floor 1024, check every 25
    assert tr.loss_scale == 65536.0 and tr.amp_min_scale == 1024.0
    images = torch.randn(2, 3, 32, 32)
    targets = torch.tensor([1, 2])
    bad = images * float('inf')
    # steps 1-4 overflow (as they would at 65536 .. 8192 for gradients that fit at 4096)
    for step in range(4):
        tr.update(bad, targets, _batch(['resnet_tiny']))
        tr.log(step)                     # (step 0 logs: first check -> 32768; from then on every step is checked)
    assert tr.loss_scale == 4096.0 and tr.skipped_updates == 4, (tr.loss_scale, tr.skipped_updates)
    for step in range(4, 7):
        tr.update(images, targets, _batch(['resnet_tiny']))
        tr.log(step)
    torch.cuda.synchronize()
    assert tr.loss_scale == 4096.0 and tr._optimizer.steps == 3 and torch.isfinite(hip._flat).all()
    # nine overflowing steps inside ONE window (no check in between): one halving, no exception
    for _ in range(9):
        tr.update(bad, targets, _batch(['resnet_tiny']))
    tr._sync_skips()
    assert tr.loss_scale == 2048.0
    tr.save(0, 299, {'config': dict(recipe.TINY_CFG)})
    hip2, _ = make_models(dict(recipe.TINY_CFG), recipe.TINY_SEED + 1)
    tr2 = Trainer(hip2, 'adamw', {'lr': 1e-4}, 'cosine', n_batches=1000, grad_clip=5, device='cuda', epochs=2, amp=True,
                  save_dir=str(tmp_path))
    assert tr2.loss_scale == 2048.0 and tr2._optimizer.steps == 3
    # skipped steps at the floor scale are ordinary skipped steps (trainer.py:364-379 clamps the scale and goes on) ...
    for _ in range(8):
        tr.update(bad, targets, _batch(['resnet_tiny']))
    assert tr.loss_scale == 1024.0
    # ... divergence is a non-finite LOSS (trainer.py:240-257) in amp_check_interval consecutive steps
    with pytest.raises(RuntimeError):
        for _ in range(tr.amp_check_interval):
            tr.update(bad, targets, _batch(['resnet_tiny']))
    assert tr.loss_scale == 1024.0


def test_eval_ghn_counterpart_script():
    """examples/eval_ghn.py: from-scratch GHN of the smallest released size, two architectures, prediction + norm print +
    evaluation forward (the eval_ghn.py:107-183 sequence)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'examples', 'eval_ghn.py'), '--ghn', 'ghn3tm8', '--arch',
                          'resnet_small', '--batch', '2'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert 'finite=True' in out.stdout and 'total norm' in out.stdout.lower(), out.stdout[-1000:]


def test_worker_precompiled_batch_gives_the_same_step():
    """A batch whose host compile ran where the batch was built (GraphBatch.precompile, then a pickle round trip as between
    a loader worker and the training process) is consumed by ghn(...) as is: same predicted parameters bit for bit, same
    gradients, and the networks receive their tensors."""
    import pickle
    from ghn3_amd import GraphBatch
    from ghn3_amd.deepnets1m import SampledNets
    hip, _ = make_models(dict(recipe.TINY_CFG), recipe.TINY_SEED)
    hip = hip.to('cuda').train()

    def batch(pre):
        src = SampledNets(seed=5, max_nodes=150)
        gb = GraphBatch([src[k] for k in range(2)], dense=True)
        gb._cat()
        gb.graphs = None
        if pre:
            gb.precompile(hip.program_config(), training=True, reduce_graph=True)
        return pickle.loads(pickle.dumps(gb))

    outs = []
    for pre in (False, True):
        gb = batch(pre)
        torch.manual_seed(11)
        hip.zero_grad(set_to_none=True)
        nets = hip(gb.nets, gb.to_device('cuda'), bn_track_running_stats=True, keep_grads=True, reduce_graph=True)
        assert getattr(gb, 'program', None) is None               # (taken, or never there)
        flat = hip._last_flat
        pred = torch.cat([flat[p['offset']:p['offset'] + p['numel']] for p in hip.last_plan.program.predicted])
        (pred * pred).sum().backward()                            # (the predicted tensors only: the gaps between them are never written)
        first = next(p for _, p in nets[0].named_parameters())
        assert torch.is_tensor(first) and first.is_cuda
        outs.append((pred.detach(), hip.embed.weight.grad.clone(), hip.decoder.conv[2].weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1:], outs[1][1:]):                    # (the backward accumulates some sums with atomics)
        assert float((a - b).norm() / b.norm()) < 1e-5
