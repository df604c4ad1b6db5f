"""
Trainer for the GHN branch of the reference's training loop, with the same call sequence and file formats
(/root/reference/ghn3/trainer.py:42-68 constructor, 238-411 ``update``, 413-432 ``save``, 434-440 ``log``; driven by
train_ghn_ddp.py:132-150):

    trainer = Trainer(ghn, opt='adamw', opt_args={'lr': 4e-4, 'weight_decay': 1e-2}, scheduler='cosine-warmup',
                      n_batches=len(queue), grad_clip=5, device=device, predparam_wd=3e-5, save_dir=..., epochs=...)
    for epoch in range(trainer.start_epoch, epochs):
        trainer.reset_metrics(epoch)
        for step, (images, targets, graphs) in enumerate(queue, start=trainer.start_step):
            trainer.update(images, targets, graphs)
            trainer.log(step)
            trainer.save(epoch, step, {'config': config})
        trainer.scheduler_step()

What is different underneath (MI355X-first, same results):
  * the GHN forward / backward are the HIP op programs of ghn3_amd.GHN3; the predicted-parameter regulariser
    (predparam_wd, trainer.py:97-98,288-294) is two streaming kernels on the flat output (GHN3.predicted_param_norm);
  * data parallel: no DistributedDataParallel wrapper -- rank 0's parameters are broadcast once (sync_parameters), the
    flat gradient buffer is averaged by FlatGradReducer inside loss.backward(), overlapped with the Graphormer backward;
  * clip_grad_norm_ + AdamW are two kernels over the flat buffers (FusedAdamW); a non-finite gradient norm (NaN / inf
    loss on ANY rank reaches every rank through the averaged gradient) skips the step on all ranks alike, on the
    device -- the reference's all-gather of the loss and host-side skip (trainer.py:240-257,397-409) without the syncs.
    Skipped steps enter neither the metric sums nor their counts; when EVERY step between two checks was skipped (and,
    under AMP, the loss scale is already at its floor) `update` / `log` raise like the reference does;
  * gradients travel in fp32 by default, as under DistributedDataParallel; grad_compress='bf16' is opt-in;
  * the training metrics (loss, top-1, top-5, regulariser) stay on the device and are averaged over the ranks with ONE
    all-reduce of a 4-vector per step instead of 4-5 scalar all-gathers; they are read at log time only.
Target networks run on stock torch ops with the predicted tensors; images may be None, in which case
the step trains on the regulariser alone (the benchmark's loss).
"""

import math
import os

import torch
import torch.nn.functional as F

from .ddp_utils import is_ddp, get_ddp_rank, sync_parameters, FlatGradReducer
from .optim import FusedAdamW, save_checkpoint
from .utils import log, Logger

import torch.distributed as dist


class _DeviceMeter:
    """Running sums kept on the device (no host read until .avg)."""

    def __init__(self, names, device):
        self.names = list(names)
        self.sum = torch.zeros(len(self.names), dtype=torch.float64, device=device)
        self.cnt = torch.zeros((), dtype=torch.float64, device=device)

    def update(self, vec, n):
        """n: sample count of the step, a Python number or a device scalar (0 for a skipped step: neither its values nor
        its count enter the averages)."""
        n = n.to(torch.float64) if torch.is_tensor(n) else float(n)
        self.sum += vec.to(torch.float64) * n
        self.cnt += n

    def avg(self):
        vals = (self.sum / torch.clamp(self.cnt, min=1.0)).tolist()
        return dict(zip(self.names, vals))


def _schedule(name, epochs, base_lr, scheduler_args):
    """Learning-rate multiplier per epoch for the schedules the reference offers (trainer.py:177-206)."""
    args = scheduler_args or {}
    if name.startswith('cosine-warmup'):
        def parse(key, default):
            p = name.find(key)
            if p < 0:
                return default
            end = name.find('-', p)
            return float(name[p + len(key):] if end < 0 else name[p + len(key):end])
        warm = int(parse('steps', 5))
        init = parse('init_lr', 1e-5) / base_lr

        def mult(e):
            if e < warm - 1:
                return init + (1.0 - init) * e / max(1, warm - 1)
            prog = float(e - warm) / float(max(1, epochs - warm))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        return mult
    if name == 'cosine':
        return lambda e: 0.5 * (1.0 + math.cos(math.pi * e / max(1, epochs)))
    if name == 'step':
        return lambda e: args.get('gamma', 0.1) ** (e // args['step_size'])
    if name == 'mstep':
        return lambda e: args.get('gamma', 0.1) ** sum(1 for m in args['milestones'] if e >= m)
    raise NotImplementedError(name)


class Trainer:
    def __init__(self, model, opt, opt_args, scheduler, n_batches, grad_clip=5, auxiliary=False, auxiliary_weight=0.4,
                 device='cuda', log_interval=100, label_smoothing=0, predparam_wd=0, scheduler_args=None, save_dir=None,
                 ckpt=None, epochs=None, verbose=False, amp=False, amp_min_scale=None, amp_growth_interval=2000,
                 grad_compress=None, amp_init_scale=65536.0, **unused):
        from .nn import GHN3
        if not isinstance(model, GHN3):
            raise NotImplementedError('ghn3_amd.Trainer drives the GHN branch (train_ghn_ddp.py); plain networks '
                                      '(train_ddp.py) train with stock PyTorch')
        if opt.lower() != 'adamw':
            raise NotImplementedError('the fused optimizer step implements AdamW (the GHN-3 recipe); got %s' % opt)
        assert 'lr' in opt_args, 'learning rate must be specified in opt_args'
        self.n_batches, self.grad_clip, self.device = n_batches, grad_clip, device
        self.auxiliary, self.auxiliary_weight = auxiliary, auxiliary_weight
        self.log_interval, self.label_smoothing = log_interval, label_smoothing
        self.predparam_wd, self.epochs, self.verbose = predparam_wd, epochs, verbose
        self.amp = amp
        # AMP (trainer.py:269,346-379): the target networks run under autocast; the GHN's own 16-bit mode is `compute`.
        # Dynamic loss scale with GradScaler's rule -- start at 65536, halve after an overflow step (which the optimizer's
        # non-finite-norm guard skips on the device), double after `amp_growth_interval` clean steps, never below
        # `amp_min_scale` (1024 in the GHN-3 recipe, trainer.py:364-379).  The overflow flags stay on the device; the host
        # copy of the scale is brought up to date every `amp_check_interval` steps and at log time (one 2-float read), so
        # the reaction to an overflow is delayed by at most that many (skipped) steps instead of a sync every step.
        self.amp_min_scale = float(amp_min_scale or 1024.0)
        self.amp_growth_interval = int(amp_growth_interval)
        self.amp_check_interval = int(unused.pop('amp_check_interval', 25))
        self.loss_scale = max(self.amp_min_scale, float(amp_init_scale)) if amp else 1.0
        self._clean_steps = 0
        self._amp_hot = False            # an overflow was seen at the last check: check every step until a clean one
        # grad_compress: None = fp32 on the wire (what DistributedDataParallel does, trainer.py:136); 'bf16' halves the xGMI
        # bytes at ~3 significant digits per gradient element (torch's bf16_compress_hook trade-off) -- opt-in
        self.ddp = is_ddp()
        self.rank = get_ddp_rank() if self.ddp else 0
        model.to(device)
        self.start_epoch = self.start_step = 0
        self.checkpoint_path = os.path.join(save_dir, 'checkpoint.pt') if save_dir else None
        state = None
        if self.checkpoint_path is not None and os.path.exists(self.checkpoint_path):
            ckpt = self.checkpoint_path
            log('Found existing checkpoint %s: resuming.' % ckpt)
        if ckpt is not None:
            if self.ddp:
                dist.barrier()
            state = torch.load(ckpt, map_location='cpu')
            model.load_state_dict(state['state_dict'])
            self.start_epoch, self.start_step = int(state.get('epoch', 0)), int(state.get('step', 0))
            if amp and 'amp_loss_scale' in state:        # (a resumed run does not repeat the warm-down from 65536)
                self.loss_scale = max(self.amp_min_scale, float(state['amp_loss_scale']))
                self._clean_steps = int(state.get('amp_clean_steps', 0))
        self._model = model
        if self.ddp:
            sync_parameters(model)                       # what DistributedDataParallel's constructor does
            model.grad_reducer = FlatGradReducer(compress=grad_compress)
        self.base_lr = float(opt_args['lr'])
        kw = {k: v for k, v in opt_args.items() if k in ('betas', 'eps', 'weight_decay')}
        self._optimizer = FusedAdamW(model, lr=self.base_lr, max_grad_norm=float(grad_clip or 0.0), **kw)
        self._lr_mult = _schedule(scheduler, epochs or 1, self.base_lr, scheduler_args)
        self._epoch = self.start_epoch
        if state is not None and 'optimizer' in state:
            self._optimizer.load_state_dict(state['optimizer'])
        self._optimizer.lr = self.base_lr * self._lr_mult(self._epoch)
        self.skipped_updates = 0
        self.reset_metrics(self.start_epoch)
        if state is not None:
            if self.start_step >= self.n_batches - 1:
                self.start_step, self.start_epoch = 0, self.start_epoch + 1
            else:
                self.start_step += 1

    # ------------------------------------------------------------------ schedule / bookkeeping
    def reset_metrics(self, epoch):
        self._step = 0
        if epoch > self.start_epoch:
            self.start_step = 0
        names = ['loss', 'top1', 'top5'] + (['loss_predwd'] if self.predparam_wd > 0 else [])
        self.metrics = _DeviceMeter(names, self.device)
        self.logger = Logger(self.n_batches, start_step=self.start_step)

    def get_lr(self):
        return self._optimizer.lr

    def scheduler_step(self):
        self._epoch += 1
        self._optimizer.lr = self.base_lr * self._lr_mult(self._epoch)

    # ------------------------------------------------------------------ one training step
    def update(self, images, targets, graphs=None):
        ghn = self._model
        if not ghn.training:
            ghn.train()
        ghn.zero_grad(set_to_none=True)                 # (p.grad are views of the previous flat gradient buffer)
        dev = self.device
        models = graphs.nets if hasattr(graphs, 'nets') and len(graphs.nets) > 0 else None
        if models is None:
            raise ValueError('graphs.nets is empty: the batch must carry the target networks (deepnets1m.py:80)')
        models = ghn(models, graphs.to_device(dev), bn_track_running_stats=True, keep_grads=True, reduce_graph=True)
        loss = torch.zeros((), device=dev)
        stats = torch.zeros(4, device=dev)               # loss, top1, top5, loss_predwd
        if images is not None:
            images = images.to(dev, non_blocking=True)
            targets = targets.to(dev, non_blocking=True)
            logits = []
            with torch.autocast('cuda', dtype=torch.float16, enabled=self.amp):
                for m in models:
                    out = m(images)
                    y = out[0] if isinstance(out, tuple) else out
                    loss = loss + F.cross_entropy(y.float(), targets, label_smoothing=self.label_smoothing)
                    if self.auxiliary and isinstance(out, tuple):
                        loss = loss + self.auxiliary_weight * F.cross_entropy(out[1].float(), targets,
                                                                              label_smoothing=self.label_smoothing)
                    logits.append(y.detach().float())
            loss = loss / len(models)
            with torch.no_grad():
                lg = torch.stack(logits)                                  # models x batch x classes
                k = min(5, lg.shape[-1])
                top = lg.topk(k, dim=-1).indices
                hit = top == targets.view(1, -1, 1)
                stats[1] = 100.0 * hit[..., :1].any(-1).float().mean()
                stats[2] = 100.0 * hit.any(-1).float().mean()
        if self.predparam_wd > 0:
            wd = self.predparam_wd * ghn.predicted_param_norm() / len(models)
            loss = loss + wd
            stats[3] = wd.detach()
        stats[0] = loss.detach()
        (loss * self.loss_scale).backward()              # backward program (+ overlapped gradient average, N > 1)
        # clip + AdamW on the flat buffers; a non-finite norm (NaN loss on any rank) skips the update everywhere
        # (overlap: the decoder parameters -- 93 % of the bytes -- are updated on the side stream while the next step's
        # Graphormer already runs; the next forward joins in front of its decoders, save() waits explicitly)
        gnorm = self._optimizer.step(ghn.last_plan.gflat, grad_scale=self.loss_scale, plan=ghn.last_plan,
                                     local_grads=not self.ddp, overlap=True)
        with torch.no_grad():
            bad = ~torch.isfinite(gnorm) if gnorm is not None else ~torch.isfinite(stats[0])
            vec = torch.cat([stats, bad.float().view(1)])
            if self.ddp:                                 # ONE small collective for all metrics
                dist.all_reduce(vec)
                vec[:4] /= dist.get_world_size()
            self._skipped = getattr(self, '_skipped', torch.zeros((), device=dev)) + (vec[4] > 0).float()
            # (steps whose AVERAGED loss is not finite, counted apart from overflowed gradients: only they can mean divergence)
            self._bad_loss = getattr(self, '_bad_loss', torch.zeros((), device=dev)) + (~torch.isfinite(vec[0])).float()
            n = int(targets.numel()) * len(models) if images is not None else len(models)
            good = (vec[4] == 0).float()                 # (skipped steps enter neither the sums nor the counts)
            self.metrics.update(torch.nan_to_num(vec[:len(self.metrics.names)]) * good, good * n)
        self._step += 1
        self._steps_since_check = getattr(self, '_steps_since_check', 0) + 1
        if self.amp and (self._amp_hot or self._steps_since_check >= self.amp_check_interval):
            self._sync_skips()
        return self.metrics

    def _sync_skips(self):
        """One host read of the device-side skip counter: loss-scale back-off / growth (GradScaler's rule), the optimizer's
        bias-correction step count (a skipped update must not advance it), and the reference's refusal to train on
        non-finite losses (trainer.py:240-257 raises).

        The host copy of the scale only changes here, so every step of the window since the last check ran at ONE scale:
        however many of them overflowed, that is one back-off (GradScaler halves once per overflowing step, each at a new
        scale).  After an overflow the check runs every step until a clean one (`_amp_hot`), so the warm-down from 65536
        takes as many steps as GradScaler's.

        Divergence is decided on the LOSS, as the reference does (trainer.py:240-257: a NaN loss raises / skips the batch;
        an fp16 overflow of the gradients at the floor scale is an ordinary skipped step there, trainer.py:364-379 -- its
        comment says the scale routinely wants to fall below 1024): the run is declared diverged when the rank-averaged loss
        was not finite in EVERY step of a window (accumulated over the one-step windows of the hot mode until
        `amp_check_interval` consecutive steps are reached), or -- without AMP, where no scale can be blamed -- when a whole
        window of updates was skipped."""
        both = torch.stack([getattr(self, '_skipped', torch.zeros(())).float().cpu(),
                            getattr(self, '_bad_loss', torch.zeros(())).float().cpu()]).tolist()
        total, bad_total = int(both[0]), int(both[1])
        new = total - self.skipped_updates
        new_bad = bad_total - getattr(self, '_bad_loss_seen', 0)
        steps = getattr(self, '_steps_since_check', 0)
        if steps == 0 and new == 0:
            return                                       # nothing ran since the last check
        self._steps_since_check = 0
        self.skipped_updates = total
        self._bad_loss_seen = bad_total
        # consecutive steps with a non-finite loss, across windows (a window with one good step resets the run)
        self._bad_run = (getattr(self, '_bad_run', 0) + new_bad) if (steps > 0 and new_bad >= steps) else 0
        if steps > 0 and self._bad_run >= (max(1, self.amp_check_interval) if self.amp else 1):
            raise RuntimeError('the loss was not finite in the last %d steps (%d skipped updates in total): the GHN has '
                               'diverged; restart from the saved checkpoint (--ckpt)' % (self._bad_run, total))
        if new > 0:
            self._optimizer.steps = max(0, self._optimizer.steps - new)
            window_scale = self.loss_scale               # the scale every step of this window used
            if self.amp:
                self.loss_scale = max(self.amp_min_scale, self.loss_scale * 0.5)
                self._clean_steps = 0
                self._amp_hot = True
            if steps > 0 and new >= steps and not self.amp:
                raise RuntimeError('the gradient norm was not finite in all of the last %d steps (%d skipped updates in '
                                   'total): the GHN has diverged' % (steps, total))
            del window_scale                             # (under AMP an overflow at the floor scale is a skipped step)
        else:
            self._amp_hot = False
            if self.amp:
                self._clean_steps += steps
                if self._clean_steps >= self.amp_growth_interval:
                    self.loss_scale *= 2.0
                    self._clean_steps = 0

    # ------------------------------------------------------------------ checkpoints / logging
    def save(self, epoch, step, config, save_freq=300, interm_epoch=5):
        if not (((step + 1) % save_freq == 0) or step == self.n_batches - 1):
            return
        # On EVERY rank (the save condition does not depend on the rank): the check moves the loss scale, the check windows and
        # the optimizer's bias-correction count -- host state all replicas must change at the same step -- and may raise.
        self._sync_skips()          # (the optimizer's bias-correction count must not include skipped steps when it is saved)
        if self.rank != 0:
            return
        self._optimizer.wait()      # (an overlapped optimizer step may still be writing the decoder parameters / moments)
        if self.amp:
            config = dict(config or {}, amp_loss_scale=self.loss_scale, amp_clean_steps=self._clean_steps)
        save_checkpoint(self.checkpoint_path, self._model, self._optimizer, epoch, step, config)
        log('\nsaved the checkpoint to {} at epoch={}, step={}'.format(self.checkpoint_path, epoch, step))
        if (epoch + 1) % interm_epoch == 0 or epoch == 0:
            path = self.checkpoint_path.replace('.pt', '_epoch%d.pt' % (epoch + 1))
            save_checkpoint(path, self._model, self._optimizer, epoch, step, config)
            log('saved the intermediate checkpoint to {}'.format(path))

    def log(self, step=None):
        step_ = self._step if step is None else (step + 1)
        if step_ % self.log_interval == 0 or step_ >= self.n_batches - 1 or step_ == 1:
            metrics = self.metrics.avg()                 # (the only host read of the step statistics)
            self._sync_skips()
            if self.skipped_updates:
                metrics['skipped'] = self.skipped_updates
            if self.amp:
                metrics['amp_scale'] = self.loss_scale
            self.logger(step_, metrics)
