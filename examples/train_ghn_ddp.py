#!/usr/bin/env python3
"""
Counterpart of /root/reference/train_ghn_ddp.py:36-150 without the DeepNets-1M / image files (none are available
offline): the same call sequence -- GHN3(**config), an architecture queue of GraphBatch objects that carry light
target networks, ``Trainer.update(images, targets, graphs=...)`` / ``log`` / ``save`` / ``scheduler_step`` -- on
architectures drawn from the DeepNets-1M search space (``ghn3_amd.deepnets1m.SampledNets``) and seeded random images.

    python examples/train_ghn_ddp.py [--model ghn3tm8] [--meta-batch-size 8] [--steps 30] [--imagenet] [--amp]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_ghn_ddp.py ...

The architecture queue runs in worker processes (graph construction is a CPU forward + autograd walk, ~0.1 s per
network: the reference reads precomputed graphs from hdf5 instead) and is a pure function of (seed, step, rank).
Prints the step time split: waiting for architectures, GHN forward (parameter prediction), target networks on the
images + loss, backward + optimizer.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MODELS = {'ghn3tm8': (64, 3, 8), 'ghn3sm8': (128, 5, 16), 'ghn3lm8': (256, 12, 16), 'ghn3xlm16': (384, 24, 16)}


def _draw(task):
    """Worker: one rank's share of one meta-batch (host only)."""
    from ghn3_amd.deepnets1m import SampledNets
    from ghn3_amd.graph import GraphBatch
    step, rank, per_rank, meta, kw, pcfg = task
    nets = SampledNets(**kw)
    base = step * meta + rank * per_rank
    gb = GraphBatch([nets[base + k] for k in range(per_rank)], dense=True)
    gb._cat()
    gb.graphs = None                 # (the per-graph copies of what _cat stacked: half of the pickle)
    if pcfg is not None:
        # the GHN's host compile for this batch (index tables, tile descriptors, the op lists: 25-30 ms of Python per step)
        # happens here too, off the training process's critical path -- Trainer.update's ghn(...) call picks it up
        gb.precompile(pcfg, training=True, predict_class_layers=True, reduce_graph=True)
    return gb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='ghn3tm8')
    ap.add_argument('--meta-batch-size', type=int, default=8)
    ap.add_argument('--batch-size', type=int, default=64, help='images per step')
    ap.add_argument('--steps', type=int, default=30, help='steps per epoch')
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--imagenet', action='store_true', help='224 x 224 inputs, 1000 classes (default: 32 x 32, 10)')
    ap.add_argument('--amp', action='store_true')
    ap.add_argument('--compute', default='f16')
    ap.add_argument('--lr', type=float, default=4e-4)
    ap.add_argument('--wd', type=float, default=1e-2)
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--max-nodes', type=int, default=400)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--save', default=None)
    ap.add_argument('--native', action='store_true',
                    help='run the target networks on ATen native convolution / batch-norm kernels instead of MIOpen')
    args = ap.parse_args()
    # Every step brings NEW architectures.  On the stock layers (GHN3_NATIVE_OPS=0, what the reference runs) that means convolution
    # configurations MIOpen may not have seen: its default (hybrid) find mode benchmarks candidate kernels on first sight of a
    # configuration -- measured at meta-batch 8, 64 images of 32 x 32, one MI355X: ~1.0 s per step on a first pass over a set of
    # architectures, 145 ms per step once their configurations are in the find-db.  With the target networks on the fused HIP
    # layers of ghn3_amd.target_ops (the default: every convolution / BatchNorm / squeeze-excitation / pooling layer of this
    # search space) there is no find step: 76-85 ms per step from a cold box, the same with --amp (the fused layers run in fp32
    # under autocast; --amp on the stock layers: 0.88 s per step cold).  profiles/r06y_train_loop_final.txt.
    # --native (for the stock path): ATen's native kernels instead of MIOpen (im2col + GEMM per sample, no build step), 1.03 s.

    import multiprocessing as mp
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(var, '2')
    pool = mp.get_context('spawn').Pool(args.workers)        # before this process touches the GPU

    from ghn3_amd import GHN3, Trainer, setup_ddp, clean_ddp, log
    ddp = setup_ddp()
    torch.backends.cudnn.enabled = not args.native
    hid, layers, heads = MODELS[args.model]
    num_classes = 1000 if args.imagenet else 10
    s = 16 if args.imagenet else 11
    config = {'max_shape': (hid, hid, s, s), 'num_classes': num_classes, 'weight_norm': True, 've': True,
              'layernorm': True, 'hid': hid, 'layers': layers, 'heads': heads}
    torch.manual_seed(args.seed)
    ghn = GHN3(**config, compute=args.compute)

    world = ddp.world_size if ddp.ddp else 1
    per_rank = args.meta_batch_size // world
    kw = dict(large_images=args.imagenet, seed=args.seed, max_nodes=args.max_nodes)
    total = args.steps * args.epochs

    trainer = Trainer(ghn, opt='adamw', opt_args={'lr': args.lr, 'weight_decay': args.wd}, scheduler='cosine',
                      n_batches=args.steps, grad_clip=5, device=ddp.device, log_interval=10, amp=args.amp,
                      amp_min_scale=1024, predparam_wd=3e-5, label_smoothing=0.1 if args.imagenet else 0.0,
                      save_dir=args.save, epochs=args.epochs, verbose=ddp.rank == 0)
    # the architecture stream is a pure function of (seed, step): a resumed run continues at the checkpointed position
    # instead of replaying the architectures of step 0 onwards
    first = trainer.start_epoch * args.steps + trainer.start_step
    pcfg = ghn.program_config() if os.environ.get('GHN3_WORKER_COMPILE', '1') != '0' else None
    queue = pool.imap(_draw, [(st, ddp.rank, per_rank, args.meta_batch_size, kw, pcfg) for st in range(first, total + 1)])
    log('training %s (%d parameters) on %d sampled architectures per step, %d x %d images'
        % (args.model, sum(p.numel() for p in ghn.parameters()), args.meta_batch_size, args.batch_size,
           224 if args.imagenet else 32))
    gen = torch.Generator().manual_seed(args.seed + 1)
    side = 224 if args.imagenet else 32
    images = torch.randn(args.batch_size, 3, side, side, generator=gen).to(ddp.device)
    targets = torch.randint(0, num_classes, (args.batch_size,), generator=gen).to(ddp.device)

    waits, t0, n_timed = 0.0, None, 0
    prof = None
    if os.environ.get('GHN3_CPROFILE'):                   # host-side profile of the loop (where the step's Python time goes)
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for epoch in range(trainer.start_epoch, args.epochs):
        trainer.reset_metrics(epoch)
        for step in range(trainer.start_step, args.steps):
            if epoch == 0 and step == 3:
                torch.cuda.synchronize()
                t0, waits, n_timed = time.perf_counter(), 0.0, 0
            w0 = time.perf_counter()
            graphs = next(queue)
            waits += time.perf_counter() - w0
            trainer.update(images, targets, graphs=graphs)
            trainer.log(step)
            n_timed += 1
            if args.save:
                trainer.save(epoch, step, {'config': config})
        trainer.scheduler_step()
    torch.cuda.synchronize()
    if prof is not None:
        import pstats
        prof.disable()
        with open(os.environ['GHN3_CPROFILE'], 'w') as fh:
            st = pstats.Stats(prof, stream=fh)
            st.sort_stats('tottime').print_stats(60)
            st.sort_stats('cumulative').print_stats(50)
    if ddp.rank == 0 and t0 is not None and n_timed:
        dt = time.perf_counter() - t0
        log('%.1f ms per step over %d steps (%.1f ms of it waiting for the architecture queue); %d updates skipped'
            % (1e3 * dt / n_timed, n_timed, 1e3 * waits / n_timed, trainer.skipped_updates))
    pool.terminate()
    if ddp.ddp:
        clean_ddp()


if __name__ == '__main__':
    main()
