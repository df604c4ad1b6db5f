#!/usr/bin/env python3
"""
One-process-per-GPU training loop over synthetic DeepNets-1M-like graphs through the PUBLIC API -- the call sequence
of /root/reference/train_ghn_ddp.py:87-150 with the image branch replaced by the predicted-parameter-norm loss
(trainer.py:97-98,288-294):

    ghn = GHN3(**config).to(device)              # train_ghn_ddp.py:87-91
    for step:  nets, graphs = next(loader)       # one fresh architecture (graph) per GPU and step
               nets = ghn(nets, graphs, keep_grads=True)        # GHN3.forward, autograd-connected predictions
               loss = sum(||p||_F) ; loss.backward()            # backward program through torch.autograd
               (gradient all-reduce inside backward, N > 1) ; clip + AdamW   # FlatGradReducer / FusedAdamW

    python examples/train_synthetic.py [--steps 20] [--model ghn3xlm16] [--nodes 256]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ghn3_amd import GHN3, FusedAdamW, setup_ddp, clean_ddp, avg_ddp_metric          # noqa: E402
from ghn3_amd.ddp_utils import FlatGradReducer, sync_parameters                                        # noqa: E402
from ghn3_amd.synthetic import synthetic_batch                                         # noqa: E402

MODELS = {'ghn3tm8': (64, 3, 8), 'ghn3sm8': (128, 5, 16), 'ghn3lm8': (256, 12, 16), 'ghn3xlm16': (384, 24, 16)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--model', default='ghn3xlm16')
    ap.add_argument('--nodes', type=int, default=256)
    ap.add_argument('--compute', default='f16')
    args = ap.parse_args()
    ddp = setup_ddp()
    hid, layers, heads = MODELS[args.model]
    torch.manual_seed(0)
    ghn = GHN3(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers, weight_norm=True,
               ve=True, layernorm=True, compute=args.compute).to(ddp.device)
    ghn.train()
    if ddp.ddp:
        sync_parameters(ghn)            # rank 0's weights everywhere (DistributedDataParallel's initial broadcast)
        # mean all-reduce of the flat gradient buffer inside loss.backward(), bf16 on the wire, overlapped with the
        # Graphormer backward (the reference wraps the GHN in DistributedDataParallel, trainer.py:136)
        ghn.grad_reducer = FlatGradReducer(compress='bf16')
    opt = FusedAdamW(ghn, lr=4e-4, weight_decay=1e-2, max_grad_norm=5.0)
    t_host = t0 = None
    for step in range(args.steps):
        if step == 2:
            torch.cuda.synchronize()
            t0, t_host = time.perf_counter(), 0.0
        h0 = time.perf_counter()
        graphs, nets = synthetic_batch([args.nodes], args.nodes * 1000 + step * ddp.world_size + ddp.rank)
        ghn.zero_grad(set_to_none=True)                # (p.grad are views of the previous step's flat gradient buffer)
        nets = ghn(nets, graphs, keep_grads=True)
        if t_host is not None:
            t_host += time.perf_counter() - h0
        # predparam_wd-style loss on the flat output buffer (two kernels); the equivalent per-tensor form is
        #   sum(torch.norm(p, p='fro') for net in nets for p in net.parameters())   (~1000 ATen launches per step)
        loss = ghn.predicted_param_norm()
        loss.backward()
        gnorm = opt.step(ghn.last_plan.gflat)          # (already averaged over the ranks by ghn.grad_reducer)
        if step % 5 == 0 or step == args.steps - 1:
            m = avg_ddp_metric(loss.detach())
            if ddp.rank == 0:
                print('step %3d  loss %.4f  grad-norm %.4f' % (step, m.item(), float(gnorm)), flush=True)
    torch.cuda.synchronize()
    if ddp.rank == 0 and t0 is not None:
        n = args.steps - 2
        dt = time.perf_counter() - t0
        print('%.1f ms per step over %d steps (of which %.1f ms host: graph generation + compile + enqueue)'
              % (1e3 * dt / n, n, 1e3 * t_host / n))
    clean_ddp()


if __name__ == '__main__':
    main()
