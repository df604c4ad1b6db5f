#!/usr/bin/env python3
"""Prints measured forward / gradient errors of the compute modes vs the CPU oracle (GPU box only; documentation
numbers for DESIGN.md, not a test).   python tools/diag/gpu_accuracy.py"""
import os
import sys

import torch

import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
from util_parity import rel_l2, make_models, synthetic_case, predicted_dict_hip   # noqa: E402

CFGS = {'ghn3tm8': dict(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3, weight_norm=True,
                        ve=True, layernorm=True),
        'ghn3sm8': dict(max_shape=(128, 128, 16, 16), num_classes=1000, hid=128, heads=16, layers=5,
                        weight_norm=True, ve=True, layernorm=True)}

NODES, SEED = 64, 6400
if '--xl' in sys.argv[1:]:
    # the headline size on a small graph (the oracle needs ~10 s per step at 32 nodes); prints the worst parameters
    CFGS = {'ghn3xlm16': dict(max_shape=(384, 384, 16, 16), num_classes=1000, hid=384, heads=16, layers=24,
                              weight_norm=True, ve=True, layernorm=True)}
    NODES, SEED = 32, 32000
    torch.set_num_threads(min(16, os.cpu_count() or 1))

for name, cfg in CFGS.items():
    for compute, kw in ((('f32', {}), ('f16', {})) if '--xl' in sys.argv[1:] else
                        (('f32', {}), ('f16', {}), ('f16', {'compute_bwd': 'f16'}), ('bf16', {}))):
        from ghn3_amd import GHN3
        hip, oracle = make_models(cfg, 7, compute=compute)
        for k, v in kw.items():
            setattr(hip, k, v)
        nets_h, gb_h, nets_o, gb_o = synthetic_case([NODES], SEED)
        hip.train()
        nets_h = hip(nets_h, gb_h, keep_grads=True)
        loss = sum(torch.norm(p, p='fro') for net in nets_h for p in net.parameters())
        loss.backward()
        torch.cuda.synchronize()
        oracle.train()
        nets_o, pred_o = oracle(nets_o, gb_o, keep_grads=True)
        loss_o = sum(torch.norm(t, p='fro') for (_, _, _, t) in pred_o)
        loss_o.backward()
        pred_h = predicted_dict_hip(hip.last_plan, hip.last_plan.out)
        ef = max(rel_l2(pred_h[k].detach().cpu(), t.detach()) for k, (_, _, _, t) in enumerate(pred_o))
        po = dict(oracle.named_parameters())
        eg, per = 0.0, []
        for k, p in hip.named_parameters():
            go = po[k].grad
            if float(go.norm()) > 1e-6:
                e = float((p.grad.cpu().double() - go.double()).norm()) / float(go.norm())
                per.append((e, k, float(go.norm())))
                eg = max(eg, e)
        if '--xl' in sys.argv[1:]:
            for e, k, nrm in sorted(per, reverse=True)[:6]:
                print('    %-40s rel-L2 %.2e  |grad| %.2e' % (k, e, nrm), flush=True)
        print('%-8s compute=%-4s %-22s worst forward rel-L2 %.2e   worst gradient rel-L2 %.2e'
              % (name, compute, str(kw), ef, eg), flush=True)
