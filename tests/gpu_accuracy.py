#!/usr/bin/env python3
"""Prints measured forward / gradient errors of the compute modes vs the CPU oracle (GPU box only; documentation
numbers for DESIGN.md, not a test).   python tests/gpu_accuracy.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from util_parity import rel_l2, make_models, synthetic_case, predicted_dict_hip   # noqa: E402

CFGS = {'ghn3tm8': dict(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3, weight_norm=True,
                        ve=True, layernorm=True),
        'ghn3sm8': dict(max_shape=(128, 128, 16, 16), num_classes=1000, hid=128, heads=16, layers=5,
                        weight_norm=True, ve=True, layernorm=True)}

for name, cfg in CFGS.items():
    for compute, kw in (('f32', {}), ('f16', {}), ('f16', {'compute_bwd': 'f16'}), ('bf16', {})):
        from ghn3_amd import GHN3
        hip, oracle = make_models(cfg, 7, compute=compute)
        for k, v in kw.items():
            setattr(hip, k, v)
        nets_h, gb_h, nets_o, gb_o = synthetic_case([64], 6400)
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
        eg = 0.0
        for k, p in hip.named_parameters():
            go = po[k].grad
            if float(go.norm()) > 1e-6:
                eg = max(eg, float((p.grad.cpu().double() - go.double()).norm()) / float(go.norm()))
        print('%-8s compute=%-4s %-22s worst forward rel-L2 %.2e   worst gradient rel-L2 %.2e'
              % (name, compute, str(kw), ef, eg), flush=True)
