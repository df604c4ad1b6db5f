#!/usr/bin/env python3
"""Diagnostic sweep on the GPU box: prints per-stage errors instead of stopping at the first failure.
    python tools/diag/gpu_diag.py [gemm] [tiny] [grads] [synth]
"""
import os
import sys
import traceback

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)

import recipe                                                     # noqa: E402
from util_parity import (rel_l2, make_models, tiny_case, synthetic_case, oracle_intermediates, ws_tensor,
                         predicted_dict_hip)                      # noqa: E402


def section(name):
    print('\n==== %s ====' % name, flush=True)


def diag_gemm():
    from ghn3_amd import _lib as L
    from gemm_cases import CASES, run_gemm_case
    ctx = L.context(0)
    for ctype, nm in ((0, 'f32'), (1, 'f16'), (2, 'bf16')):
        worst = 0
        for k, case in enumerate(CASES):
            try:
                got, exp, extra = run_gemm_case(ctx, ctype=ctype, seed=k, **case)
                e = rel_l2(got, exp)
                e2 = rel_l2(extra[0], extra[1]) if extra is not None else 0.0
                worst = max(worst, e, e2)
                flag = '' if e < (2e-6, 2e-3, 1.5e-2)[ctype] else '   <-- FAIL'
                print('%s case %2d %s err %.3e aux %.3e%s' % (nm, k, case, e, e2, flag), flush=True)
            except Exception:
                print(nm, 'case', k, case, 'EXCEPTION')
                traceback.print_exc()
        print('%s worst %.3e' % (nm, worst))


def diag_tiny(case='b2'):
    hip, oracle = make_models(recipe.TINY_CFG, recipe.TINY_SEED)
    nets_h, gb_h, nets_o, gb_o = tiny_case(case)
    plan = hip.compile(nets_h, gb_h, training=True)
    with torch.no_grad():
        flat = hip._run_forward(plan)
    torch.cuda.synchronize()
    prog = plan.program
    inter = oracle_intermediates(oracle, nets_o, gb_o)
    B, N, C, H = prog.B, prog.N, prog.C, prog.H
    print('x0   ', rel_l2(ws_tensor(plan, 'x0', (B, N, C)).cpu(), inter['x0']))
    print('bias ', rel_l2(ws_tensor(plan, 'bias', (B, H, N, N)).cpu(), inter['bias']))
    for l in range(1, prog.Lyr + 1):
        print('x%d   ' % l, rel_l2(ws_tensor(plan, 'x%d' % l, (B, N, C)).cpu(), inter['x%d' % l]))
    print('xe   ', rel_l2(hip.embeddings(plan).cpu(), inter['xe']))
    with torch.no_grad():
        _, pred_o = oracle(nets_o, gb_o, assign=False)
    pred_h = predicted_dict_hip(plan, flat)
    for k, (ind, attr, m, t) in enumerate(pred_o):
        a, b = pred_h[k].cpu(), t
        if b.dim() == 3:
            a, b = a[:, 1:], b[:, 1:]
        print('pred %2d node %3d %-14s %-18s err %.3e' % (k, ind, attr, tuple(t.shape), rel_l2(a, b)))


def diag_grads(case='b2', cfg=None, seed=None, synth=None, compute='f32'):
    cfg = cfg or recipe.TINY_CFG
    hip, oracle = make_models(cfg, seed or recipe.TINY_SEED, compute=compute)
    if synth is None:
        nets_h, gb_h, nets_o, gb_o = tiny_case(case)
    else:
        nets_h, gb_h, nets_o, gb_o = synthetic_case(*synth)
    hip.train()
    nets_h2 = hip(nets_h, gb_h, keep_grads=True)
    plan = hip.last_plan
    pred_h = predicted_dict_hip(plan, hip._last_flat)
    loss = 0
    for k in pred_h:
        q = pred_h[k]
        q = q[:, 1:] if q.dim() == 3 else q
        loss = loss + torch.norm(q, p='fro')
    loss.backward()
    torch.cuda.synchronize()
    oracle.train()
    _, pred_o = oracle(nets_o, gb_o, keep_grads=True)
    loss_o = 0
    for (ind, attr, m, t) in pred_o:
        q = t[:, 1:] if t.dim() == 3 else t
        loss_o = loss_o + torch.norm(q, p='fro')
    loss_o.backward()
    print('loss hip %.6f oracle %.6f' % (loss.item(), loss_o.item()))
    worst_f = 0
    for k, (ind, attr, m, t) in enumerate(pred_o):
        a, b = pred_h[k].detach().cpu(), t.detach()
        if b.dim() == 3:
            a, b = a[:, 1:], b[:, 1:]
        worst_f = max(worst_f, rel_l2(a, b))
    print('worst forward rel-L2 %.3e' % worst_f)
    po = dict(oracle.named_parameters())
    for k, p in hip.named_parameters():
        go = po[k].grad
        if p.grad is None:
            print('grad %-50s MISSING' % k)
            continue
        e = rel_l2(p.grad.cpu(), go)
        print('grad %-50s |g| %.4e err %.3e%s' % (k, float(go.norm()), e, '' if e < 3e-4 else '   <-- CHECK'))


if __name__ == '__main__':
    what = sys.argv[1:] or ['gemm', 'tiny', 'grads']
    torch.manual_seed(0)
    for w in what:
        try:
            if w == 'gemm':
                section('gemm unit cases')
                diag_gemm()
            elif w == 'tiny':
                for c in ('b1', 'b2'):
                    section('tiny forward ' + c)
                    diag_tiny(c)
            elif w == 'grads':
                section('tiny grads b2')
                diag_grads('b2')
            elif w == 'synth':
                section('ghn3tm8 synthetic N=48 f32')
                cfg = dict(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3,
                           weight_norm=True, ve=True, layernorm=True)
                diag_grads(cfg=cfg, seed=7, synth=([48], 4800))
                section('ghn3tm8 synthetic N=48 f16')
                diag_grads(cfg=cfg, seed=7, synth=([48], 4800), compute='f16')
        except Exception:
            traceback.print_exc()
