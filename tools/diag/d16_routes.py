#!/usr/bin/env python3
"""Direct 16-bit tile route against the fp32 route on the bench workload (GPU box): the 16-bit operand copies dth (straight)
and dthT_b (transposed bands) of both routes, family by family / band by band."""
import sys
import numpy as np
import torch
import _paths  # noqa: F401
from ghn3_amd import GHN3, _lib as L
from ghn3_amd.synthetic import synthetic_batch
import bench


def step(hip, plan, dout, fused):
    prog = plan.program
    ctx = L.context(0)
    stream = torch.cuda.current_stream().cuda_stream
    hip._run_forward(plan)
    if fused:
        ctx.run(prog.norm_fin_ops(), prog.problems, plan.bufs, stream)
        hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
    else:
        f_norm, b_norm = prog.norm_ops(1.0)
        hip._fill_bufs(plan, out=plan.out, dout=dout)
        ctx.run(f_norm, prog.problems, plan.bufs, stream)
        ctx.run(b_norm, prog.problems, plan.bufs, stream)
        hip._run_backward(plan, dout)
    torch.cuda.synchronize()


def main():
    model = sys.argv[1] if len(sys.argv) > 1 else 'ghn3xlm16'
    nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    torch.manual_seed(0)
    hip = GHN3(**bench.model_cfg(model), compute='f16').to('cuda')
    hip.train()
    gb, nets = synthetic_batch([nodes], nodes * 1000)
    plan = hip.compile(nets, gb, training=True)
    prog = plan.program
    dout = torch.empty(prog.out_numel, dtype=torch.float32, device='cuda')
    ws16 = plan.ws.view(torch.int16)
    snaps = {}
    for fused in (True, False, True):
        step(hip, plan, dout, fused)
        amax = plan.ws[prog.r_amax[1]:prog.r_amax[1] + 4].view(torch.float32).item()
        print('fused' if fused else 'stream', 'amax slot', amax, 'scale exp', np.floor(np.log2(amax)))
        s = {}
        for g in prog.gemm_groups:
            if g.get('op16'):
                s['dth%d' % g['row0']] = ws16[g['dth']:g['dth'] + g['rows'] * g['dth_ld']].clone().view(torch.float16).view(g['rows'], g['dth_ld'])
        for b in prog.wgrad_bands:
            n = b['o_max'] * b['bw'] * b['ktot']
            s['dthT%d' % b['i_lo']] = ws16[b['dthT']:b['dthT'] + n].clone().view(torch.float16).view(b['o_max'] * b['bw'], b['ktot'])
        s['gflat'] = plan.gflat.clone()
        snaps[len(snaps)] = s
    a, b, c = snaps[0], snaps[1], snaps[2]
    ea = 2.0 ** (np.floor(np.log2(1.0)))
    for k in a:
        if k == 'gflat':
            continue
        x, y, z = a[k].float(), b[k].float(), c[k].float()
        # the routes may use different power-of-two scales: compare after normalising by the ratio of the maxima's scale
        r = float(x.abs().max()) / max(float(y.abs().max()), 1e-30)
        r2 = 2.0 ** np.round(np.log2(r)) if r > 0 else 1.0
        d = (x - y * r2)
        bad = (d.abs() > 0).sum().item()
        print('%-10s shape %s  scale ratio %g  differing elements %d of %d  max |d| %g  rel %g ; rerun direct == first: %s'
              % (k, tuple(x.shape), r2, bad, x.numel(), float(d.abs().max()), float(d.norm() / max(float(x.norm()), 1e-30)),
                 bool((x == z).all())))
        if bad and k.startswith('dthT'):
            idx = (d.abs() > 0).nonzero()
            print('    first differing (row, k):', idx[:5].tolist(), ' rows range', int(idx[:, 0].min()), int(idx[:, 0].max()),
                  'k range', int(idx[:, 1].min()), int(idx[:, 1].max()))
    offs = [int(o) for o in hip._offs] + [int(hip._flat_numel)]
    for k, name in enumerate(prog.names):
        x, y = a['gflat'][offs[k]:offs[k + 1]].double(), b['gflat'][offs[k]:offs[k + 1]].double()
        e = float((x - y).norm()) / max(float(y.norm()), 1e-30)
        if e > 2e-5:
            print('grad %-40s rel diff %.3e (norm %.4g)' % (name, e, float(y.norm())))


if __name__ == '__main__':
    main()
