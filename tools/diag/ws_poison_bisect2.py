"""GPU box: which workspace regions must read as zeros -- generalised ws_poison_bisect.py: any model / synthetic batch / route.
    python ws_poison_bisect2.py ghn3xlm16 6 11541 norm      (model, nodes (comma separated), seed, route: norm | dout)"""
import sys
import torch
import _paths  # noqa: F401
import recipe
from test_gpu_configs import _cfg
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch

name, nodes, seed, route = sys.argv[1], [int(v) for v in sys.argv[2].split(',')], int(sys.argv[3]), sys.argv[4]
shapes = {k: tuple(v.shape) for k, v in GHN3(**_cfg(name)).state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=7).items()}
hip = GHN3(**_cfg(name), compute='f16'); hip.load_state_dict(sd); hip = hip.to('cuda').train()
gb, nets = synthetic_batch(nodes, seed)
plan = hip.compile(nets, gb, training=True)
prog = plan.program
names = sorted(prog._ws_names, key=lambda k: prog._ws_names[k])
offs = [prog._ws_names[k] for k in names] + [prog.ws_bytes]
zero_always = list(prog.ws_zero) + (list(getattr(prog, 'ws_zero_dout', ())) if route == 'dout' or not prog.tile_bwd_h16 else [])
req = set(o for o, _ in zero_always)
stream = torch.cuda.current_stream().cuda_stream


def run(extra=()):
    plan.ws.fill_(0xff)
    for off, n in zero_always:
        plan.ws[off:off + n].zero_()
    for k in extra:
        i = names.index(k)
        plan.ws[offs[i]:offs[i + 1]].zero_()
    plan.ws_dout_ready = True
    hip._run_forward(plan)
    if route == 'norm':
        hip._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, stream)
        hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
    else:
        hip._run_backward(plan, torch.randn(prog.out_numel, device='cuda') * 1e-3)
    torch.cuda.synchronize()
    out_ok = all(bool(torch.isfinite(plan.out[p['offset']:p['offset'] + p['numel']]).all()) for p in prog.predicted)
    bad = [n for n, off in zip(prog.names, hip._offs)
           if not torch.isfinite(plan.gflat[int(off):int(off) + dict(hip.named_parameters())[n].numel()]).all()]
    return out_ok, bad


f, bad = run()
print(name, nodes, seed, route, 'regions', len(names), 'baseline: forward finite %s, non-finite gradients: %s' % (f, bad[:6]))
if bad or not f:
    for k in names:
        if prog._ws_names[k] in req:
            continue
        f2, bad2 = run((k,))
        if f2 and not bad2:
            print('  zeroing %-14s (%d bytes) makes everything finite' % (k, offs[names.index(k) + 1] - offs[names.index(k)]))
        elif len(bad2) < len(bad):
            print('  zeroing %-14s leaves %s' % (k, bad2[:4]))
    print('all zero:', run(tuple(names)))
