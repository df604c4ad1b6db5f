import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
import network_cases, recipe
from ghn3_amd import ops
# 1. None-ness of gradients, stock vs fused
for name, (geno, kw, img) in network_cases.CASES.items():
    g = ops.Genotype(**geno)
    res = {}
    for mode in ('0', '1'):
        os.environ['GHN3_NATIVE_OPS'] = mode
        torch.manual_seed(0)
        net = ops.Network(genotype=g, **kw).cuda()
        x = torch.from_numpy(recipe.seeded_images(img, seed=7)).cuda()
        net.train(); torch.manual_seed(123)
        logits, aux = net(x)
        (logits.square().mean() + (aux.square().mean() if aux is not None else 0.)).backward()
        res[mode] = {n: (p.grad is None) for n, p in net.named_parameters()}
    diff = [n for n in res['0'] if res['0'][n] != res['1'][n]]
    print(name, 'params with different None-ness:', diff[:6], [(res['0'][n], res['1'][n]) for n in diff[:6]])
# 2. the [6]-node batch of the sweep
from test_gpu_configs import _cfg, _bench_step
from ghn3_amd import GHN3
from ghn3_amd.synthetic import synthetic_batch
shapes = {k: tuple(v.shape) for k, v in GHN3(**_cfg('ghn3xlm16')).state_dict().items()}
sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=7).items()}
m = GHN3(**_cfg('ghn3xlm16'), compute='f16'); m.load_state_dict(sd); m = m.to('cuda').train()
for rep in range(3):
    gb, nets = synthetic_batch([6], 11541)
    plan = m.compile(nets, gb, training=True)
    dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
    out, gflat, loss = _bench_step(m, plan, dout)
    torch.cuda.synchronize()
    prog = plan.program
    bad = [n for n, off in zip(prog.names, m._offs) if not torch.isfinite(gflat[int(off):int(off) + dict(m.named_parameters())[n].numel()]).all()]
    print('rep', rep, 'M', prog.M, 'n1', prog.n1, 'wgrad_cap', getattr(prog, 'wgrad_cap', None), 'out finite', bool(torch.isfinite(out[:prog.out_numel]).all()), 'bad grads:', bad[:8], len(bad))
    print([ (g['kind'], g['rows'], g['o'], g['i'], g.get('p8')) for g in prog.gemm_groups])
