"""Fused ReLU -> depthwise conv -> 1x1 conv -> BatchNorm (ghn3_dwpw_bn_fwd / _bwd) against the four stock ATen / MIOpen layers,
forward + backward, on shapes of DeepNets-1M training networks (CIFAR-10 batch 32 per GPU; ImageNet 224 batch 16):
    python tools/diag/target_ops_bench.py
Columns: stock = the four layers on NCHW tensors; fused = the op family on channels_last tensors (what the kernels do);
fused+conv = the same with the NCHW <-> NHWC copies at its boundary that ghn3_amd.target_ops.run_block makes today;
GB/s = the fused op's algorithmic traffic (x read 3x, z written + read 3x, dout read 3x, dy written + read 2x, dx / out written)
over its time."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ghn3_amd import target_ops as T  # noqa: E402

SHAPES = [  # N, C_in, C_out, H, ks, stride, pad, dil
    (32, 32, 32, 32, 3, 1, 1, 1), (32, 64, 64, 32, 3, 1, 1, 1), (32, 128, 128, 32, 3, 1, 1, 1), (32, 64, 64, 32, 5, 1, 2, 1),
    (32, 64, 64, 32, 3, 1, 2, 2), (32, 128, 128, 16, 3, 1, 1, 1), (32, 256, 256, 8, 3, 1, 1, 1), (32, 64, 128, 32, 3, 2, 1, 1),
    (16, 64, 64, 56, 3, 1, 1, 1), (16, 128, 128, 28, 5, 1, 2, 1)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print('%-44s %10s %10s %12s %8s %8s' % ('N C_in C_out H ks stride pad dil', 'stock us', 'fused us', 'fused+conv', 'speedup', 'GB/s'))
for (N, Ci, Co, H, ks, st, pad, dil) in SHAPES:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(N, Ci, H, H, device='cuda', generator=g)
    ws = [torch.randn(Ci, 1, ks, ks, device='cuda', generator=g) / ks, torch.randn(Co, Ci, 1, 1, device='cuda', generator=g) / Ci ** 0.5,
          torch.ones(Co, device='cuda'), torch.zeros(Co, device='cuda')]
    Ho = (H + 2 * pad - dil * (ks - 1) - 1) // st + 1
    up = torch.randn(N, Co, Ho, Ho, device='cuda', generator=g)
    xs = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in ws]

    def stock():
        out = T.reference(xs, *wr, stride=st, padding=pad, dilation=dil)
        torch.autograd.grad(out, [xs] + wr, up)
    xc = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    upc = up.contiguous(memory_format=torch.channels_last)

    def fused():
        out, _ = T.dwpw_bn(xc, *wr, stride=st, padding=pad, dilation=dil)
        torch.autograd.grad(out, [xc] + wr, upc)

    def fused_conv():
        out, _ = T.dwpw_bn(xs, *wr, stride=st, padding=pad, dilation=dil)
        out = out.contiguous(memory_format=torch.contiguous_format)
        torch.autograd.grad(out, [xs] + wr, up)
    t_s, t_f, t_c = timeit(stock), timeit(fused), timeit(fused_conv)
    P_in, P_out = N * H * H, N * Ho * Ho
    traffic = 4.0 * (3 * P_in * Ci + 6 * P_out * Co + 3 * P_out * Ci + P_in * Ci + P_out * Co)
    print('%-44s %10.1f %10.1f %12.1f %8.2f %8.0f' % ('%d %d %d %d %d %d %d %d' % (N, Ci, Co, H, ks, st, pad, dil), t_s, t_f, t_c,
                                                      t_s / t_f, traffic / (t_f * 1e-6) / 1e9))
