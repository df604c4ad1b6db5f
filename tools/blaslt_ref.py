"""Vendor-library reference on the lab's shapes (torch.matmul f16 -> hipBLASLt), same box, random uniform data."""
import torch, time
def bench(M, N, K, name):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).half()
    b = (torch.rand(N, K, device='cuda') * 2 - 1).half()
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            c = a @ b.t()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 8)
    print('%-14s hipblaslt f16->f16 M=%6d N=%6d K=%6d  %8.4f ms %7.1f TF' % (name, M, N, K, best, 2.0 * M * N * K / best * 1e-9), flush=True)
bench(4096, 4096, 4096, 'square4k')
bench(8192, 8192, 8192, 'square8k')
bench(768, 147456, 3072, 'w2fwd')
bench(533, 147456 // 4, 3072, 'w2fwd-ragged')
bench(65536, 3072, 576, 'wgrad-band')
bench(4096, 4096, 192, 'short-k')
