"""World-size-2 gloo tests of the data-parallel exchange (one process per rank, like one process per GPU)."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ghn3_amd.ddp_utils import setup_ddp, all_reduce_flat_grads, avg_ddp_metric, is_ddp, get_ddp_rank, clean_ddp
    from ghn3_amd.synthetic import synthetic_batch
    args = setup_ddp()
    assert args.ddp and is_ddp() and get_ddp_rank() == rank and args.world_size == world
    # the flat gradient buffer: chunked async all-reduce must equal the mean over ranks, for odd sizes too
    n = 1000003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    all_reduce_flat_grads(g, chunk_bytes=1 << 20)
    expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok_grad = bool(torch.allclose(g, expect, rtol=1e-6))
    m = avg_ddp_metric(torch.tensor(float(rank + 1)))
    ok_metric = abs(m.item() - (world + 1) / 2) < 1e-6
    # per-rank synthetic graph streams are disjoint and deterministic (seed = N*1000 + rank)
    gb, nets = synthetic_batch([24], 24000 + rank)
    sig = int(sum(n_.num_params() for n_ in nets))
    sigs = [None] * world
    dist.all_gather_object(sigs, sig)
    ret[rank] = (ok_grad, ok_metric, sigs)
    clean_ddp()


def test_flat_gradient_allreduce_and_metric_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        ok_grad, ok_metric, sigs = ret[r]
        assert ok_grad and ok_metric
        assert len(set(sigs)) == world, sigs        # different target nets per rank


def _reducer_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ghn3_amd.ddp_utils import setup_ddp, FlatGradReducer, clean_ddp
    setup_ddp()
    n, lo, hi = 300007, 1000, 250001
    base = torch.linspace(-1, 1, n)
    oks = []
    for compress, tol, algo in ((None, 1e-6, 'mesh'), ('bf16', 1e-2, 'mesh'), (None, 1e-6, 'allreduce'),
                                ('bf16', 1e-2, 'allreduce')):
        g = base * (rank + 1)
        red = FlatGradReducer(compress=compress, chunk_bytes=1 << 18, algo=algo)
        expect = base * (sum(range(1, world + 1)) / world)
        for it in range(2):                            # (two backward passes through one reducer)
            g = base * (rank + 1)
            red.begin()
            red.start(g, 120000, 200000)               # the "W2" range first ...
            red.start(g, lo, 120000)                   # ... the rest of the "decoder" around it ...
            red.start(g, 200000, hi)
            g[:lo] += 0.0                              # (the rest of the backward would run here)
            red.finish(g)                              # ... then everything else
            oks.append(bool(torch.allclose(g, expect, rtol=tol, atol=tol * 1e-2)))
    ret[rank] = oks
    clean_ddp()


def test_two_phase_flat_grad_reducer_world2():
    """FlatGradReducer (the overlapped exchange used by bench.py / GHN3._run_backward for N > 1): the W2 range
    first, the rest of the decoder in two pieces, everything else afterwards; fp32 and bf16-on-the-wire give the mean over ranks."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reducer_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert all(ret[r]), ret[r]


def _mesh_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ghn3_amd.ddp_utils import setup_ddp, mesh_all_reduce_avg, sync_parameters, clean_ddp
    setup_ddp()
    oks = []
    for n in (1, 7, 1000, 100003):                       # sizes that do not divide the world size
        x = torch.arange(n, dtype=torch.float32) * (rank + 1) + rank
        mesh_all_reduce_avg(x)
        expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world) + (world - 1) / 2
        oks.append(bool(torch.allclose(x, expect, rtol=1e-6, atol=1e-6)))
        # a NaN on one rank reaches every rank (the optimizer's NaN guard then skips the step everywhere alike)
        y = torch.ones(max(n, 4))
        if rank == world - 1:
            y[0] = float('nan')
        mesh_all_reduce_avg(y)
        oks.append(bool(torch.isnan(y[0])) and bool(torch.isfinite(y[1:]).all()))

    class _M:                                             # (sync_parameters only needs the flat buffer)
        def __init__(self):
            self._flat = torch.full((1000,), float(rank))

        def params_changed(self):
            self.changed = True
    m = sync_parameters(_M(), src=0)
    oks.append(bool((m._flat == 0).all()) and m.changed)
    ret[rank] = oks
    clean_ddp()


def test_mesh_all_reduce_world3():
    """mesh_all_reduce_avg (all-to-all + fp32 local sum + all-gather, the xGMI-mesh-shaped exchange) on 3 ranks: odd
    sizes, NaN propagation (the cross-rank NaN guard relies on it), and the initial parameter broadcast."""
    world = 3
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mesh_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert all(ret[r]), (r, ret[r])
