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


def _schedule_worker(rank, world, port, ret):
    """Two ranks compile DIFFERENT architectures; the collectives each one issues for a backward (the sequence RCCL would
    see: kind, element count, wire type) are recorded by wrapping torch.distributed's entry points."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, 'golden'))
    import recipe
    from ghn3_amd import _lib as L
    from ghn3_amd.nn import GHN3
    from ghn3_amd.program import Program
    from ghn3_amd.ddp_utils import setup_ddp, FlatGradReducer, clean_ddp
    from ghn3_amd.synthetic import synthetic_batch
    setup_ddp()
    torch.manual_seed(0)
    ghn = GHN3(**recipe.TINY_CFG)                                   # (host object: flat parameter layout only)
    gb, nets = synthetic_batch([20 + 9 * rank], 31000 + rank)       # a different target network per rank
    gb._cat()
    cfg = dict(hid=ghn.hid, heads=ghn.heads, layers=ghn.layers, num_classes=ghn.num_classes, max_shape=ghn.max_shape)
    prog = Program(cfg, gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets, training=True,
                   decoder_ctype=L.CT_F16, decoder_bwd_ctype=L.CT_F16)
    offs, total = [int(v) for v in ghn._offs], int(ghn._flat_numel)
    log = []
    real = {n: getattr(dist, n) for n in ('all_reduce', 'all_to_all_single', 'all_gather_into_tensor')}

    def wrap(name):
        def f(*a, **k):
            t = a[0]
            log.append((name, int(t.numel()), str(t.dtype)))
            return real[name](*a, **k)
        return f
    for n in real:
        setattr(dist, n, wrap(n))
    oks = []
    try:
        for algo in ('allreduce', 'mesh'):
            for compress in (None, 'bf16'):
                log.append(('config', algo, str(compress)))
                g0 = torch.randn(total, generator=torch.Generator().manual_seed(5 + rank))
                g = g0.clone()
                red = FlatGradReducer(compress=compress, chunk_bytes=1 << 16, algo=algo)
                red.begin()
                for ops, slots in prog.bwd_parts:                    # what GHN3._run_backward does between the parts
                    for s_lo, s_hi in slots:
                        if s_hi > s_lo:
                            red.start(g, offs[s_lo], offs[s_hi] if s_hi < len(offs) else total)
                red.finish(g)
                both = [torch.empty_like(g0) for _ in range(world)]
                real_gather = dist.all_gather
                real_gather(both, g0)
                mean = sum(both) / world
                tol = 1e-6 if compress is None else 2e-2
                oks.append(bool(torch.allclose(g, mean, rtol=tol, atol=tol)))
    finally:
        for n in real:
            setattr(dist, n, real[n])
    w2 = prog.slot['decoder.conv.2.weight']
    ret[rank] = (oks, log, offs[w2 + 1] - offs[w2], int(prog.M), total)
    clean_ddp()


def test_collective_schedule_is_identical_for_different_graphs_world2():
    """RCCL hangs (or silently mixes buffers) when ranks disagree on the sequence or the sizes of their collectives.  Every
    rank compiles its own architecture each step, so the schedule of the overlapped exchange must depend on the GHN's
    parameter layout only: two ranks with different graphs (different decoder row counts, different op programs) must log
    exactly the same sequence of (collective, element count, wire type), for both algorithms and both wire types, the first
    range being the W2 gradient; and the exchanged buffer must be the mean over the ranks."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_schedule_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    (ok0, log0, w2n0, rows0, total0), (ok1, log1, w2n1, rows1, total1) = ret[0], ret[1]
    assert all(ok0) and all(ok1), (ok0, ok1)
    assert rows0 != rows1, 'the two ranks were meant to compile different architectures'
    assert log0 == log1 and total0 == total1
    # allreduce / fp32: the first collectives cover exactly the W2 gradient, in 64 KB pieces
    k = log0.index(('config', 'allreduce', 'None')) + 1
    n, first = 0, []
    while n < w2n0:
        name, numel, dt = log0[k]
        assert name == 'all_reduce' and dt == 'torch.float32'
        first.append(numel)
        n += numel
        k += 1
    assert n == w2n0 and max(first) <= (1 << 16) // 4
    # mesh / bf16: all-to-all + all-gather pairs on bf16 buffers, every element of the flat buffer exactly once
    k = log0.index(('config', 'mesh', 'bf16')) + 1
    seq = log0[k:]
    assert all(a[0] == 'all_to_all_single' and b[0] == 'all_gather_into_tensor' and a[2] == b[2] == 'torch.bfloat16'
               for a, b in zip(seq[0::2], seq[1::2]))
    assert sum(a[1] for a in seq[0::2]) >= total0


def _sharded_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ghn3_amd.ddp_utils import setup_ddp, FlatGradReducer, clean_ddp
    from ghn3_amd.optim import ShardedAdamW, adamw_reference_
    setup_ddp()
    n = 64 * 1500 + 37 * 64 + 21                       # chunks that do not divide by 64 W, a ragged end
    ranges = [(64 * 700, 64 * 1200), (64 * 10, 64 * 700)]            # "W2" first, then "the rest of the decoder"
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05, max_grad_norm=0.5)
    p0 = torch.linspace(-1, 1, n)
    runs = {}
    for sharded in (True, False):
        flat = p0.clone()
        opt = ShardedAdamW(flat=flat, update=adamw_reference_, **kw)
        norms = []
        for step in range(3):
            g = torch.randn(n, generator=torch.Generator().manual_seed(100 * step + rank)) * (1.0 + step)
            if step == 2 and rank == world - 1:
                g = g * 5.0                            # (so that the clip is active in one step and the ranks differ)
            red = FlatGradReducer(chunk_bytes=4 * 64 * 190, algo='rsag', gather=not sharded)
            red.begin()
            for lo, hi in ranges:
                red.start(g, lo, hi)
            red.finish(g)
            # (gather=True: the exchange also gathered the gradients, every rank updates everything -- the replicated trainer)
            norms.append(float(opt.step(g, red)))
        runs[sharded] = (flat, norms, sorted(red.owned), sorted(red.replicated))
    (fa, na, own, rep), (fb, nb, _, _) = runs[True], runs[False]
    owned_elems = sum(hi - lo for lo, hi in own)
    ret[rank] = (bool(torch.equal(fa, fb)), na, nb, owned_elems, sum(hi - lo for lo, hi in rep), n,
                 float((fa - p0).abs().max()))
    clean_ddp()


def test_sharded_optimizer_step_equals_the_replicated_one_world2():
    """ShardedAdamW (reduce-scatter of the gradients -> every rank updates the 1 / W of each chunk it owns -> all-gather of
    the updated parameters) against the replicated step (reduce-scatter + all-gather of the gradients, every rank updates
    everything), same exchange algorithm and the same update function: parameters bit-identical after three steps incl.
    one with an active clip, the clip coefficient from the all-reduced squared norm; a rank owns ~1 / W of the elements."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sharded_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        same, na, nb, owned, rep, n, moved = ret[r]
        assert same and moved > 1e-3
        assert all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(na, nb)), (na, nb)
        assert abs(owned - (n - rep) / world) < 1 and rep < 64 * world * 12
    assert ret[0][1] == ret[1][1]                       # the same norm on both ranks


def _rsag4_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from ghn3_amd.ddp_utils import setup_ddp, FlatGradReducer, clean_ddp
    setup_ddp()
    n = 300007                                           # (not a multiple of 64 * world)
    base = torch.linspace(-1, 1, n, dtype=torch.float64).float()
    expect = base * (sum(range(1, world + 1)) / world)
    out = {}
    # chunk lengths (floats) that 64 * world = 256 does not divide: every chunk leaves a replicated tail, and ranges whose
    # length is below one shard unit are reduced everywhere
    for chunk_floats, gather in ((50001, True), (50001, False), (777, True), (65536, False)):
        g = base * (rank + 1)
        red = FlatGradReducer(chunk_bytes=4 * chunk_floats, algo='rsag', gather=gather)
        red.begin()
        red.start(g, 120000, 200003)
        red.start(g, 1000, 120000)
        red.start(g, 200003, 250001)
        red.finish(g)
        own, rep = sorted(red.owned), sorted(red.replicated)
        cover = sorted(own + rep)
        disjoint = all(a[1] <= b[0] for a, b in zip(cover, cover[1:]))
        aligned = all((hi - lo) % 64 == 0 for lo, hi in own)
        if gather:
            ok = bool(torch.allclose(g, expect, rtol=1e-6, atol=1e-8))
        else:                                            # only the rank's shards + the replicated tails hold the mean
            ok = all(bool(torch.allclose(g[lo:hi], expect[lo:hi], rtol=1e-6, atol=1e-8)) for lo, hi in cover)
        out[(chunk_floats, gather)] = (ok, disjoint, aligned, sum(hi - lo for lo, hi in own), sum(hi - lo for lo, hi in rep),
                                       g.double().sum().item() if gather else 0.0)
    ret[rank] = out
    clean_ddp()


def test_reduce_scatter_all_gather_exchange_world4_non_divisible_chunks():
    """`rsag` (the default exchange of bench.py / Trainer at N > 1) in a world of FOUR with chunk sizes that 64 * world does not
    divide: every rank ends with the mean (gather=True: bit-identical buffers on all ranks), the shards of the four ranks
    tile each chunk's divisible part exactly once, the leftovers are reduced on every rank (ddp_utils.py:257-277)."""
    world = 4
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rsag4_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    n = 300007
    for key in ret[0]:
        for r in range(world):
            ok, disjoint, aligned, owned, rep, _ = ret[r][key]
            assert ok and disjoint and aligned, (key, r, ret[r][key])
        # all ranks own equally much, and owned x world + replicated = everything
        assert len({ret[r][key][3] for r in range(world)}) == 1 and len({ret[r][key][4] for r in range(world)}) == 1
        assert world * ret[0][key][3] + ret[0][key][4] == n, (key, ret[0][key])
        if key[1]:
            assert len({ret[r][key][5] for r in range(world)}) == 1          # identical bits everywhere
