"""Diagnostic (GPU box): where the host time of the fresh-architecture loop goes.

    python tools/diag/fresh_loop_diag.py [steps]

Runs bench.py's fresh-graph loop (loader worker processes -> GHN3.plan -> forward + loss + backward) three ways and prints
per-step host timers: (a) results streamed from the pool while the loop runs (what bench.py measures), (b) the same plans
materialised BEFORE the timed loop (no pool activity: isolates the pool's result-handler thread), (c) like (b) but every
plan pre-built (no GHN3.plan in the loop: the replay cost of never-seen plans)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import _paths  # noqa: F401  (repository root, tests/, tests/golden/ on sys.path)
import multiprocessing as mp


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n_workers = int(os.environ.get('GHN3_LOADER_WORKERS', '6'))
    for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
        os.environ[var] = os.environ.get('DIAG_WORKER_THREADS', '4')     # inherited by the workers (bench.py: 4)
    pool = mp.get_context('spawn').Pool(n_workers)
    import torch
    import bench
    from ghn3_amd import GHN3, _lib as L
    dev = torch.device('cuda', 0)
    cfg = bench.model_cfg('ghn3xlm16')
    ghn = GHN3(**cfg, compute='f16').to(dev).train()
    ctx = ghn._ctx()
    stream = torch.cuda.current_stream().cuda_stream
    pcfg = ghn.program_config()
    skip = 12

    def tasks(seed0, n):
        return [(256, 1, seed0 + 7919 * (k + 1), pcfg) for k in range(n)]

    def run(pk, timers):
        prog = pk.program
        h = time.perf_counter()
        d_out = torch.empty(prog.out_numel, dtype=torch.float32, device=dev)
        norms = prog.norm_ops(1.0)
        h1 = time.perf_counter()
        ghn._run_forward(pk)
        h2 = time.perf_counter()
        ghn._fill_bufs(pk, out=pk.out, dout=d_out)
        ctx.run(norms[0], prog.problems, pk.bufs, stream)
        ctx.run(norms[1], prog.problems, pk.bufs, stream)
        h3 = time.perf_counter()
        ghn._run_backward(pk, d_out)
        h4 = time.perf_counter()
        for k, v in (('alloc+norm_ops', h1 - h), ('fwd', h2 - h1), ('loss', h3 - h2), ('bwd', h4 - h3)):
            timers[k] = timers.get(k, 0.0) + v

    def loop(name, source, planned):
        timers = {}
        spans = []
        npred = 0
        t0 = None
        for k in range(steps + skip):
            if k == skip:
                torch.cuda.synchronize()
                timers.clear()
                t0 = time.perf_counter()
            h0 = time.perf_counter()
            item = next(source)
            h1 = time.perf_counter()
            pk = item if planned else ghn.plan(item[2], item[0], item[1])
            h2 = time.perf_counter()
            ea, eb = L.Event(), L.Event()
            ea.record(stream)
            run(pk, timers)
            npred += sum(p_['numel'] for p_ in pk.program.predicted) if k >= skip else 0
            eb.record(stream)
            spans.append((ea, eb))
            timers['next'] = timers.get('next', 0.0) + h1 - h0
            timers['plan'] = timers.get('plan', 0.0) + h2 - h1
            del pk
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / steps
        gpu = sum(a.elapsed_ms(b) for a, b in spans[skip:]) / steps
        print('%-34s wall %.2f ms/step  gpu span %.2f  (%.1f M predicted params per graph)  host: %s' % (
            name, wall, gpu, npred / steps / 1e6, '  '.join('%s %.2f' % (k, 1e3 * v / steps) for k, v in timers.items())), flush=True)

    if os.environ.get('DIAG_PRELUDE', '0') == '1':          # what bench.py has done before its fresh loop: a replayed plan
        from ghn3_amd.synthetic import synthetic_batch
        gb0, nets0 = synthetic_batch([256], 256000)
        plan0 = ghn.compile(nets0, gb0, training=True)
        t0 = {}
        for _ in range(50):
            run(plan0, t0)
        torch.cuda.synchronize()
        keep = (plan0, gb0, nets0)
    n = steps + skip
    loop('(a) streamed from the pool', pool.imap(_w, tasks(256000, n)), False)
    res = list(pool.imap(_w, tasks(2000, n)))
    loop('(b) results materialised before', iter(res), False)
    res = list(pool.imap(_w, tasks(3000, n)))
    plans = [ghn.plan(r[2], r[0], r[1]) for r in res[:skip + steps]] if os.environ.get('DIAG_PREPLAN', '0') == '1' else None
    if plans is not None:
        loop('(c) plans pre-built', iter(plans), True)
    # (d) bounded in-flight submissions: at most Q results ahead of the consumer
    Q = int(os.environ.get('DIAG_AHEAD', '4'))
    tk = tasks(4000, n)
    pending = [pool.apply_async(_w, (t,)) for t in tk[:Q]]
    nxt = [Q]

    def bounded():
        while pending:
            r = pending.pop(0).get()
            if nxt[0] < len(tk):
                pending.append(pool.apply_async(_w, (tk[nxt[0]],)))
                nxt[0] += 1
            yield r
    if os.environ.get('DIAG_BOUNDED', '0') == '1':
        loop('(d) at most %d results in flight' % Q, bounded(), False)
    pool.close()


def _w(task):
    import bench
    return bench._loader_worker(task)


if __name__ == '__main__':
    main()
