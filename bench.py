#!/usr/bin/env python3
"""
bench.py -- predicted-params/sec of the GHN-3 hot path (GHN fwd+bwd) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model ghn3xlm16] [--nodes 256] [--compute f32]

One "step" = one pass of the hot path over one batch of synthetic graphs per GPU:
    GHN3 forward (Graphormer + decoders + tile/normalise -> all target-network weights)
  + loss = sum_t ||p_t||_F over the predicted tensors (the reference's predparam_wd term, trainer.py:288-294)
  + backward through the whole GHN (gradients of all GHN parameters in one flat fp32 buffer)
  + for N > 1: mean all-reduce of that gradient buffer over RCCL/xGMI (the DDP exchange of trainer.py:136).
Inputs (graph tensors, index tables, GHN weights) are resident in HBM before the timed region.

`--gpus N` with N > 1 and no RANK in the environment starts N ranks itself (fresh child processes, one per GPU, created
before this process touches the GPU); under `python -m torch.distributed.run` the ranks are the launcher's.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects, plus (N = 1)
`forward` (forward-only time and its fraction of the 16-bit MFMA peak on algorithmic FLOPs: the north star's >= 30 %
target), `fresh_graph_ms_per_step` (a NEW architecture compiled every step on a host thread, what train_ghn_ddp.py does)
and `f32_mode` (the exact-fp32 configuration of the same workload).
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODELS = {  # name: (hid, layers, heads)
    'ghn3tm8': (64, 3, 8), 'ghn3sm8': (128, 5, 16), 'ghn3lm8': (256, 12, 16), 'ghn3xlm16': (384, 24, 16)}
PEAK_TFLOPS = {'f32': 157.3, 'f16': 2500.0, 'bf16': 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks


def model_cfg(name):
    hid, layers, heads = MODELS[name]
    return dict(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
                weight_norm=True, ve=True, layernorm=True)


def usable_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup's CPU quota (the GPU boxes show 256 logical
    CPUs behind a quota of 16: more runnable threads than that are throttled, not run)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 4
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            quota, period = fh.read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as fq, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fp:
                q, per = int(fq.read()), int(fp.read())
            if q > 0:
                n = max(1, min(n, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(model, sample_nodes, seed, graphs=1):
    """Oracle (CPU restatement of the reference algorithm incl. its per-group Python loops) timed on the host, by default
    on the SAME seeded graph(s) the GPU ran (one step: ~20-40 s at ghn3xlm16 / 256 nodes)."""
    from oracle import ghn3_ref as R
    from ghn3_amd.synthetic import synthetic_batch
    # Threads = the CPUs this container may really use (GPU boxes: 256 logical CPUs behind a cgroup quota of 16; measured
    # once on such a box, profiles/r03n_cpu_baseline_threads_256_64_16.txt: 16 threads 26.1 s per step, 64 threads 39.6 s,
    # 256 threads 581 s -- threads beyond the quota are throttled), at most GHN3_CPU_THREADS (default 16: torch's CPU GEMMs
    # stop scaling and the many small per-group ops get slower beyond a few dozen threads anyway).
    cores = max(1, min(usable_cores(), int(os.environ.get('GHN3_CPU_THREADS', '16'))))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    oracle = R.GHN3Ref(**model_cfg(model))
    gb, nets = synthetic_batch([sample_nodes] * graphs, seed)
    go = R.GraphBatchRef([R.GraphRef(g.node_feat, g.node_info, g._Adj) for g in gb.graphs])
    n_pred = sum(n.num_params() for n in nets)

    def step():
        oracle.zero_grad(set_to_none=True)
        _, pred = oracle(nets, go, keep_grads=True)
        loss = sum(torch.norm(t, p='fro') for (_, _, _, t) in pred)
        loss.backward()
    # median of three runs (>= 3 at ghn3xlm16 -- ~3 x 28 s on the GPU box's 16 usable cores); samples that run
    # in seconds get an untimed warm-up first (allocator, thread pool)
    runs = int(os.environ.get('GHN3_CPU_RUNS', '3'))
    t0 = time.time()
    step()
    times = [time.time() - t0]
    if times[0] < 5:
        times = []
    while len(times) < runs:
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    t = float(np.median(times))
    return {'value': n_pred / t, 'unit': 'predicted-params/s', 'cores': cores, 'kind': 'port',
            'host_cpus': os.cpu_count(), 'usable_cpus': usable_cores(),
            'threads_note': 'threads = the cgroup CPU quota (<= 16); measured once on a GPU box, 256 logical CPUs behind a quota of '
                            '16: 16 threads 26.1 s per step, 64 threads 39.6 s, 256 threads 581 s '
                            '(profiles/r03n_cpu_baseline_threads_256_64_16.txt)',
            'sample': '%s fwd+bwd (sum of Frobenius norms loss), %d synthetic %d-node graph(s), seed %d (%d predicted '
                      'params), fp32, torch %s CPU ops, %d threads, median %.1f s per step of %d runs (%s)'
                      % (model, graphs, sample_nodes, seed, n_pred, torch.__version__, cores, t, len(times),
                         ', '.join('%.1f' % v for v in times))}


_WORKER_NICED = False


def _loader_init(ready):
    """Loader-worker start-up: the imports a task needs, then a tick on the shared counter the parent waits for."""
    try:
        os.nice(10)
    except OSError:
        pass
    global _WORKER_NICED
    _WORKER_NICED = True
    from ghn3_amd.program import Program            # noqa: F401
    from ghn3_amd.synthetic import synthetic_batch  # noqa: F401
    with ready.get_lock():
        ready.value += 1


def _loader_worker(task):
    """Loader worker (separate process, never touches the GPU): one fresh synthetic architecture per step -- graph
    generation + the host half of GHN3.compile (numpy bookkeeping -> op program), what a DeepNets-1M loader worker
    would do per batch (deepnets1m.py:84-269)."""
    nodes, graphs_per_gpu, seed, pcfg = task
    global _WORKER_NICED
    if not _WORKER_NICED:                                # the training process's enqueue thread goes first on a busy host
        try:
            os.nice(10)
        except OSError:
            pass
        _WORKER_NICED = True
    from ghn3_amd.program import Program
    from ghn3_amd.synthetic import synthetic_batch
    gb, nets = synthetic_batch([nodes] * graphs_per_gpu, seed)
    gb._cat()
    gb.graphs = None                                     # (the per-graph copies of what _cat stacked: half of the pickle)
    pcfg = dict(pcfg)
    cfg = pcfg.pop('cfg')
    prog = Program(cfg, gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets, training=True, **pcfg)
    return gb, nets, prog.strip()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: N fresh child processes (one rank per GPU) started BEFORE this
    process initialises the GPU; rank 0's stdout (the JSON line) is relayed.  Never exec()s."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread (the JSON line can exceed a pipe buffer); the loop below only watches exit codes:
    # the first rank that dies takes the others (its exact child PIDs) with it instead of leaving them in a collective forever.
    # A rank that stalls ends ITSELF (the watchdog of main(): exit code 3 after GHN3_STALL_S seconds without a finished step).
    import threading
    chunks = []
    th = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    th.start()
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad:
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode
    th.join(timeout=5)
    sys.stdout.write(b''.join(chunks).decode())
    sys.stdout.flush()
    return rc


class Watchdog:
    """N > 1: a rank that makes no progress for `limit` seconds (a hung collective, a peer that died) ends with exit code 3
    instead of holding the node until the driver's own limit -- every timed / warm-up step and every phase of the run calls
    beat().  Exits through os._exit: never an exec, never a kill by pattern."""

    def __init__(self, limit, rank):
        import threading
        self.limit, self.rank, self.last, self.where = float(limit), rank, time.time(), 'start'
        self._stop = False
        threading.Thread(target=self._run, daemon=True).start()

    def beat(self, where=None):
        self.last = time.time()
        if where is not None:
            self.where = where

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(1.0)
            if time.time() - self.last > self.limit:
                sys.stderr.write('bench.py: rank %d made no progress for %.0f s (in: %s): exiting with code 3\n'
                                 % (self.rank, self.limit, self.where))
                sys.stderr.flush()
                os._exit(3)


def link_bound_estimate(grad_bytes, world, compute_ms):
    """A-priori arithmetic (nothing measured) for the N > 1 line: a ring / reduce-scatter + all-gather exchange moves
    2 (N - 1) / N of the gradient bytes through every rank's xGMI links; MI355X: 7 links x ~153 GB/s per GPU (task statement /
    MI355X guide), of which a ring uses two neighbours' links and a direct (mesh) algorithm all N - 1."""
    if world <= 1:
        return None
    per_rank = 2.0 * (world - 1) / world * grad_bytes
    link = 153e9
    ring_ms = 1e3 * per_rank / (2 * link)                    # one ring: in + out over one link pair
    mesh_ms = 1e3 * per_rank / (min(world - 1, 7) * link)    # all peers at once
    return {'bytes_per_rank_on_the_wire': int(per_rank), 'xgmi_link_GBps': 153, 'links_per_gpu': 7,
            'single_ring_ms': ring_ms, 'all_links_ms': mesh_ms, 'compute_only_ms_per_step': compute_ms,
            'weak_scaling_ceiling_if_fully_exposed': compute_ms / (compute_ms + mesh_ms),
            'weak_scaling_ceiling_if_fully_overlapped': min(1.0, compute_ms / mesh_ms) if mesh_ms > 0 else 1.0,
            'note': 'arithmetic only: RCCL picks its own channels; `exchange_ms.exposed` is the measurement'}


CONFIGS = {   # BASELINE.json `configs` 1-4 (config 5 = the headline workload with --graphs-per-gpu 2)
    'resnet18-tm8': dict(model='ghn3tm8', fixture='resnet18', baseline_config=1),
    'tm8-128': dict(model='ghn3tm8', nodes=128, baseline_config=2),
    'lm8-200': dict(model='ghn3lm8', nodes=200, baseline_config=3),
    'resnet50-xl': dict(model='ghn3xlm16', fixture='resnet50', baseline_config=4, published_s=3.385),
    'vit-xl': dict(model='ghn3xlm16', fixture='vit_b16', baseline_config=4),
}


class _Bottleneck(torch.nn.Module):
    def __init__(self, cin, planes, stride, down, wide):
        super().__init__()
        nn = torch.nn
        if wide:
            self.conv1, self.bn1 = nn.Conv2d(cin, planes, 1, bias=False), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
            self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        else:
            self.conv1, self.bn1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False), nn.BatchNorm2d(planes)
            self.conv3 = None
        out = planes * (4 if wide else 1)
        self.downsample = nn.Sequential(nn.Conv2d(cin, out, 1, stride, bias=False), nn.BatchNorm2d(out)) if down else None

    def forward(self, x):
        F = torch.nn.functional
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.conv3 is not None:
            y = self.bn3(self.conv3(F.relu(y)))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class TorchvisionShapedResNet(torch.nn.Module):
    """torchvision.models.resnet18 / resnet50 written out by hand (torchvision is not in the image): the module `Graph(model)`
    is timed on -- eval_ghn.py:147-148 builds the graph of every evaluated network this way."""

    def __init__(self, depth=50):
        super().__init__()
        nn = torch.nn
        wide = depth == 50
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin, stages = 64, []
        for li, (nb, planes) in enumerate(zip([3, 4, 6, 3] if wide else [2, 2, 2, 2], (64, 128, 256, 512))):
            blocks = []
            for b in range(nb):
                out = planes * (4 if wide else 1)
                blocks.append(_Bottleneck(cin, planes, 2 if (b == 0 and li > 0) else 1, b == 0 and (li > 0 or wide), wide))
                cin = out
            stages.append(nn.Sequential(*blocks))
        self.layer1, self.layer2, self.layer3, self.layer4 = stages
        self.avgpool, self.fc = nn.AdaptiveAvgPool2d(1), nn.Linear(cin, 1000)

    def forward(self, x):
        x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def bench_fixture(args, cfg):
    """BASELINE configs 1 / 4: INFERENCE forward of a released GHN-3 size on a committed graph fixture (tests/golden/recipe.py:
    the torchvision-shaped ResNet-18 / ResNet-50 / ViT-B/16 graphs the reference-golden parity tests use), one GPU.
    `value` = predicted parameters / time of the forward program with graph, index tables and weights resident (the metric's
    definition); beside it the end-to-end `ghn(model, graph)` call (host compile + upload + program + assignment into the
    modules), `Graph(model)` on the host, the HBM roofline of SURVEY 8(d) (M = 196 / 229 / ~450 decoder rows: the decoder
    weight stream bounds these, not the matrix cores) and the CPU restatement on the same inputs."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import recipe
    torch.set_num_threads(int(os.environ.get('GHN3_HOST_THREADS', '1')))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the GHN-3 path has no CPU fallback')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    from ghn3_amd import GHN3, Graph, GraphBatch, _lib as L
    fixture = cfg['fixture']
    spec = recipe.vit_b16_spec() if fixture == 'vit_b16' else recipe.resnet_spec(int(fixture[6:]))
    torch.manual_seed(0)
    ghn = GHN3(**model_cfg(cfg['model']), compute=args.compute).to(dev).eval()
    net = recipe.build_torch_net(spec)
    nf, info, A = recipe.graph_arrays(spec)

    def batch():
        return GraphBatch([Graph(node_feat=nf, node_info=info, A=A)], dense=True)
    ctx = L.context(0)
    stream = torch.cuda.current_stream().cuda_stream
    with torch.no_grad():
        plan = ghn.compile([net], batch(), training=False)
        prog = plan.program
        n_pred = sum(p_['numel'] for p_ in prog.predicted)
        for _ in range(args.warmup):
            ghn._run_forward(plan)
        torch.cuda.synchronize()
        ctx.profile(2)
        ctx.profile_read_tags(reset=True)
        ea, eb = L.Event(), L.Event()
        t0 = time.perf_counter()
        ea.record(stream)
        for _ in range(args.steps):
            ghn._run_forward(plan)
        eb.record(stream)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        fwd_ms = ea.elapsed_ms(eb) / args.steps
        tags = ctx.profile_read_tags(reset=True)
        ctx.profile(0)
        # end to end, as a user calls it (eval_ghn.py:148 with a pre-built graph): host compile + uploads + program + assign
        n_e2e = max(3, min(20, args.steps))
        for _ in range(2):
            ghn(net, batch(), bn_track_running_stats=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_e2e):
            ghn(net, batch(), bn_track_running_stats=True)
        torch.cuda.synchronize()
        e2e_ms = 1e3 * (time.perf_counter() - t1) / n_e2e
    # Graph(model) on the host (the step in front of the prediction when no graph is given)
    graph_ms = None
    if fixture.startswith('resnet'):
        model = TorchvisionShapedResNet(int(fixture[6:]))
        Graph(model)
        tg = time.perf_counter()
        for _ in range(5):
            g_ = Graph(model)
        graph_ms = 1e3 * (time.perf_counter() - tg) / 5
        assert g_.n_nodes == len(spec['nodes']), (g_.n_nodes, len(spec['nodes']))
    # HBM roofline (SURVEY 8(d) "algorithmic bytes per graph"): decoder weights once in the type they are read in + Graphormer
    # weights + 4 B x predicted parameters written + the tiles written and read once
    C = prog.C
    n_pos = len(getattr(prog, 'd1', []) or [])
    w2_b = 2 if getattr(prog, 'uses_op16', False) else 4
    bytes_alg = (w2_b * C * C * 8 * C + 4.0 * n_pos * 4 * C * C + 4.0 * 8 * C * 4 * C +
                 (4.0 if prog.x3 else 4.0) * prog.Lyr * 12 * C * C + 4.0 * n_pred + 8.0 * getattr(prog, 'tiles_floats', 0))
    rows = prog.B * prog.N
    g_fl = prog.Lyr * (24.0 * rows * C ** 2 + 4.0 * prog.B * prog.N ** 2 * C)
    d_fl = sum(prog.tag_flops.get(t, 0.0) for t in (prog.TAG_D3_FWD, prog.TAG_D2_FWD, prog.TAG_D1_FWD))
    detail = {}
    for t, (tms, cnt) in sorted(tags.items()):
        detail[prog.TAG_NAMES.get(t, str(t))] = {'ms_per_step': round(tms / args.steps, 4)}
    out = {
        'metric': 'predicted-params/sec (GHN inference forward), %s -> %s' % (cfg['model'], fixture),
        'value': n_pred / (fwd_ms * 1e-3), 'unit': 'predicted-params/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': fwd_ms, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': (n_pred / (fwd_ms * 1e-3)) / (n_pred / cfg['published_s']) if 'published_s' in cfg else None,
        **({'vs_baseline_note': 'against the one published number of the reference for this path: %.3f s for this forward with '
                                'the graph pre-built, on an unspecified CPU (examples/ghn_all_pytorch.ipynb:137)'
                                % cfg['published_s']} if 'published_s' in cfg else {}),
        'dtype': args.compute, 'data': 'committed graph fixture (tests/golden/recipe.py), random-init GHN weights',
        'config': {'workload': 'BASELINE config %d: %s inference forward on the %s graph fixture (%d nodes, %d predicted '
                               'params, %d decoder rows), one MI355X' % (cfg['baseline_config'], cfg['model'], fixture, prog.N, n_pred,
                                                                         int(prog.M)),
                   'ghn_params': int(ghn._flat_numel), 'decoder_rows': int(prog.M), 'parallelism': 'dp1'},
        'roofline': {'bound': 'hbm', 'achieved': bytes_alg / (fwd_ms * 1e-3) / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                     'frac': bytes_alg / (fwd_ms * 1e-3) / 1e9 / 8000.0, 'traffic': None,
                     'algorithmic_bytes': bytes_alg,
                     'algorithmic_bytes_formula': 'W2 once (%d B / element) + used decoder.fc rows + W0 + Graphormer weights (fp32 '
                                                  'equivalents) + 4 B x predicted params + tiles written and read' % w2_b,
                     'algorithmic_gflop': (g_fl + d_fl) / 1e9,
                     'mfma_frac_of_16bit_peak': (g_fl + d_fl) / (fwd_ms * 1e-3) / 1e12 / 2500.0, 'kernels': detail},
        'end_to_end_ms': {'ghn(model, graph)': e2e_ms, 'Graph(model) host': graph_ms,
                          'note': 'ghn(model, graph) = host compile of the op program + uploads + the forward program + '
                                  'assignment of the predicted tensors into the modules (eval: clones into .data), per call'},
        'wall_ms_per_step': 1e3 * wall / args.steps,
    }
    if not args.no_cpu_baseline:
        try:
            from oracle import ghn3_ref as R
            cores = max(1, min(usable_cores(), int(os.environ.get('GHN3_CPU_THREADS', '16'))))
            torch.set_num_threads(cores)
            torch.manual_seed(0)
            oracle = R.GHN3Ref(**model_cfg(cfg['model'])).eval()
            go = R.GraphBatchRef([R.GraphRef(torch.from_numpy(nf), info, torch.from_numpy(A))])
            net_o = recipe.build_torch_net(spec)
            times = []
            with torch.no_grad():
                oracle([net_o], go)
                budget = time.time() + 25
                while len(times) < 5 and (time.time() < budget or len(times) < 2):
                    tc = time.time()
                    oracle([net_o], go)
                    times.append(time.time() - tc)
            t_med = float(np.median(times))
            out['cpu_baseline'] = {'value': n_pred / t_med, 'unit': 'predicted-params/s', 'cores': cores, 'kind': 'port',
                                   'sample': '%s inference forward on the same fixture, fp32, torch %s CPU ops, %d threads, median '
                                             '%.2f s of %d runs' % (cfg['model'], torch.__version__, cores, t_med, len(times))}
        except Exception as e:                                   # never lose the GPU line
            out['cpu_baseline'] = {'value': None, 'error': repr(e)}
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', default='ghn3xlm16')
    ap.add_argument('--nodes', type=int, default=256)
    ap.add_argument('--graphs-per-gpu', type=int, default=1)
    # f16: decoder GEMMs on f16 operands (bf16 for the W2 backward), fp32 accumulate, everything else exact fp32 --
    # the mode the 1e-3 parity tests cover (tests/test_gpu_parity.py); f32: exact fp32 MFMA everywhere
    ap.add_argument('--compute', default=os.environ.get('GHN3_COMPUTE', 'f16'), choices=['f32', 'f16', 'bf16'])
    ap.add_argument('--cpu-sample-nodes', type=int, default=0,
                    help='nodes of the CPU-baseline graph (0 = the same graph(s) the GPU ran)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the forward-only / fresh-graph / f32-mode measurements (profiling runs)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-ops', action='store_true', help='print per-op-kind time of one extra step')
    ap.add_argument('--grad-allreduce', default=os.environ.get('GHN3_GRAD_ALLREDUCE', 'f32'),
                    choices=['f32', 'bf16', 'f32-serial'],
                    help='N > 1: gradient exchange.  f32 (default: what DistributedDataParallel does, trainer.py:136) / bf16 = '
                         'all-reduce overlapped with the backward, fp32 or bf16 copies on the wire; f32-serial = one fp32 '
                         'all-reduce after the backward.  The line of the default also carries the bf16-wire timing as '
                         '`bf16_wire` (an extra, lower-precision figure: never `value`)')
    ap.add_argument('--force-ddp', action='store_true', help='run the N > 1 code path in a 1-rank group (testing)')
    ap.add_argument('--config', default=None, choices=sorted(CONFIGS),
                    help='one of the other BASELINE.json configurations instead of the headline workload: inference forward on a '
                         'committed graph fixture (resnet18-tm8 = config 1, resnet50-xl / vit-xl = config 4: the one workload the '
                         'reference publishes a number for) or the per-GPU workload of config 2 / 3 (tm8-128, lm8-200: the default '
                         'measurement on that model / graph size)')
    args = ap.parse_args()
    if args.config is not None:
        cfg_ = CONFIGS[args.config]
        if 'fixture' in cfg_:
            raise SystemExit(bench_fixture(args, cfg_))
        args.model, args.nodes = cfg_['model'], cfg_['nodes']

    if 'RANK' not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))        # (before any GPU call in this process)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: start one rank per GPU (python -m torch.distributed.run '
                         '--nproc-per-node %d ... bench.py --gpus %d) or drop RANK/WORLD_SIZE from the environment'
                         % (args.gpus, world, args.gpus, args.gpus))
    import torch.distributed as dist
    # host-side torch ops of this process (unpickled graph tensors, pinned staging): ONE thread -- torch's default
    # is one per LOGICAL CPU (256 on the GPU boxes, behind a cgroup quota of 16: a spinning OpenMP pool of that size gets the
    # whole process throttled).  cpu_baseline() sets its own count.
    torch.set_num_threads(int(os.environ.get('GHN3_HOST_THREADS', '1')))     # (4 threads measured 9.2 vs 7.4 ms per fresh step)
    pool = None
    if world == 1 and not args.no_extras and not args.force_ddp:
        import multiprocessing as mp                     # loader workers: started before ANY torch.cuda call of this process
        for var in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
            os.environ.setdefault(var, '1')              # (inherited by the workers: the host half of the compile is
            #                                              single-threaded numpy; 4 threads each measured 9.2 vs 7.4 ms per step)
        # a worker needs ~30 ms of CPU per architecture (graph + host half of the compile, ghn3xlm16 / 256 nodes) and the GPU
        # consumes one every ~6.2 ms: round 4 measured, 16 usable cores: 6 workers 8.4 ms per step (waiting for plans),
        # 10 or 12 workers 6.7 ms.  More workers than spare cores only take the cores the enqueue thread of this process
        # needs (the driver's box of round 2: 8 workers on 8 cores, enqueue 14 ms per step instead of 3)
        n_cores = usable_cores()
        n_workers = int(os.environ.get('GHN3_LOADER_WORKERS', str(max(2, min(10, n_cores - 6)))))
        # The workers import torch + the host compiler when they start (~2-4 s of CPU each).  Round 6: wait for that BEFORE the
        # timed region -- ten importing processes on the box's 16 cores slowed this process's enqueue thread during the
        # driver's short run (--steps 20: 6.02 ms per step against 5.89 with --no-extras on the same box).
        ctx_mp = mp.get_context('spawn')
        ready = ctx_mp.Value('i', 0)
        pool = ctx_mp.Pool(n_workers, initializer=_loader_init, initargs=(ready,))
        t_w = time.time()
        while ready.value < n_workers and time.time() - t_w < 120:
            time.sleep(0.05)
    if not torch.cuda.is_available():
        if pool is not None:
            pool.terminate()
        raise SystemExit('bench.py needs an MI355X: the GHN-3 path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    ddp = world > 1 or args.force_ddp
    if ddp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world)

    # (60 s without a finished step ends a rank; until the FIRST step has finished -- RCCL builds its communicators inside it,
    # tens of seconds on an 8-GPU node -- the limit is five times that)
    stall_s = float(os.environ.get('GHN3_STALL_S', '60'))
    dog = Watchdog(5 * stall_s, rank) if world > 1 else None
    beat = (lambda where=None: dog.beat(where)) if dog is not None else (lambda where=None: None)

    from ghn3_amd import GHN3, _lib as L
    from ghn3_amd.synthetic import synthetic_batch
    from ghn3_amd.ddp_utils import all_reduce_flat_grads_avg as all_reduce_flat_grads, FlatGradReducer
    reducer = None
    if ddp and args.grad_allreduce != 'f32-serial':
        reducer = FlatGradReducer(compress='bf16' if args.grad_allreduce == 'bf16' else None, force=args.force_ddp)

    torch.manual_seed(0)                                   # identical random-init GHN weights on every rank
    ghn = GHN3(**model_cfg(args.model), compute=args.compute).to(dev)
    ghn.train()
    seeds = args.nodes * 1000 + rank * args.graphs_per_gpu
    gb, nets = synthetic_batch([args.nodes] * args.graphs_per_gpu, seeds)
    plan = ghn.compile(nets, gb, training=True)
    prog = plan.program
    n_pred = sum(p['numel'] for p in prog.predicted)
    f_norm, b_norm = prog.norm_ops(1.0)
    ctx = L.context(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    dout = torch.empty(prog.out_numel, dtype=torch.float32, device=dev)

    exchange = {'on': True, 'reducer': reducer}
    # loss = sum_t ||p_t||_F (the reference's predparam_wd term, trainer.py:97-98,288-294).  Default: fused into the tile
    # kernels -- the tile forward leaves per-block sums of squares, GHN3_OP_PARAM_NORM_FIN turns them into norms + loss and
    # the tile backward forms p / ||p|| itself (what `loss.backward()` of GHN3.predicted_param_norm() runs).
    # GHN3_FUSED_LOSS=0: the streaming passes of rounds 1-3 (PARAM_NORM_FWD / BWD over the 346 MB output, materialised dout).
    fused_loss = os.environ.get('GHN3_FUSED_LOSS', '1') != '0'
    one = torch.ones(1, dtype=torch.float32, device=dev)

    def run_step(model, pl, d_out, norms):
        model._run_forward(pl)
        red = exchange['reducer'] if exchange['on'] else None
        if fused_loss:
            ctx.run(norms[2], pl.program.problems, pl.bufs, stream)
            model._run_backward(pl, None, reducer=red, norm_g=one)
        else:
            model._fill_bufs(pl, out=pl.out, dout=d_out)
            ctx.run(norms[0], pl.program.problems, pl.bufs, stream)
            ctx.run(norms[1], pl.program.problems, pl.bufs, stream)
            model._run_backward(pl, d_out, reducer=red)
        if ddp and red is None and exchange['on']:
            all_reduce_flat_grads(pl.gflat)

    fin_norm = prog.norm_fin_ops()

    def step():
        run_step(ghn, plan, dout, (f_norm, b_norm, fin_norm))
        beat()

    beat('warm-up')
    for _ in range(args.warmup):
        step()
        if dog is not None:
            torch.cuda.synchronize()                     # (N > 1: a hang shows up at the step that hangs, not 60 s later)
            beat()
            dog.limit = stall_s
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if dog is not None and args.warmup > 0:
        dog.limit = stall_s
    beat('timed region')
    # The timed region runs WITHOUT per-launch events (round 6): the HIP events of profile mode 2 around the nine tagged launch
    # groups of a step -- on two streams -- cost 0.05-0.12 ms of a 5.8 ms step (profiles/r06t_*: 5.82-5.89 with, 5.77 without, same
    # box, alternating).  `kernels_in_step` comes from the same schedule in a few instrumented steps right behind the timed
    # ones; GHN3_BENCH_TAGS_IN_TIMED=1 restores the events inside the timed region.
    tags_in_timed = os.environ.get('GHN3_BENCH_TAGS_IN_TIMED', '0') == '1'
    ctx.profile(2 if tags_in_timed else 0)
    ctx.profile_read_tags(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    beat('instrumented passes')
    n_inst = args.steps
    if not tags_in_timed:
        n_inst = max(3, min(10, args.steps))
        ctx.profile(2)
        ctx.profile_read_tags(reset=True)
        for _ in range(n_inst):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    tags = ctx.profile_read_tags(reset=True)
    ctx.profile(0)
    # The kernels' OWN durations, for `roofline`: a few untimed steps with the side stream serialised into the chain (profile
    # mode 3: nothing co-runs with a timed kernel, side-stream grid caps dropped).  The timed region above runs the fastest
    # schedule -- there the W2 weight gradient shares the chip with the Graphormer backward and its in-step duration
    # (`roofline.kernels_in_step`) says how the two streams interleave, not what the kernel can do.
    n_ser = max(3, min(10, args.steps))
    ex_on, exchange['on'] = exchange['on'], False
    ctx.profile(3)
    ctx.profile_read_tags(reset=True)
    for _ in range(n_ser):
        step()
    torch.cuda.synchronize()
    tags_ser = ctx.profile_read_tags(reset=True)
    ctx.profile(0)
    exchange['on'] = ex_on
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n_all = torch.tensor([float(n_pred)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(n_all, op=dist.ReduceOp.SUM)
    elapsed = float(t_all.item())
    total_pred = float(n_all.item())

    extras = {}
    if ddp:
        # What the N > 1 line needs to be judged: proof that RCCL spans all ranks (an all-reduce of ones), the EXPOSED cost
        # of the gradient exchange (step with it - the same step without any collective), the exchange alone (serial, not
        # overlapped) and -- for the default fp32 wire -- the bf16-wire variant as an extra figure.
        def timed(n, fn=None):
            fn = fn or step
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            beat()
            t_ = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tt = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return 1e3 * float(tt.item()) / n
        beat('exchange measurements')
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        n_x = max(5, args.steps // 4)
        exchange['on'] = False
        compute_ms = timed(n_x)
        exchange['on'] = True
        torch.cuda.synchronize()
        ea, eb = L.Event(), L.Event()
        ea.record(stream)
        for _ in range(3):
            all_reduce_flat_grads(plan.gflat)
        eb.record(stream)
        serial_ms = ea.elapsed_ms(eb) / 3
        extras['rccl_ranks'] = int(round(float(ones.item())))
        gbytes = int(plan.gflat.numel()) * 4
        wire_b = gbytes // 2 if args.grad_allreduce == 'bf16' else gbytes
        lo_d, hi_d = ghn.decoder_grad_range(prog)
        w2_b = 4 * int(prog.C) ** 2 * 8 * int(prog.C)
        extras['exchange_ms'] = {'exposed': 1e3 * elapsed / args.steps - compute_ms, 'compute_only_ms_per_step': compute_ms,
                                 'serial_fp32_allreduce': serial_ms, 'gradient_bytes': gbytes,
                                 'wire': args.grad_allreduce,
                                 'algorithm': (reducer.algo if reducer is not None else 'allreduce (one call behind the backward)'),
                                 # what one rank puts on / takes off its xGMI links per step, by phase of the exchange
                                 # (reduce-scatter + all-gather: (W - 1) / W of the buffer each way per half) and by the part
                                 # of the backward that releases it (dW2 first: it overlaps everything behind it)
                                 'bytes_per_rank': {'reduce_scatter_sent': wire_b * (world - 1) // max(world, 1),
                                                    'all_gather_received': wire_b * (world - 1) // max(world, 1)},
                                 'bytes_by_part': {'1: decoder.conv.2.weight (behind the W2 weight gradient, issued first)': w2_b,
                                                   '2: rest of the decoder': 4 * (hi_d - lo_d) - w2_b,
                                                   '3: Graphormer + embeddings (end of the backward)': gbytes - 4 * (hi_d - lo_d)}}
        extras['link_bound_estimate'] = link_bound_estimate(gbytes, world, compute_ms)
        if args.grad_allreduce == 'f32':
            beat('bf16-wire variant')
            exchange['reducer'] = FlatGradReducer(compress='bf16', force=args.force_ddp)
            for _ in range(2):
                step()
            b16 = timed(n_x)
            exchange['reducer'] = reducer
            extras['bf16_wire'] = {'ms_per_step': b16, 'value': total_pred / (b16 * 1e-3),
                                   'note': 'same step with bf16 copies of the gradients on the wire (fp32 local sums): lower '
                                           'precision than the reference DDP exchange -- reported beside, never as `value`'}
        # The reference's own per-GPU batch for this model (train_ghn_ddp.py:92: meta-batch 16 over the world -- 2 graphs per GPU
        # at N = 8, BASELINE config 5) as a second line beside the 1-graph headline: same exchange bytes, twice the compute to
        # hide them behind.
        ref_gpg = max(1, 16 // max(world, 1)) if args.model == 'ghn3xlm16' else max(1, 8 // max(world, 1))
        ref_gpg = min(ref_gpg, 4)
        if ref_gpg != args.graphs_per_gpu and os.environ.get('GHN3_BENCH_REF_BATCH', '1') != '0':
            beat('reference per-GPU batch (%d graphs)' % ref_gpg)
            gb_r, nets_r = synthetic_batch([args.nodes] * ref_gpg, args.nodes * 1000 + rank * ref_gpg)
            plan_r = ghn.compile(nets_r, gb_r, training=True)
            prog_r = plan_r.program
            norms_r = prog_r.norm_ops(1.0) + (prog_r.norm_fin_ops(),)
            dout_r = None if fused_loss else torch.empty(prog_r.out_numel, dtype=torch.float32, device=dev)

            def step_r():
                run_step(ghn, plan_r, dout_r, norms_r)
                beat()
            for _ in range(3):
                step_r()
            n_r = max(5, args.steps // 4)
            with_x = timed(n_r, step_r)
            exchange['on'] = False
            without_x = timed(n_r, step_r)
            exchange['on'] = True
            pr = torch.tensor([float(sum(p_['numel'] for p_ in prog_r.predicted))], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(pr, op=dist.ReduceOp.SUM)
            extras['reference_batch'] = {'graphs_per_gpu': ref_gpg, 'meta_batch': ref_gpg * world, 'ms_per_step': with_x,
                                         'value': float(pr.item()) / (with_x * 1e-3), 'compute_only_ms_per_step': without_x,
                                         'exchange_exposed_ms': with_x - without_x,
                                         'note': 'the per-GPU batch train_ghn_ddp.py runs for this model at this world size; '
                                                 '`value` of the line stays the 1-graph-per-GPU headline workload'}
            del plan_r
            torch.cuda.empty_cache()
        beat('tail')
    if world == 1 and not args.no_extras and not args.force_ddp:
        # (a) forward only: the north star's target is stated on the Graphormer + decoder FORWARD
        ev0, ev1 = L.Event(), L.Event()
        n_f = max(5, args.steps)
        torch.cuda.synchronize()
        ev0.record(stream)
        for _ in range(n_f):
            ghn._run_forward(plan)
        ev1.record(stream)
        fwd_ms = ev0.elapsed_ms(ev1) / n_f
        rows = prog.B * prog.N
        g_fl = prog.Lyr * (24.0 * rows * prog.C ** 2 + 4.0 * prog.B * prog.N ** 2 * prog.C)
        d_fl = sum(prog.tag_flops.get(t, 0.0) for t in (prog.TAG_D3_FWD, prog.TAG_D2_FWD, prog.TAG_D1_FWD))
        extras['forward'] = {'ms': fwd_ms, 'algorithmic_gflop': (g_fl + d_fl) / 1e9,
                             'graphormer_gflop': g_fl / 1e9, 'decoder_gflop': d_fl / 1e9,
                             'tflops': (g_fl + d_fl) / (fwd_ms * 1e-3) / 1e12,
                             'frac_of_16bit_mfma_peak': (g_fl + d_fl) / (fwd_ms * 1e-3) / 1e12 / 2500.0}
        # (b) a NEW architecture every step (train_ghn_ddp.py draws one per step): graph generation + the host half of the
        # compile in loader worker processes (like the reference's DataLoader workers), the device half + the GPU path
        # in this process
        n_fresh = max(8, min(args.steps, 40))
        pcfg = ghn.program_config()
        n_skip = 12                                        # untimed: every worker's first item carries ~0.5 s of imports
        tasks = [(args.nodes, args.graphs_per_gpu, seeds + 7919 * (k + 1), pcfg) for k in range(n_fresh + n_skip)]
        stream_it = pool.imap(_loader_worker, tasks)       # (ordered; the workers run ahead of the consumer)
        # (GHN3_SWITCH_INTERVAL: interpreter switch interval while the pool's result-handler thread unpickles the workers'
        # results inside this process; measured r03: 0.2 ms instead of the default 5 ms made the loop SLOWER, 15.4 vs 11.5 ms
        # per step -- the enqueue time is mostly the host waiting for the GPU two steps ahead, not lock hand-offs)
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(float(os.environ.get('GHN3_SWITCH_INTERVAL', str(old_switch))))
        # single consumer thread: the device half of the compile (GHN3.plan: asynchronous uploads from reusable pinned
        # slots) costs ~3 ms of host time and the enqueue of a step ~3 ms -- together less than the GPU's step, so the
        # loop stays GPU-bound without a prefetch thread (one was measured: with the pool's result thread it made three
        # Python threads compete for the interpreter lock, 12-23 ms per step)
        threaded = os.environ.get('GHN3_FRESH_PREFETCH_THREAD', '0') == '1'
        if threaded:
            import queue
            import threading
            ready = queue.Queue(maxsize=3)

            def prefetch():                                # waits for the workers + device half of the compile
                torch.cuda.set_device(local_rank)
                for _ in range(n_fresh + n_skip):
                    gbk, netsk, progk = next(stream_it)
                    ready.put(ghn.plan(progk, gbk, netsk))
            th = threading.Thread(target=prefetch, daemon=True)
            th.start()
        n_fresh_pred = 0
        t_wait = t_enq = 0.0
        gpu_spans = []
        for k in range(n_fresh + n_skip):
            if k == n_skip:
                torch.cuda.synchronize()
                t_f = time.perf_counter()
                t_wait = t_enq = 0.0
            h0 = time.perf_counter()
            if threaded:
                pk = ready.get()
            else:
                gbk, netsk, progk = next(stream_it)
                pk = ghn.plan(progk, gbk, netsk)
            h1 = time.perf_counter()
            progk = pk.program
            ea, eb = L.Event(), L.Event()
            ea.record(stream)
            run_step(ghn, pk, None if fused_loss else torch.empty(progk.out_numel, dtype=torch.float32, device=dev),
                     progk.norm_ops(1.0) + (progk.norm_fin_ops(),))
            eb.record(stream)
            gpu_spans.append((ea, eb))
            h2 = time.perf_counter()
            t_wait, t_enq = t_wait + h1 - h0, t_enq + h2 - h1
            if k >= n_skip:
                n_fresh_pred += sum(p_['numel'] for p_ in progk.predicted)
            del pk
        torch.cuda.synchronize()
        dt_f = time.perf_counter() - t_f
        if threaded:
            th.join()
        pool.close()
        sys.setswitchinterval(old_switch)
        extras['fresh_graph_host_ms'] = {'wait_for_plan': 1e3 * t_wait / n_fresh, 'enqueue': 1e3 * t_enq / n_fresh,
                                         'loader_workers': n_workers, 'usable_cores': n_cores}
        extras['fresh_graph_gpu_ms'] = sum(a_.elapsed_ms(b_) for a_, b_ in gpu_spans[n_skip:]) / max(1, len(gpu_spans) - n_skip)
        extras['fresh_graph_ms_per_step'] = 1e3 * dt_f / n_fresh
        extras['fresh_graph_value'] = n_fresh_pred / dt_f
        # (c) what a TRAINING step adds to the benchmarked forward + backward: the optimizer pass over all GHN parameters
        # (clip_grad_norm_ + AdamW fused, trainer.py:356-381) and -- because the weights now change every step -- the
        # re-cast of their 16-bit copies ("shadows") in front of the next forward
        from ghn3_amd.optim import FusedAdamW
        opt = FusedAdamW(ghn, lr=1e-6, max_grad_norm=5.0)
        n_t = max(5, min(args.steps, 30))
        def train_loop(overlap):
            for _ in range(2):
                step()
                opt.step(plan.gflat, plan=plan, local_grads=True, overlap=overlap)
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(n_t):
                step()
                opt.step(plan.gflat, plan=plan, local_grads=True, overlap=overlap)
            opt.wait()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t_) / n_t
        t_serial = train_loop(False)
        t_t = train_loop(True)           # (the decoder half of AdamW on the side stream beside the next Graphormer forward)
        ea, eb = L.Event(), L.Event()
        ea.record(stream)
        for _ in range(n_t):
            opt.step(plan.gflat, plan=plan, local_grads=True)
        eb.record(stream)
        adam_ms = ea.elapsed_ms(eb) / n_t
        ghn.params_changed()

        def refresh_ms(ops):
            # (the copies of W2 are cast on the side stream: the main stream waits for it before the second event)
            a_, b_ = L.Event(), L.Event()
            a_.record(stream)
            for _ in range(n_t):
                ctx.run(ops, prog.problems, plan.bufs, stream)
                ctx.side_wait(stream)
            b_.record(stream)
            return a_.elapsed_ms(b_) / n_t
        rest_ms, full_ms = refresh_ms(prog.shadow_ops_rest), refresh_ms(prog.shadow_ops)
        ghn.params_changed()
        extras['train_step'] = {'ms_per_step': t_t, 'value': n_pred / (t_t * 1e-3), 'adamw_ms': adam_ms,
                                'ms_per_step_serial_optimizer': t_serial,
                                'shadow_refresh_ms': rest_ms, 'shadow_refresh_without_fusion_ms': full_ms,
                                'note': 'fwd + loss + bwd + fused clip / AdamW over %d GHN parameters, the decoder half of the '
                                        'update on the side stream beside the next forward\'s Graphormer chain (overlap=True; '
                                        'ms_per_step_serial_optimizer = the same loop with the whole update on the chain\'s '
                                        'stream); the update of '
                                        'decoder.conv.2.weight writes its 16-bit copies itself (adamw_ms includes that), '
                                        'shadow_refresh_ms = re-cast of the other copies in front of the next forward (run '
                                        'alone, side stream included), shadow_refresh_without_fusion_ms = all copies incl. W2'
                                        % int(ghn._flat_numel)}
        del opt
        # (d) the real per-GPU batches of BASELINE config 5 (ghn3xlm16, meta-batch 16 on 8 GPUs = 2 graphs per GPU; 4 = the
        # same meta-batch on 4 GPUs): same measurement as the headline line, more decoder rows per launch
        for gpg in (2, 4):
            if gpg == args.graphs_per_gpu:
                continue
            gb_b, nets_b = synthetic_batch([args.nodes] * gpg, args.nodes * 1000 + rank * gpg)
            plan_b = ghn.compile(nets_b, gb_b, training=True)
            prog_b = plan_b.program
            norms_b = prog_b.norm_ops(1.0) + (prog_b.norm_fin_ops(),)
            dout_b = torch.empty(prog_b.out_numel, dtype=torch.float32, device=dev)
            n_b = max(5, args.steps // 4)
            for _ in range(3):
                run_step(ghn, plan_b, dout_b, norms_b)
            torch.cuda.synchronize()
            ctx.profile(2)
            ctx.profile_read_tags(reset=True)
            t_b = time.perf_counter()
            for _ in range(n_b):
                run_step(ghn, plan_b, dout_b, norms_b)
            torch.cuda.synchronize()
            t_b = (time.perf_counter() - t_b) / n_b
            ctx.profile_read_tags(reset=True)
            ctx.profile(3)                               # (the kernels' own durations: serialised, as for the headline line)
            for _ in range(3):
                run_step(ghn, plan_b, dout_b, norms_b)
            torch.cuda.synchronize()
            tags_b = ctx.profile_read_tags(reset=True)
            ctx.profile(0)
            ea, eb = L.Event(), L.Event()
            ea.record(stream)
            for _ in range(n_b):
                ghn._run_forward(plan_b)
            eb.record(stream)
            fwd_b = ea.elapsed_ms(eb) / n_b
            dom_b = [prog_b.TAG_D3_FWD, prog_b.TAG_D3_DGRAD, prog_b.TAG_D3_WGRAD]
            fl_b = sum(prog_b.tag_flops.get(t, 0.0) for t in dom_b)
            ms_b = sum(tags_b.get(t, (0.0, 0))[0] for t in dom_b) / 3
            rows_b = prog_b.B * prog_b.N
            gfl_b = prog_b.Lyr * (24.0 * rows_b * prog_b.C ** 2 + 4.0 * prog_b.B * prog_b.N ** 2 * prog_b.C)
            dfl_b = sum(prog_b.tag_flops.get(t, 0.0) for t in (prog_b.TAG_D3_FWD, prog_b.TAG_D2_FWD, prog_b.TAG_D1_FWD))
            n_pred_b = sum(p_['numel'] for p_ in prog_b.predicted)
            extras['b%d' % gpg] = {
                'graphs_per_gpu': gpg, 'ms_per_step': 1e3 * t_b, 'value': n_pred_b / t_b, 'predicted_params': n_pred_b,
                'decoder_rows': int(prog_b.M),
                'roofline_frac': (fl_b / (ms_b * 1e-3) / 1e12 / PEAK_TFLOPS[args.compute]) if ms_b > 0 else None,
                'w2_family_ms': ms_b, 'wgrad_side_workgroups': int(getattr(prog_b, 'wgrad_cap', 0)),
                'forward': {'ms': fwd_b, 'frac_of_16bit_mfma_peak': (gfl_b + dfl_b) / (fwd_b * 1e-3) / 1e12 / 2500.0}}
            del plan_b, dout_b
            torch.cuda.empty_cache()
        # (e) the exact-fp32 configuration of the same workload
        if args.compute != 'f32':
            torch.manual_seed(0)
            g32 = GHN3(**model_cfg(args.model), compute='f32').to(dev)
            g32.train()
            p32 = g32.compile(nets, gb, training=True)
            n32 = p32.program.norm_ops(1.0) + (p32.program.norm_fin_ops(),)
            for _ in range(2):
                run_step(g32, p32, dout, n32)
            torch.cuda.synchronize()
            t32 = time.perf_counter()
            n_32 = max(5, args.steps // 3)
            for _ in range(n_32):
                run_step(g32, p32, dout, n32)
            torch.cuda.synchronize()
            t32 = (time.perf_counter() - t32) / n_32
            extras['f32_mode'] = {'ms_per_step': 1e3 * t32, 'value': n_pred / t32, 'dtype': 'f32',
                                  'note': 'exact fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32), same workload'}
            del g32, p32

    op_breakdown = None
    phases = None
    if args.profile_ops and rank == 0:
        ev = [L.Event() for _ in range(4)]
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        ev[0].record(stream)
        ghn._run_forward(plan)
        ev[1].record(stream)
        h1 = time.perf_counter()
        if fused_loss:
            ctx.run(fin_norm, prog.problems, plan.bufs, stream)
        else:
            ghn._fill_bufs(plan, out=plan.out, dout=dout)
            ctx.run(f_norm, prog.problems, plan.bufs, stream)
            ctx.run(b_norm, prog.problems, plan.bufs, stream)
        ev[2].record(stream)
        h2 = time.perf_counter()
        ghn._run_backward(plan, None if fused_loss else dout, norm_g=one if fused_loss else None)
        ev[3].record(stream)
        h3 = time.perf_counter()
        torch.cuda.synchronize()
        phases = {'forward': round(ev[0].elapsed_ms(ev[1]), 3), 'loss_norms': round(ev[1].elapsed_ms(ev[2]), 3),
                  'backward': round(ev[2].elapsed_ms(ev[3]), 3),
                  'host_enqueue': {'forward': round(1e3 * (h1 - h0), 3), 'loss_norms': round(1e3 * (h2 - h1), 3),
                                   'backward': round(1e3 * (h3 - h2), 3)}}
        ctx.profile(1)
        ctx.profile_read(reset=True)
        step()
        torch.cuda.synchronize()
        op_breakdown = ctx.profile_read(reset=True)
        ctx.profile(0)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        # roofline of the dominant kernel family: the decoder W2 GEMM (forward, dgrad, wgrad).  Algorithmic
        # FLOPs = 2*rows*8C*(o*i) per parameter group (only the W2 rows a group consumes), DESIGN.md 4.
        dom = [prog.TAG_D3_FWD, prog.TAG_D3_DGRAD, prog.TAG_D3_WGRAD]
        fl = sum(prog.tag_flops.get(t, 0.0) for t in dom)                   # per step
        ms = sum(tags_ser.get(t, (0.0, 0))[0] for t in dom) / n_ser          # per step, kernels alone (serialised pass)
        ms_in_step = sum(tags.get(t, (0.0, 0))[0] for t in dom) / n_inst       # per step, in the step's own schedule (instrumented steps)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_TFLOPS[args.compute]

        def per_kernel(tg, n):
            d = {}
            for t, (tms, cnt) in sorted(tg.items()):
                name = prog.TAG_NAMES.get(t, str(t))
                d[name] = {'ms_per_step': round(tms / n, 4), 'launch_groups_per_step': cnt // n}
                if t in prog.tag_flops and tms > 0:
                    d[name]['tflops'] = round(prog.tag_flops[t] / (tms / n * 1e-3) / 1e12, 2)
            return d
        detail, detail_in_step = per_kernel(tags_ser, n_ser), per_kernel(tags, n_inst)
        # whole path: algorithmic FLOPs of forward + backward (3 x the forward's: Graphormer 24 N C^2 + 4 N^2 C per layer,
        # decoders on consumed rows / positions only) over the step time
        rows_ = prog.B * prog.N
        g_fl_ = prog.Lyr * (24.0 * rows_ * prog.C ** 2 + 4.0 * prog.B * prog.N ** 2 * prog.C)
        d_fl_ = sum(prog.tag_flops.get(t, 0.0) for t in (prog.TAG_D3_FWD, prog.TAG_D2_FWD, prog.TAG_D1_FWD))
        step_fl = 3.0 * (g_fl_ + d_fl_)
        # HBM bytes per step of the same kernels from the PMC counters: they cannot be collected inside this process, so
        # the number comes from the committed rocprofv3 --pmc passes over this exact workload (tools/pmc_profile.sh,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE calibrated) -- and only from a file
        # that was measured on THIS code (`code_hash` = ghn3_amd.build.source_hash() of the kernels + host compiler); null
        # otherwise.
        traffic, traffic_src = None, None
        if (args.model, args.nodes, args.graphs_per_gpu, args.compute) == ('ghn3xlm16', 256, 1, 'f16'):
            import glob
            from ghn3_amd.build import source_hash
            cur = source_hash()
            for cand in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_traffic_xl_f16.json')), reverse=True):
                try:
                    with open(cand) as fh:
                        rec = json.load(fh)
                    if rec.get('code_hash') == cur:
                        traffic = float(rec['hbm_bytes_per_step'])
                        traffic_src = 'profiles/' + os.path.basename(cand)
                        break
                except Exception:
                    continue
            if traffic is None:
                traffic_src = 'no PMC pass of the current code (source hash %s) is committed: tools/gpu_round.sh' % cur
        out = {
            'metric': 'predicted-params/sec (GHN fwd+bwd), %s, %d-node graphs' % (args.model, args.nodes),
            'value': total_pred * args.steps / elapsed,
            'unit': 'predicted-params/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.compute, 'data': 'synthetic',
            **({'precision_note': 'plain bf16 decoder operands miss the 1e-3 parity gate (2.4e-3 forward at ghn3tm8); the f16 '
                                  'mode (same MFMA rate, power-of-two gradient scaling) is the gate-meeting 16-bit mode and '
                                  'serves BASELINE config 2'} if args.compute == 'bf16' else {}),
            'config': {'workload': '%s fwd+bwd, %d synthetic %d-node graph(s) per GPU (seed %d+), %d predicted '
                                   'params per GPU, loss = sum of Frobenius norms of the predicted tensors'
                                   % (args.model, args.graphs_per_gpu, args.nodes, args.nodes * 1000, n_pred),
                       'ghn_params': int(ghn._flat_numel), 'decoder_rows': int(prog.M),
                       'workspace_bytes': int(prog.ws_bytes),
                       # every uint8 zero-fill of the run (workspace, scalar buffer, weight shadows): the known store
                       # size the WRITE_SIZE counter is calibrated against (tools/pmc_traffic.py)
                       'zero_fill_bytes': int(plan.zero_fill_bytes +
                                              (ghn._shadow.numel() if ghn._shadow is not None else 0)),
                       'parallelism': 'dp%d' % world, 'index_mode': ghn.index_mode,
                       'loss': 'fused into the tile kernels' if fused_loss else 'streaming norm passes',
                       'grad_allreduce': (args.grad_allreduce if ddp else None)},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': traffic, 'traffic_unit': 'bytes per step',
                         'traffic_source': traffic_src,
                         'kernel': 'decoder W2 grouped GEMM (fwd + dgrad + wgrad), %s MFMA operands' % args.compute,
                         'measured': 'HIP events around each launch in %d extra steps with the side stream serialised into the '
                                     'chain (nothing co-runs, side-stream grid caps dropped): the kernels\' own durations; '
                                     '`kernels_in_step` = the same events in %d steps of the timed schedule itself (the weight '
                                     'gradient shares the chip with the Graphormer backward), run right behind the timed '
                                     'region, which carries no events' % (n_ser, n_inst),
                         'algorithmic_gflop_per_step': fl / 1e9, 'kernel_ms_per_step': ms, 'kernels': detail,
                         'kernel_ms_per_step_in_step': ms_in_step, 'kernels_in_step': detail_in_step,
                         'frac_in_step': (fl / (ms_in_step * 1e-3) / 1e12 / peak) if ms_in_step > 0 else None,
                         # the whole path against the same peak: algorithmic forward + backward FLOPs / step time
                         'frac_step': step_fl / (ms_per_step * 1e-3) / 1e12 / peak,
                         'algorithmic_gflop_fwd_bwd': step_fl / 1e9,
                         'wgrad_side_workgroups': int(getattr(prog, 'wgrad_cap', 0))},
        }
        # the number the north star is written on: Graphormer + decoder FORWARD, algorithmic FLOPs / forward time / 2.5 PF
        # (N = 1 with extras; also for the per-GPU batch of BASELINE config 5, 2 graphs per GPU)
        if 'forward' in extras:
            out['roofline']['frac_forward'] = extras['forward']['frac_of_16bit_mfma_peak']
            out['roofline']['forward_ms'] = extras['forward']['ms']
            if 'b2' in extras:
                out['roofline']['frac_forward_2_graphs_per_gpu'] = extras['b2']['forward']['frac_of_16bit_mfma_peak']
        out.update(extras)
        if op_breakdown is not None:
            out['op_breakdown_ms'] = {k: round(v[0], 3) for k, v in op_breakdown.items()}
            out['phase_ms'] = phases
        if not args.no_cpu_baseline and world == 1:      # (rank 0 at N = 1 only: the other ranks would idle)
            try:
                if args.cpu_sample_nodes:
                    out['cpu_baseline'] = cpu_baseline(args.model, args.cpu_sample_nodes, args.cpu_sample_nodes * 1000)
                else:
                    out['cpu_baseline'] = cpu_baseline(args.model, args.nodes, seeds, args.graphs_per_gpu)
            except Exception as e:                                   # never lose the GPU line
                out['cpu_baseline'] = {'value': None, 'error': repr(e)}
        try:                                   # RCCL's version banner (NCCL_DEBUG=VERSION) sits in the C stdio buffer:
            import ctypes                       # flush it first so that the JSON line is the last line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if ddp:
        beat('shutdown')
        dist.barrier()
        dist.destroy_process_group()
    if dog is not None:
        dog.stop()


if __name__ == '__main__':
    main()
