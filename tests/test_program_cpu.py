"""
CPU tests of the host-side compiler: the op programs emitted by ghn3_amd/program.py are executed by the
numpy interpreter in tests/program_interp.py (test infrastructure, same semantics as include/ghn3_hip.h)
and compared with the oracle.  This validates op order, workspace offsets, GEMM addressing modes, tile
descriptors and the complete backward program without a GPU.
"""

import numpy as np
import pytest
import torch

import recipe
from oracle import ghn3_ref as R
from program_interp import Interp
from util_parity import rel_l2

from ghn3_amd import GHN3, Graph, GraphBatch, _lib as L
from ghn3_amd.program import Program


def _build(cfg, seed, index_mode):
    oracle = R.GHN3Ref(**cfg, index_mode=index_mode)
    shapes = {k: tuple(v.shape) for k, v in oracle.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=seed).items()}
    oracle.load_state_dict(sd)
    hip = GHN3(**cfg, index_mode=index_mode)
    hip.load_state_dict(sd)
    return hip, oracle


def _run_program(hip, nets, gb, training=True, **prog_kw):
    gb._cat()
    cfg = dict(hid=hip.hid, heads=hip.heads, layers=hip.layers, num_classes=hip.num_classes,
               max_shape=hip.max_shape)
    prog = Program(cfg, gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets,
                   index_mode=hip.index_mode, training=training, **prog_kw)
    P = prog.P
    pflat = hip._flat.detach().numpy().copy().view(np.uint8)
    gflat = np.zeros(hip._flat_numel, dtype=np.float32).view(np.uint8)
    bufs = [None] * prog.n_bufs
    for k, off in enumerate(hip._offs):
        bufs[k] = pflat[4 * int(off):]
        bufs[P + k] = gflat[4 * int(off):]
    bufs[prog.xbuf(prog.X_WS)] = np.zeros(prog.ws_bytes, dtype=np.uint8)
    bufs[prog.xbuf(prog.X_IDX)] = prog.idx_blob.copy()
    bufs[prog.xbuf(prog.X_EDGES)] = np.ascontiguousarray(gb.edges.numpy()).view(np.uint8).reshape(-1)
    bufs[prog.xbuf(prog.X_OUT)] = np.zeros(prog.out_numel, dtype=np.float32).view(np.uint8)
    bufs[prog.xbuf(prog.X_DOUT)] = np.zeros(prog.out_numel, dtype=np.float32).view(np.uint8)
    bufs[prog.xbuf(prog.X_TOK)] = (0.02 * np.random.RandomState(0).standard_normal(prog.tok_floats)
                                   ).astype(np.float32).view(np.uint8)
    bufs[prog.xbuf(prog.X_SCAL)] = np.zeros(prog.scal_bytes, dtype=np.uint8)
    bufs[prog.xbuf(prog.X_GRADFLAT)] = gflat
    if prog.uses_shadow:
        bufs[prog.xbuf(prog.X_SHADOW)] = np.zeros(prog.shadow_layout(prog.C, prog.max_shape, prog.Lyr)['nbytes'], dtype=np.uint8)
    it = Interp(bufs)
    it.run(prog.shadow_ops, prog.problems)      # (GHN3._refresh_shadows: only when the parameters changed)
    it.run(prog.fwd_ops, prog.problems)
    return prog, it, bufs, gflat


def _tiny(case):
    specs = recipe.edge_specs(case) if case in recipe.EDGE_CASES else \
        recipe.EXTRA_CASES[case][0] if case in recipe.EXTRA_CASES else [recipe.TINY_NETS[i] for i in recipe.TINY_CASES[case]]
    nets_h = [recipe.build_torch_net(s) for s in specs]
    nets_o = [recipe.build_torch_net(s) for s in specs]
    gh, go = [], []
    for s in specs:
        nf, info, A = recipe.graph_arrays(s)
        gh.append(Graph(node_feat=nf, node_info=info, A=A))
        go.append(R.GraphRef(torch.from_numpy(nf), info, torch.from_numpy(A)))
    return nets_h, GraphBatch(gh, dense=True), nets_o, R.GraphBatchRef(go)


@pytest.mark.parametrize('case,index_mode', [('b1', 'reference'), ('b2', 'reference'), ('b2', 'correct'),
                                             ('big', 'reference'), ('big_b2', 'reference'), ('nonorm', 'reference'),
                                             ('noln', 'reference'), ('only1d', 'reference'), ('single', 'reference'),
                                             ('ragged3', 'reference'), ('ragged3', 'correct'), ('mixed1d', 'reference')])
def test_compiled_programs_reproduce_oracle_forward_and_backward(case, index_mode):
    cfg = dict(recipe.TINY_CFG, **(recipe.EXTRA_CASES[case][1] if case in recipe.EXTRA_CASES else {}))
    hip, oracle = _build(cfg, recipe.TINY_SEED, index_mode)
    nets_h, gb_h, nets_o, gb_o = _tiny(case)
    prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, layernorm=hip.layernorm, weight_norm=hip.weight_norm)
    out = bufs[prog.xbuf(prog.X_OUT)].view(np.float32)
    oracle.train()
    _, pred_o = oracle(nets_o, gb_o, keep_grads=True)
    assert len(pred_o) == len(prog.predicted)
    loss_o = 0
    for k, (ind, attr, m, t) in enumerate(pred_o):
        p = prog.predicted[k]
        assert p['node'] == ind and p['attr'] == attr and tuple(p['tile_shape']) == tuple(t.shape)
        got = out[p['offset']:p['offset'] + p['numel']].reshape(p['tile_shape'])
        ref = t.detach().numpy()
        if ref.ndim == 3:
            got, ref = got[:, 1:], ref[:, 1:]
        assert rel_l2(got, ref) < 1e-5, (k, attr, ref.shape, rel_l2(got, ref))
        q = t[:, 1:] if t.dim() == 3 else t
        loss_o = loss_o + torch.norm(q, p='fro')
    # backward: upstream gradient = d(sum of Frobenius norms)/d(out), random row of pos-enc excluded
    dout = bufs[prog.xbuf(prog.X_DOUT)].view(np.float32)
    for p in prog.predicted:
        v = out[p['offset']:p['offset'] + p['numel']].reshape(p['tile_shape']).astype(np.float64)
        g = np.zeros_like(v)
        if v.ndim == 3:
            g[:, 1:] = v[:, 1:] / np.linalg.norm(v[:, 1:])
        else:
            g = v / np.linalg.norm(v)
        dout[p['offset']:p['offset'] + p['numel']] = g.reshape(-1)
    gflat[:] = 0x7f                      # poison: the backward program must zero / overwrite everything
    hip._patch_grad_memsets(prog)
    it.run(prog.bwd_ops, prog.problems)
    loss_o.backward()
    po = dict(oracle.named_parameters())
    g32 = gflat.view(np.float32)
    for name, off in zip(prog.names, hip._offs):
        # (a parameter the batch does not use -- the 2-D decoder of a network without weights -- has no gradient in autograd
        # and an all-zero one here)
        ref = po[name].grad.numpy() if po[name].grad is not None else np.zeros(tuple(po[name].shape), dtype=np.float32)
        got = g32[int(off):int(off) + ref.size].reshape(ref.shape)
        # (the edge-MLP output bias has an analytically zero gradient: softmax is shift invariant)
        err = float(np.linalg.norm(got.astype(np.float64) - ref))
        assert err < 5e-5 * float(np.linalg.norm(ref)) + 1e-6, (name, err, float(np.linalg.norm(ref)))


@pytest.mark.parametrize('case', ['b2', 'noln'])
def test_split_k_planes_of_the_graphormer_gemms(case, monkeypatch):
    """Program.split_small(): ff.net.3 forward, ff.net.0 dgrad and to_qkv dgrad in two K halves whose second plane the
    consuming LayerNorm op adds (threshold lowered so that the tiny configuration takes the path the 384-wide
    models take); same forward / gradients as the oracle."""
    monkeypatch.setenv('GHN3_SPLIT_K2_MIN', '64')
    cfg = dict(recipe.TINY_CFG, **(recipe.EXTRA_CASES[case][1] if case in recipe.EXTRA_CASES else {}))
    hip, _ = _build(cfg, recipe.TINY_SEED, 'reference')
    nets_h, gb_h, _, _ = _tiny(case)
    prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, layernorm=hip.layernorm, weight_norm=hip.weight_norm)
    n_fwd = sum(1 for o in prog.fwd_ops if int(o['kind']) == L.OP_LAYERNORM_FWD and int(o['r'][6]['buf']) >= 0)
    n_bwd = sum(1 for o in prog.bwd_ops if int(o['kind']) == L.OP_LAYERNORM_BWD and int(o['r'][7]['buf']) >= 0)
    layers = cfg['layers']
    assert n_fwd == (layers if cfg['layernorm'] else layers - 1) and n_bwd == 2 * layers, (n_fwd, n_bwd)
    test_compiled_programs_reproduce_oracle_forward_and_backward(case, 'reference')


@pytest.mark.parametrize('fwd_ct,bwd_ct,tol_f,tol_g,case', [(L.CT_F16, L.CT_BF16, 1e-3, 1e-2, 'b2'),
                                                             (L.CT_F16, L.CT_F16, 1e-3, 1e-3, 'b2'),
                                                             (L.CT_BF16, L.CT_BF16, 8e-3, 1e-2, 'b2'),
                                                             (L.CT_F16, L.CT_F16, 1e-3, 1e-3, 'syn'),
                                                             (L.CT_F16, L.CT_F16, 1e-3, 2e-3, 'ragged3')])
def test_16bit_operand_pipeline_program(fwd_ct, bwd_ct, tol_f, tol_g, case):
    """The 16-bit decoder pipeline (GHN3_OP_CAST16 copies + GHN3_GEMM_OP16 problems, f16 forward / bf16 backward by
    default): program structure (offsets, k-map, padding, fused bias gradient) validated against the oracle with
    the rounding emulated by the interpreter.  Tolerances are the measured rounding noise, not fp32 parity."""
    if case == 'syn':
        # seeded synthetic graphs (many (o, i) groups -> several wgrad bands and ragged families), 1000 classes
        from util_parity import synthetic_case
        cfg = dict(max_shape=(128, 128, 16, 16), num_classes=1000, hid=128, heads=8, layers=1, weight_norm=True,
                   ve=True, layernorm=True)          # widths 32 < C / 2 stay 32 (nn.py:655-661): two bands
        hip, oracle = _build(cfg, recipe.TINY_SEED, 'reference')
        nets_h, gb_h, nets_o, gb_o = synthetic_case([40], 4400)
    else:
        hip, oracle = _build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')
        nets_h, gb_h, nets_o, gb_o = _tiny(case)
    prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, decoder_ctype=fwd_ct, decoder_bwd_ctype=bwd_ct)
    assert any(g['op16'] for g in prog.gemm_groups)
    if case == 'syn':
        assert len(prog.wgrad_bands) >= 2 and any(len(b['members']) > 2 for b in prog.wgrad_bands)
    assert sum(int(p['flags']) & L.GEMM_OP16 != 0 for p in prog.problems) >= 3
    out = bufs[prog.xbuf(prog.X_OUT)].view(np.float32)
    oracle.train()
    _, pred_o = oracle(nets_o, gb_o, keep_grads=True)
    loss_o = 0
    for k, (ind, attr, m, t) in enumerate(pred_o):
        p = prog.predicted[k]
        got = out[p['offset']:p['offset'] + p['numel']].reshape(p['tile_shape'])
        ref = t.detach().numpy()
        if ref.ndim == 3:
            got, ref = got[:, 1:], ref[:, 1:]
        assert rel_l2(got, ref) < tol_f, (k, attr, ref.shape, rel_l2(got, ref))
        q = t[:, 1:] if t.dim() == 3 else t
        loss_o = loss_o + torch.norm(q, p='fro')
    dout = bufs[prog.xbuf(prog.X_DOUT)].view(np.float32)
    for p, (_, _, _, t) in zip(prog.predicted, pred_o):
        v = t.detach().numpy().astype(np.float64)
        g = np.zeros_like(v)
        if v.ndim == 3:
            g[:, 1:] = v[:, 1:] / np.linalg.norm(v[:, 1:])
        else:
            g = v / np.linalg.norm(v)
        dout[p['offset']:p['offset'] + p['numel']] = g.reshape(-1)
    gflat[:] = 0x7f
    hip._patch_grad_memsets(prog)
    it.run(prog.bwd_ops, prog.problems)
    loss_o.backward()
    po = dict(oracle.named_parameters())
    g32 = gflat.view(np.float32)
    worst = 0.0
    for name, off in zip(prog.names, hip._offs):
        ref = po[name].grad.numpy()
        got = g32[int(off):int(off) + ref.size].reshape(ref.shape)
        err = float(np.linalg.norm(got.astype(np.float64) - ref))
        assert err < tol_g * float(np.linalg.norm(ref)) + 1e-6, (name, err, float(np.linalg.norm(ref)))
        if float(np.linalg.norm(ref)) > 1e-4:
            worst = max(worst, err / float(np.linalg.norm(ref)))
    print('worst gradient rel error', worst)


def test_param_norm_ops():
    hip, oracle = _build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')
    nets_h, gb_h, _, _ = _tiny('b1')
    prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h)
    f_ops, b_ops = prog.norm_ops(1.0)
    it.run(f_ops, prog.problems)
    out = bufs[prog.xbuf(prog.X_OUT)].view(np.float32)
    expect = sum(np.linalg.norm(out[p['offset']:p['offset'] + p['numel']].astype(np.float64))
                 for p in prog.predicted)
    loss = bufs[prog.xbuf(prog.X_SCAL)][:4].view(np.float32)[0]
    assert abs(loss - expect) < 1e-4 * expect
    it.run(b_ops, prog.problems)
    dout = bufs[prog.xbuf(prog.X_DOUT)].view(np.float32)
    p = prog.predicted[3]
    v = out[p['offset']:p['offset'] + p['numel']].astype(np.float64)
    np.testing.assert_allclose(dout[p['offset']:p['offset'] + p['numel']], v / np.linalg.norm(v), rtol=1e-5,
                               atol=1e-7)


@pytest.mark.parametrize('case,with_dout', [('b2', False), ('b2', True), ('big', False), ('mixed1d', False)])
def test_fused_param_norm_loss_program(case, with_dout):
    """The predicted-parameter-norm loss fused into the tile kernels (round 4): per-work-block sums of squares from
    GHN3_OP_TILE_FWD -> GHN3_OP_PARAM_NORM_FIN (norms, loss); GHN3_OP_TILE_BWD forms  dout + g p / ||p||  itself.  The flat
    gradient must equal the one of the streaming ops (PARAM_NORM_FWD / BWD writing a materialised dout), with and without an
    additional upstream gradient."""
    cfg = dict(recipe.TINY_CFG, **(recipe.EXTRA_CASES[case][1] if case in recipe.EXTRA_CASES else {}))
    g_w = 0.37
    grads = []
    for fused in (False, True):
        hip, _ = _build(cfg, recipe.TINY_SEED, 'reference')
        nets_h, gb_h, _, _ = _tiny(case)
        prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h)
        out = bufs[prog.xbuf(prog.X_OUT)].view(np.float32)
        dout = bufs[prog.xbuf(prog.X_DOUT)].view(np.float32)
        extra = (0.01 * np.random.RandomState(5).standard_normal(prog.out_numel)).astype(np.float32)
        expect = sum(np.linalg.norm(out[p['offset']:p['offset'] + p['numel']].astype(np.float64)) for p in prog.predicted)
        if fused:
            it.run(prog.norm_fin_ops(), prog.problems)
            loss = bufs[prog.xbuf(prog.X_SCAL)][:4].view(np.float32)[0]
            assert abs(loss - expect) < 1e-5 * expect
            bufs[prog.xbuf(prog.X_NORMG)] = np.asarray([g_w], dtype=np.float32).view(np.uint8)
            it.bufs = bufs
            r = prog.bwd_ops[prog.tile_bwd_op]['r']
            for slot, (buf, off) in prog.tile_bwd_refs.items():
                on = with_dout if slot == 0 else True
                r[slot]['buf'], r[slot]['off'] = (buf, off) if on else (-1, 0)
            if with_dout:
                dout[:] = extra
        else:
            f_ops, b_ops = prog.norm_ops(g_w)
            it.run(f_ops, prog.problems)
            it.run(b_ops, prog.problems)
            if with_dout:
                dout += extra
        gflat[:] = 0x7f
        hip._patch_grad_memsets(prog)
        it.run(prog.bwd_ops, prog.problems)
        grads.append(gflat.view(np.float32).copy())
    a, b = grads
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.linalg.norm(a.astype(np.float64) - b) < 1e-5 * np.linalg.norm(a.astype(np.float64))


def test_synthetic_program_and_counts():
    from ghn3_amd.synthetic import synthetic_batch
    gb, nets = synthetic_batch([40, 31], 777)
    gb._cat()
    cfg = dict(hid=32, heads=8, layers=2, num_classes=1000, max_shape=(32, 32, 16, 16))
    prog = Program(cfg, gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets)
    assert sum(p['numel'] for p in prog.predicted) == sum(n.num_params() for n in nets)
    assert prog.B == 2 and prog.N == 40
    # every GEMM operand obeys the ABI alignment rules
    for p in prog.problems:
        assert p['lda'] % 4 == 0 and p['ldb'] % 4 == 0 and p['A']['off'] % 16 == 0 and p['B']['off'] % 16 == 0


def test_predict_class_layers_false_and_reduce_graph():
    """nn.py:301-302 (fine-tuning: the classification layers are not predicted; their shape-embedding indices use the
    GHN's own class count, ppuda ShapeEncoder) and nn.py:684-690 (reduce_graph prunes modules the graph does not
    reference): forward vs the oracle."""
    hip, oracle = _build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')
    nets_h, gb_h, nets_o, gb_o = _tiny('b2')
    prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, training=False, predict_class_layers=False)
    out = bufs[prog.xbuf(prog.X_OUT)].view(np.float32)
    oracle.eval()
    with torch.no_grad():
        _, pred_o = oracle(nets_o, gb_o, predict_class_layers=False, assign=False)
    assert len(pred_o) == len(prog.predicted) and len(pred_o) > 0
    assert not any(p['attr'] in ('weight', 'bias') and p['module'] is nets_h[0].fc for p in prog.predicted)
    for k, (ind, attr, m, t) in enumerate(pred_o):
        p = prog.predicted[k]
        assert p['node'] == ind and p['attr'] == attr
        got = out[p['offset']:p['offset'] + p['numel']].reshape(p['tile_shape'])
        ref = t.numpy()
        if ref.ndim == 3:
            got, ref = got[:, 1:], ref[:, 1:]
        assert rel_l2(got, ref) < 1e-5, (k, attr)
    # reduce_graph: same predictions; a module that no graph node references loses its parameters
    nets_r, gb_r, _, _ = _tiny('b2')
    extra = torch.nn.Conv2d(4, 4, 3)
    nets_r[0].add_module('unused', extra)
    prog_r, it_r, bufs_r, _ = _run_program(hip, nets_r, gb_r, training=False, reduce_graph=True)
    prog_f, it_f, bufs_f, _ = _run_program(hip, *_tiny('b2')[:2], training=False)
    np.testing.assert_array_equal(bufs_r[prog_r.xbuf(prog_r.X_OUT)], bufs_f[prog_f.xbuf(prog_f.X_OUT)])
    assert extra.weight is None and extra.bias is None


@pytest.mark.parametrize('nodes,staged', [([30], False), ([22, 17], False), ([30], True), ([22, 17], True)])
def test_split_bf16_graphormer_program(nodes, staged, monkeypatch):
    """GHN3_GEMM_X3: the Graphormer linears (forward and dgrad) on split-bf16 operands against persistent hi / lo weight
    copies (GHN3_CAST_SPLIT).  Round-3 plan: K splits as partial planes summed by the LayerNorm ops.  Staged plan (round 4,
    the default): fragment-major copies (GHN3_CAST_FRAG), tile codes 44 / 45, no planes, every per-layer LayerNorm (forward
    and backward) a row prologue of the GEMM that consumes it.  Decoder exact fp32, so the only deviation from the oracle is
    the dropped lo.lo term (2^-16 relative): forward 2e-5, gradients 1e-4."""
    from util_parity import synthetic_case
    monkeypatch.setenv('GHN3_X3S', '1' if staged else '0')
    cfg = dict(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=2, weight_norm=True, ve=True,
               layernorm=True)
    hip, oracle = _build(cfg, recipe.TINY_SEED, 'reference')
    nets_h, gb_h, nets_o, gb_o = synthetic_case(nodes, 3100)
    prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, graphormer_x3=True)
    assert prog.x3 and prog.uses_shadow and prog.x3s == staged
    n_x3 = sum(int(p['flags']) & L.GEMM_X3 != 0 for p in prog.problems)
    assert n_x3 >= 8 * cfg['layers']                              # 4 forward + 4 dgrad linears per layer (+ K splits)
    ln_ops = [o for o in list(prog.fwd_ops) + list(prog.bwd_ops)
              if int(o['kind']) in (L.OP_LAYERNORM_FWD, L.OP_LAYERNORM_BWD)]
    if staged:
        # the final LayerNorm forward + backward and the LayerNorm-1 backward of layer 0 (its output feeds the embedding
        # backward, not a GEMM) are the only LayerNorm launches left
        assert len(ln_ops) == 3 and all(int(o['i'][2]) == 0 for o in ln_ops)
        assert n_x3 == 8 * cfg['layers']
        n_ln = sum(int(p['ln_kind']) != 0 for p in prog.problems if int(p['flags']) & L.GEMM_X3)
        assert n_ln == 2 * cfg['layers'] + 2 * cfg['layers'] - 1
        tiles = {int(o['i'][2]) for o in list(prog.fwd_ops) + list(prog.bwd_ops) if int(o['kind']) == L.OP_GEMM and
                 any(int(p['flags']) & L.GEMM_X3 for p in prog.problems[int(o['i'][0]):int(o['i'][0]) + int(o['i'][1])])}
        assert tiles <= {44, 45} and 45 in tiles           # (44 = the wide outputs of models with 3C > 512)
    else:
        n_planes = sum(1 for o in ln_ops if int(o['i'][2]) > 0)
        assert n_planes >= 2 * cfg['layers']                      # K-split planes are consumed by LayerNorm ops
    out = bufs[prog.xbuf(prog.X_OUT)].view(np.float32)
    oracle.train()
    _, pred_o = oracle(nets_o, gb_o, keep_grads=True)
    loss_o = 0
    dout = bufs[prog.xbuf(prog.X_DOUT)].view(np.float32)
    for k, (ind, attr, m, t) in enumerate(pred_o):
        p = prog.predicted[k]
        got = out[p['offset']:p['offset'] + p['numel']].reshape(p['tile_shape'])
        assert rel_l2(got, t.detach().numpy()) < 2e-5, (k, attr, rel_l2(got, t.detach().numpy()))
        loss_o = loss_o + torch.norm(t, p='fro')
        v = t.detach().numpy().astype(np.float64)
        dout[p['offset']:p['offset'] + p['numel']] = (v / np.linalg.norm(v)).reshape(-1)
    gflat[:] = 0x7f
    hip._patch_grad_memsets(prog)
    it.run(prog.bwd_ops, prog.problems)
    loss_o.backward()
    po = dict(oracle.named_parameters())
    g32 = gflat.view(np.float32)
    for name, off in zip(prog.names, hip._offs):
        ref = po[name].grad.numpy()
        got = g32[int(off):int(off) + ref.size].reshape(ref.shape)
        err = float(np.linalg.norm(got.astype(np.float64) - ref))
        assert err < 1e-4 * float(np.linalg.norm(ref)) + 2e-6, (name, err, float(np.linalg.norm(ref)))


def _simulate_side_state(runs):
    """Host model of ghn3_run's two-stream bookkeeping (runtime.hip: marks and the pending flag belong to the context and
    survive a DETACHed run).  Returns, per WAIT op, whether the mark it names was pending when it was issued, and the
    list of (run, op) positions of main-stream ops that were issued while un-waited side work of a MARKed branch existed."""
    mark_set = [False] * 4
    pending = False
    waits = []
    for r, ops in enumerate(runs):
        touches = any((int(o['flags']) & L.OPFLAG_SIDE) or int(o['kind']) in (L.OP_JOIN, L.OP_DETACH) for o in ops)
        side_dirty = touches and pending
        if touches:
            pending = False
        detach = False
        for k, o in enumerate(ops):
            kind = int(o['kind'])
            if kind == L.OP_JOIN:
                mode, mid = int(o['i'][0]), int(o['i'][1]) & 3
                if mode == 1:
                    if side_dirty:
                        mark_set[mid] = True
                elif mode == 2:
                    waits.append((r, k, mark_set[mid]))
                    mark_set[mid] = False
                else:
                    side_dirty = False
                    mark_set = [False] * 4
                continue
            if kind == L.OP_DETACH:
                detach = True
                continue
            detach = False
            if int(o['flags']) & L.OPFLAG_SIDE:
                side_dirty = True
        if detach and side_dirty:
            pending = True
        elif touches:
            mark_set = [False] * 4
    return waits


def test_mark_wait_pairs_survive_the_data_parallel_split():
    """The 1-D decoder backward runs on the side stream behind a MARK; the node-row gather that reads its rows WAITs for
    that mark.  In the data-parallel schedule the MARK lands in part 1 and the WAIT in part 3 (separate ghn3_run calls):
    the mark must still be pending there, in the single-run order and in the two-part order alike (round-3 advisor
    finding: marks used to be local to a run, so the split runs raced)."""
    hip, _ = _build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')
    nets_h, gb_h, _, _ = _tiny('b2')
    gb_h._cat()
    cfg = dict(hid=hip.hid, heads=hip.heads, layers=hip.layers, num_classes=hip.num_classes, max_shape=hip.max_shape)
    prog = Program(cfg, gb_h.node_info, gb_h.host_n_nodes(), gb_h._node_type_host, gb_h.max_edge, nets_h,
                   decoder_ctype=L.CT_F16, decoder_bwd_ctype=L.CT_F16)
    assert prog.n1 > 0 and prog.M > 0
    for k_run, runs in enumerate(([prog.bwd_ops], [p for p, _ in prog.bwd_parts], [prog.bwd_ops_a, prog.bwd_ops_b])):
        n_marks = sum(int(o['kind']) == L.OP_JOIN and int(o['i'][0]) == 1 for ops in runs for o in ops)
        waits = _simulate_side_state(runs)
        # (the data-parallel parts carry a second pair, slot 1: the operand copies in front of the weight gradient that is
        # issued first there -- it must not consume the 1-D decoder's mark on slot 0)
        want = 1 + int(k_run == 1 and prog.ddp_wgrad_first)
        assert n_marks == want and len(waits) == want, (n_marks, waits)
        assert all(ok for (_, _, ok) in waits), waits
    # the mark is in the first part, its wait in the last one
    parts = [p for p, _ in prog.bwd_parts]
    assert any(int(o['kind']) == L.OP_JOIN and int(o['i'][0]) == 1 for o in parts[0])
    assert any(int(o['kind']) == L.OP_JOIN and int(o['i'][0]) == 2 for o in parts[-1])
    # a run without side ops (the exchange's local passes on the shared context) does not consume the pending state
    passes = np.zeros(1, dtype=L.OP_DT)
    passes['kind'] = L.OP_WIRE_PACK
    waits = _simulate_side_state([parts[0], passes, parts[1], passes, parts[2]])
    assert len(waits) == 1 + int(prog.ddp_wgrad_first) and all(w[2] for w in waits)


@pytest.mark.parametrize('case,route,order', [('b2', 'dout', None), ('b2', 'norm', None), ('syn', 'norm', None),
                                              ('ragged3', 'dout', None), ('b2', 'norm', 'first'), ('syn', 'dout', 'first')])
def test_data_parallel_parts_give_the_gradients_of_the_single_run(case, route, order, monkeypatch):
    """Program.bwd_parts (round 5: the W2 weight gradient FIRST -- behind the tile backward and its operand copies, in front
    of the W2 dgrad -- so that the exchange of dW2, 69 % of the bytes, overlaps everything else of the backward): the three
    parts run one after the other write the gradient buffer of the single-run order bit for bit, on both tile-gradient
    routes (the route switch patches ops of part 1 through Program.ddp_index), and dW2 is complete after part 1."""
    if case == 'syn':
        from util_parity import synthetic_case
        cfg = dict(max_shape=(128, 128, 16, 16), num_classes=1000, hid=128, heads=8, layers=1, weight_norm=True,
                   ve=True, layernorm=True)
        mk = lambda: (_build(cfg, recipe.TINY_SEED, 'reference')[0],) + tuple(synthetic_case([40], 4400)[:2])
    else:
        mk = lambda: (_build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')[0],) + tuple(_tiny(case)[:2])
    grads = []
    # order = 'first' (GHN3_WGRAD_ORDER, the default when the weight gradient runs on the side stream -- forced here with
    # GHN3_WGRAD_MAIN=0: the tiny models would keep it on the chain's stream): the single-process order with the weight gradient
    # in front of the W2 dgrad -- the permuted program (and the op indices the route switch patches, and the zero-fill of the
    # kernel's sum-of-squares slots) must give the bits of the 'late' order
    for split in (False, True) if order is None else (False, 'reordered', True):
        hip, nets_h, gb_h = mk()
        reordered = split == 'reordered' or (split is True and bool(order))
        if order:
            monkeypatch.setenv('GHN3_WGRAD_MAIN', '0')
            monkeypatch.setenv('GHN3_WGRAD_ORDER', order if reordered else 'late')
        prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, decoder_ctype=L.CT_F16, decoder_bwd_ctype=L.CT_F16)
        monkeypatch.delenv('GHN3_WGRAD_ORDER', raising=False)
        monkeypatch.delenv('GHN3_WGRAD_MAIN', raising=False)
        assert prog.ddp_wgrad_first and len(prog.bwd_parts) == 3
        if order:
            a_, b_ = prog.wgrad_op_range
            tag = lambda o: (int(o['flags']) >> 16) & 0xff
            w_ops = [k for k, o in enumerate(prog.bwd_ops) if int(o['kind']) == L.OP_GEMM and tag(o) == prog.TAG_D3_WGRAD]
            d_ops = [k for k, o in enumerate(prog.bwd_ops) if int(o['kind']) == L.OP_GEMM and tag(o) == prog.TAG_D3_DGRAD]
            assert prog.wgrad_cap > 0 and prog.wgrad_order == (order if reordered else 'late')
            assert (min(w_ops) < min(d_ops)) == reordered, (w_ops, d_ops)
            if prog.grad_sumsq is not None:              # its zero-fill stays in front of the launch that fills the slots
                z = [k for k, o in enumerate(prog.bwd_ops) if int(o['kind']) == L.OP_MEMSET0 and
                     int(o['r'][0]['off']) == prog.grad_sumsq['ws_off']]
                assert len(z) == 1 and z[0] < min(w_ops)
        split = bool(split is True)
        it.run(prog.norm_fin_ops(), prog.problems)
        bufs[prog.xbuf(prog.X_NORMG)] = np.asarray([0.37], dtype=np.float32).view(np.uint8)
        dout = (1e-3 * np.random.RandomState(4).standard_normal(prog.out_numel)).astype(np.float32)
        bufs[prog.xbuf(prog.X_DOUT)] = dout.view(np.uint8)
        it.bufs = bufs
        r = prog.bwd_ops[prog.tile_bwd_op]['r']
        for slot, (buf, off) in prog.tile_bwd_refs.items():
            on = (route == 'dout') if slot == 0 else (route == 'norm')
            r[slot]['buf'], r[slot]['off'] = (buf, off) if on else (-1, 0)
        patched = [prog.tile_bwd_op] + prog.set_tile_route(route == 'norm')
        gflat[:] = 0x7f
        hip._patch_grad_memsets(prog)
        if not split:
            it.run(prog.bwd_ops, prog.problems)
        else:
            k = prog.memset_grad_op                              # (what GHN3._run_backward patches into part 1)
            prog.bwd_parts[0][0][k:k + 2] = prog.bwd_ops[k:k + 2]
            for k_ in patched:
                prog.bwd_parts[0][0][prog.ddp_index.get(k_, k_)] = prog.bwd_ops[k_]
            w2 = prog.slot['decoder.conv.2.weight']
            for n_part, (ops, slots) in enumerate(prog.bwd_parts):
                it.run(ops, prog.problems)
                if n_part == 0:
                    assert slots == [(w2, w2 + 1)]
                    w2_after_part1 = gflat.view(np.float32)[int(hip._offs[w2]):int(hip._offs[w2 + 1])].copy()
        grads.append(gflat.view(np.float32).copy())
    a = grads[0]
    assert np.isfinite(a).all() and all(np.array_equal(a, b) for b in grads[1:])
    w2 = prog.slot['decoder.conv.2.weight']
    assert np.array_equal(w2_after_part1, a[int(hip._offs[w2]):int(hip._offs[w2 + 1])]) and np.abs(w2_after_part1).max() > 0


@pytest.mark.parametrize('case', ['b2', 'syn', 'ragged3'])
def test_direct_16bit_tile_gradient_route(case):
    """Direct 16-bit tiles (round 4): with the fused norm loss alone GHN3_OP_TILE_BWD writes the scaled f16 operand copy of
    the tile gradient itself, the scale taken from the a-priori bound GHN3_OP_PARAM_NORM_FIN leaves (never exceeded -- the
    interpreter asserts it), and the transposed weight-gradient operands are re-laid out from that copy
    (GHN3_CAST_SRC16).  A power-of-two scale does not change f16 rounding, so every gradient equals the fp32 route's
    (measured maximum + cast passes) except decoder.conv.2.bias, whose column sums now add f16-rounded values."""
    if case == 'syn':
        from util_parity import synthetic_case
        cfg = dict(max_shape=(128, 128, 16, 16), num_classes=1000, hid=128, heads=8, layers=1, weight_norm=True,
                   ve=True, layernorm=True)
        mk = lambda: (_build(cfg, recipe.TINY_SEED, 'reference')[0],) + tuple(synthetic_case([40], 4400)[:2])
    else:
        mk = lambda: (_build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')[0],) + tuple(_tiny(case)[:2])
    grads = []
    for direct in (False, True):
        hip, nets_h, gb_h = mk()
        prog, it, bufs, gflat = _run_program(hip, nets_h, gb_h, decoder_ctype=L.CT_F16, decoder_bwd_ctype=L.CT_F16)
        assert prog.tile_bwd_h16 > 0 and prog.d16_on_idx and prog.d16_off_idx
        it.run(prog.norm_fin_ops(), prog.problems)
        bufs[prog.xbuf(prog.X_NORMG)] = np.asarray([0.37], dtype=np.float32).view(np.uint8)
        it.bufs = bufs
        r = prog.bwd_ops[prog.tile_bwd_op]['r']
        for slot, (buf, off) in prog.tile_bwd_refs.items():
            r[slot]['buf'], r[slot]['off'] = (buf, off) if slot != 0 else (-1, 0)
        touched = prog.set_tile_route(direct)
        assert sorted(touched) == sorted(prog.d16_on_idx + prog.d16_off_idx)
        kinds = [int(prog.bwd_ops[k]['kind']) for k in prog.d16_on_idx]
        assert all(k == (L.OP_CAST16 if direct else L.OP_NOP) for k in kinds)
        if direct:
            # (poison the fp32 tile gradient of the direct rows: nothing may read it on this route)
            ws = bufs[prog.xbuf(prog.X_WS)]
            d0 = prog._ws_names['d_tiles']
            ws[d0:d0 + 4 * prog.tiles_floats].view(np.float32)[:] = np.nan
        gflat[:] = 0x7f
        hip._patch_grad_memsets(prog)
        it.run(prog.bwd_ops, prog.problems)
        grads.append(gflat.view(np.float32).copy())
        if prog.grad_sumsq is not None:
            # GHN3_GEMM_SUMSQ: the weight-gradient problems leave the sum of the squares of dW2 in their slot table
            gs = prog.grad_sumsq
            slots = bufs[prog.xbuf(prog.X_WS)][gs['ws_off']:gs['ws_off'] + 4 * gs['count']].view(np.float32)
            k2 = prog.slot[gs['name']]
            w2g = grads[-1][int(hip._offs[k2]):int(hip._offs[k2 + 1])].astype(np.float64)
            assert abs(float(slots.astype(np.float64).sum()) - float((w2g ** 2).sum())) <= 1e-5 * float((w2g ** 2).sum())
        amax = bufs[prog.xbuf(prog.X_WS)][prog.r_amax[1]:prog.r_amax[1] + 4].view(np.float32)[0]
        print('direct' if direct else 'fp32 route', 'amax slot', amax)
    a, b = grads
    assert np.isfinite(a).all() and np.isfinite(b).all()
    offs = [int(o) for o in hip._offs] + [int(hip._flat_numel)]
    for k, name in enumerate(prog.names):
        x, y = a[offs[k]:offs[k + 1]].astype(np.float64), b[offs[k]:offs[k + 1]].astype(np.float64)
        tol = 1e-3 if name == 'decoder.conv.2.bias' else 2e-6
        assert np.linalg.norm(x - y) <= tol * np.linalg.norm(x) + 1e-9, (name, np.linalg.norm(x - y), np.linalg.norm(x))


def test_weight_gradient_schedule_of_the_bench_workload(monkeypatch):
    """Program._wgrad_schedule at ghn3xlm16 (the bench workload's shape, compiled on the CPU): the side stream with 128 workgroups
    and the order 'first' for 1 / 2 graphs of 256 nodes (profiles/r05s_ab_wgrad_first_order.txt); GHN3_WGRAD_ORDER=late brings the
    round-5a model back (a multiple of 8 between 64 and 224); a small model keeps the weight gradient on the chain's stream."""
    import bench
    from ghn3_amd import GHN3
    from ghn3_amd.synthetic import synthetic_batch
    for v in ('GHN3_WGRAD_ORDER', 'GHN3_WGRAD_CAP', 'GHN3_WGRAD_MAIN', 'GHN3_WGRAD_LATE'):
        monkeypatch.delenv(v, raising=False)

    def compile_(name, nodes, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hip = GHN3(**bench.model_cfg(name), compute='f16')
        gb, nets = synthetic_batch(nodes, 256000)
        gb._cat()
        cfg = dict(hid=hip.hid, heads=hip.heads, layers=hip.layers, num_classes=hip.num_classes, max_shape=hip.max_shape)
        prog = Program(cfg, gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets, index_mode=hip.index_mode,
                       training=True, **{k: v for k, v in hip.program_config().items() if k not in ('cfg', 'index_mode')})
        for k in env:
            monkeypatch.delenv(k, raising=False)
        return prog
    for nodes in ([256], [256, 256]):
        prog = compile_('ghn3xlm16', nodes)
        assert (prog.wgrad_order, prog.wgrad_cap) == ('first', 128), (nodes, prog.wgrad_order, prog.wgrad_cap)
        tag = lambda o: (int(o['flags']) >> 16) & 0xff
        w = [k for k, o in enumerate(prog.bwd_ops) if int(o['kind']) == L.OP_GEMM and tag(o) == prog.TAG_D3_WGRAD]
        d = [k for k, o in enumerate(prog.bwd_ops) if int(o['kind']) == L.OP_GEMM and tag(o) == prog.TAG_D3_DGRAD]
        assert prog.tile_bwd_op < min(w) < min(d)
        assert int(prog.bwd_ops[min(w)]['flags']) & L.OPFLAG_SIDE and (int(prog.bwd_ops[min(w)]['i'][3]) & 0xffff) == 128
    late = compile_('ghn3xlm16', [256], GHN3_WGRAD_ORDER='late')
    assert late.wgrad_order == 'late' and 64 <= late.wgrad_cap <= 224 and late.wgrad_cap % 8 == 0
    small = compile_('ghn3tm8', [64])
    assert small.wgrad_cap == 0 and small.wgrad_order != 'first'


def test_balanced_row_tiles_and_column_ranges():
    """Round 6 tilings of the 8-phase W2 kernels (host logic only): equal row tiles per streamed panel, dense column ranges cut at
    the members' output widths, the fine-grained dynamic program of the dgrad, and the rows every range / chunk covers."""
    from ghn3_amd.program import Program
    # 533 full-width rows: 2 x 288 (not 256 + 320); 660 / 672 rows: 3 x 224; 768: 3 x 256; small families: one skinny tile
    assert Program.balanced_tiles(533) == (2, 9) and Program.balanced_tiles(660) == (3, 7) and Program.balanced_tiles(768) == (3, 8)
    assert Program.balanced_tiles(109) == (1, 4) and Program.balanced_tiles(40) == (1, 2) and Program.balanced_tiles(256) == (1, 8)
    for rows in (1, 63, 64, 65, 191, 192, 193, 320, 321, 1421, 3273):
        t, n = Program.balanced_tiles(rows)
        assert n in Program.P8_HEIGHTS and t * 32 * n >= rows and (t - 1) * 32 * n < rows
    fam = dict(i_ld=384, rows=768, subs=[dict(o=384, rows=533), dict(o=128, rows=127), dict(o=64, rows=12), dict(o=32, rows=96)])
    rng = Program.column_ranges(fam)
    assert rng == [(0, 32, 768), (32, 64, 672), (64, 128, 660), (128, 384, 533)]
    assert sum((hi - lo) * fam['i_ld'] * alive for lo, hi, alive in rng) == sum(sb['o'] * fam['i_ld'] * sb['rows'] for sb in fam['subs'])
    # a narrow range is merged into the next wider one (its rows compute a few don't-care columns)
    fam2 = dict(i_ld=8, rows=100, subs=[dict(o=64, rows=60), dict(o=16, rows=40)])
    assert Program.column_ranges(fam2) == [(0, 64, 100)]
    # row tables: equal tiles over the alive rows, 320-row zero-fill tiles behind them, extents shifted into the K chunk
    ext = np.asarray([147456] * 533 + [49152] * 127 + [24576] * 12 + [12288] * 96)
    mt = Program.range_tiles(533, 768, lambda r: ext[r], k0=49152 + 1024, kc=4096)
    assert mt.tolist() == [[0, 9, 4096], [288, 9, 4096], [576, 10, 0]]
    mt = Program.range_tiles(768, 768, lambda r: ext[r], k0=0, kc=12288)
    assert mt[:, 1].tolist() == [8, 8, 8] and mt[:, 2].tolist() == [12288, 12288, 12288]
    # the dgrad's dynamic program on 32-row positions never pays more than the 64-row one of rounds 3-5
    fine = Program.row_tiles(ext)
    assert fine[:, 0].tolist() == sorted(fine[:, 0].tolist()) and int(fine[-1, 0]) + 32 * int(fine[-1, 1]) >= 768
    cost = lambda t: sum(Program.P8_COSTN[int(n)] * float(e) for _, n, e in t)
    old = [(0, 8, 147456), (256, 10, 147456), (576, 6, 49152)]
    assert cost(fine) <= cost(old) + 1e-6
    covered = np.zeros(768, bool)
    for m0, n, e in fine:
        assert not covered[m0:m0 + 32 * n].any()
        covered[m0:m0 + 32 * n] = True
        assert int(ext[m0]) == int(e)                  # a tile pays for (and is cut at) the extent of its first row
    assert covered.all()


def test_precompiled_program_travels_with_its_batch():
    """GraphBatch.precompile (a loader worker's half of GHN3.compile): the stripped program pickles with the batch and its
    networks, still names the modules of the unpickled networks, carries the same run-time arrays as a program compiled in
    place, and is handed out once and only for the arguments it was built with."""
    import pickle
    from ghn3_amd.deepnets1m import SampledNets
    hip, _ = _build(recipe.TINY_CFG, recipe.TINY_SEED, 'reference')
    config = hip.program_config()

    def batch():
        src = SampledNets(seed=3, max_nodes=120)
        gb = GraphBatch([src[k] for k in range(2)], dense=True)
        gb._cat()
        gb.graphs = None
        return gb

    gb = batch().precompile(config, training=True, reduce_graph=True)
    gb2 = pickle.loads(pickle.dumps(gb))
    args = dict(config, training=True, predict_class_layers=True, reduce_graph=True)
    with pytest.raises(ValueError, match='training'):            # (consumed shape tables: no silent recompile)
        gb2.take_program(gb2.nets, **dict(args, training=False))
    prog = gb2.take_program(gb2.nets, **args)
    assert prog is not None and gb2.take_program(gb2.nets, **args) is None
    mods = {id(m) for net in gb2.nets for _, m in net.named_modules()}
    assert len(prog.predicted) > 0 and all(id(p['module']) in mods for p in prog.predicted)
    gb_f = batch()
    ref = Program(config['cfg'], gb_f.node_info, gb_f.host_n_nodes(), gb_f._node_type_host, gb_f.max_edge, gb_f.nets,
                  **{k: v for k, v in args.items() if k != 'cfg'})
    np.testing.assert_array_equal(prog.idx_blob, ref.idx_blob)
    np.testing.assert_array_equal(prog.problems, ref.problems)
    for a, b in ((prog.fwd_ops, ref.fwd_ops), (prog.bwd_ops, ref.bwd_ops)):
        assert a.tobytes() == b.tobytes()
    assert [(p['offset'], p['numel'], p['attr']) for p in prog.predicted] == \
        [(p['offset'], p['numel'], p['attr']) for p in ref.predicted]
    # a batch compiled without reduce_graph can fall back to a fresh compile: no program, no error
    gb_n = batch().precompile(config, training=False)
    assert gb_n.take_program(gb_n.nets, **dict(args, reduce_graph=False)) is None
