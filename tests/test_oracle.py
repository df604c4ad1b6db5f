"""
Pins the CPU oracle (oracle/) against golden vectors generated from the reference
(tests/golden/make_golden.py).  CPU only.
"""

import os
import numpy as np
import pytest
import torch

import recipe
from oracle import ppuda_base
from oracle import graphormer_ref as G
from oracle import ghn3_ref as R

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _np(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_param_counts_match_reference_printout():
    # examples/ghn_all_pytorch.ipynb:109 prints 654365184 for ghn3xlm16
    assert recipe.count_params(384, 24, 16, 1000) == 654365184
    m = R.GHN3Ref(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3,
                  weight_norm=True, ve=True, layernorm=True)
    assert sum(p.numel() for p in m.parameters()) == recipe.count_params(64, 3, 8, 1000) == 6906632


@pytest.mark.parametrize('tag', ['c32', 'c48'])
def test_graphormer_layers_match_reference(tag):
    g = _np('graphormer_layers.npz')
    C, H, N, B = [int(v) for v in g[tag + '/cfg']]
    shapes0 = {
        'ln1.weight': (C,), 'ln1.bias': (C,), 'ln2.weight': (C,), 'ln2.bias': (C,),
        'attn.to_qkv.weight': (3 * C, C), 'attn.to_out.0.weight': (C, C), 'attn.to_out.0.bias': (C,),
        'ff.net.0.weight': (4 * C, C), 'ff.net.0.bias': (4 * C,), 'ff.net.3.weight': (C, 4 * C),
        'ff.net.3.bias': (C,)}
    shapes_l0 = dict(shapes0)
    shapes_l0.update({'attn.edge_embed.embed.weight': (257, C), 'attn.proj_e.0.weight': (C, 2 * C),
                      'attn.proj_e.0.bias': (C,), 'attn.proj_e.2.weight': (H, C), 'attn.proj_e.2.bias': (H,),
                      'centrality_embed_in.weight': (101, C), 'centrality_embed_out.weight': (101, C),
                      'input_dist_embed.weight': (1001, C)})
    p0 = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes_l0, seed=100 * C).items()}
    p1 = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes0, seed=100 * C + 1).items()}
    x = torch.from_numpy(g[tag + '/x'])
    A = torch.from_numpy(g[tag + '/A'].astype(np.int64))
    n_nodes = g[tag + '/n_nodes']
    nm = torch.zeros(B, N, 1, dtype=torch.bool)
    for b in range(B):
        nm[b, :n_nodes[b]] = True
    mask = nm & nm.permute(0, 2, 1)
    y0, bias = G.transformer_layer(x, A, mask, p0, '', H, layer0=True)
    y1, _ = G.transformer_layer(y0, bias, mask, p1, '', H, layer0=False)
    np.testing.assert_allclose(bias.numpy(), g[tag + '/bias'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(y0.numpy(), g[tag + '/y0'], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(y1.numpy(), g[tag + '/y1'], rtol=1e-5, atol=2e-6)


def _tiny_model(index_mode='reference'):
    m = R.GHN3Ref(**recipe.TINY_CFG, index_mode=index_mode)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = recipe.seeded_state_dict(shapes, seed=recipe.TINY_SEED)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m, shapes


def _tiny_batch(case):
    specs = [recipe.TINY_NETS[i] for i in recipe.TINY_CASES[case]]
    nets = [recipe.build_torch_net(s) for s in specs]
    graphs = []
    for s in specs:
        nf, info, A = recipe.graph_arrays(s)
        graphs.append(R.GraphRef(torch.from_numpy(nf), info, torch.from_numpy(A)))
    return nets, R.GraphBatchRef(graphs)


def test_state_dict_layout_matches_reference():
    g = _np('ghn3_tiny.npz')
    m, shapes = _tiny_model()
    assert sorted(shapes) == [str(k) for k in g['meta/state_keys']]
    assert [str(shapes[k]) for k in sorted(shapes)] == [str(s) for s in g['meta/state_shapes']]
    assert sum(p.numel() for p in m.parameters()) == int(g['meta/n_params'][0])


@pytest.mark.parametrize('case', ['b1', 'b2', 'b2r'])
def test_full_forward_and_grads_match_reference(case):
    g = _np('ghn3_tiny.npz')
    m, _ = _tiny_model()
    nets, batch = _tiny_batch(case)
    m.train()
    nets, predicted, emb = m(nets, batch, return_embeddings=True, keep_grads=True)
    np.testing.assert_allclose(emb.detach().numpy(), g[case + '/emb'], rtol=2e-5, atol=2e-6)
    loss = 0
    n_checked = 0
    for b, net in enumerate(nets):
        for name, p in recipe.named_predicted(net):
            ref = g['%s/pred/%d/%s' % (case, b, name)]
            assert tuple(p.shape) == ref.shape, name
            q, r = (p[:, 1:], ref[:, 1:]) if p.dim() == 3 else (p, ref)   # Q3: random class-token row
            np.testing.assert_allclose(q.detach().numpy(), r, rtol=2e-5, atol=2e-6, err_msg=name)
            loss = loss + torch.norm(q, p='fro')
            n_checked += 1
    assert n_checked == sum(1 for k in g.files if k.startswith(case + '/pred/'))
    assert abs(loss.item() - float(g[case + '/loss'][0])) < 1e-4 * abs(loss.item())
    loss.backward()
    for k, p in m.named_parameters():
        ref = g['%s/grad/%s' % (case, k)]
        got = p.grad
        assert got is not None, k
        idx = recipe.sample_indices(got.numel(), 8, seed=len(k))
        mine = np.concatenate([[got.norm().item(), got.sum().item()], got.reshape(-1)[idx].numpy()])
        np.testing.assert_allclose(mine, ref, rtol=2e-4, atol=2e-5 * max(1.0, ref[0]), err_msg=k)


@pytest.mark.parametrize('case', sorted(recipe.EXTRA_CASES))
def test_extra_cases_match_reference(case):
    """Kernels larger than the 16x16 decoder grid (bilinear branch, nn.py:751-753) and the weight_norm=False /
    layernorm=False configurations: oracle vs the goldens written by the reference GHN3 class (make_golden.py extra)."""
    g = _np('ghn3_tiny_extra.npz')
    specs, over = recipe.EXTRA_CASES[case]
    m = R.GHN3Ref(**dict(recipe.TINY_CFG, **over))
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sorted(shapes) == [str(k) for k in g[case + '/state_keys']]
    m.load_state_dict({k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=recipe.TINY_SEED).items()})
    nets = [recipe.build_torch_net(s) for s in specs]
    graphs = []
    for s in specs:
        nf, info, A = recipe.graph_arrays(s)
        graphs.append(R.GraphRef(torch.from_numpy(nf), info, torch.from_numpy(A)))
    m.train()
    nets, predicted, emb = m(nets, R.GraphBatchRef(graphs), return_embeddings=True, keep_grads=True)
    np.testing.assert_allclose(emb.detach().numpy(), g[case + '/emb'], rtol=2e-5, atol=2e-6)
    loss, n_checked = 0, 0
    for b, net in enumerate(nets):
        for name, p in recipe.named_predicted(net):
            ref = g['%s/pred/%d/%s' % (case, b, name)]
            assert tuple(p.shape) == ref.shape, name
            q, r = (p[:, 1:], ref[:, 1:]) if p.dim() == 3 else (p, ref)
            np.testing.assert_allclose(q.detach().numpy(), r, rtol=2e-5, atol=2e-6, err_msg=name)
            loss = loss + torch.norm(q, p='fro')
            n_checked += 1
    assert n_checked == sum(1 for k in g.files if k.startswith(case + '/pred/'))
    assert abs(loss.item() - float(g[case + '/loss'][0])) < 1e-4 * abs(loss.item())
    loss.backward()
    for k, p in m.named_parameters():
        ref = g['%s/grad/%s' % (case, k)]
        got = p.grad
        assert got is not None, k
        idx = recipe.sample_indices(got.numel(), 8, seed=len(k))
        mine = np.concatenate([[got.norm().item(), got.sum().item()], got.reshape(-1)[idx].numpy()])
        np.testing.assert_allclose(mine, ref, rtol=2e-4, atol=2e-5 * max(1.0, ref[0]), err_msg=k)


def test_q1_index_modes_differ_only_after_a_shorter_graph():
    m_ref, _ = _tiny_model('reference')
    m_cor, _ = _tiny_model('correct')
    for case, differs in (('b1', False), ('b2r', False), ('b2', True)):
        nets_a, batch = _tiny_batch(case)
        nets_b, _ = _tiny_batch(case)
        with torch.no_grad():
            _, pa = m_ref(nets_a, batch, assign=False)
            _, pb = m_cor(nets_b, batch, assign=False)
        # b2r = [long, short]: graph 0 has n_0 == N_max, so dense-flat == sparse-flat for every node
        diff = max(float((a[3] - b[3]).abs().max()) for a, b in zip(pa, pb) if a[3].dim() != 3)
        assert (diff > 1e-6) == differs, (case, diff)


def test_tile_and_normalize_cases_match_reference():
    g = _np('tile_cases.npz')
    for i, (src, tgt) in enumerate(recipe.TILE_CASES):
        w = torch.from_numpy(g['%d/w' % i])
        gen = torch.Generator().manual_seed(11)
        t = R.tile_params(w, tgt, gen=gen)
        ref = g['%d/tiled' % i]
        assert tuple(t.shape) == ref.shape == tuple(tgt), (i, t.shape, ref.shape, tgt)
        if len(tgt) == 3 and len(src) == 4:
            t, ref = t[:, 1:], ref[:, 1:]
        np.testing.assert_array_equal(t.numpy(), ref)
        for is_w in (0, 1):
            n = R.normalize(torch.from_numpy(g['%d/tiled' % i]), bool(is_w)).numpy()
            np.testing.assert_allclose(n, g['%d/norm%d' % (i, is_w)], rtol=1e-6, atol=1e-7)


def test_group_keys():
    ms = (384, 384, 16, 16)
    assert R.group_key((64, 3, 7, 7), ms, False, False) == (64, 4, 7, 7)
    assert R.group_key((512, 256, 3, 3), ms, False, False) == (384, 384, 3, 3)
    assert R.group_key((96, 96, 1, 1), ms, False, False) == (128, 128, 1, 1)
    assert R.group_key((1000, 2048), ms, True, False) == (384, 384)
    assert R.group_key((1000,), ms, False, True) == (384, -1)
    assert R.group_key((1, 197, 768), ms, False, False) == (1, 768, 14, 14)


def test_oracle_matches_reference_on_resnet18_ghn3tm8():
    """The oracle at a released size on a real architecture: ghn3tm8 on the torchvision.resnet18-shaped graph vs the
    golden written by the reference GHN3 class (make_golden.py resnet)."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'resnet18_ghn3tm8.npz'))
    hid, layers, heads = recipe.VARIANTS['ghn3tm8']
    cfg = dict(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
               weight_norm=True, ve=True, layernorm=True)
    oracle = R.GHN3Ref(**cfg)
    shapes = {k: tuple(v.shape) for k, v in oracle.state_dict().items()}
    sd = recipe.seeded_state_dict(shapes, seed=recipe.RESNET_SEED)
    oracle.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    oracle.eval()
    spec = recipe.resnet_spec(18)
    net = recipe.build_torch_net(spec)
    nf, info, A = recipe.graph_arrays(spec)
    gb = R.GraphBatchRef([R.GraphRef(torch.from_numpy(nf), info, torch.from_numpy(A))])
    with torch.no_grad():
        _, pred, emb = oracle([net], gb, return_embeddings=True)
    assert np.linalg.norm(emb.numpy() - gold['emb']) < 1e-5 * np.linalg.norm(gold['emb'])
    total = 0
    mod_name = {id(m): n for n, m in net.named_children()}
    for (ind, attr, m, t) in pred:
        name = '%s.%s' % (mod_name[id(m)], attr)
        v = t.detach().reshape(-1)
        total += v.numel()
        idx = recipe.sample_indices(v.numel(), recipe.RESNET_SAMPLES, seed=len(name))
        ref = gold['pred/%s/sample' % name]
        err = np.linalg.norm(v[idx].numpy() - ref) / (np.linalg.norm(ref) + 1e-30)
        assert err < 1e-5, (name, err)
    assert total == int(gold['meta/n_predicted'][0]) == 11689512
