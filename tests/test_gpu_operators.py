"""GPU tests of the operator-level API: ``TransformerLayer.forward(x, edges, mask)`` and
``ConvDecoder3.forward(x, max_shape, class_pred)`` run the HIP ops through the C ABI and are compared with golden
outputs of the UNMODIFIED reference layers (tests/golden/graphormer_layers.npz) and with the oracle's restatement of
nn.py:735-762."""

import os

import numpy as np
import pytest
import torch

import recipe
from util_parity import rel_l2, make_models

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('tag', ['c32', 'c48'])
def test_transformer_layer_forward_matches_reference_goldens(tag):
    """Layer 0 (centrality / input-distance embeddings, edge bias MLP, attention, FFN) and a plain layer, called like the
    reference calls them: ``y0, bias, mask = layer0(x, A, mask); y1 = layer1(y0, bias, mask)``."""
    from ghn3_amd import GHN3
    g = np.load(os.path.join(GOLD, 'graphormer_layers.npz'))
    C, H, N, B = [int(v) for v in g[tag + '/cfg']]
    ghn = GHN3(max_shape=(C, C, 16, 16), num_classes=10, hid=C, heads=H, layers=2, layernorm=True, ve=True,
               weight_norm=True)
    sd = ghn.state_dict()
    for li in range(2):
        names = [k[len('gnn.%d.' % li):] for k in sd if k.startswith('gnn.%d.' % li)]
        shapes = {k: tuple(sd['gnn.%d.%s' % (li, k)].shape) for k in names}
        seeded = recipe.seeded_state_dict(shapes, seed=100 * C + li)
        for k, v in seeded.items():
            sd['gnn.%d.%s' % (li, k)] = torch.from_numpy(v)
    ghn.load_state_dict(sd)
    ghn = ghn.to('cuda').eval()
    assert ghn.gnn[0].return_edges and not ghn.gnn[1].return_edges
    x = torch.from_numpy(g[tag + '/x']).cuda()
    A = torch.from_numpy(g[tag + '/A'].astype(np.int64)).cuda()
    nm = torch.zeros(B, N, 1, dtype=torch.bool)
    for b, n in enumerate(g[tag + '/n_nodes']):
        nm[b, :int(n)] = True
    mask = (nm & nm.permute(0, 2, 1)).cuda()
    with torch.no_grad():
        y0, bias, m_out = ghn.gnn[0](x.clone(), A, mask)
        y1 = ghn.gnn[1](y0, bias, mask)
        y1_seq = ghn.gnn(x.clone(), A, mask)                     # SequentialMultipleInOut threads the tuple
    torch.cuda.synchronize()
    assert m_out is mask
    valid = mask.cpu().numpy()
    # (the reference leaves don't-care values in the bias of padded pairs; compare where both nodes exist)
    bias_ref, bias_got = g[tag + '/bias'], bias.cpu().numpy()
    assert rel_l2(bias_got[valid], bias_ref[valid]) < 1e-5
    rows = nm[:, :, 0].numpy()
    assert rel_l2(y0.cpu().numpy()[rows], g[tag + '/y0'][rows]) < 2e-5
    assert rel_l2(y1.cpu().numpy()[rows], g[tag + '/y1'][rows]) < 2e-5
    assert torch.equal(y1, y1_seq)


@pytest.mark.parametrize('max_shape,class_pred', [((32, 32, 3, 3), False), ((16, 8, 1, 1), False),
                                                  ((32, 20, 7, 5), False), ((32, 32, 16, 16), False),
                                                  ((10, 24, 1, 1), True)])
def test_conv_decoder3_forward_matches_oracle(max_shape, class_pred):
    """ghn.decoder(x, max_shape, class_pred) vs the oracle's conv_decoder3 (nn.py:735-762: centre crop, row-subset of
    conv.2, classifier head over the out axis)."""
    from oracle import ghn3_ref as R
    hip, oracle = make_models(recipe.TINY_CFG, recipe.TINY_SEED)
    C = recipe.TINY_CFG['hid']
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(5, C, generator=gen)
    p = dict(oracle.named_parameters())
    with torch.no_grad():
        ref = R.conv_decoder3(x, p, oracle.max_shape, max_shape, class_pred)
        got = hip.decoder(x.cuda(), max_shape, class_pred)
    torch.cuda.synchronize()
    assert tuple(got.shape) == tuple(ref.shape), (got.shape, ref.shape)
    assert rel_l2(got.cpu().numpy(), ref.numpy()) < 2e-5


def test_transformer_layer_backward_matches_oracle_autograd():
    """``y0, bias, m = layer0(x, A, mask); y1 = layer1(y0, bias, mask); loss.backward()`` -- the reference's layers are
    ordinary autograd modules (graphormer.py:208-248).  Gradients w.r.t. the input, every parameter of both layers
    (incl. the centrality / input-distance tables and the edge-bias MLP of layer 0, reached through the bias layer 1
    re-adds) against torch autograd of the oracle restatement on the same weights."""
    from ghn3_amd import GHN3
    from oracle import graphormer_ref as G
    C, H, N, B = 32, 4, 20, 2
    torch.manual_seed(3)
    ghn = GHN3(max_shape=(C, C, 16, 16), num_classes=10, hid=C, heads=H, layers=2, layernorm=True, ve=True,
               weight_norm=True).to('cuda')
    with torch.no_grad():
        for n_, p_ in ghn.named_parameters():
            if n_.startswith('gnn.'):
                p_.copy_(torch.randn_like(p_) * (0.3 if p_.dim() > 1 else 0.1) + (1.0 if n_.endswith('ln1.weight') or n_.endswith('ln2.weight') else 0.0))
    ghn.params_changed()
    gen = torch.Generator().manual_seed(5)
    n_nodes = [20, 13]
    A = torch.zeros(B, N, N, dtype=torch.int64)
    for b in range(B):
        for i in range(n_nodes[b]):
            for j in range(i + 1, n_nodes[b]):
                A[b, i, j] = min(j - i, 6) if (j - i) % 3 != 2 else 0
    nm = torch.zeros(B, N, 1, dtype=torch.bool)
    for b, n in enumerate(n_nodes):
        nm[b, :n] = True
    mask = nm & nm.permute(0, 2, 1)
    x0 = torch.randn(B, N, C, generator=gen)
    wgt = torch.randn(B, N, C, generator=gen) * nm.float()             # loss weights (padded rows do not count)
    # HIP operators
    x = x0.clone().cuda().requires_grad_(True)
    y0, bias, m_out = ghn.gnn[0](x, A.cuda(), mask.cuda())
    y1 = ghn.gnn[1](y0, bias, m_out)
    (y1 * wgt.cuda()).sum().backward()
    torch.cuda.synchronize()
    # oracle + torch autograd
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in ghn.named_parameters() if k.startswith('gnn.')}
    xo = x0.clone().requires_grad_(True)
    z0, b0 = G.transformer_layer(xo, A, mask, p, 'gnn.0.', H, True)
    z1, _ = G.transformer_layer(z0, b0, mask, p, 'gnn.1.', H, False)
    (z1 * wgt).sum().backward()
    assert rel_l2((y1.detach().cpu() * nm.float()).numpy(), (z1.detach() * nm.float()).numpy()) < 2e-5
    valid = nm[:, :, 0]
    assert rel_l2(x.grad.cpu()[valid].numpy(), xo.grad[valid].numpy()) < 5e-5
    named = dict(ghn.named_parameters())
    for k, v in p.items():
        assert named[k].grad is not None, k
        err = float((named[k].grad.cpu().double() - v.grad.double()).norm())
        # (a constant added to every score of a softmax row changes nothing: the exact gradient of the last edge-MLP bias
        # is zero, both sides hold rounding noise of the ~1e-1-sized terms that cancel)
        atol = 1e-4 if k.endswith('proj_e.2.bias') else 1e-6
        assert err < 1e-4 * float(v.grad.norm()) + atol, (k, err, float(v.grad.norm()))


@pytest.mark.parametrize('max_shape,class_pred', [((32, 32, 3, 3), False), ((16, 8, 1, 1), False), ((32, 20, 7, 5), False),
                                                  ((10, 24, 1, 1), True)])
def test_conv_decoder3_backward_matches_oracle_autograd(max_shape, class_pred):
    """``ghn.decoder(x, max_shape, class_pred)`` is differentiable like the reference's module (nn.py:735-762): gradients
    w.r.t. the node embeddings and the decoder parameters against torch autograd of the oracle."""
    from oracle import ghn3_ref as R
    hip, oracle = make_models(recipe.TINY_CFG, recipe.TINY_SEED)
    C = recipe.TINY_CFG['hid']
    gen = torch.Generator().manual_seed(11)
    x0 = torch.randn(4, C, generator=gen)
    x = x0.clone().cuda().requires_grad_(True)
    out = hip.decoder(x, max_shape, class_pred)
    wgt = torch.randn(out.shape, generator=gen)
    (out * wgt.cuda()).sum().backward()
    torch.cuda.synchronize()
    p = {k: v.detach().clone().requires_grad_(True) for k, v in oracle.named_parameters() if k.startswith('decoder.')}
    xo = x0.clone().requires_grad_(True)
    ref = R.conv_decoder3(xo, p, oracle.max_shape, max_shape, class_pred)
    (ref * wgt).sum().backward()
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().numpy()) < 2e-5
    assert rel_l2(x.grad.cpu().numpy(), xo.grad.numpy()) < 5e-5
    named = dict(hip.named_parameters())
    for k, v in p.items():
        if v.grad is None:                                        # (the classifier head is unused without class_pred)
            assert named[k].grad is None or float(named[k].grad.abs().sum()) == 0.0, k
            continue
        err = float((named[k].grad.cpu().double() - v.grad.double()).norm())
        assert err < 1e-4 * float(v.grad.norm()) + 1e-6, (k, err, float(v.grad.norm()))
