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
