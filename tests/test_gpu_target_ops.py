"""
Native target-network layers: the fused ReLU -> depthwise conv -> pointwise conv -> BatchNorm op family
(ghn3_amd/csrc/target_ops.hip through the C ABI ghn3_dwpw_bn_fwd / _bwd) against the four stock torch layers it replaces
(/root/reference/ghn3/ops.py:198-240), forward and all five gradients; then whole networks of the search space with the
fused layers switched on against the same networks on the stock path.

Tolerances: the pointwise product multiplies split-bf16 operands (hi.hi + lo.hi + hi.lo, ~1e-5 relative) and everything else
is fp32 in another summation order than torch's: 2e-4 of the tensor's scale (the GHN north star asks 1e-3).
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))

CASES = [   # N, C_in, C_out, H, W, ks, stride, pad, dil
    (4, 32, 32, 16, 16, 3, 1, 1, 1),
    (3, 48, 96, 15, 17, 3, 2, 1, 1),          # C not a multiple of 32, odd sizes, stride 2, widening
    (2, 64, 64, 12, 12, 5, 1, 4, 2),          # dil_conv_5x5: dilation 2, padding 4
    (2, 128, 64, 9, 9, 7, 1, 3, 1),           # 7 x 7 taps, narrowing
    (5, 16, 16, 8, 8, 3, 2, 2, 2),            # dil_conv_3x3 with stride 2
    (1, 256, 512, 6, 6, 3, 1, 1, 1),          # the widest tiles (32 column tiles per wave)
    (8, 80, 112, 7, 5, 5, 2, 2, 1),           # partial pixel tile (8 * 4 * 3 = 96 pixels), odd channel tiles
]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize('case', CASES)
def test_dwpw_bn_matches_the_stock_layers(case):
    from ghn3_amd import target_ops as T
    N, Ci, Co, H, W, ks, st, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w_dw = torch.randn(Ci, 1, ks, ks, generator=g) / ks
    w_pw = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    gamma = 1 + 0.3 * torch.randn(Co, generator=g)
    beta = 0.2 * torch.randn(Co, generator=g)
    up = torch.randn(N, Co, (H + 2 * pad - dil * (ks - 1) - 1) // st + 1, (W + 2 * pad - dil * (ks - 1) - 1) // st + 1,
                     generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w_dw, w_pw, gamma, beta)]
    ref = T.reference(*ref_in, stride=st, padding=pad, dilation=dil)
    (ref * up.double()).sum().backward()
    dev_in = [t.cuda().requires_grad_(True) for t in (x, w_dw, w_pw, gamma, beta)]
    assert T.DwPwBn.applicable(dev_in[0], dev_in[1], dev_in[2], dev_in[3], dev_in[4], ks)
    out, stats = T.dwpw_bn(dev_in[0].contiguous(memory_format=torch.channels_last), dev_in[1], dev_in[2], dev_in[3], dev_in[4],
                           stride=st, padding=pad, dilation=dil)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    (out * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-4, _rel(out.detach().cpu(), ref.detach())
    zz = torch.nn.functional.conv2d(torch.nn.functional.conv2d(torch.relu(x.double()), w_dw.double(), None, st, pad, dil,
                                                               groups=Ci), w_pw.double())
    assert _rel(stats[:Co].cpu(), zz.mean((0, 2, 3))) < 2e-4 and _rel(stats[2 * Co:].cpu(), zz.var((0, 2, 3), unbiased=False)) < 2e-4
    for name, a, b in zip(('dx', 'dw_dw', 'dw_pw', 'dgamma', 'dbeta'), dev_in, ref_in):
        assert a.grad is not None and torch.isfinite(a.grad).all(), name
        assert _rel(a.grad.cpu(), b.grad) < 3e-4, (name, _rel(a.grad.cpu(), b.grad))
    # deterministic: a second run gives the same bits
    out2, _ = T.dwpw_bn(dev_in[0].detach().contiguous(memory_format=torch.channels_last), dev_in[1].detach(), dev_in[2].detach(),
                        dev_in[3].detach(), dev_in[4].detach(), stride=st, padding=pad, dilation=dil)
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize('case', [(4, 32, 64, 16, 16, 1), (3, 48, 24, 9, 7, 2), (2, 256, 128, 8, 8, 1), (6, 8, 8, 5, 5, 2)])
def test_pointwise_relu_conv_bn_matches_the_stock_layers(case):
    """The same family without a depthwise stage (w_dw = NULL): ReLU -> 1 x 1 convolution (stride) -> BatchNorm, the
    preprocessing `ReLUConvBN` of every cell and the `conv_1x1` op (ops.py:180-198), against torch in fp64."""
    from ghn3_amd import target_ops as T
    F = torch.nn.functional
    N, Ci, Co, H, W, st = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=g)
    w_pw = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    gamma, beta = 1 + 0.3 * torch.randn(Co, generator=g), 0.2 * torch.randn(Co, generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w_pw, gamma, beta)]
    ref = F.batch_norm(F.conv2d(F.relu(ref_in[0]), ref_in[1], None, st), None, None, ref_in[2], ref_in[3], True, 0.1, 1e-5)
    up = torch.randn(ref.shape, generator=g)
    (ref * up.double()).sum().backward()
    dev_in = [t.cuda().requires_grad_(True) for t in (x, w_pw, gamma, beta)]
    out, stats = T.dwpw_bn(dev_in[0], None, dev_in[1], dev_in[2], dev_in[3], stride=st)
    (out * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert out.shape == ref.shape and _rel(out.detach().cpu(), ref.detach()) < 2e-4
    for name, a, b in zip(('dx', 'dw_pw', 'dgamma', 'dbeta'), dev_in, ref_in):
        assert _rel(a.grad.cpu(), b.grad) < 3e-4, (name, _rel(a.grad.cpu(), b.grad))


CONV_CASES = [   # N, C_in, C_out, H, W, (kh, kw), (sh, sw), (ph, pw), dil, relu
    (4, 32, 32, 16, 16, (3, 3), (1, 1), (1, 1), 1, True),
    (3, 48, 96, 15, 17, (3, 3), (2, 2), (1, 1), 1, True),       # C not a multiple of 32, odd sizes, stride 2, widening
    (2, 64, 64, 12, 12, (5, 5), (1, 1), (2, 2), 1, True),       # conv_5x5
    (2, 128, 64, 9, 9, (7, 7), (1, 1), (3, 3), 1, True),        # conv_7x7, narrowing
    (5, 16, 24, 8, 8, (3, 3), (2, 2), (1, 1), 1, False),        # no ReLU in front (a stem-like convolution)
    (1, 256, 512, 6, 6, (3, 3), (1, 1), (1, 1), 1, True),       # the widest tiles
    (8, 80, 112, 7, 5, (5, 5), (2, 2), (2, 2), 1, True),        # partial pixel tiles
    (2, 32, 48, 10, 10, (1, 7), (1, 2), (0, 3), 1, True),       # the 1 x k half of a `conv2` pair, asymmetric stride
    (2, 32, 48, 10, 10, (7, 1), (2, 1), (3, 0), 1, True),
    (3, 24, 40, 11, 11, (3, 3), (1, 1), (2, 2), 2, True),       # dilation 2
    (6, 576, 192, 4, 4, (1, 1), (1, 1), (0, 0), 1, True),       # a cell's preprocessing layer on three concatenated states: C_in > 512
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_bn_matches_the_stock_layers(case):
    """ghn3_conv_bn_fwd / _bwd (round 6: ReLU -> dense k x k convolution -> BatchNorm as one node, ops.py:180-198) against the
    stock torch layers in fp64: output, batch statistics and all four gradients; deterministic."""
    from ghn3_amd import target_ops as T
    N, Ci, Co, H, W, ks, st, pad, dil, relu = case
    g = torch.Generator().manual_seed(N + Ci + Co + H + W + sum(ks) + sum(st))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, ks[0], ks[1], generator=g) / (Ci * ks[0] * ks[1]) ** 0.5
    gamma = 1 + 0.3 * torch.randn(Co, generator=g)
    beta = 0.2 * torch.randn(Co, generator=g)
    ref_in = [t.clone().double().requires_grad_(True) for t in (x, w, gamma, beta)]
    ref = T.conv_reference(*ref_in, stride=st, padding=pad, dilation=dil, relu=relu)
    up = torch.randn(ref.shape, generator=g)
    (ref * up.double()).sum().backward()
    dev_in = [t.cuda().requires_grad_(True) for t in (x, w, gamma, beta)]
    assert T.ConvBn.applicable(*dev_in)
    out, stats = T.conv_bn(dev_in[0].contiguous(memory_format=torch.channels_last), dev_in[1], dev_in[2], dev_in[3], stride=st,
                           padding=pad, dilation=dil, relu=relu)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    (out * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-4, _rel(out.detach().cpu(), ref.detach())
    zz = torch.nn.functional.conv2d(torch.relu(x.double()) if relu else x.double(), w.double(), None, st, pad, dil)
    assert _rel(stats[:Co].cpu(), zz.mean((0, 2, 3))) < 2e-4 and _rel(stats[2 * Co:].cpu(), zz.var((0, 2, 3), unbiased=False)) < 2e-4
    for name, a, b in zip(('dx', 'dw', 'dgamma', 'dbeta'), dev_in, ref_in):
        assert a.grad is not None and torch.isfinite(a.grad).all(), name
        assert a.grad.shape == b.grad.shape and _rel(a.grad.cpu(), b.grad) < 3e-4, (name, _rel(a.grad.cpu(), b.grad))
    out2, _ = T.conv_bn(dev_in[0].detach().contiguous(memory_format=torch.channels_last), dev_in[1].detach(), dev_in[2].detach(),
                        dev_in[3].detach(), stride=st, padding=pad, dilation=dil, relu=relu)
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize('case', [CONV_CASES[k] for k in (0, 1, 4, 7, 8, 9)])
def test_conv_without_a_norm_layer(case):
    """GHN3_CONV_NO_NORM (the first half of the 1 x k / k x 1 pair, ops.py:186-190): [ReLU ->] convolution alone against
    torch in fp64 -- result, input gradient, weight gradient; deterministic."""
    from ghn3_amd import target_ops as T
    N, Ci, Co, H, W, ks, st, pad, dil, relu = case
    g = torch.Generator().manual_seed(3 + N + Ci + Co + H + W + sum(ks) + sum(st))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, ks[0], ks[1], generator=g) / (Ci * ks[0] * ks[1]) ** 0.5
    xr, wr = x.clone().double().requires_grad_(True), w.clone().double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(torch.relu(xr) if relu else xr, wr, None, st, pad, dil)
    up = torch.randn(ref.shape, generator=g)
    (ref * up.double()).sum().backward()
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    assert T.ConvOnly.applicable(xd, wd)
    out = T.conv_only(xd, wd, stride=st, padding=pad, dilation=dil, relu=relu)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    (out * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(out.detach().cpu(), ref.detach()) < 2e-4
    assert _rel(xd.grad.cpu(), xr.grad) < 3e-4 and _rel(wd.grad.cpu(), wr.grad) < 3e-4, (_rel(xd.grad.cpu(), xr.grad),
                                                                                       _rel(wd.grad.cpu(), wr.grad))
    assert torch.equal(T.conv_only(xd.detach(), wd.detach(), stride=st, padding=pad, dilation=dil, relu=relu), out.detach())


@pytest.mark.parametrize('stride', [1, 2])
def test_conv_pair_module_runs_on_two_dense_nodes(monkeypatch, stride):
    """`ReLUConvBN(double=True)` (the search space's `conv_1x7_7x1`, ops.py:186-190,298): ReLU + 1 x 7 convolution as one node,
    7 x 1 convolution + norm as the next; output, every gradient and the running statistics against the stock layers."""
    from ghn3_amd import ops
    res = {}
    for mode in ('stock', 'fused'):
        monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
        torch.manual_seed(6)
        m = ops.ReLUConvBN(48, 48, 7, stride, 3, double=True)
        m.op[3] = torch.nn.BatchNorm2d(48)
        m = m.cuda().train()
        x = torch.randn(5, 48, 12, 10, device='cuda', requires_grad=True)
        y = m(x)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).cuda()
        (y * up).sum().backward()
        bn = list(m.op)[-1]
        res[mode] = (y.detach().cpu(), x.grad.cpu(), [p.grad.cpu() for p in m.parameters()], bn.running_mean.cpu(), bn.running_var.cpu())
    assert res['fused'][0].shape == res['stock'][0].shape
    assert _rel(res['fused'][0], res['stock'][0]) < 2e-4 and _rel(res['fused'][1], res['stock'][1]) < 5e-4
    assert len(res['fused'][2]) == 4
    for a, b in zip(res['fused'][2], res['stock'][2]):
        assert _rel(a, b) < 5e-4, _rel(a, b)
    assert _rel(res['fused'][3], res['stock'][3]) < 1e-4 and _rel(res['fused'][4], res['stock'][4]) < 1e-4


@pytest.mark.parametrize('spec', [
    [('conv', 3, 48, 3, 1), ('bn', 48), ('id',)],                                                    # stem type 0
    [('conv', 3, 32, 7, 1), ('bn', 32), ('maxpool',)],
    [('conv', 3, 16, 3, 1), ('bn', 16), ('relu',), ('conv', 16, 32, 3, 1), ('bn', 32)],             # stem0 of the two-stem networks
    [('relu',), ('conv', 32, 32, 3, 2), ('bn', 32)],                                                 # stem1: IN-PLACE ReLU at the head
])
def test_stem_sequences_run_on_the_dense_op(monkeypatch, spec):
    """The stems (ops.py:443-463): every conv -> norm window on the fused op -- the 3-channel image padded to four channels, a
    ReLU between two windows folded into the second one, an in-place ReLU at the head applied to the caller's tensor as the
    stock layer does -- against the stock layers: output, input, every parameter gradient."""
    from ghn3_amd import ops, light_ops
    from ghn3_amd import target_ops as T
    res = {}
    c_in = spec[0][1] if spec[0][0] == 'conv' else spec[1][1]
    for mode in ('stock', 'fused'):
        monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
        torch.manual_seed(8)
        seq = ops._layer_seq(ops._TorchLayers, 'bn-track', spec).cuda().train()
        x0 = torch.randn(6, c_in, 16, 16, device='cuda', requires_grad=True)
        x = x0 * 1.0                                           # (a non-leaf: the in-place ReLU may rewrite it)
        y = ops.Network._run_stem(seq, x)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).cuda()
        ((y * up).sum() + (x * x).sum()).backward()            # (x enters the loss AFTER the stem: sees the in-place ReLU)
        res[mode] = (y.detach().cpu(), x.detach().cpu(), x0.grad.cpu(), [p.grad.cpu() for p in seq.parameters()])
    assert _rel(res['fused'][0], res['stock'][0]) < 2e-4
    assert torch.equal(res['fused'][1], res['stock'][1])
    assert _rel(res['fused'][2], res['stock'][2]) < 5e-4
    for a, b in zip(res['fused'][3], res['stock'][3]):
        assert a.shape == b.shape and _rel(a, b) < 5e-4, _rel(a, b)


def test_patch_embedding_and_wide_pointwise_layers_leave_the_stock_path(monkeypatch):
    """The two convolution shapes the stock path kept until the end of round 6: the ViT-style patch embedding (a bare strided
    Conv2d on the 3-channel image, ops.py:296) and the 1 x 1 preprocessing layer over more than 512 concatenated channels --
    both on the dense-convolution op now, equal to the stock layers."""
    from ghn3_amd import ops
    res = {}
    for mode in ('stock', 'fused'):
        monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
        torch.manual_seed(4)
        conv = torch.nn.Conv2d(3, 64, 3, stride=3, padding=1, bias=False).cuda()
        x = torch.randn(5, 3, 32, 32, device='cuda', requires_grad=True)
        from ghn3_amd import target_ops as T
        y = T.run_conv_layer(conv, x) if mode == 'fused' else conv(x)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).cuda()
        (y * up).sum().backward()
        m = ops.ReLUConvBN(576, 192, 1, 1, 0).cuda().train()
        x2 = torch.randn(4, 576, 4, 4, device='cuda', requires_grad=True)
        y2 = m(x2)
        up2 = torch.randn(y2.shape, generator=torch.Generator().manual_seed(3)).cuda()
        (y2 * up2).sum().backward()
        res[mode] = (y.detach().cpu(), x.grad.cpu(), conv.weight.grad.cpu(), y2.detach().cpu(), x2.grad.cpu(),
                     [p.grad.cpu() for p in m.parameters()], type(y2.grad_fn).__name__)
    assert 'ConvBn' in res['fused'][6] or 'Clone' in res['fused'][6] or 'Contiguous' in res['fused'][6], res['fused'][6]
    assert res['fused'][0].is_contiguous()
    for k in range(5):
        assert _rel(res['fused'][k], res['stock'][k]) < 5e-4, (k, _rel(res['fused'][k], res['stock'][k]))
    for a, b in zip(res['fused'][5], res['stock'][5]):
        assert _rel(a, b) < 5e-4


@pytest.mark.parametrize('case', [(6, 32, 16, 16, 1), (5, 48, 9, 7, 2), (3, 256, 4, 4, 1), (2, 64, 32, 32, 2), (4, 512, 2, 2, 1)])
def test_squeeze_excitation_layer_matches_the_stock_layers(case):
    """ghn3_se_fwd / _bwd (round 6: `ChannelSELayer`, ops.py:239-274 -- mean, two Linear layers, ReLU, hard-swish gate, product, stride
    slicing -- as one node) against the same module on its stock layers in fp64: output, input gradient, both weights and biases;
    deterministic."""
    from ghn3_amd import ops, target_ops as T
    N, C, H, W, stride = case
    torch.manual_seed(C + H)
    m = ops.ChannelSELayer(C, stride=stride)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(3.0)                                            # (gates on both sides of the hard-swish knees)
    x = torch.randn(N, C, H, W) * 2
    ref_m = ops.ChannelSELayer(C, stride=stride).double()
    ref_m.load_state_dict({k: v.double() for k, v in m.state_dict().items()})
    xr = x.double().requires_grad_(True)
    ref = ref_m(xr)
    up = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    (ref * up.double()).sum().backward()
    m = m.cuda()
    xd = x.cuda().requires_grad_(True)
    assert T.SqueezeExcite.applicable(xd, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
    out = m(xd)
    assert out.shape == ref.shape
    assert 'SqueezeExcite' in type(out.grad_fn).__name__ or stride > 1 or 'Clone' in type(out.grad_fn).__name__, type(out.grad_fn).__name__
    (out * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-5
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-5
    for (name, a), b in zip(m.named_parameters(), ref_m.parameters()):
        assert a.grad is not None and _rel(a.grad.cpu(), b.grad) < 2e-5, (name, _rel(a.grad.cpu(), b.grad))
    out2 = m(xd.detach())
    assert torch.equal(out2, out.detach())


@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('case', [(4, 32, 16, 16, 3, 1, 1), (3, 48, 15, 17, 3, 2, 1), (2, 64, 8, 8, 2, 2, 0), (5, 16, 7, 5, 3, 1, 1),
                                  (2, 128, 4, 4, 3, 2, 1), (2, 24, 32, 32, 5, 2, 2)])
def test_nhwc_pooling_matches_torch(case, mode):
    """ghn3_pool_fwd / _bwd (round 6: `avg_pool_3x3` with count_include_pad = False, `max_pool_3x3`, the stems' MaxPool2d(3, 2, 1);
    ops.py:289-291,452) against torch on an NCHW fp64 tensor: output (max: bit-exact) and input gradient up to fp32 rounding,
    through the light modules that route to it."""
    from ghn3_amd import light_ops
    N, C, H, W, k, s, pad = case
    g = torch.Generator().manual_seed(N * C + H + k + mode)
    x = torch.randn(N, C, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    F = torch.nn.functional
    ref = F.max_pool2d(xr, k, s, pad) if mode else F.avg_pool2d(xr, k, s, pad, count_include_pad=False)
    up = torch.randn(ref.shape, generator=g)
    (ref * up.double()).sum().backward()
    m = light_ops.MaxPool2d(k, stride=s, padding=pad) if mode else light_ops.AvgPool2d(k, stride=s, padding=pad, count_include_pad=False)
    xd = x.cuda().requires_grad_(True)
    out = m(xd)
    assert 'Pool2d' in type(out.grad_fn).__name__, type(out.grad_fn).__name__
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    (out * up.cuda()).sum().backward()
    torch.cuda.synchronize()
    if mode:
        assert torch.equal(out.detach().cpu().double(), ref.detach())
        assert _rel(xd.grad.cpu(), xr.grad) < 1e-6                 # (an input that wins several windows sums their gradients in fp32)
        assert torch.equal(xd.grad.cpu() != 0, xr.grad != 0)       # (the same winners)
    else:
        assert _rel(out.detach().cpu(), ref.detach()) < 1e-6 and _rel(xd.grad.cpu(), xr.grad) < 1e-6


def test_fused_layers_run_in_fp32_under_autocast(monkeypatch):
    """Under torch.autocast (Trainer(amp=True), trainer.py:300-319) the fused layers keep running, in fp32: the same bits as without
    autocast; GHN3_NATIVE_AMP=0 hands the layers back to the stock 16-bit autocast kernels."""
    from ghn3_amd import ops
    torch.manual_seed(12)
    m = ops.ReLUConvBN(32, 48, 3, 1, 1).cuda().train()
    x = torch.randn(4, 32, 10, 10, device='cuda')
    with torch.no_grad():
        plain = m(x)
        with torch.autocast('cuda', dtype=torch.float16):
            amp = m(x)
        monkeypatch.setenv('GHN3_NATIVE_AMP', '0')
        with torch.autocast('cuda', dtype=torch.float16):
            stock = m(x)
    assert amp.dtype == torch.float32 and torch.equal(amp, plain)
    assert _rel(stock.float().cpu(), plain.cpu()) < 2e-2 and not torch.equal(stock.float(), plain)


def test_relu_conv_bn_module_runs_on_the_dense_op(monkeypatch):
    """`ReLUConvBN` with a 3 x 3 kernel (the search space's `conv_3x3`): the module's forward goes through ONE fused node, matches the
    stock layers and updates the running statistics as torch does."""
    from ghn3_amd import ops, target_ops as T
    res = {}
    for mode in ('stock', 'fused'):
        monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
        torch.manual_seed(5)
        m = ops.ReLUConvBN(32, 48, 3, 2, 1)
        m.op[2] = torch.nn.BatchNorm2d(48)              # (a TRACKING norm: the search space's own layers do not track)
        m = m.cuda().train()
        x = torch.randn(6, 32, 12, 12, device='cuda', requires_grad=True)
        y = m(x)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).cuda()
        (y * up).sum().backward()                        # (not mean(y^2): that is constant behind a BatchNorm, its gradient pure round-off)
        bn = list(m.op)[-1]
        res[mode] = (y.detach().cpu(), x.grad.cpu(), [p.grad.cpu() for p in m.parameters()], bn.running_mean.cpu(), bn.running_var.cpu(),
                     type(y.grad_fn).__name__)
    assert 'ConvBn' in res['fused'][5] or 'Clone' in res['fused'][5] or 'Contiguous' in res['fused'][5], res['fused'][5]
    assert _rel(res['fused'][0], res['stock'][0]) < 2e-4 and _rel(res['fused'][1], res['stock'][1]) < 5e-4
    for a, b in zip(res['fused'][2], res['stock'][2]):
        assert _rel(a, b) < 5e-4
    assert _rel(res['fused'][3], res['stock'][3]) < 1e-4 and _rel(res['fused'][4], res['stock'][4]) < 1e-4


def test_factorized_reduce_runs_as_one_dense_convolution(monkeypatch):
    """`FactorizedReduce` (ops.py:163-178) on the fused path = a 2 x 2 stride-2 convolution assembled from the two 1 x 1 weights:
    output and every gradient (input, both convolution weights, norm) against the stock layers."""
    from ghn3_amd import ops
    res = {}
    for mode in ('stock', 'fused'):
        monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
        torch.manual_seed(11)
        m = ops.FactorizedReduce(24, 40).cuda().train()
        x = torch.randn(5, 24, 14, 10, device='cuda', requires_grad=True)
        y = m(x)
        up = torch.randn(y.shape, generator=torch.Generator().manual_seed(2)).cuda()
        (y * up).sum().backward()
        res[mode] = (y.detach().cpu(), x.grad.cpu(), [p.grad.cpu() for p in m.parameters()])
    assert res['fused'][0].shape == res['stock'][0].shape == (5, 40, 7, 5)
    assert _rel(res['fused'][0], res['stock'][0]) < 2e-4 and _rel(res['fused'][1], res['stock'][1]) < 5e-4
    for a, b in zip(res['fused'][2], res['stock'][2]):
        assert a.shape == b.shape and _rel(a, b) < 5e-4


def test_descriptor_limits_are_refused_loudly():
    from ghn3_amd import target_ops as T, _lib as L
    x = torch.randn(1, 6, 4, 4, device='cuda')
    with pytest.raises(L.Ghn3Error):          # C % 4 != 0 straight through the ABI
        T.dwpw_bn(x, torch.randn(6, 1, 3, 3, device='cuda'), torch.randn(8, 6, device='cuda'), torch.ones(8, device='cuda'),
                  torch.zeros(8, device='cuda'), padding=1)
    assert not T.DwPwBn.applicable(x, torch.randn(6, 1, 3, 3, device='cuda'), torch.randn(8, 6, device='cuda'),
                                   torch.ones(8, device='cuda'), torch.zeros(8, device='cuda'), 3)
    with pytest.raises(L.Ghn3Error):
        T.dwpw_bn(torch.randn(1, 8, 4, 4), torch.randn(8, 1, 3, 3), torch.randn(8, 8), torch.ones(8), torch.zeros(8))


@pytest.mark.parametrize('light', [False, True])
def test_networks_on_the_fused_layers_match_the_stock_path(light, monkeypatch):
    """Whole networks of the search space (tests/golden/network_cases.py: separable / dilated convolutions of every size among
    the other ops): logits and every parameter gradient with the fused HIP layers against the same network, same tensors, on
    ATen / MIOpen; light flavour = the tensors are assigned as a GHN would (views of one flat buffer)."""
    import network_cases
    import recipe
    from ghn3_amd import ops
    used = 0
    for name, (geno, kw, img) in network_cases.CASES.items():
        g = ops.Genotype(**geno)
        if not any(n[0].startswith(('sep_conv', 'dil_conv', 'conv_')) for n in g.normal + g.reduce):
            continue
        used += 1
        res = {}
        for mode in ('stock', 'fused'):
            monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
            torch.manual_seed(0)
            kws = {k: ('bn' if (k == 'norm' and v and light) else v) for k, v in kw.items()}
            net = (ops.NetworkLight if light else ops.Network)(genotype=g, **kws)
            x = torch.from_numpy(recipe.seeded_images(img, seed=7)).cuda()
            if light:
                table = {}
                for cell in net._layered_modules:
                    table.update(cell)
                shapes = [(n, tuple(e['sz'])) for n, e in table.items()]
                params = recipe.seeded_net_params(shapes, seed=len(name))
                total = sum(int(np.prod(s)) for _, s in shapes)
                flat = torch.zeros(total, device='cuda', requires_grad=True)
                off, views = 0, {}
                with torch.no_grad():
                    for n, s in shapes:
                        k = int(np.prod(s))
                        flat[off:off + k] = torch.from_numpy(params[n]).reshape(-1).cuda()
                        off += k
                off = 0
                for n, s in shapes:
                    k = int(np.prod(s))
                    views[n] = flat[off:off + k].view(s)
                    off += k
                for n, e in table.items():
                    setattr(e['module'], 'weight' if e['is_w'] else 'bias', views[n])
                leaves = [flat]
                for attr in ('auxiliary_head',):                  # (a torch.nn module with parameters of its own, ops.py:506-510)
                    if hasattr(net, attr):
                        getattr(net, attr).cuda()
                        leaves += list(getattr(net, attr).parameters())
            else:
                net = net.cuda()
                params = recipe.seeded_net_params([(n, tuple(p.shape)) for n, p in net.named_parameters()], seed=len(name))
                with torch.no_grad():
                    for n, p in net.named_parameters():
                        p.copy_(torch.from_numpy(params[n]))
                leaves = [p for _, p in net.named_parameters()]
            net.train()
            torch.manual_seed(123)
            logits, aux = net(x)
            loss = logits.square().mean() + (aux.square().mean() if aux is not None else 0.)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = (logits.detach().cpu(), [p.grad.detach().cpu() if p.grad is not None else None for p in leaves])
        (l0, g0), (l1, g1) = res['stock'], res['fused']
        assert _rel(l1, l0) < 1e-3, (name, _rel(l1, l0))
        for a, b in zip(g1, g0):
            assert (a is None) == (b is None)
            if a is not None and float(b.norm()) > 0:
                assert _rel(a, b) < 2e-3, (name, _rel(a, b))
    assert used >= 3


def test_sampled_architectures_on_the_fused_layers_match_the_stock_path(monkeypatch):
    """Twelve architectures of the training loop's own stream (ghn3_amd.deepnets1m.SampledNets as examples/train_ghn_ddp.py draws
    them: whatever mix of stems, conv / pair / separable / dilated ops, pooling, squeeze-excitation, factorized reductions and wide
    preprocessing layers the sampler produces), tensors assigned as a GHN would: logits and the gradient of the flat parameter
    buffer on the fused HIP layers against the stock ATen / MIOpen layers."""
    import recipe
    from ghn3_amd.deepnets1m import SampledNets
    worst = (0.0, 0.0)
    x = torch.from_numpy(recipe.seeded_images((8, 3, 32, 32), seed=3)).cuda()
    for k in range(12):
        res = {}
        for mode in ('stock', 'fused'):
            monkeypatch.setenv('GHN3_NATIVE_OPS', '0' if mode == 'stock' else '1')
            net = SampledNets(large_images=False, seed=0, max_nodes=400)[k].net
            table = {}
            for cell in net._layered_modules:
                table.update(cell)
            shapes = [(n, tuple(e['sz'])) for n, e in table.items()]
            params = recipe.seeded_net_params(shapes, seed=100 + k)
            total = sum(int(np.prod(s)) for _, s in shapes)
            flat = torch.zeros(total, device='cuda')
            off = 0
            for n, s in shapes:
                cnt = int(np.prod(s))
                flat[off:off + cnt] = torch.from_numpy(params[n]).reshape(-1).cuda()
                off += cnt
            flat.requires_grad_(True)
            off = 0
            for n, e in table.items():
                cnt = int(np.prod(e['sz']))
                setattr(e['module'], 'weight' if e['is_w'] else 'bias', flat[off:off + cnt].view(tuple(e['sz'])))
                off += cnt
            net.train()
            torch.manual_seed(5)
            logits, _ = net(x)
            up = torch.randn(logits.shape, generator=torch.Generator().manual_seed(9)).cuda()
            (logits * up).sum().backward()
            torch.cuda.synchronize()
            res[mode] = (logits.detach().cpu(), flat.grad.detach().cpu())
        e_l, e_g = _rel(res['fused'][0], res['stock'][0]), _rel(res['fused'][1], res['stock'][1])
        worst = (max(worst[0], e_l), max(worst[1], e_g))
        assert e_l < 1e-3 and e_g < 5e-3, (k, e_l, e_g)
    print('worst logits / gradient deviation over the stream: %.2e / %.2e' % worst)
