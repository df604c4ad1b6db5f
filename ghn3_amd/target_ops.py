"""
Native execution of target-network layers with GHN-predicted weights, first slice: the chain

    ReLU -> depthwise k x k convolution -> pointwise 1 x 1 convolution -> BatchNorm (batch statistics)

of ``DilConv`` and of each half of ``SepConv`` (/root/reference/ghn3/ops.py:198-240; run at trainer.py:308-319), forward and
backward, as ONE autograd node on the HIP op family ``ghn3_dwpw_bn_fwd / _bwd`` (include/ghn3_hip.h,
ghn3_amd/csrc/target_ops.hip) instead of four ATen / MIOpen modules per direction.

Activations are torch ``channels_last`` tensors (NHWC in memory, NCHW in shape) inside the op family: the kernels take such
storage as it is and return it, so consecutive fused blocks (the two halves of a SepConv) never convert.  At the boundary to
the stock layers the activations are converted (``run_block``): the stock ATen / MIOpen layers of this ROCm build return
wrong gradients for channels_last inputs, so the layout is not allowed to leak into them.  The weights are the views of the GHN's flat prediction buffer the GHN assigned to the
layers -- read in place, no copy; their gradients leave as dense tensors for autograd to route back into that buffer.

``DwPwBn.applicable`` states what the kernels take (fp32 CUDA tensors, batch statistics, C <= 512, ks <= 7); a layer
outside of it keeps the stock path.  There is no CPU implementation: on a CPU tensor the stock path runs (the target
networks themselves are torch modules), and ``dwpw_bn`` raises without the library.
"""

import ctypes
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L


class _Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('N', 'H', 'W', 'C_in', 'C_out', 'ks', 'stride', 'pad', 'dil', 'Ho', 'Wo')] + \
        [('eps', ctypes.c_float)]


def _desc(x, C_out, ks, stride, pad, dil, eps):
    N, C, H, W = x.shape
    Ho = (H + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (ks - 1) - 1) // stride + 1
    return _Desc(N, H, W, C, C_out, ks, stride, pad, dil, Ho, Wo, float(eps))


def _ptr(t):
    return t.data_ptr()               # (an int: ctypes converts it for the c_void_p parameters)


def _autocast_excludes():
    """Under torch.autocast the fused layers still run -- in fp32, on the fp32 tensors the GHN predicted and the fp32 activations the
    previous fused layer left (a precision at or above what the stock fp16 / bf16 autocast kernels would use; a 16-bit activation from
    a stock autocast layer makes the next fused layer inapplicable by its dtype check).  GHN3_NATIVE_AMP=0: leave every layer to the
    stock path while autocast is on (the behaviour until round 6)."""
    return torch.is_autocast_enabled() and os.environ.get('GHN3_NATIVE_AMP', '1') == '0'


def _stream():
    """The current HIP stream of the current device as an integer handle (torch.cuda.current_stream() builds a Stream object
    per call: 11 us; this is called twice per fused layer, ~700 times per training step)."""
    raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
    if raw is None:                                       # (a torch build without the raw accessor)
        return torch.cuda.current_stream().cuda_stream
    return raw(torch.cuda.current_device())


_SCRATCH = {}


def _scratch_floats(lib, d, backward):
    key = (d.N, d.H, d.W, d.C_in, d.C_out, d.ks, d.stride, d.pad, d.dil, backward)
    n = _SCRATCH.get(key)
    if n is None:
        n = int(lib.ghn3_dwpw_scratch_floats(ctypes.byref(d), backward))
        if n < 0:
            raise L.Ghn3Error('ghn3_dwpw_scratch_floats: %s' % lib.ghn3_last_error().decode())
        _SCRATCH[key] = n
    return n


def lazy_layout(*layers):
    """True when a fused block may hand its NHWC output on as it is: every layer of the block is of the light flavour (whose stock
    Conv2d / BatchNorm2d convert a channels_last input themselves, light_ops._stock_layout) and GHN3_NATIVE_LAZY_LAYOUT is not 0.
    torch.nn-flavour networks keep the conversion at the fused block's boundary."""
    if os.environ.get('GHN3_NATIVE_LAZY_LAYOUT', '1') == '0':
        return False
    return all(not isinstance(m, torch.nn.Module) for m in layers)


def enabled():
    """GHN3_NATIVE_OPS=0 keeps every target-network layer on the stock ATen / MIOpen path (A/B measurements)."""
    return os.environ.get('GHN3_NATIVE_OPS', '1') != '0'


class DwPwBn(torch.autograd.Function):
    @staticmethod
    def applicable(x, w_dw, w_pw, gamma, beta, ks, training_stats=True):
        if not (enabled() and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            return False
        if _autocast_excludes():
            return False                       # (GHN3_NATIVE_AMP=0: under AMP the stock path decides the types)
        if not training_stats or not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32
                                         for t in (w_pw, gamma, beta) + (() if w_dw is None else (w_dw,))):
            return False
        C_in, C_out = x.shape[1], w_pw.shape[0]
        return C_in % 4 == 0 and C_out % 4 == 0 and C_in <= 512 and C_out <= 512 and ks <= 7 and \
            x.numel() < 2 ** 31 and (w_dw is None or w_dw.shape[1] == 1) and w_pw.numel() == C_out * C_in

    @staticmethod
    def forward(ctx, x, w_dw, w_pw, gamma, beta, stride, pad, dil, eps):
        lib = L.load()
        ks = 1 if w_dw is None else int(w_dw.shape[-1])
        C_out = int(w_pw.shape[0])
        xc = x.contiguous(memory_format=torch.channels_last)          # (a no-op inside a channels_last network)
        d = _desc(xc, C_out, ks, stride, pad, dil, eps)
        wd, wp = (None if w_dw is None else w_dw.contiguous()), w_pw.contiguous()
        g, b = gamma.contiguous(), beta.contiguous()
        dev = x.device
        out = torch.empty((d.N, C_out, d.Ho, d.Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        z = torch.empty_like(out)
        stats = torch.empty(3 * C_out, dtype=torch.float32, device=dev)
        scratch = torch.empty(_scratch_floats(lib, d, 0), dtype=torch.float32, device=dev)
        stream = _stream()
        L._check(lib.ghn3_dwpw_bn_fwd(ctypes.byref(d), _ptr(xc), _ptr(wd) if wd is not None else None, _ptr(wp), _ptr(g), _ptr(b),
                                      _ptr(z), _ptr(out), _ptr(stats), _ptr(scratch), stream), 'ghn3_dwpw_bn_fwd')
        ctx.has_dw = wd is not None
        ctx.save_for_backward(xc, z, stats, wd if wd is not None else stats, wp, g)
        ctx.cfg = (stride, pad, dil, eps)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dout, _dstats):
        lib = L.load()
        xc, z, stats, wd, wp, g = ctx.saved_tensors
        if not ctx.has_dw:
            wd = None
        stride, pad, dil, eps = ctx.cfg
        C_out, ks = int(wp.shape[0]), (1 if wd is None else int(wd.shape[-1]))
        d = _desc(xc, C_out, ks, stride, pad, dil, eps)
        dev = xc.device
        do = dout.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(xc)
        # (one allocation for the four parameter gradients and the scratch area: a call is launch- and host-bound for the
        # small layers of a CIFAR network)
        n_wd = 0 if wd is None else wd.numel()
        n_par = n_wd + wp.numel() + 2 * C_out
        buf = torch.empty(n_par + 64 + _scratch_floats(lib, d, 1), dtype=torch.float32, device=dev)
        dwd = None if wd is None else buf[:n_wd].view(wd.shape)
        dwp = buf[n_wd:n_wd + wp.numel()].view(wp.shape)
        db = buf[n_par - 2 * C_out:n_par - C_out]            # (dbeta directly followed by dgamma: the kernels' own pair of sums)
        dg = buf[n_par - C_out:n_par]
        scratch = buf[(n_par + 63) // 64 * 64:]
        stream = _stream()
        L._check(lib.ghn3_dwpw_bn_bwd(ctypes.byref(d), _ptr(do), _ptr(xc), _ptr(z), _ptr(stats),
                                      _ptr(wd) if wd is not None else None, _ptr(wp), _ptr(g), _ptr(dx),
                                      _ptr(dwd) if dwd is not None else None, _ptr(dwp), _ptr(dg), _ptr(db), _ptr(scratch), stream),
                 'ghn3_dwpw_bn_bwd')
        return dx, dwd, dwp, dg, db, None, None, None, None


class _ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('N', 'H', 'W', 'C_in', 'C_out', 'kh', 'kw', 'stride_h', 'stride_w', 'pad_h', 'pad_w',
                                              'dil', 'Ho', 'Wo', 'relu')] + [('eps', ctypes.c_float)]


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


CONV_NO_NORM = 2                  # include/ghn3_hip.h GHN3_CONV_NO_NORM


def _conv_max_in():
    """Widest input of the dense-convolution op (include/ghn3_hip.h: the second-version kernels walk C_in in chunks)."""
    return 4096 if os.environ.get('GHN3_TNET_CONV2', '1') != '0' else 512


def _conv_desc(x, w, stride, pad, dil, relu, eps, no_norm=False):
    N, C, H, W = x.shape
    C_out, _, kh, kw = w.shape
    (sh, sw), (ph, pw) = _pair(stride), _pair(pad)
    Ho = (H + 2 * ph - dil * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dil * (kw - 1) - 1) // sw + 1
    return _ConvDesc(N, H, W, C, C_out, kh, kw, sh, sw, ph, pw, int(dil), Ho, Wo,
                     int(bool(relu)) | (CONV_NO_NORM if no_norm else 0), float(eps))


_CONV_SCRATCH = {}


def _conv_scratch_floats(lib, d, backward):
    key = tuple(getattr(d, f[0]) for f in d._fields_[:-1]) + (backward,)
    n = _CONV_SCRATCH.get(key)
    if n is None:
        n = int(lib.ghn3_conv_scratch_floats(ctypes.byref(d), backward))
        if n < 0:
            raise L.Ghn3Error('ghn3_conv_scratch_floats: %s' % lib.ghn3_last_error().decode())
        _CONV_SCRATCH[key] = n
    return n


class ConvBn(torch.autograd.Function):
    """[ReLU ->] dense kh x kw convolution -> BatchNorm (batch statistics) as ONE autograd node on ghn3_conv_bn_fwd / _bwd
    (round 6: `ReLUConvBN` with a k x k kernel, ops.py:180-198).  Same conventions as DwPwBn: channels_last storage inside,
    the weight [C_out][C_in][kh][kw] read in place, its gradient written in the same order."""

    @staticmethod
    def applicable(x, w, gamma, beta, training_stats=True):
        if not (enabled() and os.environ.get('GHN3_NATIVE_CONV', '1') != '0' and torch.is_tensor(x) and x.is_cuda and
                x.dtype == torch.float32 and x.dim() == 4):
            return False
        if _autocast_excludes() or not training_stats:
            return False
        if not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in (w, gamma, beta)) or w.dim() != 4:
            return False
        C_in, C_out = x.shape[1], w.shape[0]
        return C_in % 4 == 0 and C_out % 4 == 0 and C_in <= _conv_max_in() and C_out <= 512 and w.shape[1] == C_in and \
            max(w.shape[2], w.shape[3]) <= 7 and x.numel() < 2 ** 31 and gamma.numel() == C_out

    @staticmethod
    def forward(ctx, x, w, gamma, beta, stride, pad, dil, relu, eps):
        lib = L.load()
        xc = x.contiguous(memory_format=torch.channels_last)
        wc, g, b = w.contiguous(), gamma.contiguous(), beta.contiguous()
        d = _conv_desc(xc, wc, stride, pad, dil, relu, eps)
        dev, C_out = x.device, int(wc.shape[0])
        out = torch.empty((d.N, C_out, d.Ho, d.Wo), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        z = torch.empty_like(out)
        stats = torch.empty(3 * C_out, dtype=torch.float32, device=dev)
        scratch = torch.empty(_conv_scratch_floats(lib, d, 0), dtype=torch.float32, device=dev)
        stream = _stream()
        L._check(lib.ghn3_conv_bn_fwd(ctypes.byref(d), _ptr(xc), _ptr(wc), _ptr(g), _ptr(b), _ptr(z), _ptr(out), _ptr(stats),
                                      _ptr(scratch), stream), 'ghn3_conv_bn_fwd')
        ctx.save_for_backward(xc, z, stats, wc, g)
        ctx.cfg = (stride, pad, dil, relu, eps)
        ctx.mark_non_differentiable(stats)
        return out, stats

    @staticmethod
    def backward(ctx, dout, _dstats):
        lib = L.load()
        xc, z, stats, wc, g = ctx.saved_tensors
        stride, pad, dil, relu, eps = ctx.cfg
        d = _conv_desc(xc, wc, stride, pad, dil, relu, eps)
        dev, C_out = xc.device, int(wc.shape[0])
        do = dout.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(xc)
        n_par = wc.numel() + 2 * C_out
        buf = torch.empty((n_par + 63) // 64 * 64 + _conv_scratch_floats(lib, d, 1), dtype=torch.float32, device=dev)
        dw = buf[:wc.numel()].view(wc.shape)
        db = buf[wc.numel():wc.numel() + C_out]              # (dbeta directly followed by dgamma: the kernels' own pair of sums)
        dg = buf[wc.numel() + C_out:n_par]
        scratch = buf[(n_par + 63) // 64 * 64:]
        stream = _stream()
        L._check(lib.ghn3_conv_bn_bwd(ctypes.byref(d), _ptr(do), _ptr(xc), _ptr(z), _ptr(stats), _ptr(wc), _ptr(g), _ptr(dx), _ptr(dw),
                                      _ptr(dg), _ptr(db), _ptr(scratch), stream), 'ghn3_conv_bn_bwd')
        return dx, dw, dg, db, None, None, None, None, None


class ConvOnly(torch.autograd.Function):
    """[ReLU ->] dense kh x kw convolution WITHOUT a norm layer (ghn3_conv_bn_fwd / _bwd with GHN3_CONV_NO_NORM): the first half
    of the 1 x k / k x 1 pair of `ReLUConvBN(double=True)` (ops.py:186-190).  Same storage conventions as ConvBn."""

    @staticmethod
    def applicable(x, w):
        if not (enabled() and os.environ.get('GHN3_NATIVE_CONV', '1') != '0' and torch.is_tensor(x) and x.is_cuda and
                x.dtype == torch.float32 and x.dim() == 4 and not _autocast_excludes()):
            return False
        if not (torch.is_tensor(w) and w.is_cuda and w.dtype == torch.float32 and w.dim() == 4):
            return False
        C_in, C_out = x.shape[1], w.shape[0]
        return C_in % 4 == 0 and C_out % 4 == 0 and C_in <= _conv_max_in() and C_out <= 512 and w.shape[1] == C_in and \
            max(w.shape[2], w.shape[3]) <= 7 and x.numel() < 2 ** 31

    @staticmethod
    def forward(ctx, x, w, stride, pad, dil, relu):
        lib = L.load()
        xc = x.contiguous(memory_format=torch.channels_last)
        wc = w.contiguous()
        d = _conv_desc(xc, wc, stride, pad, dil, relu, 0.0, no_norm=True)
        z = torch.empty((d.N, int(wc.shape[0]), d.Ho, d.Wo), dtype=torch.float32, device=x.device,
                        memory_format=torch.channels_last)
        scratch = torch.empty(_conv_scratch_floats(lib, d, 0), dtype=torch.float32, device=x.device)
        stream = _stream()
        L._check(lib.ghn3_conv_bn_fwd(ctypes.byref(d), _ptr(xc), _ptr(wc), None, None, _ptr(z), None, None, _ptr(scratch), stream),
                 'ghn3_conv_bn_fwd')
        ctx.save_for_backward(xc, wc)
        ctx.cfg = (stride, pad, dil, relu)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = L.load()
        xc, wc = ctx.saved_tensors
        stride, pad, dil, relu = ctx.cfg
        d = _conv_desc(xc, wc, stride, pad, dil, relu, 0.0, no_norm=True)
        do = dz.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(xc)
        n_par = (wc.numel() + 63) // 64 * 64
        buf = torch.empty(n_par + _conv_scratch_floats(lib, d, 1), dtype=torch.float32, device=xc.device)
        dw = buf[:wc.numel()].view(wc.shape)
        stream = _stream()
        L._check(lib.ghn3_conv_bn_bwd(ctypes.byref(d), _ptr(do), _ptr(xc), None, None, _ptr(wc), None, _ptr(dx), _ptr(dw), None,
                                      None, _ptr(buf[n_par:]), stream), 'ghn3_conv_bn_bwd')
        return dx, dw, None, None, None, None


def conv_only(x, w, stride=1, padding=0, dilation=1, relu=False):
    """conv2d(relu(x) if relu else x, w) on the dense-convolution kernels (no bias, groups = 1)."""
    if not x.is_cuda:
        raise L.Ghn3Error('conv_only runs on an MI355X only (no CPU implementation: use the stock torch layers)')
    return ConvOnly.apply(x, w, _pair(stride), _pair(padding), int(dilation), bool(relu))


def conv_bn(x, w, gamma, beta, stride=1, padding=0, dilation=1, relu=True, eps=1e-5):
    """out = batch_norm(conv2d(relu(x) if relu else x, w)) with batch statistics; returns (out, stats) as dwpw_bn does.
    x: (N, C, H, W) fp32 CUDA tensor (channels_last preferred), w (C_out, C, kh, kw)."""
    if not x.is_cuda:
        raise L.Ghn3Error('conv_bn runs on an MI355X only (no CPU implementation: use the stock torch layers)')
    return ConvBn.apply(x, w, gamma, beta, _pair(stride), _pair(padding), int(dilation), bool(relu), float(eps))


class SqueezeExcite(torch.autograd.Function):
    """y = x * hardswish(W2 relu(W1 mean_hw(x) + b1) + b2) as ONE autograd node on ghn3_se_fwd / _bwd (round 6:
    `ChannelSELayer`, ops.py:239-274).  NHWC storage inside; the Linear layers' weights and biases are read in place."""

    @staticmethod
    def applicable(x, w1, b1, w2, b2):
        if not (enabled() and os.environ.get('GHN3_NATIVE_SE', '1') != '0' and torch.is_tensor(x) and x.is_cuda and
                x.dtype == torch.float32 and x.dim() == 4 and not _autocast_excludes()):
            return False
        if not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in (w1, b1, w2, b2)):
            return False
        C, J = x.shape[1], w1.shape[0]
        return C % 4 == 0 and C <= 1024 and J <= 1024 and tuple(w1.shape) == (J, C) and tuple(w2.shape) == (C, J) and \
            b1.numel() == J and b2.numel() == C and x.numel() < 2 ** 31

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        lib = L.load()
        xc = x.contiguous(memory_format=torch.channels_last)
        w1c, b1c, w2c, b2c = w1.contiguous(), b1.contiguous(), w2.contiguous(), b2.contiguous()
        N, C, H, W = xc.shape
        J = int(w1c.shape[0])
        y = torch.empty_like(xc)
        save = torch.empty(N * (2 * C + J), dtype=torch.float32, device=x.device)
        L._check(lib.ghn3_se_fwd(N, H * W, C, J, _ptr(xc), _ptr(w1c), _ptr(b1c), _ptr(w2c), _ptr(b2c), _ptr(y), _ptr(save), _stream()),
                 'ghn3_se_fwd')
        ctx.save_for_backward(xc, w1c, w2c, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        xc, w1c, w2c, save = ctx.saved_tensors
        N, C, H, W = xc.shape
        J = int(w1c.shape[0])
        do = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(xc)
        n_par = 2 * C * J + C + J
        buf = torch.empty((n_par + 63) // 64 * 64 + N * (C + J), dtype=torch.float32, device=xc.device)
        dw1 = buf[:C * J].view(J, C)
        dw2 = buf[C * J:2 * C * J].view(C, J)
        db1 = buf[2 * C * J:2 * C * J + J]
        db2 = buf[2 * C * J + J:n_par]
        L._check(lib.ghn3_se_bwd(N, H * W, C, J, _ptr(do), _ptr(xc), _ptr(w1c), _ptr(w2c), _ptr(save), _ptr(dx), _ptr(dw1), _ptr(db1),
                                 _ptr(dw2), _ptr(db2), _ptr(buf[(n_par + 63) // 64 * 64:]), _stream()), 'ghn3_se_bwd')
        return dx, dw1, db1, dw2, db2


def se_layer(x, w1, b1, w2, b2):
    """Squeeze-and-excitation with a hard-swish gate on the fused op: x (N, C, H, W) fp32 CUDA; w1 (J, C), b1 (J), w2 (C, J), b2 (C)."""
    if not x.is_cuda:
        raise L.Ghn3Error('se_layer runs on an MI355X only (no CPU implementation: use the stock torch layers)')
    return SqueezeExcite.apply(x, w1, b1, w2, b2)


def run_se_layer(fc1, fc2, x, keep_layout=False):
    """`ChannelSELayer` body (before its stride slicing) on the fused op where it applies; None otherwise (the caller keeps its
    stock layers)."""
    w1, b1, w2, b2 = (getattr(fc1, 'weight', None), getattr(fc1, 'bias', None), getattr(fc2, 'weight', None),
                      getattr(fc2, 'bias', None))
    if not SqueezeExcite.applicable(x, w1, b1, w2, b2):
        return None
    y = se_layer(x, w1, b1, w2, b2)
    return y if (keep_layout or lazy_layout(fc1, fc2)) else y.contiguous(memory_format=torch.contiguous_format)


class _PoolDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('N', 'H', 'W', 'C', 'k', 'stride', 'pad', 'Ho', 'Wo', 'mode')]


class Pool2d(torch.autograd.Function):
    """k x k max (mode 1) / average over the valid taps (mode 0) pooling on NHWC storage: ghn3_pool_fwd / _bwd (round 6; the
    `max_pool_3x3` / `avg_pool_3x3` ops and the stems' MaxPool2d, ops.py:289-291,452)."""

    @staticmethod
    def applicable(x, k, stride, pad):
        return enabled() and os.environ.get('GHN3_NATIVE_POOL', '1') != '0' and torch.is_tensor(x) and x.is_cuda and \
            x.dtype == torch.float32 and x.dim() == 4 and not _autocast_excludes() and x.shape[1] % 4 == 0 and \
            isinstance(k, int) and isinstance(stride, int) and isinstance(pad, int) and 0 < k <= 15 and 2 * pad <= k and \
            stride > 0 and x.shape[2] + 2 * pad >= k and x.shape[3] + 2 * pad >= k and x.numel() < 2 ** 31

    @staticmethod
    def forward(ctx, x, k, stride, pad, mode):
        lib = L.load()
        xc = x.contiguous(memory_format=torch.channels_last)
        N, C, H, W = xc.shape
        d = _PoolDesc(N, H, W, C, k, stride, pad, (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1, mode)
        y = torch.empty((N, C, d.Ho, d.Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty(y.numel(), dtype=torch.uint8, device=x.device) if mode else None
        L._check(lib.ghn3_pool_fwd(ctypes.byref(d), _ptr(xc), _ptr(y), _ptr(idx) if mode else None, _stream()), 'ghn3_pool_fwd')
        ctx.desc, ctx.idx = d, idx
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        d = ctx.desc
        do = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((d.N, d.C, d.H, d.W), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
        L._check(lib.ghn3_pool_bwd(ctypes.byref(d), _ptr(do), _ptr(ctx.idx) if d.mode else None, _ptr(dx), _stream()), 'ghn3_pool_bwd')
        return dx, None, None, None, None


def _as_int(v):
    """An int for a square kernel / stride / padding given as int or equal pair, else None."""
    if isinstance(v, (tuple, list)):
        return int(v[0]) if len(v) == 2 and v[0] == v[1] else None
    return int(v) if isinstance(v, int) else None


def run_pool(x, kernel_size, stride, padding, mode):
    """max (mode 1) / average (mode 0, valid taps only) pooling on the fused op where it applies; None otherwise."""
    k, s, p = _as_int(kernel_size), _as_int(stride), _as_int(padding)
    if k is None or s is None or p is None or not Pool2d.applicable(x, k, s, p):
        return None
    return Pool2d.apply(x, k, s, p, mode)


def conv_reference(x, w, gamma, beta, stride=1, padding=0, dilation=1, relu=True, eps=1e-5):
    """The stock layers ConvBn replaces (ops.py:186-193), functional form -- the parity reference of the tests."""
    y = F.conv2d(F.relu(x) if relu else x, w, None, stride, padding, dilation)
    return F.batch_norm(y, None, None, gamma, beta, True, 0.1, eps)


def run_conv_block(layers, x, keep_layout=False):
    """[ReLU, k x k Conv2d, BatchNorm2d] -- `ReLUConvBN` (ops.py:180-198) -- on the fused dense-convolution op where it applies,
    else layer by layer.  Same layout contract as run_block."""
    relu, conv, bn = layers
    w, gamma, beta = getattr(conv, 'weight', None), getattr(bn, 'weight', None), getattr(bn, 'bias', None)
    has_run = getattr(bn, 'running_mean', None) is not None
    batch_stats = getattr(bn, 'training', True) or not has_run
    ok = hasattr(bn, 'eps') and getattr(conv, 'bias', None) is None and hasattr(conv, 'kernel_size') and \
        not isinstance(conv.padding, str) and getattr(conv, 'groups', 1) == 1 and torch.is_tensor(w) and \
        len(set(_pair(getattr(conv, 'dilation', 1)))) == 1 and ConvBn.applicable(x, w, gamma, beta, batch_stats)
    if not ok:
        for m in layers:
            x = m(x)
        return x
    out, stats = conv_bn(x, w, gamma, beta, conv.stride, conv.padding, _pair(conv.dilation)[0], True, bn.eps)
    _update_running_stats(bn, stats, out, has_run)
    return out if (keep_layout or lazy_layout(conv, bn)) else out.contiguous(memory_format=torch.contiguous_format)


def _plain_conv(conv):
    """A bias-free, ungrouped convolution with numeric padding and one dilation for both axes (what the dense kernels take)."""
    return hasattr(conv, 'kernel_size') and getattr(conv, 'bias', None) is None and not isinstance(conv.padding, str) and \
        getattr(conv, 'groups', 1) == 1 and torch.is_tensor(getattr(conv, 'weight', None)) and \
        len(set(_pair(getattr(conv, 'dilation', 1)))) == 1


def run_conv_pair_block(layers, x, keep_layout=False):
    """[ReLU, 1 x k Conv2d, k x 1 Conv2d, BatchNorm2d] -- `ReLUConvBN(double=True)`, the `conv_1x7_7x1` op (ops.py:186-190,
    298) -- as two dense-convolution nodes: ReLU + the first convolution alone (ConvOnly), then the second one with the norm
    (ConvBn without a ReLU); the intermediate stays NHWC.  Else layer by layer."""
    relu, conv_a, conv_b, bn = layers
    gamma, beta = getattr(bn, 'weight', None), getattr(bn, 'bias', None)
    has_run = getattr(bn, 'running_mean', None) is not None
    batch_stats = getattr(bn, 'training', True) or not has_run
    ok = hasattr(bn, 'eps') and _plain_conv(conv_a) and _plain_conv(conv_b) and ConvOnly.applicable(x, conv_a.weight) and \
        conv_b.weight.shape[1] == conv_a.weight.shape[0]
    if ok:
        # (the second node's input has conv_a's channel count and x's type: checked on a stand-in of that shape)
        probe = x if conv_a.weight.shape[0] == x.shape[1] else x.new_empty((1, conv_a.weight.shape[0], 1, 1))
        ok = ConvBn.applicable(probe, conv_b.weight, gamma, beta, batch_stats)
    if not ok:
        for m in layers:
            x = m(x)
        return x
    y = conv_only(x, conv_a.weight, conv_a.stride, conv_a.padding, _pair(conv_a.dilation)[0], relu=True)
    out, stats = conv_bn(y, conv_b.weight, gamma, beta, conv_b.stride, conv_b.padding, _pair(conv_b.dilation)[0], False, bn.eps)
    _update_running_stats(bn, stats, out, has_run)
    return out if (keep_layout or lazy_layout(conv_a, conv_b, bn)) else out.contiguous(memory_format=torch.contiguous_format)


def run_conv_layer(conv, x):
    """A bare Conv2d without bias (the patch embedding of the ViT-style networks, ops.py:296 `conv_stride`) on the
    dense-convolution op -- 3-channel images padded as in run_layer_seq -- handing an NCHW tensor on; else the stock layer."""
    if _plain_conv(conv) and torch.is_tensor(x) and x.is_cuda and x.dim() == 4:
        w, xin = conv.weight, x
        if x.shape[1] == 3 and w.shape[1] == 3:
            xin, w = F.pad(x, (0, 0, 0, 0, 0, 1)), F.pad(w, (0, 0, 0, 0, 0, 1))
        if ConvOnly.applicable(xin, w):
            y = conv_only(xin, w, conv.stride, conv.padding, _pair(conv.dilation)[0], relu=False)
            return y.contiguous(memory_format=torch.contiguous_format)
    return conv(x)


def _is_kind(m, name):
    return type(m).__name__ == name


def run_layer_seq(seq, x):
    """A stem (`nn.Sequential` of Conv2d / BatchNorm2d / ReLU / MaxPool2d / Identity, ops.py:443-463) with every
    [Conv2d, BatchNorm2d] and [ReLU, Conv2d, BatchNorm2d] window on the fused dense-convolution op and the rest layer by layer.
    A 3-channel image (the first convolution of every network) is given a zero fourth channel -- and the weight a zero fourth
    input slice, through autograd -- because the kernels read channels in groups of four."""
    layers = list(seq)
    k, n = 0, len(layers)
    while k < n:
        m = layers[k]
        relu = _is_kind(m, 'ReLU') and k + 2 < n
        conv = layers[k + 1] if relu else m
        bn = layers[k + 2] if relu else (layers[k + 1] if k + 1 < n else None)
        done = False
        if _is_kind(conv, 'Conv2d') and bn is not None and _is_kind(bn, 'BatchNorm2d') and hasattr(bn, 'eps') and \
                _plain_conv(conv) and torch.is_tensor(x) and x.is_cuda and x.dim() == 4:
            w, gamma, beta = conv.weight, getattr(bn, 'weight', None), getattr(bn, 'bias', None)
            has_run = getattr(bn, 'running_mean', None) is not None
            batch_stats = getattr(bn, 'training', True) or not has_run
            xin = x
            if x.shape[1] == 3 and w.shape[1] == 3:
                xin = F.pad(x, (0, 0, 0, 0, 0, 1))
                w = F.pad(w, (0, 0, 0, 0, 0, 1))
            if ConvBn.applicable(xin, w, gamma, beta, batch_stats):
                fold = bool(relu)
                if relu and k == 0 and getattr(m, 'inplace', False):
                    # an in-place ReLU at the head of the sequence rewrites the CALLER's tensor (stem1 of the two-stem networks:
                    # the cells read relu(stem0's output) afterwards, ops.py:449,545-546): kept as the layer it is
                    xin, fold = m(xin), False
                x, stats = conv_bn(xin, w, gamma, beta, conv.stride, conv.padding, _pair(conv.dilation)[0], fold, bn.eps)
                _update_running_stats(bn, stats, x, has_run)
                if not lazy_layout(conv, bn):
                    x = x.contiguous(memory_format=torch.contiguous_format)
                k += 3 if relu else 2
                done = True
        if not done:
            x = m(x)
            k += 1
    return x


def run_factorized_reduce(relu, conv_1, conv_2, bn, x, stride=2, keep_layout=False):
    """`FactorizedReduce` (ops.py:163-178: ReLU, two 1 x 1 convolutions of stride 2 on the even and the odd pixel grid, concat,
    norm) as ONE dense-convolution node: a 2 x 2 kernel of stride 2 whose tap (0, 0) carries conv_1's weights for the first half
    of the output channels and whose tap (1, 1) carries conv_2's for the second half (the other entries are zeros) reads exactly
    the pixels the two strided convolutions read.  The 2 x 2 weight is assembled by differentiable torch ops, so the gradients
    reach conv_1 / conv_2 (views of the GHN's prediction buffer) through autograd.  Returns None when the fused op does not
    apply (the caller keeps the stock layers)."""
    w1, w2 = getattr(conv_1, 'weight', None), getattr(conv_2, 'weight', None)
    gamma, beta = getattr(bn, 'weight', None), getattr(bn, 'bias', None)
    has_run = getattr(bn, 'running_mean', None) is not None
    batch_stats = getattr(bn, 'training', True) or not has_run
    if not (stride == 2 and hasattr(bn, 'eps') and torch.is_tensor(w1) and torch.is_tensor(w2) and x.dim() == 4 and
            x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and getattr(conv_1, 'bias', None) is None and
            getattr(conv_2, 'bias', None) is None and w1.shape == w2.shape and tuple(w1.shape[2:]) == (1, 1)):
        return None
    half, C_in = int(w1.shape[0]), int(w1.shape[1])
    w = torch.zeros(2 * half, C_in, 2, 2, dtype=w1.dtype, device=w1.device)
    w[:half, :, 0, 0] = w1[:, :, 0, 0]
    w[half:, :, 1, 1] = w2[:, :, 0, 0]
    if not ConvBn.applicable(x, w, gamma, beta, batch_stats):
        return None
    out, stats = conv_bn(x, w, gamma, beta, 2, 0, 1, True, bn.eps)
    _update_running_stats(bn, stats, out, has_run)
    return out if (keep_layout or lazy_layout(conv_1, conv_2, bn)) else out.contiguous(memory_format=torch.contiguous_format)


def dwpw_bn(x, w_dw, w_pw, gamma, beta, stride=1, padding=0, dilation=1, eps=1e-5):
    """out = batch_norm(conv1x1(depthwise_conv(relu(x)))) with batch statistics; returns (out, stats) with
    stats = [mean | 1 / sqrt(var + eps) | biased variance] per output channel.  x: (N, C, H, W) fp32 CUDA tensor
    (channels_last preferred), w_dw (C, 1, ks, ks), w_pw (C_out, C, 1, 1) or (C_out, C)."""
    if not x.is_cuda:
        raise L.Ghn3Error('dwpw_bn runs on an MI355X only (no CPU implementation: use the stock torch layers)')
    return DwPwBn.apply(x, w_dw, w_pw.reshape(w_pw.shape[0], -1), gamma, beta, int(stride), int(padding), int(dilation),
                        float(eps))


def run_pointwise_block(layers, x, keep_layout=False):
    """[ReLU, 1 x 1 Conv2d, BatchNorm2d] -- `ReLUConvBN` with a 1 x 1 kernel (ops.py:180-198: the preprocessing layer of every
    cell, the `conv_1x1` op) -- on the fused op without a depthwise stage where it applies, else layer by layer."""
    relu, pw, bn = layers
    w_pw, gamma, beta = getattr(pw, 'weight', None), getattr(bn, 'weight', None), getattr(bn, 'bias', None)
    has_run = getattr(bn, 'running_mean', None) is not None
    batch_stats = getattr(bn, 'training', True) or not has_run
    ok = hasattr(bn, 'eps') and getattr(pw, 'bias', None) is None and hasattr(pw, 'kernel_size') and \
        tuple(pw.kernel_size) == (1, 1) and pw.stride[0] == pw.stride[1] and not isinstance(pw.padding, str) and \
        tuple(pw.padding) == (0, 0) and getattr(pw, 'groups', 1) == 1 and torch.is_tensor(w_pw) and \
        DwPwBn.applicable(x, None, w_pw, gamma, beta, 1, batch_stats)
    if not ok:
        # (e.g. more than 512 input channels -- the concatenated states of a wide cell: the dense-convolution op takes those)
        return run_conv_block(layers, x, keep_layout)
    out, stats = dwpw_bn(x, None, w_pw, gamma, beta, pw.stride[0], 0, 1, bn.eps)
    _update_running_stats(bn, stats, out, has_run)
    return out if (keep_layout or lazy_layout(pw, bn)) else out.contiguous(memory_format=torch.contiguous_format)


def _update_running_stats(bn, stats, out, has_run):
    if has_run and getattr(bn, 'training', True) and getattr(bn, 'track_running_stats', False):
        with torch.no_grad():
            C = bn.weight.numel()
            n = out.numel() // C
            # torch.nn.modules.batchnorm._BatchNorm.forward: the counter moves first; momentum=None = cumulative average
            nbt = getattr(bn, 'num_batches_tracked', None)
            if nbt is not None:
                nbt += 1
            if bn.momentum is not None:
                mom = float(bn.momentum)
            else:
                mom = 1.0 / float(nbt) if nbt is not None else 0.1
            bn.running_mean.mul_(1 - mom).add_(stats[:C], alpha=mom)
            bn.running_var.mul_(1 - mom).add_(stats[2 * C:] * (n / max(n - 1, 1)), alpha=mom)


def reference(x, w_dw, w_pw, gamma, beta, stride=1, padding=0, dilation=1, eps=1e-5):
    """The four stock layers the op replaces (ops.py:205-212), functional form -- the parity reference of the tests."""
    y = F.conv2d(F.relu(x), w_dw, None, stride, padding, dilation, groups=x.shape[1])
    zz = F.conv2d(y, w_pw.reshape(w_pw.shape[0], -1, 1, 1))
    return F.batch_norm(zz, None, None, gamma, beta, True, 0.1, eps)


def run_block(layers, x, keep_layout=False):
    """[ReLU, depthwise Conv2d, pointwise Conv2d, BatchNorm2d] (light or torch.nn flavour) on the fused op when it applies,
    else layer by layer.  Running statistics of a tracking BatchNorm are updated as torch does (momentum, unbiased variance).

    The fused op works on NHWC storage.  Its output is handed to the neighbouring (stock ATen / MIOpen) layers as a plain
    NCHW-contiguous tensor unless keep_layout is set (the next layer is another fused block): on this ROCm build the stock
    layers compute wrong parameter gradients for channels_last activations (tools/diag/target_ops_diag.py,
    profiles/r05c_target_ops_channels_last_diag.txt), so the layout must not leak into them -- one transposing copy per
    direction at the op's boundary until the neighbouring layers are native as well."""
    relu, dw, pw, bn = layers
    w_dw, w_pw = getattr(dw, 'weight', None), getattr(pw, 'weight', None)
    gamma, beta = getattr(bn, 'weight', None), getattr(bn, 'bias', None)
    has_run = getattr(bn, 'running_mean', None) is not None
    batch_stats = getattr(bn, 'training', True) or not has_run
    ks = dw.kernel_size[0] if hasattr(dw, 'kernel_size') else 0
    ok = hasattr(bn, 'eps') and getattr(dw, 'bias', None) is None and getattr(pw, 'bias', None) is None and \
        hasattr(dw, 'kernel_size') and dw.kernel_size[0] == dw.kernel_size[1] and dw.stride[0] == dw.stride[1] and \
        not isinstance(dw.padding, str) and dw.padding[0] == dw.padding[1] and dw.dilation[0] == dw.dilation[1] and \
        getattr(dw, 'groups', 1) == x.shape[1] and tuple(pw.kernel_size) == (1, 1) and \
        tuple(getattr(pw, 'stride', (1, 1))) == (1, 1) and not isinstance(getattr(pw, 'padding', 0), str) and \
        tuple(getattr(pw, 'padding', (0, 0))) == (0, 0) and getattr(pw, 'groups', 1) == 1 and \
        tuple(getattr(pw, 'dilation', (1, 1))) == (1, 1) and \
        DwPwBn.applicable(x, w_dw, w_pw, gamma, beta, ks, batch_stats)
    if not ok:
        for m in layers:
            x = m(x)
        return x
    out, stats = dwpw_bn(x, w_dw, w_pw, gamma, beta, dw.stride[0], dw.padding[0], dw.dilation[0], bn.eps)
    _update_running_stats(bn, stats, out, has_run)
    return out if (keep_layout or lazy_layout(dw, pw, bn)) else out.contiguous(memory_format=torch.contiguous_format)
