"""
Light target-network layers (behaviour of /root/reference/ghn3/light_ops.py:26-337 and the two
base classes of /root/reference/ghn3/ops.py:28-101).

A target network whose weights a GHN predicts does not need ``nn.Module``: no buffers, hooks, state dicts or
parameter registration -- constructing the thousands of ``nn.Module`` layers of a training meta-batch costs more than
running them.  A light layer only remembers its hyper-parameters and the SHAPES of its ``weight`` / ``bias`` (lists)
until the GHN assigns tensors (``layer.weight = tensor``: views of the flat prediction buffer of
``ghn3_amd.GHN3.forward``, autograd-connected in training).  ``forward`` is the functional form of the torch layer.

Differences from the reference, all additive: the classes live at module level (they pickle without the reference's
``_InitializeModule`` shim, ops.py:588-597), ``to()`` returns the module, and ``train()`` / ``eval()`` /
``named_parameters()`` exist.
"""

import numbers
import os
import operator
from collections import OrderedDict
from itertools import islice

import torch
import torch.nn as nn
import torch.nn.functional as F

__all__ = ['ModuleEmpty', 'Module', 'Sequential', 'ModuleList', 'AvgPool2d', 'MaxPool2d', 'AdaptiveAvgPool2d', 'ReLU',
           'GELU', 'Hardswish', 'Identity', 'Dropout', 'Conv2d', 'Linear', 'BatchNorm2d', 'LayerNorm']

_PARAM_NAMES = ('weight', 'bias')


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class ModuleEmpty:
    """Layer without parameters (ops.py:28-53): a training flag, child table and ``__call__`` -> ``forward``."""

    def __init__(self):
        object.__setattr__(self, 'training', True)
        object.__setattr__(self, '_modules', {})

    def add_module(self, name, module):
        self.__dict__['_modules'][name] = module

    def named_modules(self, memo=None, prefix=''):
        # (the reference's parameter-free base reports no modules at all, ops.py:46-47)
        return iter(())

    def children(self):
        return iter(self.__dict__['_modules'].values())

    def to(self, *args, **kwargs):
        return self

    def train(self, mode=True):
        self.training = mode
        for m in self.__dict__['_modules'].values():
            if hasattr(m, 'train'):
                m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def __call__(self, *inputs, **kwargs):
        return self.forward(*inputs, **kwargs)

    def __repr__(self):
        return '%s()' % type(self).__name__


class Module(ModuleEmpty):
    """Layer with parameters and / or children (ops.py:55-101).  Attribute assignment files children under
    ``_modules`` and ``weight`` / ``bias`` (tensors, shape lists or None) under ``_parameters``."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, '_parameters', {})

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, ModuleEmpty)):
            self.__dict__['_modules'][name] = value
        elif isinstance(value, torch.Tensor) or \
                (name in _PARAM_NAMES and (value is None or isinstance(value, (list, tuple)))):
            self.__dict__['_parameters'][name] = value
        object.__setattr__(self, name, value)

    def parameters(self, recurse=True):
        for _, p in self.named_parameters(recurse=recurse):
            yield p

    def named_parameters(self, prefix='', recurse=True):
        """Assigned tensors only (a shape list is not a parameter yet)."""
        for n, p in self.__dict__['_parameters'].items():
            if isinstance(p, torch.Tensor):
                yield prefix + n, p
        if recurse:
            # (torch.nn children -- the auxiliary heads -- own their parameters and are not listed, ops.py:74-79)
            for name, m in self.__dict__['_modules'].items():
                if isinstance(m, Module):
                    yield from m.named_parameters(prefix + name + '.', True)

    def named_modules(self, memo=None, prefix='', remove_duplicate=True):
        """Depth-first over the layers that can carry parameters (parameter-free children are skipped, ops.py:84-90)."""
        memo = set() if memo is None else memo
        if id(self) in memo:
            return
        if remove_duplicate:
            memo.add(id(self))
        yield prefix, self
        for name, m in self.__dict__['_modules'].items():
            if isinstance(m, Module):
                yield from m.named_modules(memo, prefix + ('.' if prefix else '') + name, remove_duplicate)

    def shapes(self):
        """name -> shape of every weight / bias below this module, assigned or not."""
        out = OrderedDict()
        for name, m in self.named_modules():
            for n, p in m.__dict__['_parameters'].items():
                if p is not None:
                    out[name + ('.' if name else '') + n] = tuple(p) if isinstance(p, (list, tuple)) else tuple(p.shape)
        return out


# ---- containers ------------------------------------------------------------------------------------
class _Container(Module):
    def __len__(self):
        return len(self._modules)

    def __iter__(self):
        return iter(self._modules.values())

    def __dir__(self):
        return [k for k in super().__dir__() if not k.isdigit()]

    def append(self, module):
        self.add_module(str(len(self)), module)
        return self


class Sequential(_Container):
    """light_ops.py:28-77."""

    def __init__(self, *args):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for k, module in enumerate(args):
                self.add_module(str(k), module)

    def forward(self, x):
        for module in self._modules.values():
            x = module(x)
        return x

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return type(self)(OrderedDict(list(self._modules.items())[idx]))
        n = len(self)
        i = operator.index(idx)
        if not -n <= i < n:
            raise IndexError('index {} is out of range'.format(idx))
        return next(islice(self._modules.values(), i % n, None))


class ModuleList(_Container):
    """light_ops.py:79-124."""

    def __init__(self, modules=None):
        super().__init__()
        if modules is not None:
            self.extend(modules)

    def _key(self, idx):
        i = operator.index(idx)
        if not -len(self) <= i < len(self):
            raise IndexError('index {} is out of range'.format(idx))
        return str(i + len(self) if i < 0 else i)

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return type(self)(list(self._modules.values())[idx])
        return self._modules[self._key(idx)]

    def __iadd__(self, modules):
        return self.extend(modules)

    def __add__(self, other):
        return ModuleList(list(self) + list(other))

    def extend(self, modules):
        for m in modules:
            self.append(m)
        return self


# ---- parameter-free layers ---------------------------------------------------------------------------
def _stock_layout(x):
    """Round 6 ("lazy layout"): the fused HIP layers of ghn3_amd.target_ops hand their NHWC (channels_last) activations on as
    they are -- element-wise ops, concatenation and the next fused layer take them without a copy -- and the stock layers
    that are WRONG on channels_last tensors on this ROCm build convert, here: MIOpen convolution / BatchNorm (wrong parameter
    gradients, profiles/r05c_*) and ATen's max / average pooling (wrong input gradients: tools/diag/lazy_layout_diag.py,
    profiles/r06v_*).  A no-op for the NCHW tensors of a network without fused layers."""
    if x.is_cuda and x.dim() == 4 and not x.is_contiguous():
        return x.contiguous()
    return x


class AvgPool2d(ModuleEmpty):
    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True,
                 divisor_override=None):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, kernel_size if stride is None else stride
        self.padding, self.ceil_mode = padding, ceil_mode
        self.count_include_pad, self.divisor_override = count_include_pad, divisor_override

    def forward(self, x):
        if x.is_cuda and not self.ceil_mode and not self.count_include_pad and self.divisor_override is None:
            from . import target_ops                     # (round 6: NHWC pooling on the fused op; its output stays NHWC)
            y = target_ops.run_pool(x, self.kernel_size, self.stride, self.padding, 0)
            if y is not None:
                return y
        return F.avg_pool2d(_stock_layout(x), self.kernel_size, self.stride, self.padding, self.ceil_mode,
                            self.count_include_pad, self.divisor_override)


class MaxPool2d(ModuleEmpty):
    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False):
        super().__init__()
        self.kernel_size, self.stride = kernel_size, kernel_size if stride is None else stride
        self.padding, self.dilation = padding, dilation
        self.return_indices, self.ceil_mode = return_indices, ceil_mode

    def forward(self, x):
        if x.is_cuda and not self.ceil_mode and not self.return_indices and self.dilation in (1, (1, 1)):
            from . import target_ops
            y = target_ops.run_pool(x, self.kernel_size, self.stride, self.padding, 1)
            if y is not None:
                return y
        return F.max_pool2d(_stock_layout(x), self.kernel_size, self.stride, self.padding, self.dilation,
                            ceil_mode=self.ceil_mode, return_indices=self.return_indices)


class AdaptiveAvgPool2d(ModuleEmpty):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return F.adaptive_avg_pool2d(_stock_layout(x), self.output_size)


class ReLU(ModuleEmpty):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return F.relu(x, inplace=self.inplace)


class GELU(ModuleEmpty):
    def __init__(self, approximate='none'):
        super().__init__()
        self.approximate = approximate

    def forward(self, x):
        return F.gelu(x, approximate=self.approximate)


class Hardswish(ModuleEmpty):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return F.hardswish(x, self.inplace)


class Identity(ModuleEmpty):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x


class Dropout(ModuleEmpty):
    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p, self.inplace = p, inplace

    def forward(self, x):
        return F.dropout(x, self.p, self.training, self.inplace)


# ---- layers with predicted parameters -----------------------------------------------------------------
class Conv2d(Module):
    """light_ops.py:213-245: ``weight`` = [out, in / groups, kh, kw] until a tensor is assigned."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode='zeros', device=None, dtype=None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding = padding if isinstance(padding, str) else _pair(padding)
        self.dilation, self.groups, self.padding_mode = _pair(dilation), groups, padding_mode
        self.weight = [out_channels, in_channels // groups, *self.kernel_size]
        self.bias = [out_channels] if bias else None

    def forward(self, x):
        return F.conv2d(_stock_layout(x), self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class Linear(Module):
    """light_ops.py:247-264."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = [out_features, in_features]
        self.bias = [out_features] if bias else None

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class BatchNorm2d(Module):
    """light_ops.py:266-311: affine, batch statistics whenever no running statistics exist (they never do while a GHN
    trains: ``track_running_stats`` must be off, light_ops.py:283)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=False, device=None,
                 dtype=None):
        super().__init__()
        assert affine and not track_running_stats, 'assumed affine and that running stats is not updated'
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.running_mean = self.running_var = self.num_batches_tracked = None
        self.weight = [num_features]
        self.bias = [num_features]

    def forward(self, x):
        use_running = (not self.training) or self.track_running_stats
        batch_stats = self.training or (self.running_mean is None and self.running_var is None)
        x = _stock_layout(x)
        return F.batch_norm(x, self.running_mean if use_running else None, self.running_var if use_running else None,
                            self.weight, self.bias, batch_stats, 0.0 if self.momentum is None else self.momentum,
                            self.eps)


class LayerNorm(Module):
    """light_ops.py:313-331."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True, device=None, dtype=None):
        super().__init__()
        assert elementwise_affine
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        self.normalized_shape = tuple(normalized_shape)
        self.eps, self.elementwise_affine = eps, elementwise_affine
        self.weight = list(self.normalized_shape)
        self.bias = list(self.normalized_shape)

    def forward(self, x):
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
