"""CPU checks of the C-ABI boundary: the shared library loads, exports every symbol include/ghn3_hip.h declares,
the numpy/ctypes struct mirrors have the C sizes, and the product fails loudly without a GPU."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch

from ghn3_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'ghn3_hip.h')).read()
    return sorted(set(re.findall(r'\b(ghn3_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        from ghn3_amd import build
        build.build(verbose=False)
    h = ctypes.CDLL(L.LIB_PATH)
    syms = _declared_symbols()
    assert set(L.EXPORTS) == set(syms), (sorted(set(syms) ^ set(L.EXPORTS)))
    for s in syms:
        assert hasattr(h, s), s
    assert L.load().ghn3_abi_version() == L.ABI_VERSION


def test_struct_mirrors_match_header_layout(tmp_path):
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include "ghn3_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %d %zu\\n",'
                   'sizeof(ghn3_ref),sizeof(ghn3_gemm_problem),sizeof(ghn3_tile_desc),sizeof(ghn3_op),'
                   'sizeof(ghn3_cast_desc),'
                   '(int)GHN3_OP_KIND_COUNT,sizeof(ghn3_dwpw_desc));return 0;}\n')
    exe = tmp_path / 'sz'
    import subprocess
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    from ghn3_amd import target_ops
    assert [int(v) for v in out] == [L.REF_DT.itemsize, L.PROBLEM_DT.itemsize, L.TILE_DT.itemsize,
                                     L.OP_DT.itemsize, L.CAST_DT.itemsize, L.OP_KIND_COUNT,
                                     ctypes.sizeof(target_ops._Desc)]


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_no_cpu_fallback():
    from ghn3_amd import GHN3
    from ghn3_amd.synthetic import synthetic_batch
    ghn = GHN3(max_shape=(32, 32, 16, 16), num_classes=10, hid=32, heads=8, layers=1, weight_norm=True, ve=True,
               layernorm=True)
    gb, nets = synthetic_batch([8], 1)
    with pytest.raises(L.Ghn3Error):
        ghn(nets, gb)
    with pytest.raises(L.Ghn3Error):
        L.context(0)


def test_state_dict_layout_and_param_count():
    from ghn3_amd import GHN3
    import recipe
    ghn = GHN3(max_shape=(64, 64, 16, 16), num_classes=1000, hid=64, heads=8, layers=3, weight_norm=True, ve=True,
               layernorm=True)
    assert sum(p.numel() for p in ghn.parameters()) == 6906632 == recipe.count_params(64, 3, 8, 1000)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'ghn3_tiny.npz'))
    tiny = GHN3(**recipe.TINY_CFG)
    sd = tiny.state_dict()
    assert sorted(sd) == [str(k) for k in g['meta/state_keys']]
    assert [str(tuple(sd[k].shape)) for k in sorted(sd)] == [str(s) for s in g['meta/state_shapes']]
    # all parameters are views of one flat buffer, and survive load_state_dict / HF-style key layout
    hf = {k.replace('gnn.0.centrality', 'centrality').replace('gnn.0.input_dist', 'input_dist'): v.clone()
          for k, v in sd.items()}
    tiny.load_state_dict(hf)
    base = tiny._flat.data_ptr()
    for p in tiny.parameters():
        assert base <= p.data_ptr() < base + 4 * tiny._flat_numel
