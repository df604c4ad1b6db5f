"""
ctypes binding of libghn3_hip.so (include/ghn3_hip.h).  The library is the only compute path of this
package: importing works anywhere (so that host logic can be tested on CPU), but every compute entry
point raises if the shared object or a GPU is missing -- there is no CPU fallback.
"""

import ctypes
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libghn3_hip.so')
ABI_VERSION = 19

# ---- numpy mirrors of the C structs -------------------------------------------------------------
REF_DT = np.dtype([('buf', '<i4'), ('_pad', '<i4'), ('off', '<i8')])
_REF_NAMES = ('A', 'B', 'C', 'bias', 'residual', 'aux_in', 'aux_out', 'a_gather', 'b_gather', 'c_gather')
_INT_NAMES = ('M', 'N', 'K', 'lda', 'ldb', 'ldc', 'a_mode', 'b_mode', 'a_q', 'a_s', 'b_q', 'b_s', 'c_q', 'c_s',
              'bias_q', 'bias_s', 'bias_stride', 'act', 'dact', 'flags')
PROBLEM_DT = np.dtype([(n, REF_DT) for n in _REF_NAMES] + [(n, '<i4') for n in _INT_NAMES] +
                      [('alpha', '<f4'), ('ksplit', '<i4'), ('b_kq', '<i4'), ('b_ks', '<i4'), ('lim', REF_DT), ('lim_kind', '<i4'),
                       ('xcd_pin', '<i4'), ('alpha_amax', REF_DT), ('ln_p', REF_DT, 6), ('ln_kind', '<i4'),
                       ('ln_eps', '<f4'), ('B2', REF_DT), ('x3_slice', '<i4'), ('_pad3', '<i4'), ('mtiles', REF_DT),
                       ('n_mtiles', '<i4'), ('_pad4', '<i4')])
TILE_DT = np.dtype([('dst_off', '<i8'), ('src_off', '<i8'), ('S', '<i8', 4), ('T', '<i4', 4), ('E', '<i4', 4),
                    ('R', '<i4', 4), ('src_buf', '<i4'), ('mode', '<i4'), ('scale', '<f4'), ('_pad', '<i4')])
CAST_DT = np.dtype([('src_off', '<i8'), ('dst_off', '<i8'), ('dstT_off', '<i8'), ('rows', '<i4'), ('cols', '<i4'),
                    ('ld_src', '<i4'), ('ld_dst', '<i4'), ('ld_dstT', '<i4'), ('flags', '<u4'), ('bias_q', '<i4'),
                    ('bias_s', '<i4'), ('block_start', '<i4'), ('bias_off', '<i4'), ('src_q', '<i4'), ('src_s', '<i4'),
                    ('lo_off', '<i8'), ('part_off', '<i8')])
ROWSET_DT = np.dtype([('off', '<i8'), ('rows', '<i4'), ('o', '<i4'), ('i', '<i4'), ('ld', '<i4'), ('i0', '<i4'),
                      ('_pad', '<i4')])
OP_DT = np.dtype([('kind', '<i4'), ('flags', '<i4'), ('i', '<i8', 8), ('f', '<f4', 4), ('r', REF_DT, 16)])
assert REF_DT.itemsize == 16 and PROBLEM_DT.itemsize == 448 and TILE_DT.itemsize == 112 and OP_DT.itemsize == 344
assert CAST_DT.itemsize == 88 and ROWSET_DT.itemsize == 32

MODE_ROW, MODE_COL = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
DACT_NONE, DACT_RELU, DACT_GELU = 0, 1, 2
GEMM_ACCUM = 1
GEMM_BIASGRAD = 2
GEMM_OP16 = 4
GEMM_X3 = 8
GEMM_SUMSQ = 16
GEMM_X3F16 = 32
X3F16_WSHIFT = 6
CAST_STRAIGHT, CAST_TRANSPOSED, CAST_STRAIGHT_BF16, CAST_TRANSPOSED_BF16, CAST_COLSUM, CAST_SCALED = 1, 2, 4, 8, 16, 32
CAST_TIGHT = 64
CAST_SPLIT = 128
CAST_COLSUM_PARTS = 256
CAST_FRAG = 512
CAST_SRC16 = 1024
CAST_SPLIT_F16 = 2048
CT_F32, CT_F16, CT_BF16 = 0, 1, 2
COMPUTE_TYPES = {'f32': CT_F32, 'f16': CT_F16, 'bf16': CT_BF16}

(OP_NOP, OP_GEMM, OP_GRAPH_PROLOGUE, OP_EMBED_NODES, OP_EDGE_HIDDEN, OP_BIAS_GATHER, OP_LAYERNORM_FWD,
 OP_ATTN_FWD, OP_TILE_FWD, OP_PARAM_NORM_FWD, OP_PARAM_NORM_BWD, OP_TILE_BWD, OP_COLSUM, OP_ROWSEG_SUM,
 OP_LAYERNORM_BWD, OP_LN_PARAM_GRAD, OP_ATTN_BWD, OP_BIAS_HIST, OP_EDGE_HIDDEN_BWD, OP_EMBED_BWD, OP_MEMSET0,
 OP_ADD, OP_DACT, OP_CAST16, OP_JOIN, OP_DETACH, OP_SUMSQ, OP_ADAMW, OP_RELU_FIX, OP_ROWSET_COLSUM, OP_WIRE_PACK,
 OP_RANK_REDUCE, OP_TRANSPOSE32, OP_PARAM_NORM_FIN, OP_LN_PARAM_GRAD_BATCH, OP_ADAMW_CAST16, OP_KIND_COUNT) = range(37)
OP_NAMES = ['nop', 'gemm', 'graph_prologue', 'embed_nodes', 'edge_hidden', 'bias_gather', 'layernorm_fwd',
            'attn_fwd', 'tile_fwd', 'param_norm_fwd', 'param_norm_bwd', 'tile_bwd', 'colsum', 'rowseg_sum',
            'layernorm_bwd', 'ln_param_grad', 'attn_bwd', 'bias_hist', 'edge_hidden_bwd', 'embed_bwd', 'memset0',
            'add', 'dact', 'cast16', 'join', 'detach', 'sumsq', 'adamw', 'relu_fix', 'rowset_colsum', 'wire_pack',
            'rank_reduce', 'transpose32', 'param_norm_fin', 'ln_param_grad_batch', 'adamw_cast16']

EXPORTS = ['ghn3_abi_version', 'ghn3_last_error', 'ghn3_ctx_create', 'ghn3_ctx_destroy',
           'ghn3_ctx_set_compute_type', 'ghn3_ctx_side_wait', 'ghn3_ctx_side_pending', 'ghn3_ctx_cache_stats', 'ghn3_run', 'ghn3_event_create', 'ghn3_event_record',
           'ghn3_event_elapsed_ms', 'ghn3_event_destroy', 'ghn3_profile_enable', 'ghn3_profile_read',
           'ghn3_profile_read_tags', 'ghn3_dwpw_scratch_floats', 'ghn3_dwpw_bn_fwd', 'ghn3_dwpw_bn_bwd',
           'ghn3_conv_scratch_floats', 'ghn3_conv_bn_fwd', 'ghn3_conv_bn_bwd', 'ghn3_se_fwd', 'ghn3_se_bwd', 'ghn3_pool_fwd', 'ghn3_pool_bwd']
OPFLAG_TIMED = 0x100
OPFLAG_SIDE = 0x200

_lib = None
_lock = threading.Lock()


class Ghn3Error(RuntimeError):
    pass


def load():
    """dlopen the library (no GPU needed) and declare prototypes."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise Ghn3Error('%s is missing: build it with `python -m ghn3_amd.build` (hipcc, gfx950). '
                            'ghn3_amd has no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.ghn3_abi_version.restype = ctypes.c_int
        lib.ghn3_last_error.restype = ctypes.c_char_p
        lib.ghn3_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        lib.ghn3_ctx_destroy.argtypes = [ctypes.c_void_p]
        lib.ghn3_ctx_destroy.restype = None
        lib.ghn3_ctx_set_compute_type.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ghn3_ctx_side_wait.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.ghn3_ctx_side_pending.argtypes = [ctypes.c_void_p]
        lib.ghn3_ctx_cache_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        lib.ghn3_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        lib.ghn3_event_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        lib.ghn3_event_record.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.ghn3_event_elapsed_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        lib.ghn3_event_destroy.argtypes = [ctypes.c_void_p]
        lib.ghn3_profile_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ghn3_profile_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ghn3_profile_read_tags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        lib.ghn3_dwpw_scratch_floats.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ghn3_dwpw_scratch_floats.restype = ctypes.c_int64
        lib.ghn3_dwpw_bn_fwd.argtypes = [ctypes.c_void_p] * 11
        lib.ghn3_dwpw_bn_bwd.argtypes = [ctypes.c_void_p] * 15
        lib.ghn3_conv_scratch_floats.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lib.ghn3_conv_scratch_floats.restype = ctypes.c_int64
        lib.ghn3_conv_bn_fwd.argtypes = [ctypes.c_void_p] * 10
        lib.ghn3_conv_bn_bwd.argtypes = [ctypes.c_void_p] * 13
        lib.ghn3_se_fwd.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 8
        lib.ghn3_se_bwd.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 12
        lib.ghn3_pool_fwd.argtypes = [ctypes.c_void_p] * 5
        lib.ghn3_pool_bwd.argtypes = [ctypes.c_void_p] * 5
        if lib.ghn3_abi_version() != ABI_VERSION:
            raise Ghn3Error('libghn3_hip.so ABI %d != expected %d: rebuild' % (lib.ghn3_abi_version(), ABI_VERSION))
        _lib = lib
        return lib


def _check(rc, what):
    if rc != 0:
        raise Ghn3Error('%s failed (%d): %s' % (what, rc, load().ghn3_last_error().decode()))


class Context:
    """One per process and device (one process per GPU)."""

    def __init__(self):
        lib = load()
        h = ctypes.c_void_p()
        _check(lib.ghn3_ctx_create(ctypes.byref(h)), 'ghn3_ctx_create')
        self._h = h
        self._lib = lib

    def set_compute_type(self, name):
        _check(self._lib.ghn3_ctx_set_compute_type(self._h, COMPUTE_TYPES[name]), 'ghn3_ctx_set_compute_type')

    def run(self, ops, problems, buf_ptrs, stream):
        """ops: OP_DT array; problems: PROBLEM_DT array; buf_ptrs: uint64 array of device pointers."""
        assert ops.dtype == OP_DT and problems.dtype == PROBLEM_DT and buf_ptrs.dtype == np.uint64
        rc = self._lib.ghn3_run(self._h, ops.ctypes.data, len(ops),
                                problems.ctypes.data if len(problems) else None, len(problems),
                                buf_ptrs.ctypes.data, len(buf_ptrs), ctypes.c_void_p(stream))
        _check(rc, 'ghn3_run')

    def side_wait(self, stream):
        """Make `stream` (a hipStream_t as int) wait for the side-stream work issued so far."""
        _check(self._lib.ghn3_ctx_side_wait(self._h, ctypes.c_void_p(stream)), 'ghn3_ctx_side_wait')

    def side_pending(self):
        """True when a DETACHed run's side-stream work has not been joined by a later run yet."""
        return self._lib.ghn3_ctx_side_pending(self._h) == 1

    def cache_stats(self):
        """(hits, misses) of ghn3_run's store of resolved problem tables since the context was created."""
        h, m = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.ghn3_ctx_cache_stats(self._h, ctypes.byref(h), ctypes.byref(m)), 'ghn3_ctx_cache_stats')
        return int(h.value), int(m.value)

    def profile(self, mode):
        """0 off, 1 every op (synchronising), 2 only ops flagged OPFLAG_TIMED (no sync until read_tags)."""
        _check(self._lib.ghn3_profile_enable(self._h, int(mode)), 'ghn3_profile_enable')

    def profile_read_tags(self, reset=True):
        ms = np.zeros(256, dtype=np.float64)
        n = np.zeros(256, dtype=np.int64)
        _check(self._lib.ghn3_profile_read_tags(self._h, ms.ctypes.data, n.ctypes.data, int(reset)),
               'ghn3_profile_read_tags')
        return {k: (float(ms[k]), int(n[k])) for k in range(256) if n[k]}

    def profile_read(self, reset=True):
        ms = np.zeros(OP_KIND_COUNT, dtype=np.float64)
        n = np.zeros(OP_KIND_COUNT, dtype=np.int64)
        _check(self._lib.ghn3_profile_read(self._h, ms.ctypes.data, n.ctypes.data, int(reset)), 'ghn3_profile_read')
        return {OP_NAMES[k]: (float(ms[k]), int(n[k])) for k in range(1, OP_KIND_COUNT) if n[k]}

    def __del__(self):
        try:
            if self._h:
                self._lib.ghn3_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass


_ctx = {}


def context(device_index=0):
    import torch
    if not torch.cuda.is_available():
        raise Ghn3Error('ghn3_amd needs an MI355X (HIP device); no CPU fallback exists. '
                        'Use oracle/ only as a test checker.')
    if device_index not in _ctx:
        with torch.cuda.device(device_index):
            _ctx[device_index] = Context()
    return _ctx[device_index]


class Event:
    """HIP event on an explicit stream (torch.cuda.Event only sees torch's current stream)."""

    def __init__(self):
        self._lib = load()
        h = ctypes.c_void_p()
        _check(self._lib.ghn3_event_create(ctypes.byref(h)), 'ghn3_event_create')
        self._h = h

    def record(self, stream):
        _check(self._lib.ghn3_event_record(self._h, ctypes.c_void_p(stream)), 'ghn3_event_record')

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _check(self._lib.ghn3_event_elapsed_ms(self._h, stop._h, ctypes.byref(ms)), 'ghn3_event_elapsed_ms')
        return ms.value

    def __del__(self):
        try:
            self._lib.ghn3_event_destroy(self._h)
        except Exception:
            pass
