#!/usr/bin/env python3
"""Host-only: dumps the 8-phase (tile code 28) problems of the bench workload's W2 forward or dgrad launch as text for
tools/p8_replay (GPU box): the exact problem table the step runs -- row-tile tables, XCD pins, K chunks, k-map -- so that
tilings can be compared kernel-alone with in-kernel cycle stamps.
    python tools/p8_dump.py fwd|dgrad <out.txt> [graphs per GPU]      (tiling switches: the GHN3_P8_* environment)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ghn3_amd.program import Program
from ghn3_amd.synthetic import synthetic_batch
from ghn3_amd import _lib as L
import bench

which, out = sys.argv[1], sys.argv[2]
graphs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = bench.model_cfg('ghn3xlm16')
gb, nets = synthetic_batch([256] * graphs, 256000)
gb._cat()
prog = Program(dict(hid=cfg['hid'], heads=cfg['heads'], layers=cfg['layers'], num_classes=10, max_shape=cfg.get('max_shape')),
               gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets, training=True,
               decoder_ctype=L.CT_F16, decoder_bwd_ctype=L.CT_F16, direct16=True, side_stream=True, graphormer_x3=True)
P, idx = prog.problems, prog.idx_blob
ws, sh = prog.xbuf(prog.X_WS), prog.xbuf(prog.X_SHADOW)
sel = [p for p in P if int(p['n_mtiles']) > 0 and (int(p['N']) == 8 * prog.C) == (which == 'dgrad')]
kind = {ws: 0, sh: 1}
flops = 0.0
with open(out, 'w') as fh:
    fh.write('%d %d %d\n' % (len(sel), prog.ws_bytes, prog.shadow_layout(prog.C, prog.max_shape, prog.Lyr)['nbytes']))
    for p in sel:
        n = int(p['n_mtiles'])
        off = int(p['mtiles']['off'])
        mt = np.frombuffer(idx[off:off + 12 * n].tobytes(), dtype=np.int32)
        vals = [int(p[k]) for k in ('M', 'N', 'K', 'lda', 'ldb', 'ldc')] + \
            [kind[int(p['A']['buf'])], int(p['A']['off']), kind[int(p['B']['buf'])], int(p['B']['off']),
             kind[int(p['C']['buf'])], int(p['C']['off'])] + \
            [int(p[k]) for k in ('b_q', 'b_s', 'b_kq', 'b_ks', 'c_q', 'c_s', 'lim_kind', 'xcd_pin')] + [n] + mt.tolist()
        fh.write(' '.join(str(v) for v in vals) + '\n')
        for (m0, code, ext) in mt.reshape(-1, 3):
            h = 64 * code if code <= 5 else 32 * code
            rows = max(0, min(int(p['M']) - m0, h))
            if int(p['lim_kind']) == 2:
                flops += 2.0 * h * int(p['N']) * min(int(p['K']), max(int(ext), 0))
            else:
                flops += 2.0 * h * min(int(p['N']), max(int(ext), 0)) * int(p['K'])
print(which, 'problems', len(sel), 'padded GFLOP', flops / 1e9, 'algorithmic GFLOP',
      prog.tag_flops[prog.TAG_D3_DGRAD if which == 'dgrad' else prog.TAG_D3_FWD] / 1e9, 'sub', getattr(prog, 'dgrad_sub', None))
