"""Host-only: decoder families of the bench workload (rows, widths, row tiles of the 8-phase kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ghn3_amd.program import Program
from ghn3_amd.synthetic import synthetic_batch
from ghn3_amd import _lib as L
import bench

model = sys.argv[1] if len(sys.argv) > 1 else 'ghn3xlm16'
nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 256
graphs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = bench.model_cfg(model)
gb, nets = synthetic_batch([nodes] * graphs, nodes * 1000)
gb._cat()
prog = Program(dict(hid=cfg['hid'], heads=cfg['heads'], layers=cfg['layers'], num_classes=10, max_shape=cfg.get('max_shape', (64, 64, 11, 11)) if isinstance(cfg, dict) else None),
               gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, nets, training=True,
               decoder_ctype=L.COMPUTE_TYPES['f16'], decoder_bwd_ctype=L.COMPUTE_TYPES['f16'], direct16=True, side_stream=True,
               graphormer_x3=True)
print('decoder rows', prog.M)
for gg in prog.gemm_groups:
    print('i_ld', gg['i_ld'], 'rows', gg['rows'], 'cols', gg['cols'], 'family', gg['family'], 'p8', gg.get('p8'),
          'subs', [(s['rows'], s['cols']) for s in gg['subs']], 'mtiles', gg.get('mtiles', np.zeros((0, 3))).tolist())
tot = {}
for p in prog.predicted:
    sh = tuple(p['shape']) if 'shape' in p else None
    k = 'hw>1' if sh and len(sh) == 4 and sh[2] * sh[3] > 1 else ('1x1' if sh and len(sh) == 4 else ('2d' if sh and len(sh) == 2 else '1d'))
    tot[k] = tot.get(k, 0) + p['numel']
print('predicted elements by kind', tot, 'fwd blocks', prog.fwd_blk[0], 'bwd blocks', prog.bwd_blk[0], 'tile lds', prog.tile_lds)
import collections
wg = [p for p in prog.problems if int(p['lda']) == int(p['ldb']) and (int(p['flags']) & L.GEMM_OP16) and int(p['N']) == 8 * prog.C and int(p['c_q']) > 0]
tot_tiles = 0; tot_fl = 0
print('wgrad band problems (M, K, tiles of 256x256, k-tiles):')
for p in wg:
    M, N, K = int(p['M']), int(p['N']), int(p['K'])
    t = ((M + 255) // 256) * ((N + 255) // 256)
    tot_tiles += t; tot_fl += 2.0 * M * N * K
    print('  M %6d K %4d tiles %5d ktiles %2d' % (M, K, t, (K + 63) // 64))
print('total tiles', tot_tiles, 'GFLOP (padded K not counted)', tot_fl / 1e9)
