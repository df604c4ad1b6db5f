#!/usr/bin/env python3
"""HBM traffic of the decoder W2 GEMM family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch.db> <write.db> <steps in each run> [bench line of the WRITE_SIZE run]  > profiles/<tag>_pmc_traffic_xl_f16.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB of memory-side (fabric)
requests; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B / lane) loads at 64 bytes, so it is doubled
-- every load of gemm_h16d_kernel is a 16 B / lane global_load_lds_dwordx4 or dwordx4 load.  WRITE_SIZE is calibrated
in the same run against a store of known size: the zero-fill of the program workspace (torch FillFunctor, 16 B / lane
stores): counter / true bytes is printed as `write_calibration` and divided out.
Infinity-Cache hits are counted, so this is traffic at the L2 <-> fabric boundary (an upper bound of HBM bytes)."""
import json
import sqlite3
import sys
from collections import defaultdict


def sums(db, counter):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(counters_collection)')]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    agg, calls = defaultdict(float), defaultdict(set)
    for name, did, cname, val in cur.execute('select %s, dispatch_id, counter_name, value from counters_collection'
                                              % name_col):
        if cname != counter:
            continue
        key = name.replace('(anonymous namespace)::', '').split('(')[0]
        agg[key] += val
        calls[key].add(did)
    return agg, {k: len(v) for k, v in calls.items()}


def main():
    fdb, wdb, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f, fc = sums(fdb, 'FETCH_SIZE')
    w, wc = sums(wdb, 'WRITE_SIZE')
    fam = [k for k in f if 'gemm_h16d_kernel' in k or 'gemm_h16w_kernel' in k or 'gemm_p8_kernel' in k or 'gemm_p8w_kernel' in k]
    fetch_kib = sum(f[k] for k in fam)
    write_kib = sum(w.get(k, 0.0) for k in fam)
    launches = sum(fc[k] for k in fam)
    # write calibration: the uint8 FillFunctor launch zero-fills the workspace once per run (known size from WAVES)
    fill = [k for k in w if 'FillFunctor<unsigned char>' in k]
    out = {
        'workload': 'bench.py defaults: ghn3xlm16, one synthetic 256-node graph, f16 mode, side stream serialised',
        'kernel': 'gemm_p8_kernel + gemm_p8w_kernel + gemm_h16d_kernel (decoder W2 forward / dgrad / wgrad + decoder.conv.0 backward)',
        'steps_in_run': steps, 'launches_per_step': launches / steps,
        'fetch_size_kib_per_step_raw': fetch_kib / steps, 'write_size_kib_per_step_raw': write_kib / steps,
        'fetch_correction': 2.0,
        'per_kernel_raw_kib_per_step': {k: {'fetch': f[k] / steps, 'write': w.get(k, 0.0) / steps,
                                            'launches': fc[k] / steps} for k in fam},
    }
    write_cal = 1.0
    if fill:
        out['fill_write_size_kib_raw'] = sum(w[k] for k in fill)
        out['fill_launches'] = sum(wc[k] for k in fill)
        if len(sys.argv) > 4:                      # bench line of the WRITE_SIZE run: config.workspace_bytes
            line = [l for l in open(sys.argv[4]).read().splitlines() if l.startswith('{')][-1]
            cfg = json.loads(line)['config']
            ws_bytes = cfg.get('zero_fill_bytes', cfg['workspace_bytes'])
            out['fill_true_bytes'] = ws_bytes
            write_cal = out['fill_write_size_kib_raw'] * 1024.0 / ws_bytes
    out['write_calibration'] = write_cal
    try:                                           # the code this was measured on (bench.py refuses a figure from other code)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from ghn3_amd.build import source_hash
        out['code_hash'] = source_hash()
    except Exception as e:                         # pragma: no cover
        out['code_hash'] = None
        out['code_hash_error'] = repr(e)
    out['hbm_bytes_per_step'] = (2.0 * fetch_kib + write_kib / write_cal) * 1024.0 / steps
    out['hbm_read_bytes_per_step'] = 2.0 * fetch_kib * 1024.0 / steps
    out['hbm_write_bytes_per_step'] = write_kib / write_cal * 1024.0 / steps
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
