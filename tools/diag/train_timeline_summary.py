"""Summary of a training-step timeline written by tools/rocprof_timeline.py <db> <csv> 2 (see train_step_diag.py): where the W2
cast of the shadow refresh runs and when the decoder forward starts, relative to the end of AdamW."""
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ia=[i for i,r in enumerate(rows) if 'adamw' in r['kernel']][0]
t_end=float(rows[ia]['start_us'])+float(rows[ia]['dur_us'])
big=max([r for r in rows[ia:ia+120] if 'cast16' in r['kernel']], key=lambda r: float(r['dur_us']))
x=[r for r in rows[ia:] if 'gemm_x3s' in r['kernel']][0]
pro=[r for r in rows[ia:] if 'graph_prologue' in r['kernel']][0]
p8=[r for r in rows[ia:] if 'gemm_p8_kernel' in r['kernel']][0]
print(sys.argv[1], 'big cast start %.0f dur %.0f (queue %s) | prologue at %.0f | first x3s dur %.0f | W2 fwd starts %.0f dur %.0f' % (float(big['start_us'])-t_end, float(big['dur_us']), big['queue'], float(pro['start_us'])-t_end, float(x['dur_us']), float(p8['start_us'])-t_end, float(p8['dur_us'])))
