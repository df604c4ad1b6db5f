cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof16
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof16 -o r -- python3 bench.py --compute f16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench16.json 2> gpurun_out/prof_err.log
F=$(find /tmp/prof16 -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t_end = int(rows[-1]['End_Timestamp'])
out = open('gpurun_out/trace_last_step.csv', 'w')
out.write('start_us,dur_us,queue,kernel\n')
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t_end - s < 20e6:
        out.write('%.2f,%.2f,%s,%s\n' % ((s - (t_end - 20e6)) / 1e3, (e - s) / 1e3, r.get('Queue_Id', ''), r['Kernel_Name'][:60].replace(',', ';')))
out.close()
PY
wc -l gpurun_out/trace_last_step.csv
