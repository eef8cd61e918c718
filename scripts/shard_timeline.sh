# dispatch timeline of one headline iteration (rocprofv3 --kernel-trace): kernel, start offset, duration, queue
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tls && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tls -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --total-envs 8192 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tls/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rollout' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]   # (the last rollout of the trace belongs to the untimed read-back iteration)
t0 = int(rows[a]['Start_Timestamp']); n_upd = 0
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = r['Kernel_Name'].split('(')[0][-48:]
    if 'update_x2' in nm:
        n_upd += 1
        if n_upd > 2 and n_upd < 16: continue
    print("%-48s start %9.1f us  dur %8.1f us  queue %s" % (nm, (s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?')))
print("iteration span %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
