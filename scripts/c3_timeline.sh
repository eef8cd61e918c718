# dispatch timeline of one C3 optimiser step (rocprofv3 --kernel-trace): kernel, start offset, duration, gap to the previous kernel's end
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# find the last-but-one wide_fused_fwd launch and print until the next one
idx = [i for i, r in enumerate(rows) if 'wide_fused_fwd' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp']); prev_end = None; busy = 0
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-52s start %8.1f us  dur %7.1f us  gap %6.1f us  stream/queue %s" % (r['Kernel_Name'].split('(')[0][-52:], (s - t0) / 1e3, (e - s) / 1e3, gap, r.get('Queue_Id', '?')))
    prev_end = max(prev_end or 0, e)
print("step span %.1f us" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
PY
