"""Dispatch timeline of the C3 update pass from a rocprofv3 kernel trace: busy time, gaps between consecutive kernels, per-kernel totals over the optimiser steps of one iteration.
   rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline ; python scripts/c3_step_gaps.py /tmp/kt"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void crl::", "")) for r in rows), key=lambda x: x[0])
# the last iteration: from the last rollout kernel to the end
starts = [i for i, e in enumerate(ev) if "rollout" in e[2]]
i0 = starts[-1]
seg = ev[i0:]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy = 0; cur_end = t0; gaps = []; gap_by = collections.Counter()
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, n)); gap_by[n[:40]] += s - cur_end
    else:
        pass
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print("last iteration: %.3f ms from rollout start to last kernel end; some kernel running %.3f ms; idle %.3f ms in %d gaps" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))
print("idle time by the kernel that FOLLOWS the gap (ms):")
for n, g in gap_by.most_common(12): print("   %-42s %.3f" % (n, g / 1e6))
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n in seg: tot[n[:40]] += e - s; cnt[n[:40]] += 1
print("kernel time (ms, calls):")
for n, g in tot.most_common(14): print("   %-42s %.3f  %d" % (n, g / 1e6, cnt[n]))
