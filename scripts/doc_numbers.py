"""The figures BASELINE.md §5 / README.md / DESIGN.md quote, from the committed measurement set: python scripts/doc_numbers.py [tag]"""
import csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda n: os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_{n}")
last = lambda f: json.loads(open(f).read().strip().splitlines()[-1])
d = last(P("bench_n1.json"))
r = d["roofline"]
print("headline: %.4g env-steps/s, %.2f ms (regions %s); update launch %.4f ms (rocprof %.4f), frac %.3f (rocprof %.3f); clock %.0f MHz, at rated %.4g" % (
    d["value"], d["ms_per_step"], " / ".join("%.2f" % x for x in d["ms_per_step_runs"]), r["avg_launch_ms"], r.get("avg_launch_ms_rocprof", 0), r["frac"], r.get("frac_rocprof", 0),
    d["clock"]["shader_mhz_under_vector_load"], d["clock"]["value_at_rated_clock"]))
s = d["strict_f32"]; print("strict_f32: %.4g, %.2f ms, launch %.4f ms" % (s["value"], s["ms_per_step"], s.get("update_kernel_avg_launch_ms", 0)))
w = d["with_stats_readback"]; print("read-back: async %.4g / %.2f ms, synchronous %.4g / %.2f ms" % (w["value"], w["ms_per_step"], w["synchronous"]["value"], w["synchronous"]["ms_per_step"]))
for f in ("bijection", "minibatches1", "envs32768", "envs16384", "envs8192", "rccl_forced", "peer_forced", "envs8192_peer_forced", "envs8192_peer_forced_three_launch"):
    x = last(P(f"bench_n1_{f}.json")); print("%-36s %.4g  %.3f ms  launch %.4f ms" % (f, x["value"], x["ms_per_step"], x["roofline"].get("avg_launch_ms") or 0))
for f in ("envs8192_breakdown", "breakdown"):
    x = last(P(f"bench_n1_{f}.json")); print(f, "%.3f ms" % x["ms_per_step"], {k: round(v, 3) for k, v in x["kernel_ms_per_step"].items() if v})
for k in ("c2", "c3", "shard_8192"):
    c = d["configs"][k]; print("%-10s %.4g  %.3f ms (regions %s)  strict %.2f ms  launch %.4f ms" % (k, c["value"], c["ms_per_step"], c.get("ms_per_step_runs"), c["strict_f32"]["ms_per_step"], c["roofline"]["avg_launch_ms"]))
print("c3 roofline:", json.dumps(d["configs"]["c3"]["roofline"]))
g = d["roofline_gae"]; print("gae:", {k: {a: round(b, 4) for a, b in g[k].items() if a in ("frac", "avg_launch_ms")} for k in g if isinstance(g[k], dict) and g[k]}, "traffic %.1f MB" % (g["traffic"] / 1e6))
print("cpu: %.4g, batched %.4g; a2c %.4g, dqn %.4g" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["batched"]["value"], d["configs"]["a2c"]["value"], d["configs"]["dqn"]["value"]))
for row in list(csv.DictReader(open(P("c3_rocprof_kernel_stats.csv"))))[:10]:
    print("   %-72s %5s x %9.1f us" % (row["Name"][:72], row["Calls"], float(row["AverageNs"]) / 1e3))
print("c3 stand-alone line: %.4g, %.2f ms" % (last(P("bench_c3_n1.json"))["value"], last(P("bench_c3_n1.json"))["ms_per_step"]))
for f in ("readback_cost.txt", "c3_iter_times.txt"):
    if os.path.exists(P(f)): print(open(P(f)).read()[:1400])
