"""Condense rocprofv3 CSV output (kernel stats + the counter passes) into one small JSON + the
raw kernel_stats CSV, ready to be copied into profiles/.

    python3 scripts/summarise_prof.py <out dir> <workload tag> [commit]

Every `pmc_*` sub-directory holds one counter pass; each counter found is averaged per launch and
per kernel.  `launches_per_update` = launches of the kernel / launches of strategy_mask_kernel
(one per decision update)."""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    ksrc = bench.kernel_sources_hash()
except Exception:
    ksrc = None
res = {"workload": tag, "commit": sys.argv[3] if len(sys.argv) > 3 else None, "kernel_sources": ksrc, "kernels": {}}
def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None
f = find("trace", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0]
        res["kernels"][name] = dict(calls=int(r["Calls"]), total_ns=float(r["TotalDurationNs"]),
                                    avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
    os.system("cp %s %s" % (f, os.path.join(out, "%s_kernel_stats.csv" % tag)))
    n_upd = max([v["calls"] for k, v in res["kernels"].items() if "strategy_mask_kernel" in k] or [0])
    if n_upd:
        for v in res["kernels"].values():
            v["launches_per_update"] = v["calls"] / n_upd
# The trace holds EVERY loop of bench.py (lone updates, pipelined, resident, every tile swept, the loop without events): a kernel's
# average over all of them mixes regimes — the state saturates as the script goes on — so the launches of the TIMED region (the lone
# loop: launches [warmup, warmup + steps) of a once-per-update kernel; profile_gpu.sh runs --warmup 3 --steps 10) are averaged on their own.
f = find("trace", "*kernel_trace.csv")
if f:
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    per = {}
    for r in rows:
        per.setdefault(r["Kernel_Name"].split("(")[0], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    W, K = 3, 10
    for name, ds in per.items():
        v = res["kernels"].get(name)
        if v is not None and v.get("launches_per_update", 0) >= 0.9 and len(ds) >= W + K:
            v["avg_ns_timed_region"] = sum(ds[W:W + K]) / K
            v["timed_region_launches"] = "launches %d..%d of %d in start order (the lone-update loop)" % (W, W + K - 1, len(ds))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    f = find(os.path.basename(d), "*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        key = r.get("Counter_Name")
        name = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault((name, key), [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for (name, key), (tot, n) in acc.items():
        res["kernels"].setdefault(name, {})[key + "_avg_per_launch"] = tot / max(n, 1)
        res["kernels"][name][key + "_launches"] = n
# derived figures for the kernels of interest
for name, v in res["kernels"].items():
    g = lambda k: v.get(k + "_avg_per_launch")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        # KB as reported; FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B requests at 64 B)
        v["hbm_bytes_per_launch"] = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        v["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
        v["mean_waves_in_flight_per_busy_sq_cycle"] = g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES")
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if g(k) is not None:
                v[k + "_frac_of_wave_cycles"] = g(k) / g("SQ_WAVE_CYCLES")
json.dump(res, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
KEYS = ("calls", "avg_ns", "avg_ns_timed_region", "pct", "hbm_bytes_per_launch", "l2_hit_rate", "SQ_WAIT_ANY_frac_of_wave_cycles",
        "SQ_ACTIVE_INST_VALU_frac_of_wave_cycles", "SQ_ACTIVE_INST_ANY_frac_of_wave_cycles")
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("total_ns", 0))[:8]:
    print(k[:48], {a: (round(v[a], 3) if isinstance(v[a], float) else v[a]) for a in KEYS if a in v})
