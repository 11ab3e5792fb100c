"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into
one small JSON + the raw kernel_stats CSV, ready to be copied into profiles/."""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"workload": tag, "kernels": {}}
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
for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != key:
            continue
        name = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(name, [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for name, (tot, n) in acc.items():
        res["kernels"].setdefault(name, {})[key + "_avg_per_launch"] = tot / max(n, 1)
        res["kernels"][name][key + "_launches"] = n
json.dump(res, open(os.path.join(out, "%s_summary.json" % tag), "w"), indent=1)
for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("total_ns", 0))[:12]:
    print(k[:60], {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
