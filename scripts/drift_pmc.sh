#!/bin/bash
# Counters of the ingest sweep launch by launch over scripts/drift_steps.py (what changes as coverage accumulates?): one --pmc pass,
# kernel trace only.   gpurun -- bash scripts/drift_pmc.sh "<counters>" [kernel substring]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/drift_pmc"; mkdir -p "$OUT"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches BOSSX_NO_OVERLAP=1
python3 "$ROOT/scripts/drift_steps.py" chr20_21 60 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d "$OUT/pmc" -o drift -- python3 "$ROOT/scripts/drift_steps.py" chr20_21 60 > "$OUT/drift.log" 2>&1
python3 - "$OUT" "${2:-site_sweep1_kernel<true}" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True)[0]
by = collections.defaultdict(lambda: collections.defaultdict(float))
order = {}
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        d = int(r["Dispatch_Id"]); order[d] = 1
        by[d][r["Counter_Name"]] += float(r["Counter_Value"])
ds = sorted(order)
names = sorted({n for d in ds for n in by[d]})
def mean(sel, n): return sum(by[d][n] for d in sel) / max(len(sel), 1)
early, late = ds[5:15], ds[-10:]
print("%d launches of %s; mean of launches 6-15 | last 10 | ratio" % (len(ds), sys.argv[2]))
for n in names:
    a, b = mean(early, n), mean(late, n)
    print("%-28s %14.0f %14.0f  %.3f" % (n, a, b, b / a if a else 0))
PY
