#!/bin/bash
# Which kernels grow as coverage accumulates: a kernel trace of scripts/drift_steps.py, per kernel the mean duration of its first and its last ten
# launches of the lone-update loop.   gpurun -- bash scripts/drift_trace.sh [workload] [steps]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/drift"; mkdir -p "$OUT"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
python3 "$ROOT/scripts/drift_steps.py" "${1:-chr20_21}" "${2:-60}" > /dev/null 2>&1      # (batches into the cache, outside the profiler)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o drift -- python3 "$ROOT/scripts/drift_steps.py" "${1:-chr20_21}" "${2:-60}" > "$OUT/drift.log" 2>&1
tail -6 "$OUT/drift.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][-48:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-50s %6s %10s %10s" % ("kernel", "calls", "first 10", "last 10"))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1][-10:])):
    if len(v) >= 30:
        lone = v[5:]            # (five warm-up updates)
        print("%-50s %6d %9.1f %9.1f us" % (k, len(v), sum(lone[:10]) / 10, sum(lone[-10:]) / 10))
PY
