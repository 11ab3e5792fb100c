#!/bin/bash
# kernel trace of the default bench: per-kernel averages (gpurun -- bash scripts/r4_trace.sh [bench args])
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/r4"
cd /tmp && export TMPDIR=/tmp
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
ARGS="--no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 $*"
python3 "$ROOT/bench.py" $ARGS --prepare-only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r4 -o r4 -- python3 "$ROOT/bench.py" $ARGS > /tmp/prof_r4.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_r4/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:16]:
    print("%-64s calls %6s avg %9.1f us  total %8.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_r4/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for k, v in by.items():
    if any(x in k for x in ("chain", "sweep1", "hist", "mask")):
        v.sort()
        d = [x[1] for x in v]
        print("%-42s n=%3d  first25 avg %8.1f us | 26-60 %8.1f | 61+ %8.1f" % (k, len(d), sum(d[:25]) / max(len(d[:25]), 1), sum(d[25:60]) / max(len(d[25:60]), 1), sum(d[60:]) / max(len(d[60:]), 1)))
PY
