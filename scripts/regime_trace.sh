#!/bin/bash
# rocprofv3 kernel stats of the lone-update loop started from a given preload depth (8 = the headline regime, 28 = the late one):
#   gpurun -- bash scripts/regime_trace.sh [depth] [workload]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=${1:-28}; W=${2:-chr20_21}
OUT="$ROOT/gpurun_out/regime_$D"; rm -rf "$OUT"; mkdir -p "$OUT"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
cat > /tmp/regime_run.py <<PY
import os, sys, tempfile
sys.path.insert(0, "$ROOT")
import bench
w = "$W"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, bench.WORKLOADS[w][3]) for i in range(25)])
os.chdir(tempfile.mkdtemp())
print(bench.late_regime_run(w, bench._GEN[w], 0, batches, 5, 20, True, depth=float($D)))
PY
cd "$ROOT" && python3 bench.py --no-cpu-baseline --no-large --no-others --no-entropy-off-run --no-cold --no-late --prepare-only > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o r -- python3 /tmp/regime_run.py > "$OUT/run.log" 2>&1
grep "preload_depth" "$OUT/run.log" | cut -c1-400
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-62s calls %4s avg %9.1f us  %5s %%" % (r["Name"].replace("bossx::", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
