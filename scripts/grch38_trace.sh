#!/bin/bash
# GRCh38 on one GPU: chain counters + rocprofv3 kernel trace (gpurun -- bash scripts/grch38_trace.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/grch38"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_grch38
ARGS="--workload grch38 --steps 8 --warmup 3"
python3 "$ROOT/bench.py" $ARGS --prepare-only > /dev/null 2>&1
BOSSX_SPEC_STATS=1 python3 "$ROOT/bench.py" $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
grep -i -E "spec|chain|paus" "$OUT/bench.err" | tail -8
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o grch38 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
