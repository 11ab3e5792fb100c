#!/bin/bash
# kernel-trace stats of the default bench (gpurun -- bash scripts/r3_trace.sh [workload])
W=${1:-chr20_21}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/r3trace_$W"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --no-cpu-baseline --no-large --no-others --steps 10 --warmup 3"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_$W
python3 "$ROOT/bench.py" $ARGS --prepare-only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv" 2>/dev/null
column -s, -t "$f" | cut -c1-200 | head -30
