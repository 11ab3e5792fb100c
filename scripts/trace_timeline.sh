#!/bin/bash
# Kernel start / end timestamps of the LAST lone update of scripts/e2e_timeline.py (14 lone updates, PAF text -> masks), from a
# rocprofv3 kernel trace: what runs when, and the gaps between launches.   gpurun -- bash scripts/trace_timeline.sh [workload]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/trace_tl"
rm -rf "$OUT"; mkdir -p "$OUT"
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
cd "$ROOT" && python3 bench.py --no-cpu-baseline --no-large --no-others --no-entropy-off-run --no-cold --no-late --prepare-only > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o tl -- python3 scripts/e2e_timeline.py ${1:-chr20_21} > "$OUT/run.log" 2>&1 )
tail -8 "$OUT/run.log"
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/scripts/update_timeline.py" "$F" 2
