#!/bin/bash
# The GPU-side span of lone updates (scripts/gpu_span.py) per environment variant, one rocprofv3 kernel trace of scripts/lone_steps.py each:
#   gpurun -- bash scripts/span_ab.sh "-" "BOSSX_SPLIT_PLAN_UPLOAD=1" ...        ("-" = no variable; REPS=n repeats the set)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
cd "$ROOT" && python3 bench.py --no-cpu-baseline --no-large --no-others --no-entropy-off-run --no-cold --no-late --prepare-only > /dev/null 2>&1
[ $# -eq 0 ] && set -- "-"
for rep in $(seq 1 ${REPS:-1}); do
for arm in "$@"; do
  OUT="$ROOT/gpurun_out/span_ab/$(echo "$arm" | tr ' =' '__')_$rep"; rm -rf "$OUT"; mkdir -p "$OUT"
  (
    if [ "$arm" != "-" ]; then for kv in $arm; do export "$kv"; done; fi
    cd /tmp && export TMPDIR=/tmp
    cd "$ROOT" && rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o tl -- python3 scripts/lone_steps.py ${WORKLOAD:-chr20_21} > "$OUT/run.log" 2>&1
  )
  F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
  echo "[$arm] $(grep 'loop 0' "$OUT/run.log" | cut -d'|' -f1)"
  python3 "$ROOT/scripts/gpu_span.py" "$F" 8 | head -${LINES_OUT:-1}
done
done
