#!/bin/bash
# lone update against the runtime's wait mode (gpurun -- bash scripts/r4_wait_mode.sh)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
python bench.py --no-cpu-baseline --no-others --no-entropy-off-run --no-large --steps 4 --warmup 2 > /dev/null 2>&1   # batch cache
for rep in 1 2; do
for w in 0 200 3000; do
  ROC_ACTIVE_WAIT_TIMEOUT=$w timeout 600 python bench.py --no-cpu-baseline --no-others --no-entropy-off-run --no-large --steps 30 --warmup 6 > gpurun_out/r4/bench_w$w.json 2> gpurun_out/r4/bench_w$w.err || { echo "bench w=$w failed"; continue; }
  python3 - "$w" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r4/bench_w%s.json' % sys.argv[1]))
h = d["host"]
print("active wait %-5s lone %.3f ms  pipelined %.3f  resident %.3f  stage mean %.3f | busy cpus %.1f throttled %.1f ms" % (sys.argv[1], d["lone_update_ms"], d["pipelined_ms_per_step"], d["kernels_only_ms"], h["stage_batch_ms_mean"], h["cpus_busy_in_timed_region"], h["throttled_ms_in_timed_region"]))
PY
done
done
