#!/bin/bash
# host side of the lone update against the number of parse threads (gpurun -- bash scripts/r4_host_threads.sh)
mkdir -p gpurun_out/r4
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc $(nproc)  load $(cut -d' ' -f1-3 /proc/loadavg)"
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
for t in "" 8 16 def2; do
  [ "$t" = def2 ] && { unset BOSSX_PARSE_THREADS; export OMP_NUM_THREADS=4 OPENBLAS_NUM_THREADS=4 MKL_NUM_THREADS=4; }
  [ -n "$t" ] && [ "$t" != def2 ] && export BOSSX_PARSE_THREADS=$t
  timeout 600 python bench.py --no-cpu-baseline --no-others --no-entropy-off-run --steps 24 --warmup 6 > gpurun_out/r4/bench_t${t:-def}.json 2> gpurun_out/r4/bench_t${t:-def}.err || { echo "bench t=$t failed"; continue; }
  python3 - "${t:-def}" <<'PY'
import json, sys
d = json.load(open('gpurun_out/r4/bench_t%s.json' % sys.argv[1]))
h = d["host"]
print("threads %-4s lone %.3f ms  pipelined %.3f  resident %.3f  stage mean %.3f | quota %s busy cpus %.1f throttled %.1f ms in %d periods" % (sys.argv[1], d["lone_update_ms"], d["pipelined_ms_per_step"], d["kernels_only_ms"], h["stage_batch_ms_mean"], h["cpu_quota_cores"], h["cpus_busy_in_timed_region"], h["throttled_ms_in_timed_region"], h["throttled_periods_in_timed_region"]))
PY
done
grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
