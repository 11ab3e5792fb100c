#!/bin/bash
# round 3: sweep-kernel experiments — parity subset with the default build, then the default bench with
# every library variant given (gpurun -- bash scripts/r3_sweep_exp.sh "" o56 o57 ...)
mkdir -p gpurun_out/r3exp
if [ -z "$SKIP_TESTS" ]; then
( timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "end_to_end or incremental or deep or counters or awkward or random or two_batches or saturated or simulation or emulated" 2>&1 | tail -8 ) > gpurun_out/r3exp/parity.log 2>&1
tail -4 gpurun_out/r3exp/parity.log
fi
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_exp
for v in "$@"; do
  lib=boss-runs_amd/csrc/libbossx${v:+_$v}.so
  [ -f "$lib" ] || { echo "no $lib"; continue; }
  BOSSX_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --no-others --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/r3exp/bench_${v:-default}.json 2> gpurun_out/r3exp/bench_${v:-default}.err || { echo "bench $v failed"; tail -5 gpurun_out/r3exp/bench_${v:-default}.err; continue; }
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1] or "default"
d = json.load(open('gpurun_out/r3exp/bench_%s.json' % v))
r = d["roofline"]
print("%-8s step %.3f ms resident %.3f | sweep %.4f ms frac %.3f | full %.4f ms frac %.3f | chain %.3f | stage %.3f | large stream %.3f gather %.3f" % (
    v, d["ms_per_step"], d["kernels_only_ms"], r["avg_launch_ms"], r["frac"], r["full_sweep"]["avg_launch_ms"], r["full_sweep"]["frac"],
    d["kernels"]["benefit_chain"]["avg_ms"], d["host"]["stage_batch_ms_mean"],
    d.get("roofline_large", {}).get("stream", {}).get("frac", 0), d.get("roofline_large", {}).get("gather", {}).get("frac", 0)))
PY
done
