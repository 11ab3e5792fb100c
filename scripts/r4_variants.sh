#!/bin/bash
# round 4: the default bench (entropy on) for a list of library variants (gpurun -- bash scripts/r4_variants.sh v1 v2 ...)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
for v in "" "$@"; do
  lib=boss-runs_amd/csrc/libbossx${v:+_$v}.so
  [ -f "$lib" ] || { echo "no $lib"; continue; }
  BOSSX_LIB=$PWD/$lib timeout 900 python bench.py --no-cpu-baseline --no-others --no-entropy-off-run --steps 12 --warmup 4 $BENCH_ARGS > gpurun_out/r4/bench_${v:-default}.json 2> gpurun_out/r4/bench_${v:-default}.err || { echo "bench $v failed"; tail -5 gpurun_out/r4/bench_${v:-default}.err; continue; }
  python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1] or "default"
d = json.load(open('gpurun_out/r4/bench_%s.json' % v))
r = d["roofline"]
print("%-8s step %.3f ms lone %.3f resident %.3f | sweep %.4f ms frac %.3f | full %.4f ms frac %.3f | chain %.3f | stage %.3f | large stream %.3f gather %.3f" % (
    v, d["ms_per_step"], d.get("lone_update_ms", 0), d["kernels_only_ms"], r["avg_launch_ms"], r["frac"], r["full_sweep"]["avg_launch_ms"], r["full_sweep"]["frac"],
    d["kernels"]["benefit_chain"]["avg_ms"], d["host"]["stage_batch_ms_mean"],
    d.get("roofline_large", {}).get("stream", {}).get("frac", 0), d.get("roofline_large", {}).get("gather", {}).get("frac", 0)))
PY
done
