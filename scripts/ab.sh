#!/bin/bash
# One box, several library variants (and/or environment settings): parity subset first, then the default bench per arm.
#   gpurun --timeout 1500 -- bash scripts/ab.sh [arm ...]
# arm = <variant>[@ENV=VAL[,ENV=VAL...]] ; variant "" / "default" = libbossx.so, else libbossx_<variant>.so
#   (make -C boss-runs_amd/csrc variant NAME=x DEFS=-D...).  Environment: TESTS="<pytest -k expr>" (parity subset on
#   every arm's library, skipped when empty), TEST_FILE (default tests/test_parity_gpu.py), BENCH_ARGS (extra bench flags),
#   REPS (default 1), OUT (default gpurun_out/ab).
OUT=${OUT:-gpurun_out/ab}; mkdir -p $OUT
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
[ $# -eq 0 ] && set -- default
for rep in $(seq 1 ${REPS:-1}); do
for arm in "$@"; do
  v=${arm%%@*}; envs=""; [ "$arm" != "$v" ] && envs=${arm#*@}
  [ "$v" = default ] && v=""
  lib=$PWD/boss-runs_amd/csrc/libbossx${v:+_$v}.so
  [ -f "$lib" ] || { echo "no $lib"; continue; }
  tag=$(echo "${arm}" | tr '@,=' '___')
  (
    export BOSSX_LIB=$lib
    for kv in $(echo "$envs" | tr ',' ' '); do export "$kv"; done
    if [ -n "$TESTS" ] && [ "$rep" = 1 ]; then
      timeout 1500 python -m pytest ${TEST_FILE:-tests/test_parity_gpu.py} -m gpu -x -q -k "$TESTS" > $OUT/test_$tag.log 2>&1
      echo "[$arm] tests: $(tail -1 $OUT/test_$tag.log)"
      grep -E "^(FAILED|ERROR)|Error|assert " $OUT/test_$tag.log | head -8
    fi
    timeout 900 python bench.py --no-cpu-baseline --no-others --no-entropy-off-run --no-cold --no-late --steps 12 --warmup 4 $BENCH_ARGS > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || { echo "[$arm] bench failed"; tail -5 $OUT/bench_$tag.err; exit 0; }
    python3 - "$arm" "$OUT/bench_$tag.json" <<'PY'
import json, sys
arm, f = sys.argv[1], sys.argv[2]
d = json.loads(open(f).read().strip().splitlines()[-1])
r = d["roofline"]; k = d.get("kernels", {})
fs = r.get("full_sweep") or {}
lg = d.get("roofline_large") or {}
print("[%s] lone %.3f ms resident %.3f | ingest sweep %.4f ms frac %.3f | full %.4f frac %.3f | chain %.3f | stage %.3f | large stream %.3f gather %.3f" % (
    arm, d["ms_per_step"], d.get("kernels_only_ms") or 0, r["avg_launch_ms"], r["frac"], fs.get("avg_launch_ms", 0), fs.get("frac", 0),
    (k.get("benefit_chain") or {}).get("avg_ms", 0), (d.get("host") or {}).get("stage_batch_ms_mean", 0),
    (lg.get("stream") or {}).get("frac", 0), (lg.get("gather") or {}).get("frac", 0)))
PY
  )
done
done
