#!/bin/bash
# rocprofv3 summaries of the default bench (run on the GPU box through gpurun):
#   gpurun -- bash scripts/profile_gpu.sh [workload] [round]
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2/3: PMC FETCH_SIZE / WRITE_SIZE in
# their own runs (never combined with trace domains other than kernel-trace).
# Summaries land in gpurun_out/prof_<workload>/ and are condensed by scripts/summarise_prof.py.
W=${1:-ecoli}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/prof_$W"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --no-cpu-baseline --no-large ${BENCH_ARGS}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
# counter collection serialises kernels: keep the chain after the sweep (it would time out and fall back anyway)
export BOSSX_NO_OVERLAP=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_write.log" 2>&1
tail -1 "$OUT/bench_trace.log" | cut -c1-300
find "$OUT" -name "*.csv" | head -20
python3 "$ROOT/scripts/summarise_prof.py" "$OUT" "$W"
