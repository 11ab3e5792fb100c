#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (run on the GPU box through gpurun):
#   gpurun -- bash scripts/profile_gpu.sh [workload]
# Summaries land in gpurun_out/prof_<workload>/; copy the *_kernel_stats.csv into profiles/.
W=${1:-ecoli}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$W" -o "$W" -- python3 "$ROOT/bench.py" --workload "$W" --no-cpu-baseline ${BENCH_ARGS} > "$ROOT/gpurun_out/prof_${W}_bench.log" 2>&1
tail -1 "$ROOT/gpurun_out/prof_${W}_bench.log" | cut -c1-400
find "$ROOT/gpurun_out/prof_$W" -name "*stats*" | head
