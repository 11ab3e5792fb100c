#!/bin/bash
# rocprofv3 summaries of the default bench (run on the GPU box through gpurun):
#   gpurun -- bash scripts/profile_gpu.sh [workload] [extra bench args]
# Pass 1: --kernel-trace --stats (per-kernel time).  Further passes: PMC counters, each set in
# its own run with --kernel-trace only (never combined with other trace domains):
#   FETCH_SIZE | WRITE_SIZE | wave / issue counters (SQ) | L2 hit-miss + LDS conflicts.
# Summaries land in gpurun_out/prof_<workload>/ and are condensed by scripts/summarise_prof.py.
W=${1:-chr20_21}
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT="$ROOT/gpurun_out/prof_$W"
mkdir -p "$OUT"
COMMIT=$(cat "$ROOT/.bossx_commit" 2>/dev/null)
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $W --no-cpu-baseline --no-large --no-others --no-entropy-off-run --no-cold --no-late --steps 10 --warmup 3 $*"
# the batches are generated once, outside the profiler (forked workers + the profiler's preloaded tool
# have deadlocked at exit); every pass below loads them from the cache
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_$W
python3 "$ROOT/bench.py" $ARGS --prepare-only
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
# counter collection serialises kernels: keep the chain after the sweep (it would time out and fall back anyway)
export BOSSX_NO_OVERLAP=1
pass() {   # name, counters...
    local name=$1; shift
    timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -o "$W" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_$name.log" 2>&1 \
        || echo "counter pass $name failed (see bench_$name.log)"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
pass mem TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS
tail -1 "$OUT/bench_trace.log" | cut -c1-300
python3 "$ROOT/scripts/summarise_prof.py" "$OUT" "$W" "$COMMIT"
