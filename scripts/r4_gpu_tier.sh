#!/bin/bash
# round 4: the whole GPU tier with per-test durations, then the smoke entry (gpurun -- bash scripts/r4_gpu_tier.sh)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -40 ) > gpurun_out/r4/gpu_tier.log 2>&1
tail -40 gpurun_out/r4/gpu_tier.log
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -5 ) > gpurun_out/r4/smoke.log 2>&1
tail -5 gpurun_out/r4/smoke.log
