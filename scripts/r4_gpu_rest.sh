#!/bin/bash
# round 4: the GPU tier from a given test on (gpurun -- bash scripts/r4_gpu_rest.sh)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( timeout 3000 python -m pytest tests/test_parity_gpu.py -m gpu -q --durations=8 -k "device_cigar_walk or stagewise or fuzz or native_driver or lookahead or error_classes or deep_saturation or saturation" 2>&1 | tail -60 ) > gpurun_out/r4/gpu_rest.log 2>&1
tail -60 gpurun_out/r4/gpu_rest.log
