#!/bin/bash
# gpurun -- bash scripts/r4_variant_test.sh <variant> <pytest -k expression>
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx_$1.so timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "$2" 2>&1 | tail -30 ) > gpurun_out/r4/variant_test.log 2>&1
tail -30 gpurun_out/r4/variant_test.log
BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx_$1.so timeout 600 python bench.py --no-cpu-baseline --no-others --no-entropy-off-run --steps 12 --warmup 4 2>&1 | tail -5 | cut -c1-600
