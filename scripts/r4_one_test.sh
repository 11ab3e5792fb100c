#!/bin/bash
# gpurun -- bash scripts/r4_one_test.sh <pytest -k expression>
mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q --durations=5 -k "$1" 2>&1 | tail -25 ) > gpurun_out/r4/one_test.log 2>&1
tail -25 gpurun_out/r4/one_test.log
nproc; uptime
