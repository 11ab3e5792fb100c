#!/bin/bash
mkdir -p gpurun_out/r4
( timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "fuzz or error_classes or device_cigar_walk" 2>&1 | tail -12 ) > gpurun_out/r4/fuzz_gpu.log 2>&1
tail -12 gpurun_out/r4/fuzz_gpu.log
bash scripts/r4_variants.sh "$@"
