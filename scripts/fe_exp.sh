#!/bin/bash
# front-end experiment: task trace + stage timing (gpurun -- bash scripts/fe_exp.sh)
mkdir -p gpurun_out/fe
BOSSX_STAGE_TIMING=1 timeout 300 python3 scripts/front_end_timing.py chr20_21 > gpurun_out/fe/trace.log 2>&1
grep -E "stage_batch|process_batch" gpurun_out/fe/trace.log | tail -30
grep -E "pass1" gpurun_out/fe/trace.log | sed -n '25,36p'
