#!/bin/bash
mkdir -p gpurun_out/r2d
python3 scripts/front_end_timing.py chr20_21 > gpurun_out/r2d/fe_chr20_21.log 2>&1
( time python bench.py ) > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
grep -E "^threads|^python|^process|stage_batch: host" gpurun_out/r2d/fe_chr20_21.log | tail -30
cut -c1-600 gpurun_out/r2d/bench.json
