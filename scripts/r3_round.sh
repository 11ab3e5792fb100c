#!/bin/bash
# round 3 evidence in one call: the whole GPU tier, then the chr20+21 rocprof passes
mkdir -p gpurun_out/r3round
( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > gpurun_out/r3round/pytest.log 2>&1
tail -5 gpurun_out/r3round/pytest.log
for w in ${WORKLOADS:-chr20_21}; do timeout 900 bash scripts/profile_gpu.sh $w > gpurun_out/prof_$w.log 2>&1; tail -6 gpurun_out/prof_$w.log | cut -c1-400; done
