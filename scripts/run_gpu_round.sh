#!/bin/bash
# first GPU run of round 2: machine facts, the whole GPU tier, the new default bench
mkdir -p gpurun_out/r2a
(nproc; free -g | head -2; rocm-smi --showmeminfo vram | head -8; lscpu | grep "Model name") > gpurun_out/r2a/machine.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r2a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
echo "bench rc=$?" >> gpurun_out/r2a/bench.err
tail -5 gpurun_out/r2a/pytest.log
tail -3 gpurun_out/r2a/bench.err
cut -c1-1500 gpurun_out/r2a/bench.json
