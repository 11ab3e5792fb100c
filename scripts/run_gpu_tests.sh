#!/bin/bash
# the whole GPU tier with durations (gpurun -- bash scripts/run_gpu_tests.sh [pytest args])
mkdir -p gpurun_out/tests
( time timeout 2400 python -m pytest tests -m gpu -q --durations=25 "$@" ) > gpurun_out/tests/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/tests/pytest.log
tail -45 gpurun_out/tests/pytest.log
