#!/bin/bash
# round 3: the scan chain — parity subset, then bench with the scan chain and with the serial chain (same box)
mkdir -p gpurun_out/r3scan
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "${TESTS:-scan_chain or end_to_end or random or awkward or saturated or chain_kernels}" > gpurun_out/r3scan/pytest.log 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/r3scan/pytest.log | tail -8
fi
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_scan
for sc in ${SCANS:-1 0}; do
  BOSSX_SCAN_STATS=1 BOSSX_CHAIN_SCAN=$sc timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/r3scan/bench_sc$sc.json 2> gpurun_out/r3scan/bench_sc$sc.err || tail -5 gpurun_out/r3scan/bench_sc$sc.err
  grep "scan chain" gpurun_out/r3scan/bench_sc$sc.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r3scan/bench_sc$sc.json')); print('scan=$sc step %.3f lone %.3f resident %.3f chain %.3f sweep %.4f'%(d['ms_per_step'], d['lone_update_ms'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['roofline']['avg_launch_ms']))"
done
