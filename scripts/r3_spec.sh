#!/bin/bash
# round 3: the chunk-parallel chain — parity subset, then bench with it and with the serial chain (same box)
mkdir -p gpurun_out/r3spec
if [ -z "$SKIP_TESTS" ]; then
BOSSX_SPEC_STATS=1 timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "${TESTS:-end_to_end or random or awkward or saturated or chain_kernels or lookahead or incremental}" > gpurun_out/r3spec/pytest.log 2>&1
grep -E "passed|failed|Error|error|assert" gpurun_out/r3spec/pytest.log | tail -8
grep "chunk-parallel" gpurun_out/r3spec/pytest.log | sort | uniq -c | sort -rn | head -5
fi
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_spec
for sp in ${SPECS:-1 0}; do
  BOSSX_SPEC_STATS=1 BOSSX_CHAIN_SPEC=$sp timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/r3spec/bench_sp$sp.json 2> gpurun_out/r3spec/bench_sp$sp.err || tail -5 gpurun_out/r3spec/bench_sp$sp.err
  grep "chunk-parallel" gpurun_out/r3spec/bench_sp$sp.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r3spec/bench_sp$sp.json')); print('spec=$sp step %.3f lone %.3f resident %.3f chain %.3f sweep %.4f'%(d['ms_per_step'], d['lone_update_ms'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['roofline']['avg_launch_ms']))"
done
