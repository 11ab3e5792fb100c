#!/bin/bash
# round 3: staging ahead — parity subset, then the default bench with and without lookahead (same box)
mkdir -p gpurun_out/r3la
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "lookahead or end_to_end or two_batches or error or empty or incremental or nccl" 2>&1 | tail -5
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_la
for la in 1 0 1 0; do
  if [ $la = 0 ]; then export BOSSX_NO_LOOKAHEAD=1; else unset BOSSX_NO_LOOKAHEAD; fi
  timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 20 --warmup 5 > gpurun_out/r3la/bench_la$la.json 2> gpurun_out/r3la/bench_la$la.err || tail -5 gpurun_out/r3la/bench_la$la.err
  python3 -c "
import json; d=json.load(open('gpurun_out/r3la/bench_la$la.json')); print('lookahead=$la step %.3f lone %.3f resident %.3f chain %.3f sweep %.4f stage %.3f'%(d['ms_per_step'], d['lone_update_ms'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['roofline']['avg_launch_ms'], d['host']['stage_batch_ms_mean']))"
done
