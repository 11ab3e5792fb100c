#!/bin/bash
bash scripts/run_gpu_tests.sh -x
python bench.py --workload grch38 --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['grch38']
print('grch38 ms_per_step %.3f chain %.3f sweep %.3f' % (d['ms_per_step'], g['benefit_chain_ms_rank0'], g['site_sweep_rank0']['avg_ms']))"
python bench.py --no-cpu-baseline --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('chr20_21 ms_per_step %.3f kernels_only %.3f' % (d['ms_per_step'], d['kernels_only_ms']))"
BOSSX_STAGE_TIMING=1 timeout 300 python3 scripts/front_end_timing.py chr20_21 2>&1 | grep -E "stage_batch:|\[parse\]|pass1\] region" | sed -n '20,34p'
