#!/bin/bash
mkdir -p gpurun_out/inc
( BOSSX_INCREMENTAL=1 timeout 2400 python -m pytest tests -m gpu -q -x ) > gpurun_out/inc/pytest.log 2>&1
tail -3 gpurun_out/inc/pytest.log
(python bench.py --workload grch38 --steps 6 --warmup 2) > gpurun_out/inc/grch38.json 2> gpurun_out/inc/grch38.err
python3 -c "
import json
d=json.load(open('gpurun_out/inc/grch38.json'))['grch38']
print('grch38 ms_per_step', d['ms_per_step'], 'sweep', d['site_sweep_rank0'], 'chain', d['benefit_chain_ms_rank0'])
"
bash scripts/quick_bench.sh
