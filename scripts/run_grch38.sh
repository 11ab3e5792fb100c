#!/bin/bash
mkdir -p gpurun_out/grch38
(python bench.py --workload grch38 --steps 8 --warmup 3) > gpurun_out/grch38/bench.json 2> gpurun_out/grch38/bench.err
python3 -c "
import json
d=json.load(open('gpurun_out/grch38/bench.json'))['grch38']
print('grch38 ms_per_step', d['ms_per_step'], 'sweep', d['site_sweep_rank0'], 'chain', d['benefit_chain_ms_rank0'])
"
