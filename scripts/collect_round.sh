#!/bin/bash
# everything the round's profiles/ directory holds (gpurun -- bash scripts/collect_round.sh):
# rocprofv3 summaries of three workloads + the bench lines of the same commit
for w in chr20_21 ecoli barcoded; do bash scripts/profile_gpu.sh $w > gpurun_out/prof_$w.log 2>&1; tail -9 gpurun_out/prof_$w.log; done
mkdir -p gpurun_out/bench
python bench.py > gpurun_out/bench/default.json 2> gpurun_out/bench/default.err
python bench.py --workload ecoli --no-others --no-large > gpurun_out/bench/ecoli.json 2> gpurun_out/bench/ecoli.err
python bench.py --workload barcoded --no-others --no-large > gpurun_out/bench/barcoded.json 2> gpurun_out/bench/barcoded.err
python bench.py --workload grch38 --steps 8 --warmup 3 > gpurun_out/bench/grch38.json 2> gpurun_out/bench/grch38.err
python3 scripts/probe_chain.py > gpurun_out/bench/chain_probe.txt 2>&1
for f in default ecoli barcoded grch38; do python3 -c "
import json,sys
d=json.load(open('gpurun_out/bench/$f.json'))
print('$f', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], 'kernels_only', d.get('kernels_only_ms'), 'roofline', d.get('roofline',{}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('ms_per_update'))
"; done
