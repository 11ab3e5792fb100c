#!/bin/bash
# everything the round's profiles/ directory holds, in two calls (each well under its time limit):
#   gpurun --timeout 1200 -- bash scripts/collect_round.sh profiles
#   gpurun --timeout 1200 -- bash scripts/collect_round.sh bench
case "$1" in
profiles)
  for w in chr20_21 ecoli barcoded; do timeout 900 bash scripts/profile_gpu.sh $w > gpurun_out/prof_$w.log 2>&1; tail -6 gpurun_out/prof_$w.log | cut -c1-260; done
  ;;
bench)
  mkdir -p gpurun_out/bench
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench/default.json 2> gpurun_out/bench/default.err     # the driver's command
  timeout 300 python bench.py --workload ecoli --no-others --no-large > gpurun_out/bench/ecoli.json 2> gpurun_out/bench/ecoli.err
  timeout 300 python bench.py --workload barcoded --no-others --no-large > gpurun_out/bench/barcoded.json 2> gpurun_out/bench/barcoded.err
  timeout 300 python bench.py --workload grch38 --steps 8 --warmup 3 > gpurun_out/bench/grch38.json 2> gpurun_out/bench/grch38.err




  for f in default ecoli barcoded grch38; do python3 -c "
import json,sys
d=json.load(open('gpurun_out/bench/$f.json'))
print('$f', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], 'kernels_only', d.get('kernels_only_ms'), 'roofline', d.get('roofline',{}).get('frac'), 'cpu', (d.get('cpu_baseline') or {}).get('ms_per_update'), 'equal', (d.get('cpu_baseline') or {}).get('masks_and_threshold_equal_to_gpu'))
"; done
  ;;
esac
