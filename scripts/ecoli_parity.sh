#!/bin/bash
# E. coli bench line with the CPU port compared on the same batch, under several schedules
for env in "X=1" "BOSSX_NO_OVERLAP=1" "BOSSX_CHAIN_BARRIER=1" "BOSSX_OVERLAP=1"; do
  echo "== ecoli $env"
  env $env timeout 900 python bench.py --workload ecoli --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
c=d['cpu_baseline']
print('ms_per_step %.3f kernels_only %.3f chain %.3f equal %s cpu %.0f' % (d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], c['masks_and_threshold_equal_to_gpu'], c['ms_per_update']))"
done
