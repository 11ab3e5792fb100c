#!/bin/bash
# quick look at the chain kernel: cycle probe (E. coli) + chr20_21 bench line
timeout 120 python3 scripts/probe_chain.py 2>&1 | grep -E "wave 0:|wave 1:|wave 11:|wave 14:|benefit ms" | tail -6
for env in "X=1" "BOSSX_OVERLAP=1" "BOSSX_FLOW_BUFS4=1"; do
  echo "== chr20_21 $env"
  env $env timeout 600 python bench.py --workload chr20_21 --no-cpu-baseline --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step %.3f kernels_only %.3f chain %.3f sweep %.3f ns/bin %.2f stage %.3f' % (d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['kernels']['site_sweep']['avg_ms'], d['chain_latency']['ns_per_bin_longest'], d['host']['stage_batch_ms_mean']))"
done
