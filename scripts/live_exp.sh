#!/bin/bash
for env in "X=1" "BOSSX_LIVE_AFTER=1" "BOSSX_NO_OVERLAP=1"; do
  echo "== $env"
  env $env python bench.py --no-cpu-baseline --no-others --no-large 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step %.3f kernels_only %.3f chain %.3f sweep %.3f' % (d['ms_per_step'], d['kernels_only_ms'], d['kernels']['benefit_chain']['avg_ms'], d['kernels']['site_sweep']['avg_ms']))"
done
