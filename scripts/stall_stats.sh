#!/bin/bash
# How often a lone update of bench.py's timed region stalls (> 2.5 ms), over N short runs per arm:  gpurun -- bash scripts/stall_stats.sh N "<VAR=val ...>" ...
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
N=${1:-4}; shift
for arm in "$@"; do
  for i in $(seq 1 $N); do
    if [ "$arm" = "-" ]; then timeout 300 python3 bench.py --steps 20 --warmup 5 --no-others --no-large --no-cpu-baseline --no-cold --no-late --no-entropy-off-run > /tmp/ss.json 2>/dev/null
    else env $arm timeout 300 python3 bench.py --steps 20 --warmup 5 --no-others --no-large --no-cpu-baseline --no-cold --no-late --no-entropy-off-run > /tmp/ss.json 2>/dev/null; fi
    python3 -c "
import json,numpy as np
d=json.load(open('/tmp/ss.json')); e=d['host']['lone_update_ms_each']
print('[$arm] mean %.3f median %.3f stalls %s' % (d['ms_per_step'], np.median(e), [x for x in e if x > 2.5]))"
  done
done
