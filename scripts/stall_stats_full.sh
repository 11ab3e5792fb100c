#!/bin/bash
# The driver's command N times: mean / median / stalled updates of the timed region:  gpurun -- bash scripts/stall_stats_full.sh N
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
for i in $(seq 1 ${1:-3}); do
  timeout 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/ss.json 2>/dev/null
  python3 -c "
import json,numpy as np
d=json.load(open('/tmp/ss.json')); e=d['host']['lone_update_ms_each']
print('full bench: mean %.3f median %.3f stalls %s late %.3f' % (d['ms_per_step'], np.median(e), [x for x in e if x > 2.5], d['late_regime']['ms_per_step']))"
done
