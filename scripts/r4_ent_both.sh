#!/bin/bash
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
for rep in 1 2; do
for v in "" cond; do
BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx${v:+_$v}.so timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 12 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${v:-default} entropy on: sweep %.4f frac %.3f lone %.3f | off: sweep %.4f frac %.3f lone %.3f' % (d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['track_entropy_false']['site_sweep_avg_ms'], d['track_entropy_false']['site_sweep_frac'], d['track_entropy_false']['ms_per_step']))"
done
done
