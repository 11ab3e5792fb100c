#!/bin/bash
# A/B of library variants against the default, both entropy modes (gpurun -- bash scripts/r4_ab.sh <variant> ...)
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
for rep in 1 2; do
for v in "" "$@"; do
BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx${v:+_$v}.so timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 12 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-9s entropy on: sweep %.4f frac %.3f full %.4f | off: sweep %.4f frac %.3f | chain %.3f resident %.3f' % ('${v:-default}', d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['full_sweep']['avg_launch_ms'], d['track_entropy_false']['site_sweep_avg_ms'], d['track_entropy_false']['site_sweep_frac'], d['kernels']['benefit_chain']['avg_ms'], d['kernels_only_ms']))"
done
done
