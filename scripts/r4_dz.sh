#!/bin/bash
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( timeout 1500 python -m pytest tests -m gpu -x -q -k "incremental or random_scenarios or chr20_21_full or forty or saturated or golden" 2>&1 | tail -5 ) > gpurun_out/r4/parity_dz.log 2>&1
tail -5 gpurun_out/r4/parity_dz.log
BOSSX_BATCH_CACHE=/tmp/bossx_batches_grch38 timeout 900 python bench.py --workload grch38 --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['grch38']; print('grch38 lone', d['ms_per_step'], 'chain', g['benefit_chain_ms_rank0'], 'sweep', g['site_sweep_rank0']['avg_ms'])"
bash scripts/r4_variants.sh
