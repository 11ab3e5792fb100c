#!/bin/bash
# two windows per candidates wave against four (gpurun -- bash scripts/r4_cw2.sh)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx.so timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "chain or saturat or movesum" 2>&1 | tail -4 ) > gpurun_out/r4/parity_cw2.log 2>&1
tail -4 gpurun_out/r4/parity_cw2.log
bash scripts/r4_variants.sh cw4
for v in ""; do
BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx${v:+_$v}.so BOSSX_BATCH_CACHE=/tmp/bossx_batches_grch38 timeout 900 python bench.py --workload grch38 --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['grch38']; print('${v:-default} grch38 lone', d['ms_per_step'], 'chain', g['benefit_chain_ms_rank0'], 'sweep', g['site_sweep_rank0']['avg_ms'])"
done
