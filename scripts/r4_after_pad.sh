#!/bin/bash
# chain subset of the parity tier, then the default bench and GRCh38 (gpurun -- bash scripts/r4_after_pad.sh)
mkdir -p gpurun_out/r4
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
( timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "chain or saturat or random_scenarios or incremental or movesum or stagewise" 2>&1 | tail -6 ) > gpurun_out/r4/parity_chain.log 2>&1
tail -6 gpurun_out/r4/parity_chain.log
bash scripts/r4_variants.sh
timeout 600 python bench.py --no-cpu-baseline --no-others --no-large --steps 12 --warmup 4 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('entropy off:', d.get('track_entropy_false'))
print('chain', d['kernels']['benefit_chain'])"
BOSSX_BATCH_CACHE=/tmp/bossx_batches_grch38 timeout 900 python bench.py --workload grch38 --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['grch38']; print('grch38 lone', d['ms_per_step'], 'chain', g['benefit_chain_ms_rank0'], 'sweep', g['site_sweep_rank0']['avg_ms'])"
