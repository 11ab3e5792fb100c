#!/bin/bash
# GRCh38 strong-scaling workload on ONE GPU: the fused single-GPU path, and the multi-GPU protocol
# with its collectives forced on one rank (what every rank of an N-GPU run executes)
mkdir -p gpurun_out/grch38
i=0
for env in "X=1" "BOSSX_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571" "BOSSX_TORCH_COLLECTIVES=1 BOSSX_FORCE_COLLECTIVES=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29572"; do
  echo "== $env"
  env $env timeout 600 python bench.py --workload grch38 --steps 8 --warmup 3 > gpurun_out/grch38/out_$i.txt 2> gpurun_out/grch38/err_$i.txt
  python3 -c "
import json
for line in open('gpurun_out/grch38/out_$i.txt'):
    line=line.strip()
    if not line.startswith('{'): print('  extra:', line[:200]); continue
    d=json.loads(line); g=d['grch38']
    print('grch38 ms_per_step %.3f chain %.3f sweep %.3f collectives %s ranks %s' % (d['ms_per_step'], g['benefit_chain_ms_rank0'], g['site_sweep_rank0']['avg_ms'], g.get('collectives_per_update'), g.get('n_ranks_seen')))"
  tail -3 gpurun_out/grch38/err_$i.txt | cut -c1-200
  i=$((i+1))
done
