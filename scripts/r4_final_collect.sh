#!/bin/bash
# end of round: GPU tier + smoke, profiles, bench lines — one call (gpurun --timeout 3300 -- bash scripts/r4_final_collect.sh)
bash scripts/r4_gpu_tier.sh 2>&1 | tail -12
bash scripts/collect_round.sh profiles 2>&1 | grep -E "site_sweep1_kernel<true|chain_candidates|chain_stitch|failed" | cut -c1-260
bash scripts/collect_round.sh bench 2>&1 | tail -5
