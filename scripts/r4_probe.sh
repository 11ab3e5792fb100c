#!/bin/bash
# sweep cycle probes + sweep-only timings (gpurun -- bash scripts/r4_probe.sh)
export SWEEP_REPS=2 BOSSX_INCREMENTAL=0
for ch in "" 1; do
echo "== chunk '$ch'"
BOSSX_SWEEP_CHUNK=$ch BOSSX_SWEEP_PROBE=1 python3 scripts/sweep_only.py 2>&1 | grep -E "probe|ms" | sed -n '3,4p;7,8p'
done
echo "== no probe"; python3 scripts/sweep_only.py 2>&1 | grep -E "ms"
unset BOSSX_INCREMENTAL
python3 scripts/sweep_probe.py chr20_21 2>&1 | grep probe | tail -2
