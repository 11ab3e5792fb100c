#!/bin/bash
# Lone-update wall time under environment variants, interleaved on one box:  gpurun -- bash scripts/ab_lone_env.sh "<VAR=val ...>" ...   ("-" = none)
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
for round in 1 2; do
  for arm in "$@"; do
    if [ "$arm" = "-" ]; then r=$(python3 scripts/lone_steps.py 2>&1 | grep "loop 0" | cut -c1-60); else r=$(env $arm python3 scripts/lone_steps.py 2>&1 | grep "loop 0" | cut -c1-60); fi
    echo "[$arm] $r"
  done
done
