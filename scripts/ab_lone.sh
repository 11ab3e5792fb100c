#!/bin/bash
# Lone-update wall time of several library variants on ONE box, interleaved (the box-to-box spread of this number is larger than
# most changes to it):   gpurun -- bash scripts/ab_lone.sh [variant ...]     (variant "default" = libbossx.so; REPS, default 3)
export BOSSX_BATCH_CACHE=/tmp/bossx_batches
[ $# -eq 0 ] && set -- default
for rep in $(seq 1 ${REPS:-3}); do
  for v in "$@"; do
    lib=$PWD/boss-runs_amd/csrc/libbossx$([ "$v" = default ] || echo _$v).so
    echo "[$v] $(BOSSX_LIB=$lib python3 scripts/lone_steps.py 2>/dev/null | cut -d'|' -f1 | tr '\n' ' ')"
  done
done
for v in "$@"; do
  lib=$PWD/boss-runs_amd/csrc/libbossx$([ "$v" = default ] || echo _$v).so
  echo "[$v] stage timing:"; BOSSX_LIB=$lib BOSSX_STAGE_TIMING=1 python3 scripts/lone_steps.py 2>&1 | grep "stage_batch:" | tail -6 | cut -c1-400
done
