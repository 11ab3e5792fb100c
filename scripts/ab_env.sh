#!/bin/bash
# Kernel-trace A/B of the lone-update loop under environment variants (one box):  gpurun -- bash scripts/ab_env.sh "<VAR=val ...>" "<VAR=val ...>" ...
# ("-" = no variable).  Prints the top kernels of every arm (scripts/regime_trace.sh at depth 8).
for arm in "$@"; do
  echo "=== arm: $arm"
  if [ "$arm" = "-" ]; then bash scripts/regime_trace.sh ${DEPTH:-8} 2>&1 | tail -15; else env $arm bash scripts/regime_trace.sh ${DEPTH:-8} 2>&1 | tail -15; fi
done
