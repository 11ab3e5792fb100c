#!/bin/bash
# cycle probes of the one-barcode sweep on the bench workload, entropy on and off (gpurun -- bash scripts/r4_probe.sh)
# needs the probe build: make -C boss-runs_amd/csrc variant NAME=probe DEFS=-DBOSSX_SWEEP_PROBE_BUILD
export BOSSX_BATCH_CACHE=/tmp/bossx_batches_r4
export BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx_probe.so
for ent in 1 0; do
echo "== entropy $ent"
ENT=$ent python3 scripts/sweep_probe.py chr20_21 2>&1 | grep probe | tail -3
done
