#!/bin/bash
# sweep-only timing on a 200 Mb contig (every tile swept each time) for library variants / grid settings
export SWEEP_REPS=3 BOSSX_INCREMENTAL=0
run() { echo "== $*"; env "$@" timeout 300 python3 scripts/sweep_only.py 2>&1 | grep -E "bossx\]|ms" | awk '{printf "%s | ", $0} END {print ""}'; }
for v in "$@"; do
  lib=$PWD/boss-runs_amd/csrc/libbossx${v:+_$v}.so
  [ "$v" = "default" ] && lib=$PWD/boss-runs_amd/csrc/libbossx.so
  run BOSSX_LIB=$lib
  run BOSSX_LIB=$lib BOSSX_SWEEP_ONE_PER_BLOCK=1
done
