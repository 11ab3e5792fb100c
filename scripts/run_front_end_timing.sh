#!/bin/bash
mkdir -p gpurun_out/fe
make -C boss-runs_amd/csrc clean > /dev/null; make -C boss-runs_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -pthread -Wall -I../../include -I. -DBOSSX_PARSE_TIMING" > gpurun_out/fe/build.log 2>&1
for w in chr20_21 ecoli; do python3 scripts/front_end_timing.py $w > gpurun_out/fe/$w.log 2>&1; done
tail -40 gpurun_out/fe/chr20_21.log
