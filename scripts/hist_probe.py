"""Cycle stamps of the histogram kernel's phases (a -DBOSSX_HIST_PROBE build: make -C boss-runs_amd/csrc variant NAME=hprobe DEFS=-DBOSSX_HIST_PROBE):
   gpurun -- env BOSSX_LIB=$PWD/boss-runs_amd/csrc/libbossx_hprobe.so python3 scripts/hist_probe.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 1000 + i, 4000, nb) for i in range(8)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
R = bench.Runner(w, runs, nb, batches, False)
for b in batches:
    R.step_e2e(b)
