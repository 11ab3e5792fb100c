"""cProfile of pipelined end-to-end steps (host side)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(30)])
import torch
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
R = bench.Runner(w, runs, nb, batches, False)
for b in batches[:4]:
    R.step_e2e(b)
R.prime(batches[4])
runs.engine.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
R.run_e2e(batches[4:29], tail=batches[29])
pr.disable()
runs.engine.synchronize()
print("ms per step", 1e3 * (time.perf_counter() - t0) / 25)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
