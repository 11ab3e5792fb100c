"""Dump the downsampled scores (scores_ds) and the move_sum windows of a bench-like run, for offline analysis."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(8)])
import torch
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
R = bench.Runner(w, runs, nb, batches, False)
for b in batches:
    R.step_e2e(b)
os.makedirs("gpurun_out/r3ds", exist_ok=True)
name = list(runs.contigs_filt.keys())[0]
c = runs.contigs_filt[name]
ds = np.asarray(c.scores_ds)[:, 0].copy()
np.save("gpurun_out/r3ds/ds_%s.npy" % name, ds[:200000])
print(name, ds.shape, "windows", getattr(runs, "_last_windows", None))
import inspect
print("approx_ccl", runs.rl_dist.approx_ccl, "mu", runs.args.optional.mu if hasattr(runs.args.optional, "mu") else None)
print("ds stats", ds.min(), ds.max(), np.mean(ds == 0), np.percentile(ds, [1, 10, 50, 90, 99]))
