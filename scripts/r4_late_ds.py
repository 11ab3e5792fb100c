"""Bin sums of a bench-like run late in the run (the regime where the chunk-parallel chain meets capped sites):
python3 scripts/r4_late_ds.py [n_updates]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 90
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(8)])
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
R = bench.Runner(w, runs, nb, batches, False)
for i in range(n):
    R.step_e2e(batches[i % len(batches)])
    if i in (5, 30, 60, n - 1):
        name = list(runs.contigs_filt.keys())[0]
        ds = np.asarray(runs.contigs_filt[name].scores_ds)[:, 0].copy()
        lg = np.log2(np.maximum(ds, 1e-320))
        jumps = np.abs(np.diff(lg))
        print("update", i, "ds percentiles", np.percentile(ds, [0, 1, 10, 50, 90, 99, 100]), "tiny frac", float(np.mean(ds < 1e-200)),
              "zero frac", float(np.mean(ds == 0)), "bin-to-bin |log2 ratio| > 2:", float(np.mean(jumps > 2)), "> 8:", float(np.mean(jumps > 8)),
              "chain", runs.engine.chain_stats() if hasattr(runs, "engine") else None)
os.makedirs("gpurun_out/r4", exist_ok=True)
np.save("gpurun_out/r4/ds_late.npy", ds[:300000])
