"""Host-side timeline of the pipelined end-to-end step (process_batch_paf(lookahead=...)): where the wall time of a
step goes, section by section.  gpurun -- python3 scripts/r3_la_timeline.py [workload]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nbm = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 1000 + i, 4000, nbm) for i in range(16)])
import torch
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
eng = runs.engine
T = {}
def wrap(obj, name, key=None):
    f = getattr(obj, name)
    key = key or name
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T.setdefault(key, []).append(time.perf_counter() - t0)
    setattr(obj, name, g)
wrap(runs, "_ingest_batch"); wrap(runs, "launch_benefit"); wrap(runs, "_stage_ahead"); wrap(runs, "_account_reads")
wrap(runs, "update_wrapper"); wrap(eng, "update_begin"); wrap(eng, "update", "engine.update"); wrap(runs.rl_dist, "update", "rl_dist.update")
if hasattr(runs, "_publish_masks"): wrap(runs, "_publish_masks")
R = bench.Runner(w, runs, nb, batches, False)
for b in batches[:4]:
    R.step_e2e(b)
R.prime(batches[4])
T.clear()
eng.synchronize()
steps = []
for i in range(4, 15):
    t0 = time.perf_counter()
    R.step_e2e(batches[i], batches[i + 1])
    steps.append(time.perf_counter() - t0)
eng.synchronize()
print("step ms: mean %.3f  min %.3f  max %.3f" % (1e3 * np.mean(steps), 1e3 * min(steps), 1e3 * max(steps)))
for k, v in T.items():
    print("  %-18s mean %.3f ms  (n=%d)" % (k, 1e3 * np.mean(v), len(v)))
