"""Per-step wall time of lone updates, several loops in a row (is the first timed loop of bench.py representative?).
   gpurun -- python3 scripts/lone_steps.py [workload]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
    if os.environ.get("WITH_TORCH") == "1thread":
        torch.set_num_threads(1)
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 1000 + i, 4000, nb) for i in range(25)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
R = bench.Runner(w, runs, nb, batches, False)
eng = runs.engine
for b in batches[:5]:
    R.step_e2e(b)
for loop in range(2):
    if loop == 1:
        eng.enable_timing(True)
    if loop == 3:
        eng.enable_timing(False)
    ts = []
    eng.synchronize()
    for b in batches[5:25]:
        t0 = time.perf_counter()
        R.step_e2e(b)
        ts.append(1e3 * (time.perf_counter() - t0))
    print("loop %d (timing %s): mean %.3f median %.3f  | %s" % (loop, "on" if loop in (1, 2) else "off", np.mean(ts), np.median(ts), " ".join("%.2f" % t for t in ts)))
