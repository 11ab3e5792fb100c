import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(25)])
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0)
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
eng = runs.engine
ts = []
for i in range(40):
    b = batches[i % 25]
    t0 = time.perf_counter()
    eng.stage_batch(b["paf"], b["seqs"])
    ts.append(1e3 * (time.perf_counter() - t0))
print("stage only:", " ".join("%.1f" % t for t in ts))
R = bench.Runner(w, runs, nb, batches, False)
if os.environ.get("WITH_TORCH"):
    torch.cuda.synchronize()
eng.enable_timing(bool(os.environ.get("WITH_TIMING")))
ts = []
for i in range(25):
    t0 = time.perf_counter(); R.step_e2e(batches[i]); ts.append(1e3 * (time.perf_counter() - t0))
print("updates:", " ".join("%.1f" % t for t in ts))
if os.environ.get("WITH_TORCH"):
    torch.cuda.synchronize()
eng.synchronize(); eng.kernel_stats()
ts = []
for i in range(25):
    t0 = time.perf_counter(); R.step_e2e(batches[i]); ts.append(1e3 * (time.perf_counter() - t0))
print("updates behind a synchronize + kernel_stats:", " ".join("%.1f" % t for t in ts))

if os.environ.get("WITH_TORCH"):
    for rep in range(3):
        torch.cuda.synchronize(); eng.synchronize()
        ts = []
        eng.select_batch(5)
        for i in range(3):
            t0 = time.perf_counter(); eng.stage_batch(batches[i]["paf"], batches[i]["seqs"]); ts.append(1e3 * (time.perf_counter() - t0))
        eng.select_batch(0)
        for i in range(4):
            t0 = time.perf_counter(); R.step_e2e(batches[i]); ts.append(1e3 * (time.perf_counter() - t0))
        print("behind torch.cuda.synchronize(): three stage-only calls, then four updates:", " ".join("%.1f" % t for t in ts))
    for rep in range(3):
        torch.cuda.synchronize(); eng.synchronize()
        time.sleep(0.05)
        ts = []
        for i in range(4):
            t0 = time.perf_counter(); R.step_e2e(batches[i]); ts.append(1e3 * (time.perf_counter() - t0))
        print("behind torch.cuda.synchronize() + 50 ms: four updates:", " ".join("%.1f" % t for t in ts))
