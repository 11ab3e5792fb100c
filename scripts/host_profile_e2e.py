"""cProfile of the host side of lone end-to-end decision updates (bench-style step, nothing staged ahead):
    python3 scripts/host_profile_e2e.py [workload]        (BOSSX_STAGE_TIMING=1 adds the library's own split)"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 100 + i, 4000, nb) for i in range(16)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
R = bench.Runner(w, runs, nb, batches, False)
for b in batches[:4]:
    R.step_e2e(b)
runs.engine.synchronize()
t = time.perf_counter()
for b in batches[4:10]:
    R.step_e2e(b)
print("ms per lone step (no profiler) %.3f" % (1e3 * (time.perf_counter() - t) / 6))
if os.environ.get("BOSSX_STAGE_TIMING"):
    sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for b in batches[10:16]:
    R.step_e2e(b)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
