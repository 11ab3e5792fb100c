"""Where do lone updates get slower as coverage accumulates?  60 lone updates with the engine's HIP events on: wall time and the
per-kernel time of every update (sweep / benefit chain / histogram / masks), plus the chain's counters every ten updates.
   gpurun -- python3 scripts/drift_steps.py [workload] [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 1000 + i, 4000, nb) for i in range(steps + 5)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
R = bench.Runner(w, runs, nb, batches, False)
eng = runs.engine
for b in batches[:5]:
    R.step_e2e(b)
eng.enable_timing(True)
eng.synchronize()
rows = []
stats_from = int(os.environ.get("DRIFT_STATS_FROM", "-1"))      # the chain's own counters (BOSSX_SPEC_STATS, printed at exit) from this update on only
for i, b in enumerate(batches[5:]):
    if i == stats_from:
        os.environ["BOSSX_SPEC_STATS"] = "1"
    t0 = time.perf_counter()
    R.step_e2e(b)
    dt = 1e3 * (time.perf_counter() - t0)
    ks = eng.kernel_stats()
    rows.append((dt, ks["site_sweep"]["ms_last"], ks["benefit_chain"]["ms_last"], ks["threshold_hist"]["ms_last"], ks["strategy_mask"]["ms_last"], ks["site_sweep"]["bytes_last"] / 1e6))
    if i % 10 == 9:
        print("update %3d: wall %.2f ms | sweep %.3f chain %.3f hist %.3f mask %.3f (sweep: %.0f MB) | %s" % ((i + 1,) + rows[-1] + (eng.chain_stats(),)))
r = np.array(rows)
for lo in range(0, steps, 10):
    m = np.median(r[lo:lo + 10], axis=0)
    print("updates %2d-%2d median: wall %.3f | sweep %.3f chain %.3f hist %.3f mask %.3f | rest %.3f | sweep %.0f MB = %.2f TB/s" % ((lo + 1, lo + 10) + tuple(m[:5]) + (m[0] - m[1:5].sum(), m[5], m[5] / m[1] / 1e3)))
