"""A/B of the staging knobs on ONE box (boxes differ by 2x): mean / median wall-clock of
Engine.stage_batch over the chr20+21 bench batches per setting of the environment knobs."""
import itertools, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 100 + i, 4000, 1) for i in range(12)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
eng = runs.engine
print("cpus", os.cpu_count(), "load", os.getloadavg())
for b in batches[:3]:
    runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"])
def measure(env):
    for k in ("BOSSX_UP_STREAMS", "BOSSX_POOL_THREADS", "BOSSX_LINE_TASK_KB", "BOSSX_PARSE_THREADS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for rep in range(3):
        for b in batches[3:]:
            t = time.perf_counter(); eng.stage_batch(b["paf"], b["seqs"]); ts.append(time.perf_counter() - t)
            time.sleep(0.003)          # the workers go back to sleep between batches, as in a real update
    ts = 1e3 * np.array(ts[3:])
    return float(np.mean(ts)), float(np.median(ts)), float(np.min(ts))
configs = [{}]
for up in ("1", "4"):
    for pool in ("16", "32", "64"):
        configs.append({"BOSSX_UP_STREAMS": up, "BOSSX_POOL_THREADS": pool})
configs += [{"BOSSX_LINE_TASK_KB": "24"}, {"BOSSX_LINE_TASK_KB": "96"}, {"BOSSX_LINE_TASK_KB": "384"}, {}]
for env in configs:
    m, med, mn = measure(env)
    print("%-60s mean %.3f median %.3f min %.3f ms" % (env or "default", m, med, mn), flush=True)
