"""Lone-update wall time under environment variants that the library reads PER CALL, alternated update by update inside ONE process (boxes and
processes differ by ~0.1 ms: this is the comparison that resolves less):
   gpurun -- python3 scripts/ab_inproc.py "<VAR=val ...>" "<VAR=val ...>" ...     ("-" = none)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
arms = sys.argv[1:] or ["-"]
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
N = 8 + int(os.environ.get("AB_PER_ARM", "12")) * len(arms)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(N)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
R = bench.Runner(w, runs, nb, batches, False)
for b in batches[:8]:
    R.step_e2e(b)
ts = {a: [] for a in arms}
keys = sorted({kv.split("=")[0] for a in arms if a != "-" for kv in a.split()})
for i, b in enumerate(batches[8:]):
    a = arms[i % len(arms)]
    for k in keys:
        os.environ.pop(k, None)
    if a != "-":
        for kv in a.split():
            k, v = kv.split("=", 1)
            os.environ[k] = v
    runs.engine.synchronize()
    t0 = time.perf_counter()
    R.step_e2e(b)
    ts[a].append(1e3 * (time.perf_counter() - t0))
for a in arms:
    print("[%s] median %.3f mean %.3f min %.3f ms over %d updates" % (a, np.median(ts[a]), np.mean(ts[a]), np.min(ts[a]), len(ts[a])))
