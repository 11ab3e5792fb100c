"""Where the host front end of one update goes (run on the GPU box): per-phase wall-clock of
bossx_stage_batch_ptrs (BOSSX_STAGE_TIMING / a -DBOSSX_PARSE_TIMING build print the native
phases to stderr) and of the Python layers around it, per parser thread count.
    python3 scripts/front_end_timing.py [workload]"""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from boss_runs_amd import synth

w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 100 + i, 4000, nb) for i in range(8)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, False)
eng = runs.engine
for b in batches[:3]:
    runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None)
print("PAF text MB %.2f, reads MB %.2f" % (len(batches[0]["paf"]) / 1e6, sum(len(s) for s in batches[0]["seqs"].values()) / 1e6))
os.environ["BOSSX_STAGE_TIMING"] = "1"
for th in (8, 16, 32, 64):
    os.environ["BOSSX_PARSE_THREADS"] = str(th)
    ts = []
    for b in batches[3:]:
        t = time.perf_counter(); eng.stage_batch(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None); ts.append(time.perf_counter() - t)
    print("threads %2d  stage_batch ms: %s" % (th, " ".join("%.2f" % (1e3 * x) for x in ts)), flush=True)
os.environ.pop("BOSSX_STAGE_TIMING")
os.environ["BOSSX_PARSE_THREADS"] = "16"
# python-side pieces
b = batches[4]
t = time.perf_counter(); ids = list(b["seqs"].keys()); vals = list(b["seqs"].values()); t1 = time.perf_counter()
p = eng._str_pointers(ids); q = eng._str_pointers(vals); t2 = time.perf_counter()
enc = b["paf"].encode(); t3 = time.perf_counter()
print("python: key/value lists %.3f ms, str pointers %.3f ms, paf encode %.3f ms" % (1e3 * (t1 - t), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
for rep in range(2):
    ts = []
    for b in batches[3:]:
        t = time.perf_counter()
        runs.rl_dist.update(b["read_lengths_arr"]); runs.process_batch_paf(b["paf"], b["seqs"], barcodes=b["barcodes"] if nb > 1 else None)
        ts.append(time.perf_counter() - t)
    print("process_batch_paf ms: %s" % " ".join("%.2f" % (1e3 * x) for x in ts))
