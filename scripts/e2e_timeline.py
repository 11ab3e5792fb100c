"""Host-side timeline of the end-to-end update (python3 scripts/e2e_timeline.py [workload])."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
nb = bench.WORKLOADS[w][3]
batches = bench.generate_batches([(w, 100 + i, 4000, nb) for i in range(14)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, os.environ.get("ENT", "1") == "1")
eng = runs.engine
T = {}
for it, b in enumerate(batches):
    bcs = b["barcodes"] if nb > 1 else None
    eng.synchronize()
    t0 = time.perf_counter()
    runs.rl_dist.update(b["read_lengths_arr"])
    t1 = time.perf_counter()
    summ = eng.ingest_paf(b["paf"], b["seqs"], barcodes=bcs)
    t2 = time.perf_counter()
    eng.update_begin(runs.args.optional.bucket_threshold)
    runs.launch_benefit()
    t3 = time.perf_counter()
    runs._account_reads(summ, len(b["seqs"]))
    t4 = time.perf_counter()
    fh = runs.read_starts.fhat_compact()
    t5 = time.perf_counter()
    runs.update_wrapper()
    t6 = time.perf_counter()
    if it >= 4:
        for k, v in (("rl", t1 - t0), ("stage", t2 - t1), ("launch", t3 - t2), ("account", t4 - t3), ("fhat(extra)", t5 - t4),
                     ("update_wrapper", t6 - t5), ("total-minus-extra", t6 - t0 - (t5 - t4))):
            T.setdefault(k, []).append(1e3 * v)
for k, v in T.items():
    print("%-18s median %.3f   %s" % (k, float(np.median(v)), " ".join("%.2f" % x for x in v)))
