"""Host-side timeline of one update at 3.1 Gb on one GPU (where do the ms outside the chain go?)."""
import os, sys, tempfile, time, argparse
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
a = argparse.Namespace(steps=4, warmup=2, reads=4000, track_entropy=False)
# reuse bench.run_grch38's set-up by monkeypatching Runner.step_e2e to a timed version
T = {}
def timed_step(self, b):
    runs, eng = self.runs, self.eng
    t0 = time.perf_counter()
    runs.rl_dist.update(b["read_lengths_arr"])
    t1 = time.perf_counter()
    summ = eng.ingest_paf(b["paf"], b["seqs"])
    t2 = time.perf_counter()
    eng.update_begin(runs.args.optional.bucket_threshold)
    runs.launch_benefit()
    t3 = time.perf_counter()
    runs._account_reads(summ, len(b["seqs"]))
    t4 = time.perf_counter()
    runs.update_wrapper()
    t5 = time.perf_counter()
    for k, v in (("rl", t1 - t0), ("stage", t2 - t1), ("launch", t3 - t2), ("account", t4 - t3), ("update_wrapper", t5 - t4), ("total", t5 - t0)):
        T.setdefault(k, []).append(1e3 * v)
bench.Runner.step_e2e = timed_step
import torch
torch.cuda.set_device(0)
os.chdir(tempfile.mkdtemp())
mode = sys.argv[1] if len(sys.argv) > 1 else "bits"
res = bench.run_grch38(a, 0, 1, 0)
for k, v in T.items():
    print("%-15s %s" % (k, " ".join("%.2f" % x for x in v)))
print(res["ms_per_step"], res["benefit_chain_ms_rank0"], res["site_sweep_rank0"]["avg_ms"])
