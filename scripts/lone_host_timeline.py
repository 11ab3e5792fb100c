"""Where the HOST's wall time goes inside lone updates (chr20+21): stamps around every call of process_batch_paf's body.
   gpurun -- python3 scripts/lone_host_timeline.py"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(25)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
eng = runs.engine
rows = []
def step(b):
    t = [time.perf_counter()]
    runs.rl_dist.update(b["read_lengths_arr"]); t.append(time.perf_counter())
    summ = eng.stage_batch(b["paf"], b["seqs"]); t.append(time.perf_counter())
    eng.ingest_staged(); t.append(time.perf_counter())
    eng.update_begin(runs.args.optional.bucket_threshold); t.append(time.perf_counter())
    runs.launch_benefit(); t.append(time.perf_counter())
    runs._account_reads(summ, len(b["seqs"])); t.append(time.perf_counter())
    runs.update_wrapper(); t.append(time.perf_counter())
    rows.append(np.diff(t) * 1e3)
for b in batches[:5]:
    step(b)
rows.clear()
eng.synchronize()
for b in batches[5:25]:
    step(b)
m = np.array(rows)
names = ["rl_dist.update", "stage_batch", "ingest_staged", "update_begin", "launch_benefit", "account_reads", "update_wrapper"]
for n, v in zip(names, np.median(m, axis=0)):
    print("%-16s %.3f ms" % (n, v))
print("total            %.3f ms (median of sums), min %.3f" % (np.median(m.sum(axis=1)), m.sum(axis=1).min()))
