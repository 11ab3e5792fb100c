"""Which call of a lone update takes the milliseconds when an update stalls (bench.py's conditions: torch's context alive, a
torch.cuda.synchronize() + engine synchronize in front of the loop):   gpurun -- python3 scripts/stall_hunt.py [rounds]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import torch
torch.cuda.set_device(0)
w = "chr20_21"
bench._GEN[w] = bench.make_reference(w, 0)
batches = bench.generate_batches([(w, 1000 + i, 4000, 1) for i in range(25)])
os.chdir(tempfile.mkdtemp())
runs, nb = bench.make_runs(w, bench._GEN[w], 0, 1, 0, True)
eng = runs.engine
eng.enable_timing(True)
names = ["rl_dist", "stage_batch", "ingest", "update_begin", "launch_benefit", "account_reads", "update_wrapper"]
def step(b):
    t = [time.perf_counter()]
    runs.rl_dist.update(b["read_lengths_arr"]); t.append(time.perf_counter())
    summ = eng.stage_batch(b["paf"], b["seqs"]); t.append(time.perf_counter())
    eng.ingest_staged(); t.append(time.perf_counter())
    eng.update_begin(runs.args.optional.bucket_threshold); t.append(time.perf_counter())
    runs.launch_benefit(); t.append(time.perf_counter())
    runs._account_reads(summ, len(b["seqs"])); t.append(time.perf_counter())
    runs.update_wrapper(); t.append(time.perf_counter())
    return np.diff(t) * 1e3
torch.cuda.synchronize(); eng.synchronize()
for b in batches[:5]:
    step(b)
eng.enable_timing(True, only="site_sweep")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tot = []
for r in range(rounds):
    import gc; gc.disable()
    torch.cuda.synchronize(); eng.synchronize()
    for i, b in enumerate(batches[5:25]):
        d = step(b)
        tot.append(d.sum())
        if d.sum() > 2.6:
            print("round %d step %2d: %.2f ms  " % (r, i, d.sum()) + "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d)))
    gc.enable()
print("updates %d, median %.3f, mean %.3f, > 2.6 ms: %d" % (len(tot), np.median(tot), np.mean(tot), sum(1 for x in tot if x > 2.6)))
