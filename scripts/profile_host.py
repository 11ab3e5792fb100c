"""cProfile of the host side of one decision update (E. coli workload), bench-style step."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.runs import BossRuns
os.chdir(tempfile.mkdtemp())
contigs = synth.make_reference([4_641_652], seed=1, names=["e"])
a = BossConfig(); a.optional.bucket_threshold = 0
r = BossRuns(a); r.write_masks = False; r.log_fractions = False
r.init(contigs=[(n, synth.codes_to_str(c)) for n, c in contigs])
bs = [synth.make_batch(contigs, 4000, seed=100 + i, extras=False) for i in range(8)]
ss = []
for i, b in enumerate(bs):
    r.engine.select_batch(i); ss.append(r.engine.stage_batch(b["paf"], b["seqs"]))
    b["rl"] = np.fromiter(b["read_lengths"].values(), dtype=np.int64)
def step(i):
    r.engine.ingest_staged(slot=i)
    r.engine.update_begin(0)
    r.rl_dist.update(bs[i]["rl"]); r.launch_benefit()
    r._account_reads(ss[i], 4000); r.update_wrapper()
step(0); step(1)
t = time.perf_counter()
for i in range(2, 5): step(i)
print("ms per step (no profiler)", 1e3 * (time.perf_counter() - t) / 3)
pr = cProfile.Profile(); pr.enable()
for i in range(5, 8): step(i)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
