"""cProfile of the host side of one decision update (E. coli workload)."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.runs import BossRuns
os.chdir(tempfile.mkdtemp())
contigs = synth.make_reference([4_641_652], seed=1, names=["e"])
a = BossConfig(); a.optional.bucket_threshold = 0
r = BossRuns(a); r.write_masks = False
r.init(contigs=[(n, synth.codes_to_str(c)) for n, c in contigs])
bs = [synth.make_batch(contigs, 4000, seed=100 + i, extras=False) for i in range(6)]
ss = []
for i, b in enumerate(bs):
    r.engine.select_batch(i); ss.append(r.engine.stage_batch(b["paf"], b["seqs"]))
def step(i):
    r.rl_dist.update(bs[i]["read_lengths"]); r.engine.ingest_staged(slot=i)
    r._account_reads(ss[i], 4000); r.update_wrapper()
step(0); step(1)
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for i in range(2, 6): step(i)
el = time.perf_counter() - t
pr.disable()
print("ms per step", 1e3 * el / 4)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
