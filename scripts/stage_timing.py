"""Wall-clock of the PCIe-inclusive path on the E. coli workload: bossx_stage_batch_ptrs (PAF
text + reads in host memory -> emit runs, segments and read blob in HBM) and a whole
process_batch_paf (stage + decision update), per parser thread count."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.runs import BossRuns
os.chdir(tempfile.mkdtemp())
contigs = synth.make_reference([4_641_652], seed=1, names=["e"])
a = BossConfig(); a.optional.bucket_threshold = 0
r = BossRuns(a); r.log_fractions = False
r.init(contigs=[(n, synth.codes_to_str(c)) for n, c in contigs])
r.write_masks = False
bs = [synth.make_batch(contigs, 4000, seed=100 + i, extras=False) for i in range(6)]
for th in (1, 4, 8, 16, 32):
    os.environ["BOSSX_PARSE_THREADS"] = str(th)
    ts = []
    for b in bs:
        t = time.perf_counter(); r.engine.stage_batch(b["paf"], b["seqs"]); ts.append(time.perf_counter() - t)
    print("threads %2d  stage_batch ms: %s" % (th, " ".join("%.2f" % (1e3 * x) for x in ts)))
os.environ.pop("BOSSX_PARSE_THREADS")
for rep in range(2):
    ts = []
    for b in bs:
        r.rl_dist.update(b["read_lengths"])
        t = time.perf_counter(); r.process_batch_paf(b["paf"], b["seqs"]); ts.append(time.perf_counter() - t)
    print("process_batch_paf (PAF text -> masks on host) ms: %s" % " ".join("%.2f" % (1e3 * x) for x in ts))
