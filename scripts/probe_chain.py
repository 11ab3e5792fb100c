import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np
os.environ["BOSSX_CHAIN_PROBE"] = "1"
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.runs import BossRuns, MULT
os.chdir(tempfile.mkdtemp())
contigs = synth.make_reference([int(os.environ.get("PROBE_L", "4641652"))], seed=1, names=["e"])
a = BossConfig(); a.optional.bucket_threshold = 0
r = BossRuns(a); r.write_masks = False
r.init(contigs=[(n, synth.codes_to_str(c)) for n, c in contigs])
r.engine.preload_coverage(6.0, seed=3)
r.engine.sweep()
w = np.concatenate(([4], r.rl_dist.approx_ccl // 100)).astype(np.int32)
for i in range(3):
    t = time.perf_counter(); mx = r.engine.benefit(w, MULT); print("benefit ms", 1e3 * (time.perf_counter() - t), mx)
