"""Sweep-kernel experiments on a large shard: fresh state (no LUT gathers) vs preloaded state."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_runs_amd import synth
from boss_runs_amd.engine import Engine
from boss_runs_amd.scoring import SiteScoring
L = int(os.environ.get("SWEEP_L", "200000000"))
ploidy = int(os.environ.get("SWEEP_PLOIDY", "2"))
reps = int(os.environ.get("SWEEP_REPS", "5"))
depth = float(os.environ.get("SWEEP_DEPTH", "8"))
rng = np.random.default_rng(1)
codes = rng.integers(0, 4, size=L, dtype=np.uint8)
seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes()
e = Engine(nbarcodes=1, track_entropy=True)
e.add_contig("big", seq)
hap = SiteScoring(1); sc = hap if ploidy == 1 else SiteScoring(ploidy)
e.finalize(hap.score0[0], hap.ent0[0]); e.set_lut(*sc.tables())
e.enable_timing(True)
def run(tag):
    for i in range(reps):
        e.sweep(); e.synchronize()
        st = e.kernel_stats()["site_sweep"]
        print(tag, i, "ms", round(st["ms_last"], 4), "GB/s", round(st["bytes_last"] / 1e6 / st["ms_last"], 1), flush=True)
run("fresh(no gathers)")
if depth > 0:
    e.preload_coverage(depth, seed=3)
    run("preloaded depth %g (first sweep touches all)" % depth)
