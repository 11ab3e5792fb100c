"""Host-side timeline of one bench-style decision update (E. coli): where the wall-clock goes
between the launches and the final synchronisation."""
import os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.runs import BossRuns
os.chdir(tempfile.mkdtemp())
contigs = synth.make_reference([4_641_652], seed=1, names=["e"])
N = 13
bs = [synth.make_batch(contigs, 4000, seed=100 + i, extras=False) for i in range(N)]
for b in bs:
    b["rl"] = np.fromiter(b["read_lengths"].values(), dtype=np.int64)


def make(name, overlap):
    if overlap:
        os.environ.pop("BOSSX_NO_OVERLAP", None)
        os.environ["BOSSX_OVERLAP"] = "1"
    else:
        os.environ.pop("BOSSX_OVERLAP", None)
        os.environ["BOSSX_NO_OVERLAP"] = "1"
    a = BossConfig(); a.optional.bucket_threshold = 0; a.general.name = name
    r = BossRuns(a); r.log_fractions = False
    r.init(contigs=[(n, synth.codes_to_str(c)) for n, c in contigs])
    r.write_masks = False
    ss = []
    for i, b in enumerate(bs):
        r.engine.select_batch(i); ss.append(r.engine.stage_batch(b["paf"], b["seqs"]))
    r.engine.enable_timing(True)
    return r, ss


variants = {"overlap": make("ov", True), "serial": make("se", False)}
results = {}
for vname, (r, ss) in variants.items():
  marks = []
  def step(i):
    t = [time.perf_counter()]
    r.engine.ingest_staged(slot=i); t.append(time.perf_counter())
    r.engine.update_begin(0); t.append(time.perf_counter())
    r.rl_dist.update(bs[i]["rl"]); t.append(time.perf_counter())
    r.launch_benefit(); t.append(time.perf_counter())
    r._account_reads(ss[i], 4000); t.append(time.perf_counter())
    r.update_wrapper(); t.append(time.perf_counter())
    marks.append(np.diff(t) * 1e3)
  for i in range(N): step(i)
  m = np.array(marks[3:])
  names = ["ingest_staged", "update_begin", "rl_dist.update", "launch_benefit", "account_reads", "update_wrapper"]
  print("== %s" % vname)
  for n, v in zip(names, np.median(m, axis=0)):
    print("%-16s %.3f ms" % (n, v))
  print("total            %.3f ms  (min %.3f)" % (np.median(m.sum(axis=1)), m.sum(axis=1).min()))
  ks = r.engine.kernel_stats()
  print({k: round(v["ms_last"], 4) for k, v in ks.items()})
