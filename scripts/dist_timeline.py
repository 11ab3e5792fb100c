"""Host timeline of the distributed (in-stream collectives) update on ONE rank with the
collectives forced: run under torchrun --nproc-per-node 1 with BOSSX_FORCE_COLLECTIVES=1."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ["BOSSX_FORCE_COLLECTIVES"] = "1"
torch.cuda.set_device(0)
dist.init_process_group("nccl")
from boss_runs_amd import synth
from boss_runs_amd.config import BossConfig
from boss_runs_amd.parallel import DistributedBossRuns
os.chdir(tempfile.mkdtemp())
contigs = synth.make_reference([4_641_652], seed=1, names=["e"])
a = BossConfig(); a.optional.bucket_threshold = 0
r = DistributedBossRuns(a)
r.init(contigs=[(n, synth.codes_to_str(c)) for n, c in contigs], sharded_reads=True, gather_masks=False)
r.write_masks = False; r.log_fractions = False
N = 13
bs = [synth.make_batch(contigs, 4000, seed=100 + i, extras=False) for i in range(N)]
ss = []
for i, b in enumerate(bs):
    r.engine.select_batch(i); ss.append(r.engine.stage_batch(b["paf"], b["seqs"]))
    b["rl"] = np.fromiter(b["read_lengths"].values(), dtype=np.int64)
marks = []
def step(i):
    t = [time.perf_counter()]
    r.engine.ingest_staged(slot=i); r.begin_update(); t.append(time.perf_counter())
    r.account_batch(ss[i], bs[i]["rl"], 4000); t.append(time.perf_counter())
    r.update_wrapper(); t.append(time.perf_counter())
    marks.append(np.diff(t) * 1e3)
for i in range(8): step(i)
m = np.array(marks[3:])
for n, v in zip(["ingest+begin", "account_batch", "update_wrapper"], np.median(m, axis=0)):
    print("%-16s %.3f ms" % (n, v))
print("total            %.3f ms" % np.median(m.sum(axis=1)))
# sub-steps of account_batch
import boss_runs_amd.parallel as par
orig_gather = r.comm.allgather
tg = []
def timed_gather(a):
    t0 = time.perf_counter(); o = orig_gather(a); tg.append(time.perf_counter() - t0); return o
r.comm.allgather = timed_gather
orig_rl = r.rl_dist.update
trl = []
def timed_rl(x):
    t0 = time.perf_counter(); orig_rl(x); trl.append(time.perf_counter() - t0)
r.rl_dist.update = timed_rl
orig_early = r._launch_chain_early
te = []
def timed_early():
    t0 = time.perf_counter(); orig_early(); te.append(time.perf_counter() - t0)
r._launch_chain_early = timed_early
pr = cProfile.Profile(); pr.enable()
for i in range(8, 13): step(i)
pr.disable()
print('allgather %.3f  rl %.3f  early-chain %.3f ms' % (1e3*np.median(tg), 1e3*np.median(trl), 1e3*np.median(te)))
dist.destroy_process_group()
