"""The multi-rank scenario shared by the CPU protocol test (gloo, oracle-backed engine double) and the
multi-GPU test (nccl = RCCL, the real engine, one process per device): three sharded batches on the
end-to-end reference, every rank's masks / threshold / statistics against the single-process oracle."""
import os
import sys
import tempfile

import numpy as np

from scenarios import REPO, E2E_REJECT, e2e_batch, e2e_contig_strings, e2e_reference


# BOSSX_DIST_SCENARIO=four: four contigs whose longest-first packing onto two ranks is NOT contiguous (owners 0 1 0 1): the row-drift
# halo of every contig must travel whoever owns its neighbours.  Default: the end-to-end reference (two kept contigs).
# BOSSX_DIST_SCENARIO=grch27: BASELINE configs[3]'s GEOMETRY for the eight-rank protocol — the 24 chromosome lengths of GRCh38 scaled by 1/400
# (chr1 622 kb ... chr21 117 kb) + three scaffolds just above the 100-kb filter = 27 contigs, 8 Mb: longest-first packing onto eight ranks
# is non-contiguous, every rank owns three or four contigs, the 26 rows of drift of `_distribute_strategy` cross ranks at nearly every
# block boundary, the halo block is 27^2 * 2 * nb doubles and the summaries' all-gather runs at world 8.
GRCH27_SCALE = 400


def grch27_lengths():
    from boss_runs_amd import synth
    return [L // GRCH27_SCALE for L in synth.GRCH38_LENS] + [100_300, 104_100, 110_700]


def _zero_bucket_threshold():
    return os.environ.get("BOSSX_DIST_SCENARIO") in ("four", "grch27")


def _scenario():
    from boss_runs_amd import synth
    if os.environ.get("BOSSX_DIST_SCENARIO") == "grch27":
        lens = grch27_lengths()
        names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY", "scafA", "scafB", "scafC"]
        contigs = synth.make_reference(lens, seed=6, names=names)
        return contigs, [(n, synth.codes_to_str(c)) for n, c in contigs], "", \
            (lambda b, nb: synth.make_batch(contigs, 1500, seed=700 + b, mean_len=3000.0, nbarcodes=nb))
    if os.environ.get("BOSSX_DIST_SCENARIO") == "four":
        contigs = synth.make_reference([300_000, 120_000, 110_000, 290_000], seed=5, names=["f0", "f1", "f2", "f3"])
        return contigs, [(n, synth.codes_to_str(c)) for n, c in contigs], "", \
            (lambda b, nb: synth.make_batch(contigs, 900, seed=500 + b, mean_len=3000.0, nbarcodes=nb))
    contigs = e2e_reference()
    return contigs, e2e_contig_strings(contigs), E2E_REJECT, (lambda b, nb: e2e_batch(contigs, b, nb))


def worker(rank, world, port, tmp, nb, ploidy, ret, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
    import torch
    import torch.distributed as dist
    engine = None
    if backend == "nccl":
        # one process per GPU; the engine is created by DistributedBossRuns on torch's stream and the
        # update's collectives run in-stream over RCCL (boss_runs_amd/parallel.py)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        from fake_engine import FakeEngine
        dist.init_process_group("gloo", rank=rank, world_size=world)
        engine = FakeEngine(nbarcodes=nb, ploidy=ploidy)
    os.chdir(tempfile.mkdtemp(dir=tmp))
    out, runs = run_rank(rank, nb, ploidy, engine=engine, device=rank if backend == "nccl" else None)
    ret[rank] = out
    dist.barrier()
    if backend == "nccl":
        runs.engine.close()
    dist.destroy_process_group()


def run_rank(rank, nb, ploidy, engine=None, device=None, comm=None):
    """One rank of the scenario: three sharded batches through DistributedBossRuns; what it ends each update with."""
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.parallel import DistributedBossRuns
    contigs, strings, reject, make_batch = _scenario()
    args = BossConfig()
    args.general.name = "dist%d" % rank
    args.optional.ploidy = ploidy
    args.optional.reject_refs = reject
    if _zero_bucket_threshold():
        args.optional.bucket_threshold = 0
    if device is not None:
        args.gpu.device = device
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = DistributedBossRuns(args)
    runs.init(contigs=strings, engine=engine, sharded_reads=True, comm=comm)
    out = []
    for b in range(3):
        batch = make_batch(b, nb)
        # shard the reads: a read goes to the rank owning the target of the mapping the update will CHOOSE for it (filters, then
        # the best of what is left, paf.py:631-722 — a read whose first line is filtered out is decided by a later one, which may lie
        # on another rank's contig); reads on the short / rejected contigs and reads without a surviving mapping go to rank 0
        from oracle.pafcigar import parse_paf, best_mapper
        chosen = {rid: (best_mapper(recs) if len(recs) > 1 else recs[0]).tname for rid, recs in parse_paf(batch["paf"], min_len=200).items()}
        by_read = {}
        for line in batch["paf"].split("\n"):
            by_read.setdefault(line.split("\t")[0], []).append(line)
        lines, seqs, lens = [], {}, {}
        for rid, seq in batch["seqs"].items():
            ls = by_read.get(rid, [])
            owner = runs.owner_of.get(chosen.get(rid), 0)
            if owner == rank:
                lines.extend(ls)
                seqs[rid] = seq
                lens[rid] = batch["read_lengths"][rid]
        bcs = {k: batch["barcodes"][k] for k in seqs} if nb > 1 else None
        runs.process_batch_paf("\n".join(lines), seqs, barcodes=bcs, read_lengths=lens)
        out.append(dict(threshold=runs.threshold, stats={k: np.asarray(v) for k, v in runs.last_stats.items()},
                        strat={n: np.array(c.strat, copy=True) for n, c in runs.contigs.items()},
                        approx_ccl=runs.rl_dist.approx_ccl.copy(), starts=runs.read_starts.merge().copy(),
                        collectives=runs.n_collectives))
    return out, runs


def oracle_expected(nb, ploidy):
    from oracle.pipeline import OracleRuns
    contigs, strings, reject, make_batch = _scenario()
    kw = dict(bucket_threshold=0) if _zero_bucket_threshold() else {}
    o = OracleRuns(strings, ploidy=ploidy, reject_refs={reject} if reject else set(), nbarcodes=nb, **kw)
    expected = []
    for b in range(3):
        batch = make_batch(b, nb)
        # multi-mapper second lines target another contig; the oracle sees the whole batch
        o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"],
                        barcodes=batch["barcodes"] if nb > 1 else None)
        expected.append(dict(threshold=o.threshold, detail=dict(o.detail),
                             strat={n: c.strat.copy() for n, c in o.contigs.items()},
                             approx_ccl=o.rl_dist.approx_ccl.copy(), starts=o.read_starts.merge().copy()))
    return expected


def check(ret, expected, world, full_stats=True):
    assert set(ret.keys()) == set(range(world))
    for rank in range(world):
        for b in range(3):
            got, exp = ret[rank][b], expected[b]
            assert np.array_equal(got["approx_ccl"], exp["approx_ccl"])
            assert np.array_equal(got["starts"], exp["starts"])
            if exp["threshold"] is None:
                assert got["threshold"] is None
                continue
            assert got["threshold"] == exp["threshold"], (rank, b)
            assert got["stats"]["normaliser"] == exp["detail"]["normaliser"]
            assert got["stats"]["strat_size"] == exp["detail"]["strat_size"]
            if full_stats:       # the host-side protocol carries the whole histogram; the in-stream one keeps it in HBM
                assert np.array_equal(got["stats"]["counts"], exp["detail"]["counts"])
                assert np.array_equal(got["stats"]["exponents"], exp["detail"]["exponents"])
                assert np.allclose(got["stats"]["f_grid"], exp["detail"]["f_grid"], rtol=1e-11)
            for n, st in exp["strat"].items():
                assert np.array_equal(got["strat"][n], st), (rank, b, n)


class ThreadComm:
    """Stand-in for parallel.Comm when the ranks are THREADS of one process (tests/rccl_loopback): the few host-side
    exchanges of DistributedBossRuns (the communicator id at init, the finished masks) go through a shared
    barrier; everything per update goes through the library's own communicator (BOSSX_RCCL_LIB)."""

    class Shared:
        def __init__(self, world):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world, timeout=120)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        import torch
        self.shared, self.rank, self.world = shared, rank, shared.world
        self.on, self.device, self.torch, self.dist, self.force = True, "cuda", torch, None, False
        self.n_collectives = 0

    def allgather(self, arr):
        sh = self.shared
        sh.slots[self.rank] = np.ascontiguousarray(arr).copy()
        sh.barrier.wait()
        out = np.stack(sh.slots)
        sh.barrier.wait()
        self.n_collectives += 1
        return out

    def allreduce(self, arr, op="sum"):
        allv = self.allgather(arr)
        return allv.sum(axis=0).astype(allv.dtype) if op == "sum" else allv.max(axis=0)


def main_threads(nb, ploidy, tmp, out_path, world=2):
    """`world` ranks as threads of THIS process, every engine on device 0, the native driver's collectives through the
    loopback library: run with BOSSX_RCCL_LIB set (test_parity_gpu.py builds it and starts this as a subprocess)."""
    import pickle
    import threading
    assert os.environ.get("BOSSX_RCCL_LIB"), "BOSSX_RCCL_LIB must name the loopback library"
    sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
    os.chdir(tmp)
    shared = ThreadComm.Shared(world)
    ret, errs = {}, []

    def go(rank):
        try:
            ret[rank], runs = run_rank(rank, nb, ploidy, device=0, comm=ThreadComm(shared, rank))
            shared.barrier.wait()
            runs.engine.close()
        except BaseException as e:      # noqa: BLE001 - reported by the parent
            import traceback
            errs.append((rank, traceback.format_exc()))
            shared.barrier.abort()

    threads = [threading.Thread(target=go, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    with open(out_path, "wb") as fh:
        pickle.dump(dict(ret=ret, errs=errs), fh)
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main_threads(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], world=int(sys.argv[5]) if len(sys.argv) > 5 else 2))
