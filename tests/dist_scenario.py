"""The multi-rank scenario shared by the CPU protocol test (gloo, oracle-backed engine double) and the
multi-GPU test (nccl = RCCL, the real engine, one process per device): three sharded batches on the
end-to-end reference, every rank's masks / threshold / statistics against the single-process oracle."""
import os
import sys
import tempfile

import numpy as np

from scenarios import REPO, E2E_REJECT, e2e_batch, e2e_contig_strings, e2e_reference


def worker(rank, world, port, tmp, nb, ploidy, ret, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
    import torch
    import torch.distributed as dist
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.parallel import DistributedBossRuns
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
    contigs = e2e_reference()
    args = BossConfig()
    args.general.name = "dist%d" % rank
    args.optional.ploidy = ploidy
    args.optional.reject_refs = E2E_REJECT
    if backend == "nccl":
        args.gpu.device = rank
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = DistributedBossRuns(args)
    runs.init(contigs=e2e_contig_strings(contigs), engine=engine, sharded_reads=True)
    out = []
    for b in range(3):
        batch = e2e_batch(contigs, b, nb)
        # shard the reads: a read goes to the rank owning the target of its first PAF line
        # (reads on the short / rejected contigs and unmapped reads go to rank 0)
        by_read = {}
        for line in batch["paf"].split("\n"):
            by_read.setdefault(line.split("\t")[0], []).append(line)
        lines, seqs, lens = [], {}, {}
        for rid, seq in batch["seqs"].items():
            ls = by_read.get(rid, [])
            tgt = ls[0].split("\t")[5] if ls else None
            owner = runs.owner_of.get(tgt, 0)
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
    ret[rank] = out
    dist.barrier()
    if backend == "nccl":
        runs.engine.close()
    dist.destroy_process_group()


def oracle_expected(nb, ploidy):
    from oracle.pipeline import OracleRuns
    contigs = e2e_reference()
    o = OracleRuns(e2e_contig_strings(contigs), ploidy=ploidy, reject_refs={E2E_REJECT}, nbarcodes=nb)
    expected = []
    for b in range(3):
        batch = e2e_batch(contigs, b, nb)
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
