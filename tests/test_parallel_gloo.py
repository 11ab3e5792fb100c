"""The multi-rank protocol (boss_runs_amd.parallel) on CPUs: 2 processes, gloo backend, each
rank driving a FakeEngine (oracle numerics) for its own contigs.  The stitched result must be
bit-identical to the single-process oracle — masks, threshold, statistics — which pins the
partitioning, the collectives' payloads, the exact limb arithmetic and the halo patching."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from scenarios import REPO, E2E_REJECT, e2e_batch, e2e_contig_strings, e2e_reference

from boss_runs_amd.parallel import partition_contigs, fx_to_limbs, limbs_to_float


def test_partition_contigs():
    assert partition_contigs([5, 5], 4) == [0, 1]
    assert partition_contigs([1, 1, 1, 1], 2) == [0, 0, 1, 1]
    o = partition_contigs([10, 1, 1, 1, 10], 3)
    assert o == sorted(o) and len(set(o)) == 3
    assert partition_contigs([3, 2, 1], 1) == [0, 0, 0]
    assert partition_contigs([], 2) == []


def test_limb_roundtrip():
    rng = np.random.default_rng(0)
    fx = rng.integers(0, 2 ** 63, size=(50, 2), dtype=np.uint64)
    limbs = fx_to_limbs(fx)
    summed = limbs * 3                      # as if three ranks contributed the same value
    back = limbs_to_float(summed)
    for (lo, hi), v in zip(fx.tolist(), back):
        assert v == 3 * ((hi << 64) + lo) / (1 << 100)


def _worker(rank, world, port, tmp, nb, ploidy, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
    import torch.distributed as dist
    from fake_engine import FakeEngine
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.parallel import DistributedBossRuns
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.chdir(tempfile.mkdtemp(dir=tmp))
    contigs = e2e_reference()
    args = BossConfig()
    args.general.name = "dist%d" % rank
    args.optional.ploidy = ploidy
    args.optional.reject_refs = E2E_REJECT
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = DistributedBossRuns(args)
    runs.init(contigs=e2e_contig_strings(contigs), engine=FakeEngine(nbarcodes=nb, ploidy=ploidy),
              sharded_reads=True)
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
                        strat={n: c.strat.copy() for n, c in runs.contigs.items()},
                        approx_ccl=runs.rl_dist.approx_ccl.copy(), starts=runs.read_starts.merge().copy()))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nb,ploidy,world", [(1, 1, 2), (2, 2, 2), (1, 1, 3)])
def test_two_ranks_equal_single_process_oracle(nb, ploidy, world, tmp_path):
    """2 ranks (each owns one contig) and 3 ranks (the third owns nothing: it still takes part in
    every collective and ends with the same global threshold and masks)."""
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
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), nb, ploidy, ret), nprocs=world, join=True)
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
            assert np.array_equal(got["stats"]["counts"], exp["detail"]["counts"])
            assert np.array_equal(got["stats"]["exponents"], exp["detail"]["exponents"])
            assert np.allclose(got["stats"]["f_grid"], exp["detail"]["f_grid"], rtol=1e-11)
            for n, st in exp["strat"].items():
                assert np.array_equal(got["strat"][n], st), (rank, b, n)
