"""The multi-rank protocol (boss_runs_amd.parallel) on CPUs: 2 processes, gloo backend, each
rank driving a FakeEngine (oracle numerics) for its own contigs.  The stitched result must be
bit-identical to the single-process oracle — masks, threshold, statistics — which pins the
partitioning, the collectives' payloads, the exact limb arithmetic and the halo patching."""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

from boss_runs_amd.parallel import partition_contigs, fx_to_limbs, limbs_to_float


def test_partition_contigs():
    assert partition_contigs([5, 5], 4) == [0, 1]
    assert partition_contigs([1, 1, 1, 1], 2) == [0, 0, 1, 1]
    o = partition_contigs([10, 1, 1, 1, 10], 3, "linear")
    assert o == sorted(o) and len(set(o)) == 3
    assert partition_contigs([3, 2, 1], 1) == [0, 0, 0]
    assert partition_contigs([], 2) == []
    # longest-first packing against the contiguous partition on the GRCh38 contig set of bench.py
    from boss_runs_amd.parallel import shard_balance
    import bench
    lens = [L for L in bench.GRCH38_LENGTHS if L >= 100_000]
    for world, lin_max, lpt_max in ((8, 1.19, 1.02), (4, 1.09, 1.01), (2, 1.005, 1.001)):
        lin, lpt = partition_contigs(lens, world, "linear"), partition_contigs(lens, world, "lpt")
        assert lin == sorted(lin) and set(lin) == set(range(world)) == set(lpt)
        assert shard_balance(lens, lpt, world) <= lpt_max < shard_balance(lens, lin, world) + 0.1
        assert shard_balance(lens, lin, world) >= lin_max
        auto = partition_contigs(lens, world)
        assert shard_balance(lens, auto, world) == min(shard_balance(lens, lin, world), shard_balance(lens, lpt, world))
    # equal shards repeated per rank (the weak-scaling bench): the tie goes to the contiguous partition, each rank its own copy
    assert partition_contigs([64, 46] * 4, 4) == [0, 0, 1, 1, 2, 2, 3, 3]
    assert partition_contigs([64, 46] * 4, 4, "lpt") == [0, 0, 1, 1, 2, 2, 3, 3]


def test_limb_roundtrip():
    rng = np.random.default_rng(0)
    fx = rng.integers(0, 2 ** 63, size=(50, 2), dtype=np.uint64)
    limbs = fx_to_limbs(fx)
    summed = limbs * 3                      # as if three ranks contributed the same value
    back = limbs_to_float(summed)
    for (lo, hi), v in zip(fx.tolist(), back):
        assert v == 3 * ((hi << 64) + lo) / (1 << 100)


@pytest.mark.parametrize("nb,ploidy,world", [(1, 1, 2), (2, 2, 2), (1, 1, 3)])
def test_two_ranks_equal_single_process_oracle(nb, ploidy, world, tmp_path):
    """2 ranks (each owns one contig) and 3 ranks (the third owns nothing: it still takes part in
    every collective and ends with the same global threshold and masks)."""
    import dist_scenario
    expected = dist_scenario.oracle_expected(nb, ploidy)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(dist_scenario.worker, args=(world, port, str(tmp_path), nb, ploidy, ret), nprocs=world, join=True)
    dist_scenario.check(ret, expected, world)


@pytest.mark.parametrize("method", ["lpt", "linear"])
def test_noncontiguous_partition_equals_single_process_oracle(method, tmp_path, monkeypatch):
    """Four contigs (300 / 120 / 110 / 290 kb) on two ranks: longest-first packing gives owners 0 1 0 1 — every contig's
    predecessor in the merged array lives on the OTHER rank, so each halo row of `_distribute_strategy` (core.py:125-155)
    crosses ranks — and the contiguous partition 0 0 1 1.  Thresholds, statistics and masks equal the single-process oracle
    either way."""
    import dist_scenario
    monkeypatch.setenv("BOSSX_DIST_SCENARIO", "four")
    monkeypatch.setenv("BOSSX_PARTITION", method)
    lens = [300_000, 120_000, 110_000, 290_000]
    assert partition_contigs(lens, 2) == ([0, 1, 0, 1] if method == "lpt" else [0, 0, 1, 1])
    expected = dist_scenario.oracle_expected(1, 2)
    assert expected[-1]["threshold"] is not None
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(dist_scenario.worker, args=(2, port, str(tmp_path), 1, 2, ret), nprocs=2, join=True)
    dist_scenario.check(ret, expected, 2)


def test_eight_ranks_grch38_geometry_equal_single_process_oracle(tmp_path, monkeypatch):
    """the EIGHT-rank protocol at BASELINE configs[3]'s geometry (dist_scenario: the 27-contig GRCh38 set
    scaled to 8 Mb), run for real — eight processes over gloo, owners from `partition_contigs(..., "lpt")`: non-contiguous, every
    rank three or four contigs, 26 rows of drift crossing ranks, the summaries' all-gather, the MAX and the exact SUM at world
    8.  Thresholds, statistics and every contig's mask on every rank equal the single-process oracle's, update by update."""
    import dist_scenario
    monkeypatch.setenv("BOSSX_DIST_SCENARIO", "grch27")
    monkeypatch.setenv("BOSSX_PARTITION", "lpt")
    lens = dist_scenario.grch27_lengths()
    owners = partition_contigs(lens, 8, "lpt")
    assert len(lens) == 27 and min(lens) >= 100_000 and set(owners) == set(range(8))
    assert owners != sorted(owners)                                       # non-contiguous
    assert sum(1 for a, b in zip(owners, owners[1:]) if a != b) >= 20     # nearly every block boundary separates two ranks
    assert min(owners.count(r) for r in range(8)) >= 2
    expected = dist_scenario.oracle_expected(1, 2)
    assert expected[-1]["threshold"] is not None
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(dist_scenario.worker, args=(8, port, str(tmp_path), 1, 2, ret), nprocs=8, join=True)
    dist_scenario.check(ret, expected, 8)


def test_native_update_reraises_what_staging_ahead_raised():
    """Engine.update holds an exception of `between()` (the next batch staged while the update runs) back so that the
    update can be collected; `_update_native` must re-raise it on every way out — strategy not armed yet, read lengths missing,
    normal completion — after applying the update's results, and count the update."""
    from types import SimpleNamespace
    from boss_runs_amd.parallel import DistributedBossRuns

    class Boom(RuntimeError):
        pass

    def make(res, have_rl):
        r = object.__new__(DistributedBossRuns)
        r.args = SimpleNamespace(optional=SimpleNamespace(bucket_threshold=0))
        r.engine = SimpleNamespace(update_begin=lambda thr: None, update=lambda *a, **k: dict(res), strat_view=lambda i: None)
        r._begun, r._chain_early, r.armed, r.local_filt = False, False, False, {}
        r.rl_dist = SimpleNamespace(time_cost=500, approx_ccl=np.arange(10) * 100 + 900) if have_rl else SimpleNamespace()
        r.read_starts = SimpleNamespace(_engine=None, fhat_compact=lambda: (np.zeros((1, 2)), 1))
        r.gather_masks, r.write_masks = False, False
        r.comm = SimpleNamespace(world=1, force=False, rank=0)
        return r
    base = dict(contig_on=np.zeros(0, bool), any_on=False, threshold=1.0, normaliser=1.0, ubar0=0.0, strat_size=1, n_bins=1)
    for res, have_rl in ((dict(base), True), (dict(base, any_on=True), True)):
        r = make(dict(res, between_error=Boom("staging failed")), have_rl)
        with pytest.raises(Boom):
            r._update_native(between=lambda: None)
        assert r.n_updates == 1 and r.armed == res["any_on"]
    # a failure while the results are APPLIED (here the reference's AttributeError for a missing time_cost) is the
    # primary one — it is not replaced by the held staging error, which travels along as its cause
    r = make(dict(base, any_on=True, between_error=Boom("staging failed")), False)
    with pytest.raises(AttributeError) as ei:
        r._update_native(between=lambda: None)
    assert isinstance(ei.value.__cause__, Boom) and r.n_updates == 1 and r.armed
    r = make(dict(base, any_on=True), True)                  # nothing held back: nothing raised, results applied
    r._update_native()
    assert r.threshold == 1.0 and r.n_updates == 1
    r = make(dict(base, any_on=True, between_error=KeyboardInterrupt()), True)
    with pytest.raises(KeyboardInterrupt):
        r._update_native(between=lambda: None)
