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
