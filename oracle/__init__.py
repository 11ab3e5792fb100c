"""CPU oracle for the BOSS-RUNS decision-update path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  It is a numpy/C restatement of
the reference algorithm (goldman-gp-ebi/BOSS-RUNS v0.4.0, `boss/runs/*`), written to be
structurally faithful to the reference (same whole-array passes, same per-read
`np.add.at` loops, same 12-chunk bincounts) so that it can serve as

  * the checker for the HIP path in `tests/` and `__graft_entry__.smoke()`, and
  * the `cpu_baseline` leg ("port") of `bench.py`.

Nothing under `boss-runs_amd/` may import it.  Every function cites the reference
file:line it follows.

Pinning status:
  * pinned against the reference's own data-free known answers
    (tests/base/test_runs_sequences.py:113-126, tests/base/test_readlengthdist.py:21-32,
    tests/base/test_reference.py:10-36) and against golden vectors produced by importing
    the reference in the build container (tests/golden/make_golden.py);
  * `move_sum` is pinned to Bottleneck ITSELF: the library (~=1.3.7, reference pyproject.toml:19) is
    absent from /root/reference and from the system Python, but /opt/conda/bin/python3.9 of the
    build image ships the real compiled Bottleneck 1.3.2.  `tests/golden/make_movesum_golden.py`
    runs it on every `scores_ds` column of the golden runs (at the windows those runs used, both
    directions) and on random arrays over forty decades, and re-derives the golden fixtures'
    `additional_benefit` with it (equal, 52 columns); `oracle/movesum.c` is held to
    `tests/golden/g_movesum.npz` bit for bit (tests/test_oracle_golden.py).
"""
