#!/opt/conda/bin/python3.9
"""Pins `move_sum` to Bottleneck ITSELF (runs ONLY in the build container).

The reference calls `bottleneck.move_sum(a, window, min_count=1)` at boss/runs/reference.py:233-234
(calc_smu) and :259-260 (calc_u); pyproject.toml:19 pins `bottleneck~=1.3.7`.  The system Python of this
image has no Bottleneck (so `make_golden.py` imports the reference through `_shims/bottleneck.py`, a
restatement), but /opt/conda/bin/python3.9 ships the real compiled library (Bottleneck 1.3.2, numpy 1.26):

    cd /tmp && /opt/conda/bin/python3.9 /root/repo/tests/golden/make_movesum_golden.py

writes `g_movesum.npz` next to this script: for every `scores_ds` column of the end-to-end and saturated
fixtures (the arrays the reference really handed to Bottleneck in those runs) at the windows those runs
used, and for seeded random arrays over forty decades with runs of `tiny` and of zeros, the SHA-256 of
`bn.move_sum(a, w, min_count=1)` forward and on the reversed view (full outputs for the short arrays), the
`additional_benefit` of calc_smu + calc_u formed from Bottleneck's sums (reference.py:215-269: the value the
HIP chain must reproduce), and Bottleneck's behaviour at the window edges (w = 0, w > n: ValueError).

Consumers: tests/test_oracle_golden.py (oracle/movesum.c and the import shim, bit for bit, CPU tier) and
tests/test_parity_gpu.py::test_benefit_chain_equals_bottleneck (bossx_benefit on imported bin sums, GPU
tier).  The e2e fixtures' own `additional_benefit` columns are re-derived here with the real library and
asserted equal, so everything `make_golden.py` produced through the shim is Bottleneck's result.
"""
import glob
import hashlib
import json
import os
import sys

import numpy as np
import bottleneck as bn

HERE = os.path.dirname(os.path.abspath(__file__))
MU_W = 4                                   # mu // window = 400 // 100 (reference.py:214, 233)
DEFAULT_CCL = np.array([1167, 2729, 3903, 4918, 5866, 6808, 7797, 8912, 10321, 12713])      # readlengthdist default (tests/base/test_readlengthdist.py:21-32)
MULT = np.arange(0.05, 1, 0.1)[::-1]       # reference.py:253


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()


def benefit_from_bottleneck(ds, windows):
    """Contig.calc_smu + Contig.calc_u for one barcode column (reference.py:215-269) on the real library."""
    n = ds.shape[0]
    smu = np.zeros((n, 2))
    smu[:, 0] = bn.move_sum(ds[::-1], window=MU_W, min_count=1)[::-1]
    smu[:, 1] = bn.move_sum(ds, window=MU_W, min_count=1)
    tmp = np.zeros((n, 2))
    for i in range(10):
        fwd = bn.move_sum(ds[::-1], window=int(windows[i]), min_count=1)[::-1]
        rev = bn.move_sum(ds, window=int(windows[i]), min_count=1)
        tmp[:, 0] += fwd * MULT[i]
        tmp[:, 1] += rev * MULT[i]
    ab = tmp - smu
    ab[ab < 0] = 0
    return ab


def random_arrays():
    """Seeded arrays shaped like bin sums of every regime (each long enough to be a contig: >= 1001 bins)."""
    tiny = np.finfo(float).tiny
    out = []
    for s in range(24):
        rng = np.random.default_rng(7000 + s)
        n = int(rng.integers(1001, 2400))
        kind = s % 6
        if kind == 0:                                   # ordinary scores: 100 sites x 0.01 .. 0.2
            a = rng.uniform(0.5, 20.0, n)
        elif kind == 1:                                 # forty decades
            a = 10.0 ** rng.uniform(-20, 20, n)
        elif kind == 2:                                 # saturated: sums of `tiny` with islands of real scores
            a = np.full(n, 100 * tiny)
            for _ in range(6):
                p = int(rng.integers(0, n - 40)); a[p:p + int(rng.integers(1, 40))] = 10.0 ** rng.uniform(-44, -18)
        elif kind == 3:                                 # runs of zeros (dropout rows) between ordinary stretches
            a = rng.uniform(0.5, 20.0, n)
            for _ in range(10):
                p = int(rng.integers(0, n - 60)); a[p:p + int(rng.integers(1, 60))] = 0.0
        elif kind == 4:                                 # climbs and drops of dozens of binades, `tiny` runs in between
            a = 10.0 ** np.cumsum(rng.normal(0, 1.5, n)).clip(-300, 300)
            for _ in range(5):
                p = int(rng.integers(0, n - 30)); a[p:p + int(rng.integers(1, 30))] = tiny
        else:                                           # near-capped sites: 1e-20 .. 1e-40 next to ordinary ones
            a = np.where(rng.random(n) < 0.5, 10.0 ** rng.uniform(-40, -20, n), rng.uniform(0.5, 20.0, n))
        out.append(("rand%02d" % s, np.ascontiguousarray(a, dtype=np.float64)))
    return out


def main():
    out = {}
    cases = []                 # (name, source, n, windows)
    n_checked_vs_fixture = 0

    def add_case(name, a, windows, source, fixture_benefit=None):
        nonlocal n_checked_vs_fixture
        wins = [MU_W] + [int(w) for w in windows]
        rec = dict(name=name, source=source, n=int(a.shape[0]), windows=wins, fwd={}, rev={})
        for w in sorted(set(wins)):
            rec["rev"][str(w)] = sha(bn.move_sum(a, window=w, min_count=1))                   # as reference.py:234,260
            rec["fwd"][str(w)] = sha(bn.move_sum(a[::-1], window=w, min_count=1)[::-1])       # as reference.py:233,259
        ab = benefit_from_bottleneck(a, wins[1:])
        rec["benefit_sha"] = sha(ab)
        if fixture_benefit is not None:
            assert np.array_equal(ab, fixture_benefit), "fixture additional_benefit is not Bottleneck's: " + name
            n_checked_vs_fixture += 1
        cases.append(rec)
        return ab

    # (1) what the reference handed to Bottleneck in the golden runs
    for f in sorted(glob.glob(os.path.join(HERE, "g_e2e_*.npz")) + glob.glob(os.path.join(HERE, "g_sat_*.npz"))):
        d = np.load(f)
        fname = os.path.basename(f)
        for k in sorted(d.files):
            if not k.endswith("_scores_ds"):
                continue
            tag = k.split("_")[0]                                     # b1 .. b4 / sat / sat1
            ccl = d[tag + "_approx_ccl"]
            ds = d[k]
            ab_fix = d[k[:-len("scores_ds")] + "additional_benefit"]
            for b in range(ds.shape[1]):
                add_case("%s:%s:%d" % (fname, k, b), np.ascontiguousarray(ds[:, b]), ccl // 100,
                         "fixture", fixture_benefit=ab_fix[:, :, b])

    # (2) seeded arrays over every regime; inputs and (for four of them) full outputs are stored
    for name, a in random_arrays():
        rng = np.random.default_rng(int(name[4:]) + 99)
        wins = np.sort(rng.integers(5, min(300, a.shape[0] - 1), size=10))
        if name in ("rand00", "rand01"):
            wins = DEFAULT_CCL // 100
        ab = add_case(name, a, wins, "inline")
        out["in_" + name] = a
        if name in ("rand01", "rand02", "rand04", "rand05"):
            out["benefit_" + name] = ab
            out["rev_w%d_%s" % (int(wins[3]), name)] = bn.move_sum(a, window=int(wins[3]), min_count=1)
            out["fwd_w%d_%s" % (int(wins[3]), name)] = bn.move_sum(a[::-1], window=int(wins[3]), min_count=1)[::-1]

    # (3) short arrays with full outputs at every window 1..n (the min_count=1 head, w = n)
    short = []
    for s in range(6):
        rng = np.random.default_rng(8000 + s)
        n = int(rng.integers(5, 40))
        a = 10.0 ** rng.uniform(-12, 12, n) if s % 2 else rng.uniform(0, 10, n)
        full = np.stack([bn.move_sum(a, window=w, min_count=1) for w in range(1, n + 1)])
        full_r = np.stack([bn.move_sum(a[::-1], window=w, min_count=1)[::-1] for w in range(1, n + 1)])
        out["short%d_in" % s] = a
        out["short%d_rev" % s] = full
        out["short%d_fwd" % s] = full_r
        short.append(n)

    # (4) the window edges: what Bottleneck raises
    edges = {}
    a = np.arange(10, dtype=np.float64)
    for w in (0, -1, 11, 10, 1):
        try:
            bn.move_sum(a, window=w, min_count=1)
            edges[str(w)] = "ok"
        except Exception as e:      # noqa: BLE001
            edges[str(w)] = type(e).__name__
    out["meta"] = np.array(json.dumps(dict(bottleneck=bn.__version__, numpy=np.__version__, python=sys.version.split()[0],
                                           n_fixture_columns_equal_to_bottleneck=n_checked_vs_fixture,
                                           edges_n10=edges, short_n=short, cases=cases)))
    np.savez_compressed(os.path.join(HERE, "g_movesum.npz"), **out)
    print("bottleneck", bn.__version__, "numpy", np.__version__, ":", len(cases), "cases,", n_checked_vs_fixture,
          "fixture columns re-derived with the real library and equal; edges", edges)


if __name__ == "__main__":
    main()
