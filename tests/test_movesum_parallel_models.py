"""CPU pins of the arithmetic behind the two parallel forms of bottleneck.move_sum built in round 3
 — against the sequential recurrence of oracle/movesum.c (= reference.py:233-234, 259-260).

1. `movesum_scan_model.move_sum_scan`: the exact scan inside two-binade bands (the executable model of
   `movesum_scan_kernel`): integer prefix sums + a scan of 4-state residue maps, every label verified.
2. The candidate prediction of the chunk-parallel chain (`chain_candidates_kernel` / `chain_stitch_kernel`): two
   start values a multiple of K ulps apart, K = 4 * 2^rise, go through the same roundings and arrive exactly that
   far apart — so a table over the K residues predicts a chunk's end value from its exact start value."""
import ctypes
import os

import numpy as np
import pytest

from movesum_scan_model import move_sum_scan, move_sum_serial

HERE = os.path.dirname(os.path.abspath(__file__))


def _oracle_move_sum(a, w):
    lib = ctypes.CDLL(os.path.join(HERE, "..", "oracle", "_build", "liboracle_movesum.so"))
    lib.oracle_move_sum.argtypes = [ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_long, ctypes.c_long, ctypes.c_void_p]
    a = np.ascontiguousarray(a, dtype=np.float64)
    y = np.empty_like(a)
    assert lib.oracle_move_sum(a.ctypes.data, 1, a.shape[0], w, y.ctypes.data) == 0
    return y


def _arrays():
    rng = np.random.default_rng(3)
    yield "gamma", rng.gamma(2.0, 3.0, size=6000)
    yield "hover", np.abs(rng.normal(8.0, 0.8, size=6000))            # window 64: sums next to 512
    yield "ties", rng.integers(0, 1 << 10, size=6000) / 16.0           # few mantissa bits: ties everywhere
    z = rng.gamma(2.0, 3.0, size=6000); z[1500:3500] = 0.0
    yield "zero run", z
    c = rng.gamma(2.0, 0.05, size=6000); c[2000:2600] = 1e-8; c[4000:4100] = 1e-8      # capped stretches: 10^7 apart
    yield "capped", c
    yield "signed", rng.normal(0.0, 1.0, size=4000)


@pytest.mark.parametrize("w", [1, 4, 20, 64, 141])
def test_band_scan_model_equals_the_sequential_recurrence(w):
    for name, a in _arrays():
        want = _oracle_move_sum(a, w)
        assert np.array_equal(move_sum_serial(a, w), want), name          # the python loop is the C recurrence
        st = {}
        got = move_sum_scan(a, w, N=512, stats=st)
        assert np.array_equal(got, want), (name, w, st)
    # where the running sum stays in one band the stretches are long (the model is not plain adds in disguise)
    st = {}
    a = np.abs(np.random.default_rng(0).normal(8.0, 0.8, size=20000))
    move_sum_scan(a, 64, N=512, stats=st)
    assert st["plain"] < 40 and st["stretches"] < 80, st


def _predict_chunk_ends(a, w, L, kmul=4):
    """The stitch of DESIGN §4.3.1 in numpy: per chunk K = kmul * 2^rise candidates around the approximate start,
    run through the chunk the plain way; the exact start picks its residue.  Returns (#chunks, #wrong, mean K)."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    d = a.copy(); d[w:] = a[w:] - a[:-w]
    true = move_sum_serial(a, w)
    cs = np.concatenate([[0.0], np.cumsum(a)])
    wrong, ks, chunks = 0, [], 0
    for lo in range(L, n, L):
        hi = min(n, lo + L)
        start = true[lo - 1]
        A = cs[lo] - cs[max(lo - w, 0)]                           # approximate value before the chunk
        seg = cs[lo + 1:hi + 1] - cs[np.maximum(np.arange(lo + 1, hi + 1) - w, 0)]
        top = max(float(np.max(np.abs(seg))), abs(A)) * (1 + 2.0 ** -20)
        if not (A > 0 and np.isfinite(A)) or np.frexp(A)[1] != np.frexp(start)[1]:
            continue                                              # (the kernels add such a chunk the plain way)
        rise = max(0, int(np.frexp(top)[1] - np.frexp(A)[1]))
        if rise > 4:
            continue                                              # (cut into pieces / plain in the kernels)
        K = max(16, kmul << rise) if kmul == 4 else max(1, kmul << rise)
        u = np.spacing(A)
        base = np.floor(A / (u * K)) * (u * K)
        ends = base + np.arange(K) * u
        for j in range(lo, hi):
            ends = ends + d[j]                                    # every candidate: plain IEEE adds
        t = (start - base) / u
        if t != np.floor(t):
            continue
        r = int(t % K)
        pred = ends[r] + (start - (base + r * u))
        chunks += 1
        ks.append(K)
        wrong += int(pred != true[hi - 1])
    return chunks, wrong, float(np.mean(ks)) if ks else 0.0


@pytest.mark.parametrize("w", [4, 20, 141])
def test_candidate_tables_predict_chunk_ends_exactly(w):
    rng = np.random.default_rng(11)
    # bin sums like the bench's: relative s.d. ~0.7 with slow trends, plus a capped stretch
    trend = np.exp(np.cumsum(rng.normal(0, 0.02, size=40000)))
    a = rng.gamma(2.2, 0.07, size=40000) * trend / trend.mean()
    a[9000:9800] = 1e-8
    chunks, wrong, kmean = _predict_chunk_ends(a, w, 1024)
    assert chunks >= 25 and wrong == 0, (chunks, wrong, kmean)
    # the parity of the start matters: one candidate per 2^rise is NOT enough (this is what the residues are for)
    chunks1, wrong1, _ = _predict_chunk_ends(a, w, 1024, kmul=1)
    assert wrong1 > chunks1 // 8, (chunks1, wrong1)


def _predict_strided(a, w, L):
    """Strided rows (csrc/kernels.hip.inc: kStridedK) in numpy: a chunk whose sum climbs more than four binades keeps 64 of its
    4 * 2^rise residues, 2^(rise - 4) ulps apart; the exact start, reduced modulo 4 * 2^rise ulps, lies between two of them and
    takes their common end — exact because the recurrence is monotone in its start.  Returns (#strided chunks, #decided, #wrong)."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    d = a.copy(); d[w:] = a[w:] - a[:-w]
    true = move_sum_serial(a, w)
    cs = np.concatenate([[0.0], np.cumsum(a)])
    strided = decided = wrong = 0
    for lo in range(L, n, L):
        hi = min(n, lo + L)
        start = true[lo - 1]
        A = cs[lo] - cs[max(lo - w, 0)]
        seg = cs[lo + 1:hi + 1] - cs[np.maximum(np.arange(lo + 1, hi + 1) - w, 0)]
        top = max(float(np.max(np.abs(seg))), abs(A)) * (1 + 2.0 ** -20)
        if not (A > 0 and np.isfinite(A)) or np.frexp(A)[1] != np.frexp(start)[1]:
            continue
        rise = int(np.frexp(top)[1] - np.frexp(A)[1])
        if rise <= 4 or rise > 28:
            continue
        strided += 1
        u = np.spacing(A)
        Kfull, stride = 4 << rise, 1 << (rise - 4)
        base = np.floor(A / (u * Kfull)) * (u * Kfull)
        ends = base + np.arange(64) * (stride * u)
        for j in range(lo, hi):
            ends = ends + d[j]
        assert np.all(np.diff(ends) >= 0)                         # monotone in the start, whatever the magnitudes
        t = int(round((start - base) / u))
        tm = t % Kfull
        r, rem = tm // stride, tm % stride
        if rem == 0 or (r < 63 and ends[r] == ends[r + 1]):
            decided += 1
            wrong += int(ends[r] + (t - tm) * u != true[hi - 1])
    return strided, decided, wrong


@pytest.mark.parametrize("w", [11, 39, 127])
def test_strided_rows_decide_exactly_or_not_at_all(w):
    """Bins of deep sites, about one per window holding a shallow one (a run from ~12x coverage on): most chunks climb 10-25 binades.
    A decided strided row must give the sequential recurrence's value bit for bit; most rows must be decided (the 64 ends take a
    handful of values: an exact start falls between two different ones a few times in a hundred)."""
    rng = np.random.default_rng(77 + w)
    n = 30 * 1024
    a = rng.uniform(0.5e-9, 2e-9, n) * 10.0 ** np.cumsum(rng.normal(0, 0.01, n)).clip(-2, 2)
    sp = rng.random(n) < 1.0 / w                                  # (about one shallow bin per window: the sum falls back between them)
    a[sp] = 10.0 ** rng.uniform(-5, -2, int(sp.sum()))
    strided, decided, wrong = _predict_strided(a, w, 1024)
    assert strided >= 10 and wrong == 0 and decided >= 0.75 * strided, (strided, decided, wrong)


def _saturated_bins(rng, n):
    """Bin sums of a genome part of which is capped: ordinary bins (a gamma around 0.6) with long stretches of 100 x tiny
    (fully capped), stretches with deep sites only (1e-44 .. 1e-36), and a few bins with a handful of uncapped sites."""
    a = rng.gamma(2.0, 0.3, size=n)
    i = int(rng.integers(200, 3000))
    while i < n:
        kind = rng.integers(0, 4)
        ln = int(rng.integers(150, 2500))
        if kind <= 1:
            a[i:i + ln] = 2.2250738585072014e-306
        elif kind == 2:
            a[i:i + ln] = rng.uniform(1e-44, 1e-36, size=len(a[i:i + ln]))
        else:
            a[i:i + ln] = 10.0 ** rng.uniform(-3, 0, size=len(a[i:i + ln]))
        i += ln + int(rng.integers(1500, 6000))
    return a


@pytest.mark.parametrize("w", [4, 11, 27, 127])
def test_chunk_values_from_tables_pieces_and_bands_are_exact(w):
    """movesum_pieces_model (= chain_candidates_kernel / candidates_cut / chain_stitch_kernel / stitch_eval_range): the value
    behind every chunk — from an ordinary table, from RUN / ONE / CAND pieces, from the one-grid integer sum of a stretch
    the exact value dominates, or from plain adds — is the sequential recurrence's, bit for bit, over window sums that
    drop from 1 to 1e-306 and to the rounding residue of what went before, and climb back; and the tables and the
    one-grid sums (not plain adds) carry most of the chunks."""
    from movesum_pieces_model import stitch_chunk
    rng = np.random.default_rng(100 + w)
    a = _saturated_bins(rng, 60000)
    d = a.copy(); d[w:] = a[w:] - a[:-w]
    true = move_sum_serial(a, w)
    cs = np.concatenate([[0.0], np.cumsum(a)])
    L = 1024
    st = {}
    chunks = 0
    for lo in range(L, len(a), L):
        hi = min(len(a), lo + L)
        A = cs[lo] - cs[max(lo - w, 0)]
        pred = stitch_chunk(true[lo - 1], d[lo:hi], A, st)
        assert pred == true[hi - 1], (w, lo, st)
        chunks += 1
    assert chunks >= 50 and st.get("table", 0) > 10 and st.get("band", 0) > 0, st


def test_band_eval_is_the_recurrence_where_the_start_dominates():
    """One grid: a value of 1e-13 (the residue of earlier adds) through a thousand differences of 1e-20 .. 1e-30."""
    from movesum_pieces_model import band_eval, run_plain
    rng = np.random.default_rng(5)
    hits = 0
    for trial in range(300):
        s = np.float64(rng.uniform(-1, 1) * 10.0 ** rng.uniform(-16, -10))
        d = rng.uniform(-1, 1, size=1000) * 10.0 ** rng.uniform(-45, -18, size=1000)
        got = band_eval(s, d)
        if got is None:
            continue
        hits += 1
        assert got == run_plain(s, d), trial
    assert hits > 150, hits
