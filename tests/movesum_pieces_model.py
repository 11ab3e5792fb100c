"""Executable model (numpy) of how the chunk-parallel chain serves a chunk — `chain_candidates_kernel` /
`candidates_cut` and `chain_stitch_kernel` / `stitch_slow_row` of boss-runs_amd/csrc/kernels.hip.inc:

  * an ORDINARY row: K = 4 * 2^rise candidates around the approximate start.  A table follows the sum up to 2^rise x its
    start and DOWN as far as it goes — with one exception: while the sum is below 2^-8 of the start the candidates sit on
    their rounding residues, sixteen ulps of the start apart and no longer close to each other relatively; an add that
    does not vanish against such a residue (|d| >= 2^-105 x the start) rounds differently for each of them.  The table ends
    in front of such a bin (a stretch of fully capped bins — differences of 0 or a few times 1e-308 — is followed exactly);
  * a CUT row: pieces — CAND (sixteen candidates, the same rule; the sum below 4 x the piece's start), ONE (a single bin
    added the plain way), RUN (a stretch where the approximation is far below the chunk's scale);
  * whatever the tables do not serve is evaluated from the exact value itself: while every difference of a stretch is small
    against it, the recurrence rounds on ONE grid (the spacing of its binade) and is an integer sum — `band_eval`, a
    parallel reduction on the device — and only what is left is added the plain way.

Pinned on the CPU against the sequential recurrence by tests/test_movesum_parallel_models.py."""
import numpy as np

T_REL = 2.0 ** -24          # "far below the chunk's scale"
DIP = 2.0 ** -8             # below this fraction of a table's start its candidates sit on their rounding residues ...
VANISH = 2.0 ** -105        # ... and only adds below this fraction of the start leave every one of them as it is
LOOKUP = 1 << 30            # |exact start - table base| in ulps the tables are used for at most (see table_reach)


def _bits(x):
    return int(np.float64(x).view(np.int64))


def table_reach(v, ref, u, vmax):
    """How many ulps of the table's start the exact value may lie from the table's base.  The exact trajectory and its
    candidate stay (t - r) ulps apart as long as they round on the same grid, i.e. lie in the same binade at every step:
    half the smallest distance of the approximate sums to a binade edge (those on the residues, below 2^-8 of the start,
    aside: only vanishing adds happen there), less what the approximation itself may be off."""
    a = np.abs(np.asarray(v, dtype=np.float64))
    a = a[a >= DIP * abs(ref)]
    if a.size == 0:
        return float(LOOKUP)
    m, e = np.frexp(a)                                   # a = m 2^e, m in [0.5, 1)
    dist = np.minimum(m - 0.5, 1.0 - m) * np.ldexp(1.0, e)
    return float(0.5 * np.min(dist) / u - 4096.0 * (vmax / abs(ref)))


def stitch_step(s, base, u, K, ends, reach=float(LOOKUP)):
    sb, bb = _bits(s), _bits(base)
    t = sb - bb
    if ((sb ^ bb) >> 52) != 0 or not (abs(t) < min(float(LOOKUP), reach)):
        return None
    r = t & (K - 1)
    return np.float64(ends[r]) + np.float64(t - r) * np.float64(u)


def run_plain(s, d):
    s = np.float64(s)
    for x in d:
        s = s + np.float64(x)
    return s


def band_eval(s, d):
    """s + d_0 + d_1 + ... as the recurrence rounds it, WITHOUT walking it: valid while s stays inside its binade —
    every add then rounds to the one spacing q of that binade, s = n q, and n + rint(d / q) is what IEEE gives unless
    d / q is exactly half-way (ties look at n's parity).  None where that cannot be guaranteed."""
    s = np.float64(s)
    d = np.asarray(d, dtype=np.float64)
    if d.size == 0:
        return s
    ex = (_bits(abs(s)) >> 52) & 0x7ff
    if ex < 120 or ex > 1950:
        return None
    q = np.ldexp(1.0, ex - 1023 - 52)
    x = d / q if s > 0 else -d / q                    # (a power of two: exact)
    if not np.all(np.abs(x) < 2.0 ** 51):
        return None
    r = np.rint(x)
    if np.any(np.abs(x - np.trunc(x)) == 0.5):
        return None
    n0 = abs(s) / q                                   # integer in [2^52, 2^53)
    room = min(n0 - 2.0 ** 52, 2.0 ** 53 - 1 - n0)
    if not (np.sum(np.abs(r)) < room):
        return None
    n1 = n0 + np.sum(r)
    return np.float64(n1 * q if s > 0 else -(n1 * q))


def eval_range(s, d, stats=None):
    """A stretch no table serves, from the exact value in front of it."""
    v = band_eval(s, d)
    if v is not None:
        if stats is not None: stats["band"] = stats.get("band", 0) + 1
        return v
    if stats is not None: stats["plain"] = stats.get("plain", 0) + 1
    return run_plain(s, d)


def ordinary_row(d, A):
    """(K, base, u, ends) of an ordinary row, or None where the chunk needs pieces."""
    d = np.asarray(d, dtype=np.float64)
    v = np.float64(A) + np.cumsum(d)
    top = max(float(np.max(np.abs(v))), abs(float(A))) * (1 + 2.0 ** -20)
    if not (A > 0 and 2.0 ** -900 < A < 2.0 ** 900):
        return None
    rise = int(np.frexp(top)[1] - np.frexp(A)[1])
    if rise > 4 or np.any(_bad_bins(d, v, A, A)):
        return None
    K = max(16, 4 << max(rise, 0))
    u = np.spacing(np.float64(A))
    base = np.floor(A / (u * K)) * (u * K)
    return K, base, u, [run_plain(base + r * u, d) for r in range(K)], table_reach(v, A, u, top)


def _bad_bins(d, v, A, ref):
    """bins whose add happens on the residues (the sum below DIP x ref before and behind it) without vanishing"""
    vp = np.concatenate([[np.float64(A)], v[:-1]])
    lo = DIP * abs(ref)
    return (np.abs(v) < lo) & (np.abs(vp) < lo) & (np.abs(d) >= VANISH * abs(ref))


def cut_pieces(d, A, max_payload=124):
    """The pieces of one chunk with differences `d` and approximate start `A`; None where the kernel gives up."""
    d = np.asarray(d, dtype=np.float64)
    n = d.shape[0]
    v = np.float64(A) + np.cumsum(d)
    vmax = max(float(np.max(np.abs(v))), abs(float(A)))
    T = vmax * T_REL
    pieces, pos, ref, used = [], 0, np.float64(A), 0

    def first(pos, pred):
        hit = np.nonzero(pred(np.abs(v[pos:])))[0]
        return pos + int(hit[0]) if hit.size else n

    while pos < n:
        if not (abs(ref) >= T) or not (2.0 ** -900 < ref < 2.0 ** 900):
            e1 = first(pos, lambda x: x >= T)
            if e1 > pos:
                used += 3
                pieces.append(("run", pos, e1 - pos))
            if e1 < n:
                used += 2
                pieces.append(("one", float(d[e1])))
                ref = v[e1]
            pos = e1 + 1 if e1 < n else n
        else:
            hi_lim = 4.0 * ref * (1.0 - 2.0 ** -20)
            bad = _bad_bins(d, v, A, ref)
            hit = np.nonzero((np.abs(v[pos:]) >= hi_lim) | bad[pos:])[0]
            e2 = pos + int(hit[0]) if hit.size else n
            if e2 > pos:
                used += 22
                u = np.spacing(ref)
                base = np.floor(ref / (u * 16)) * (u * 16)
                pieces.append(("cand", pos, e2 - pos, float(base), float(u), [run_plain(base + r * u, d[pos:e2]) for r in range(16)],
                               table_reach(v[pos:e2], ref, u, vmax)))
            if e2 < n:
                refn = v[e2 - 1] if e2 > 0 else np.float64(A)
                ve = v[e2]
                if e2 == pos or not (abs(ve) < 4.0 * abs(refn) * (1.0 - 2.0 ** -20)) or not (abs(refn) >= T):
                    # bin e2 on its own leaves the band of its predecessor: a plain add
                    used += 2
                    pieces.append(("one", float(d[e2])))
                    ref, pos = ve, e2 + 1
                else:
                    ref, pos = refn, e2
            else:
                pos = n
        if used > max_payload:
            return None
    return pieces


def stitch_pieces(s, pieces, d, stats=None):
    """The exact value behind the chunk from the exact value `s` in front of it."""
    v = np.float64(s)
    for p in pieces:
        if p[0] == "one":
            v = v + np.float64(p[1])
        elif p[0] == "run":
            v = eval_range(v, d[p[1]:p[1] + p[2]], stats)
        else:
            t = stitch_step(v, p[3], p[4], 16, p[5], p[6])
            if t is None:
                t = eval_range(v, d[p[1]:p[1] + p[2]], stats)
            elif stats is not None:
                stats["table"] = stats.get("table", 0) + 1
            v = t
    return v


def stitch_chunk(s, d, A, stats=None):
    """One chunk as the stitch walks it: ordinary row, cut row, or no table at all."""
    if not np.any(d != 0):
        return np.float64(s)
    row = ordinary_row(d, A)
    if row is not None:
        K, base, u, ends, reach = row
        t = stitch_step(s, base, u, K, ends, reach)
        if t is not None:
            if stats is not None: stats["table"] = stats.get("table", 0) + 1
            return t
        return eval_range(s, d, stats)
    pieces = cut_pieces(d, A)
    if pieces is None:
        return eval_range(s, d, stats)
    return stitch_pieces(s, pieces, d, stats)
