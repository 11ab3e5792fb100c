"""Executable model (numpy, test infrastructure) of the exact PARALLEL move_sum that
`movesum_scan_kernel` (boss-runs_amd/csrc/kernels.hip.inc) implements — the same steps, stretch by
stretch, with vector operations where the kernel uses scans.  It exists to pin the arithmetic
argument on the CPU: tests compare it bit for bit with the sequential recurrence of
oracle/movesum.c (= bottleneck.move_sum, reference.py:233-234, 259-260).

The argument.  The reference's accumulator obeys s_j = fl(s_{j-1} + d_j), d_j = fl(a_j - a_{j-w}).
Write everything in units of q = 2^(eb-52), the spacing of doubles in binade eb: while the exact sum
z_j = s_{j-1} + d_j stays inside the two binades {eb, eb+1}, s_j/q is an integer n_j in [2^52, 2^54)
and the IEEE rounding is a rounding of the real number n_{j-1} + x_j (x_j = d_j/q, exact) to an
integer (lower binade) or to an even integer (upper binade), ties to even in the result's own
spacing.  With fl = floor(x), fr = x - fl that rounding is  n_j = n_{j-1} + fl + e + c  where
  e = [lower and fr > 1/2]                          (state-independent), and
  c in {-1, 0, +1} depends ONLY on (n_{j-1} + fl) mod 4, the label (lower / upper) and the class of fr.
So the recurrence is (i) an integer prefix sum of fl + e, exact in float64 because every partial sum
stays below 2^53, (ii) a scan of 4-state maps rho -> rho' (n mod 4), which yields every c, and
(iii) a prefix sum of the c.  The labels come from an approximate prefix sum (off by at most the
stretch length) and every one of them is then VERIFIED exactly (x_j against 2^52 / 2^53 / 2^54 minus
n_{j-1}, all exactly representable); a stretch ends before the first element that fails, that one
element is done with a plain floating-point add, and the next stretch starts behind it."""
import numpy as np

M_CUT = 2.0 ** 16
TWO52, TWO53, TWO54, TWO51 = 2.0 ** 52, 2.0 ** 53, 2.0 ** 54, 2.0 ** 51


def _luts():
    maps = np.zeros((5, 4, 4), dtype=np.int64)      # [class][fl mod 4][rho] -> rho'
    cfun = np.zeros((5, 4, 4), dtype=np.int64)      # ... -> c
    for cls in range(5):
        for flm in range(4):
            for rho in range(4):
                t = (rho + flm) & 3
                if cls == 0:
                    c = 0
                elif cls == 1:
                    c = 0
                elif cls == 2:
                    c = t & 1
                elif cls == 3:
                    c = 0 if not (t & 1) else (-1 if t == 1 else 1)
                else:
                    c = t & 1
                inc = flm + (1 if cls == 1 else 0) + c
                maps[cls, flm, rho] = (rho + inc) & 3
                cfun[cls, flm, rho] = c
    return maps, cfun


MAPS, CFUN = _luts()


def move_sum_scan(a, w, N=512, stats=None):
    """move_sum(a, window=w, min_count=1) by stretches of N elements; returns float64 array."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    out = np.empty(n)
    d_all = a.copy()
    d_all[w:] = a[w:] - a[:-w]
    i, s = 0, 0.0
    st = stats if stats is not None else {}
    st.setdefault("stretches", 0); st.setdefault("plain", 0); st.setdefault("violations", 0); st.setdefault("cuts", 0)
    while i < n:
        m, ex = np.frexp(abs(s))              # |s| = m * 2^ex, m in [0.5, 1)
        e0 = int(ex) - 1
        if s == 0.0 or e0 < -900 or e0 > 900:
            # plain steps (the start of a chain, exact zeros)
            s = s + d_all[i]; out[i] = s; i += 1; st["plain"] += 1
            continue
        upper_half = m >= 0.75
        eb = e0 if upper_half else e0 - 1
        sh = 52 - eb
        sgn = -1.0 if s < 0 else 1.0
        n0 = np.ldexp(abs(s), sh)
        rho0 = int(n0 % 4)
        hi = min(n, i + N)
        d = d_all[i:hi]
        x = np.ldexp(d, sh) * sgn
        cnt = hi - i
        cut = np.zeros(cnt, dtype=bool)
        cut |= (d != 0) & (np.abs(x) < 2.0 ** -900)
        cut |= ~(np.abs(x) < TWO52)
        xs = np.where(np.isfinite(x), x, 0.0)
        fl = np.floor(xs); fr = xs - fl
        F = np.cumsum(fl)                     # exact while |F| < 2^52: every partial sum of a scan is then below 2^53 (checked below)
        Fprev = np.concatenate([[0.0], F[:-1]])
        zap = n0 + Fprev + xs
        cut |= ~((zap > TWO52 + M_CUT) & (zap < TWO54 - M_CUT)) | (np.abs(zap - TWO53) < M_CUT) | ~(np.abs(F) < TWO52)
        up = zap >= TWO53
        cls = np.where(up, np.where(fr == 0.0, 3, 4), np.where(fr == 0.5, 2, np.where(fr > 0.5, 1, 0)))
        flm = (fl - 4.0 * np.floor(fl * 0.25)).astype(np.int64)
        # (ii) map scan — sequential here, a composition scan on the device
        rho = rho0
        c = np.zeros(cnt, dtype=np.int64)
        for g in range(cnt):
            c[g] = CFUN[cls[g], flm[g], rho]
            rho = MAPS[cls[g], flm[g], rho]
        K = np.cumsum((cls == 1).astype(np.int64) + c)
        nn = n0 + (F + K.astype(np.float64))
        nprev = np.concatenate([[n0], nn[:-1]])
        ok = np.where(up, (xs >= TWO53 - nprev) & (xs < TWO54 - nprev), (xs >= TWO52 - nprev) & (xs < TWO53 - nprev))
        bad = cut | ~ok
        p = int(np.argmax(bad)) if bad.any() else cnt
        if bad.any():
            st["cuts" if cut[p] else "violations"] += 1
        st["stretches"] += 1
        out[i:i + p] = sgn * np.ldexp(nn[:p], -sh)
        if p > 0:
            s = out[i + p - 1]
        if p < cnt:
            s = s + d[p]; out[i + p] = s; st["plain"] += 1
            i += p + 1
        else:
            i += p
    return out


def move_sum_serial(a, w):
    a = np.asarray(a, dtype=np.float64)
    out = np.empty(a.shape[0])
    s = 0.0
    for i in range(a.shape[0]):
        s = s + (a[i] - a[i - w] if i >= w else a[i])
        out[i] = s
    return out
