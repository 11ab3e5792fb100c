"""Read-start distribution F-hat — host side, G/2000 x 2 values.

Same model as /root/reference/boss/runs/readstartdist.py:12-152, fed from the per-mapping
summary arrays the native PAF front end returns instead of PafLine objects.  The posterior is
kept COMPACT (one value per 2-kb window and strand); the repeat(20) / pad / trim / rescale of
`_expand_fhat` and the later `adjust_length` are pure index arithmetic that the histogram
kernel applies on the fly (csrc/kernels.hip.inc, threshold_hist_kernel).
"""

import numpy as np
from scipy.special import betaln


def _sum(a):
    """Sum in extended precision, rounded once (what an exact sum gives in all but pathological
    cases; a C loop instead of math.fsum's Python-level iteration)."""
    return float(np.sum(a, dtype=np.longdouble))


class ReadStartDist:
    def __init__(self, contigs, window_size=2000, alpha=1.0, p0=0.1):
        """`contigs`: {name: object with .length}, the non-rejected contigs in merge order."""
        self.alpha, self.p0, self.window_size = alpha, p0, window_size
        # one merged array (merge order) with per-contig views, so that a batch is counted with ONE
        # bincount and merge() is free
        sizes = [int(c.length / window_size) for c in contigs.values()]
        self._offs = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        self._merged = np.zeros((int(self._offs[-1]), 2))
        self.read_starts = {n: self._merged[self._offs[i]:self._offs[i + 1]] for i, n in enumerate(contigs)}
        self._order = {n: i for i, n in enumerate(contigs)}
        self._lookup = (None, None, None)
        self.total_len = int(np.sum([a.shape[0] for a in self.read_starts.values()]))
        self.target_size = int(np.sum([c.length for c in contigs.values()]) // 100)
        self.on_target = 1
        self._engine = None          # attach_engine: the counts are mirrored in HBM, a batch adds its starts there too
        self._csum = 0.0

    def attach_engine(self, engine):
        """Keep the counts resident on the device as well (bossx_fhat_reset / bossx_fhat_add): the
        fused update then rebuilds the posterior there from the O(1) part of the model
        (`fhat_model`), without the O(windows) host pass and upload of `fhat_compact`."""
        self._engine = engine
        self.resync_engine()

    def resync_engine(self):
        """After the host arrays were written directly (checkpoint restore)."""
        self._csum = float(np.sum(self._merged))
        if self._engine is not None:
            self._engine.fhat_reset(self._merged, self._merged.shape[0])

    def fhat_model(self):
        """The scalars of update_f_pointmass (readstartdist.py:95-117) for the device build."""
        n = self._merged.shape[0]
        csum = self._csum
        rhs = (self.alpha / (2 * n * self.alpha + csum))
        beta_num = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha + csum)))
        beta_denom = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha))) or 1e-20
        p0_bit = self.p0 / (self.p0 + (1 - self.p0))
        expected = (1 - p0_bit * (beta_num / beta_denom)) * rhs
        return dict(n_windows=n, target_rs=self.target_size, alpha=self.alpha, den=2 * n * self.alpha + csum,
                    expected=float(expected), on_target=self.on_target)

    def merge(self):
        return self._merged

    def count_starts(self, names, contig_idx, rev, tstart, tend):
        """readstartdist.py:43-82 on arrays: forward reads count at tstart, reverse at tend, in
        2-kb windows with np.histogram's closed last edge and out-of-range drop."""
        if self._lookup[0] is not names:        # contig index (add order) -> window offset / count
            off = np.full(len(names) + 1, -1, dtype=np.int64)
            cnt = np.zeros(len(names) + 1, dtype=np.int64)
            for i, nm in enumerate(names):
                k = self._order.get(nm)
                if k is not None:
                    off[i], cnt[i] = self._offs[k], self._offs[k + 1] - self._offs[k]
            self._lookup = (names, off, cnt)
        _, off, cnt = self._lookup
        ci = np.asarray(contig_idx, dtype=np.int64)
        rev = np.asarray(rev) != 0
        x = np.where(rev, tend, tstart).astype(np.int64)
        ws = self.window_size
        n = cnt[ci]                              # ci == -1 picks the sentinel (count 0)
        # np.histogram(x, bins=n, range=(0, ws * n)) per contig and strand: windows [k*ws, (k+1)*ws),
        # the last one closed on the right, anything outside dropped
        keep = (n > 0) & (x >= 0) & (x <= ws * n)
        if not keep.any():
            return
        w = np.minimum(x[keep] // ws, n[keep] - 1)
        key = (off[ci[keep]] + w) * 2 + rev[keep]
        # O(batch), not O(windows): 1.5 M windows at 3.1 Gb
        np.add.at(self._merged.reshape(-1), key, 1.0)
        self._csum += float(key.size)
        if self._engine is not None:
            self._engine.fhat_add(key)

    def fhat_compact(self):
        """-> (fhat_c float64[n_windows, 2] already multiplied by the on-target normaliser,
        target_size).  readstartdist.py:86-152."""
        merged = self.merge()
        n = merged.shape[0]
        nzmask = merged != 0
        csum = np.sum(merged)                    # counts are integers: the zeros add nothing, any order is exact
        rhs = (self.alpha / (2 * n * self.alpha + csum))
        beta_num = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha + csum)))
        beta_denom = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha))) or 1e-20
        p0_bit = self.p0 / (self.p0 + (1 - self.p0))
        expected = (1 - p0_bit * (beta_num / beta_denom)) * rhs
        # (alpha + C) / (2 N alpha + sum C) where reads started, the point-mass expectation elsewhere
        fhat = np.where(nzmask, np.divide(np.add(self.alpha, merged), 2 * n * self.alpha + csum), expected)
        # sum of the expanded array without building it: 20 copies of every row, plus the
        # re-appended tail (or minus the trimmed tail) of _expand_fhat
        rep = int(self.window_size // 100)
        d = self.target_size - rep * n
        assert d < self.window_size
        total = rep * _sum(fhat)
        if d != 0:
            k = abs(d)
            full, part = divmod(k, rep)            # tail rows of the expanded array
            tail = rep * _sum(fhat[n - full:]) if full else 0.0
            if part:
                tail += part * _sum(fhat[n - full - 1])
            total = total + tail if d > 0 else total - tail
        if total != 0:
            fhat = np.multiply(fhat, self.on_target / total)
        return fhat, self.target_size

    def fhat_expanded_reference_order(self):
        """What ReadStartDist.update_f_pointmass returns in the reference (readstartdist.py:86-152), with ITS arithmetic: the expanded
        array is built and normalised by numpy's pairwise sum of that array — `fhat_compact` (and the device) normalise by an
        extended-precision sum of the compact one, which can differ in the last bit.  Materialises target_size x 2 doubles: used only
        where the last bit may matter (BossRuns._resolve_near_tie)."""
        merged = self.merge()
        n = merged.shape[0]
        nzmask = merged != 0
        csum = np.sum(merged[nzmask])
        rhs = (self.alpha / (2 * n * self.alpha + csum))
        beta_num = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha + csum)))
        beta_denom = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha))) or 1e-20
        p0_bit = self.p0 / (self.p0 + (1 - self.p0))
        expected = (1 - p0_bit * (beta_num / beta_denom)) * rhs
        fhat = np.where(nzmask, np.divide(np.add(self.alpha, merged), 2 * n * self.alpha + csum), expected)
        out = np.repeat(fhat, int(self.window_size // 100), axis=0)
        d = self.target_size - out.shape[0]
        assert d < self.window_size
        if d > 0:
            out = np.append(out, out[-d:], axis=0)
        elif d < 0:
            out = out[:-abs(d)]
        total = np.sum(out)
        if total != 0:
            out = np.multiply(out, self.on_target / total)
        return out

    def expand(self, fhat_c, target):
        """Materialise what the reference would pass to find_strat_thread (tests / small
        genomes only): repeat, pad/trim to target_size, then adjust_length to `target`."""
        rep = int(self.window_size // 100)
        out = np.repeat(fhat_c, rep, axis=0)
        for size in (self.target_size, target):
            d = size - out.shape[0]
            if d > 0:
                out = np.append(out, out[-d:], axis=0)
            elif d < 0:
                out = out[:-abs(d)]
        return out
