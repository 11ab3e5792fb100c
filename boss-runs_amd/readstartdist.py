"""Read-start distribution F-hat — host side, G/2000 x 2 values.

Same model as /root/reference/boss/runs/readstartdist.py:12-152, fed from the per-mapping
summary arrays the native PAF front end returns instead of PafLine objects.  The posterior is
kept COMPACT (one value per 2-kb window and strand); the repeat(20) / pad / trim / rescale of
`_expand_fhat` and the later `adjust_length` are pure index arithmetic that the histogram
kernel applies on the fly (csrc/kernels.hip.inc, threshold_hist_kernel).
"""
import math

import numpy as np
from scipy.special import betaln


class ReadStartDist:
    def __init__(self, contigs, window_size=2000, alpha=1.0, p0=0.1):
        """`contigs`: {name: object with .length}, the non-rejected contigs in merge order."""
        self.alpha, self.p0, self.window_size = alpha, p0, window_size
        self.read_starts = {n: np.zeros((int(c.length / window_size), 2)) for n, c in contigs.items()}
        self.total_len = int(np.sum([a.shape[0] for a in self.read_starts.values()]))
        self.target_size = int(np.sum([c.length for c in contigs.values()]) // 100)
        self.on_target = 1

    def merge(self):
        return np.concatenate(list(self.read_starts.values()))

    def count_starts(self, names, contig_idx, rev, tstart, tend):
        """readstartdist.py:43-82 on arrays: forward reads count at tstart, reverse at tend, in
        2-kb windows with np.histogram's closed last edge and out-of-range drop."""
        contig_idx = np.asarray(contig_idx)
        pos = np.where(np.asarray(rev) != 0, tend, tstart)
        for ci in np.unique(contig_idx):
            if ci < 0 or names[ci] not in self.read_starts:
                continue
            arr = self.read_starts[names[ci]]
            n = arr.shape[0]
            sel = contig_idx == ci
            for strand in (0, 1):
                x = pos[sel & ((np.asarray(rev) != 0) == bool(strand))]
                if x.size == 0 or n == 0:
                    continue
                arr[:, strand] += np.histogram(x, bins=n, range=(0, self.window_size * n))[0].astype('float')

    def fhat_compact(self):
        """-> (fhat_c float64[n_windows, 2] already multiplied by the on-target normaliser,
        target_size).  readstartdist.py:86-152."""
        merged = self.merge()
        n = merged.shape[0]
        fhat = np.zeros(merged.shape)
        nzi = np.nonzero(merged)
        nz = merged[nzi]
        csum = np.sum(nz)
        fhat[nzi] = np.divide(np.add(self.alpha, nz), 2 * n * self.alpha + csum)
        rhs = (self.alpha / (2 * n * self.alpha + csum))
        beta_num = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha + csum)))
        beta_denom = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha))) or 1e-20
        p0_bit = self.p0 / (self.p0 + (1 - self.p0))
        expected = (1 - p0_bit * (beta_num / beta_denom)) * rhs
        zero = np.ones(fhat.shape, dtype="bool")
        zero[nzi] = 0
        fhat[zero] = expected
        # sum of the expanded array without building it: 20 copies of every row, plus the
        # re-appended tail (or minus the trimmed tail) of _expand_fhat
        rep = int(self.window_size // 100)
        d = self.target_size - rep * n
        assert d < self.window_size
        total = rep * math.fsum(fhat.reshape(-1))
        if d != 0:
            k = abs(d)
            full, part = divmod(k, rep)            # tail rows of the expanded array
            tail = rep * math.fsum(fhat[n - full:].reshape(-1)) if full else 0.0
            if part:
                tail += part * math.fsum(fhat[n - full - 1])
            total = total + tail if d > 0 else total - tail
        if total != 0:
            fhat = np.multiply(fhat, self.on_target / total)
        return fhat, self.target_size

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
