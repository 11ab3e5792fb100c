"""Read-length and read-start distributions.

Follows /root/reference/boss/readlengthdist.py:8-97 and
/root/reference/boss/runs/readstartdist.py:12-152.
"""
import numpy as np
from scipy.special import betaln

from .pafcigar import best_mapper


class OReadlengthDist:
    def __init__(self, mu=400, sd=4000, lam=6000, eta=11):
        """readlengthdist.py:10-32: truncated-normal prior."""
        self.sd, self.lam, self.eta, self.mu = sd, lam, eta, mu
        self.read_lengths = np.zeros(int(1e6), dtype='uint16')
        longest = int(lam + 10 * sd)
        x = np.arange(longest, dtype='int')
        L = np.exp(-((x - lam + 1) ** 2) / (2 * (sd ** 2))) / (sd * np.sqrt(2 * np.pi))
        L /= sum(L)
        self.L = L
        self.approx_ccl = self.ccl_approx_constant()

    def update(self, read_lengths):
        """readlengthdist.py:36-68. `read_lengths`: {read id: length}."""
        for _, length in read_lengths.items():
            if length > self.mu * 2:
                if length >= 1e6:
                    length = int(1e6) - 1
                self.read_lengths[length] += 1
        obs = np.nonzero(self.read_lengths)
        if len(obs[0]) == 0:
            return
        length_sum = np.sum(obs * self.read_lengths[obs])
        self.lam = length_sum / np.sum(self.read_lengths[obs])
        self.longest_read = np.max(np.where(self.read_lengths))
        self.L = np.copy(self.read_lengths[:self.longest_read + 1]).astype('float64')
        self.L /= sum(self.L)
        self.approx_ccl = self.ccl_approx_constant()
        self.time_cost = self.lam - 400 - 300

    def ccl_approx_constant(self):
        """readlengthdist.py:72-97."""
        ccl = np.zeros(len(self.L) + 1)
        ccl[0] = 1
        ccl[1:] = 1 - np.concatenate((self.L[1:].cumsum(), np.ones(1)))
        ccl[ccl < 1e-6] = 0
        ccl = np.concatenate((np.trim_zeros(ccl, trim='b'), np.zeros(1)))
        self.ccl = ccl
        approx = np.zeros(self.eta - 1, dtype='int32')
        i = 0
        for part in range(self.eta - 1):
            prob = 1 - (part + 0.5) / (self.eta - 1)
            while (ccl[i] > prob) and (len(ccl) > i):
                i += 1
            approx[part] = i
        return approx


class OReadStartDist:
    def __init__(self, contigs, window_size=2000, alpha=1.0, p0=0.1):
        """readstartdist.py:14-32. `contigs`: {name: object with .length} (non-rejected)."""
        self.alpha, self.p0, self.window_size = alpha, p0, window_size
        self.read_starts = {n: np.zeros((int(c.length / window_size), 2)) for n, c in contigs.items()}
        self.target_size = int(np.sum([c.length for c in contigs.values()]) // 100)
        self.on_target = 1
        self.fhat = self.update_f_pointmass()

    def merge(self):
        return np.concatenate(list(self.read_starts.values()))

    def count_read_starts(self, paf_dict):
        """readstartdist.py:43-82."""
        fwd, rev = {}, {}
        for rid in paf_dict.keys():
            recs = paf_dict[rid]
            rec = best_mapper(recs) if len(recs) > 1 else recs[0]
            if rec.rev:
                rev.setdefault(rec.tname, []).append(rec.tend)
            else:
                fwd.setdefault(rec.tname, []).append(rec.tstart)
        for cname, arr in self.read_starts.items():
            n = int(arr.shape[0])
            rng = (0, self.window_size * n)
            arr[:, 0] += np.histogram(fwd.get(cname, []), bins=n, range=rng)[0].astype('float')
            arr[:, 1] += np.histogram(rev.get(cname, []), bins=n, range=rng)[0].astype('float')

    def update_f_pointmass(self):
        """readstartdist.py:86-117."""
        merged = self.merge()
        n = merged.shape[0]
        fhat = np.zeros(merged.shape)
        nzi = np.nonzero(merged)
        nz = merged[nzi]
        csum = np.sum(nz)
        denom = 2 * n * self.alpha + csum
        fhat[nzi] = np.divide(np.add(self.alpha, nz), denom)
        rhs = (self.alpha / (2 * n * self.alpha + csum))
        beta_num = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha + csum)))
        beta_denom = np.exp(betaln(self.alpha, ((2 * n - 1) * self.alpha))) or 1e-20
        p0_bit = self.p0 / (self.p0 + (1 - self.p0))
        expected = (1 - p0_bit * (beta_num / beta_denom)) * rhs
        zero = np.ones(fhat.shape, dtype="bool")
        zero[nzi] = 0
        fhat[zero] = expected
        return self._expand(fhat)

    def _expand(self, fhat, ds=100):
        """readstartdist.py:121-152."""
        out = np.repeat(fhat, int(self.window_size // ds), axis=0)
        d = self.target_size - out.shape[0]
        assert d < self.window_size
        if d > 0:
            out = np.append(out, out[-d:], axis=0)
        elif d < 0:
            out = out[:-abs(d)]
        s = np.sum(out)
        if s != 0:
            out = np.multiply(out, self.on_target / s)
        return out
