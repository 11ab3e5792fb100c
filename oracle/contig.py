"""Per-contig state and the per-site / per-bin stages.

Follows /root/reference/boss/runs/reference.py:18-269 (Contig) and
/root/reference/boss/runs/sequences.py:398-455 (Scoring.update_scores).
"""
from string import ascii_letters

import numpy as np

from .model import SiteModel
from .movesum import move_sum

TINY = np.finfo(float).tiny          # sequences.py:430
_HAPLOID = None


def _haploid_model():
    global _HAPLOID
    if _HAPLOID is None:
        _HAPLOID = SiteModel(1)
    return _HAPLOID


def seq_to_int(seq):
    """reference.py:46-68: ACGT -> 0..3, every other ASCII letter -> 0."""
    table = {ord(c): '0' for c in ascii_letters}
    table.update({ord('A'): '0', ord('C'): '1', ord('G'): '2', ord('T'): '3'})
    return np.frombuffer(seq.translate(table).encode(), 'u1') - ord('0')


def adjust_length(original_size, expanded):
    """utils.py:206-226."""
    d = original_size - expanded.shape[0]
    if d > 0:
        out = np.append(expanded, expanded[-d:], axis=0)
    elif d < 0:
        out = expanded[:-abs(d)]
    else:
        out = expanded
    assert out.shape[0] == original_size
    return out


class OContig:
    def __init__(self, name, seq, rej=False, nbarcodes=1):
        self.name = name.strip().split(" ")[0]
        seq = seq.upper()
        self.length = len(seq)
        self.rej = rej
        self.nb = nbarcodes
        self.seq_int = seq_to_int(seq)
        L, nb = self.length, self.nb
        self.coverage = np.zeros((L, 5, nb), dtype="uint16")           # reference.py:78
        self.change_mask = np.zeros((L, nb), dtype="bool")             # reference.py:80
        self.bucket_size = 20_000                                      # reference.py:84
        self.bucket_switches = np.zeros((L // 20_000 + 1, nb), dtype="bool")
        self.switched_on = np.zeros(nb, dtype="bool")
        m = _haploid_model()          # Contig is always built with ploidy=1: reference.py:314-334
        self.scores = np.full((L, nb), m.score0[0])                    # reference.py:104
        self.entropy = np.full((L, nb), m.ent0[0])
        if rej:
            self.strat = np.zeros(1, dtype="bool")                     # reference.py:116
        else:
            self.strat = np.ones((L // 100, 2, nb), dtype="bool")      # reference.py:118

    # -- a6 ------------------------------------------------------------------------------
    def increment_coverage(self, increments):
        """reference.py:122-145."""
        self.change_mask.fill(0)
        tmp = np.zeros(self.coverage.shape, dtype="uint16")
        for (start, end, q, bc) in increments:
            idx = np.arange(q.shape[0])
            np.add.at(tmp[start:end], (idx, q, 0 if bc is None else bc), 1)
        self.change_mask[np.where(tmp)[0]] = 1
        self.coverage += tmp

    # -- a7 ------------------------------------------------------------------------------
    def update_scores(self, cache):
        """sequences.py:398-455; `cache.score_arr` / `cache.entropy_arr` are the 40^5 x 4 tables."""
        for b in range(self.nb):
            scores = self.scores[:, b]
            entropy = self.entropy[:, b]
            cov = self.coverage[:, :, b]
            cm = self.change_mask[:, b]
            maxed = np.where(cov.sum(axis=1) >= 30)[0]
            cm[maxed] = False
            pos = np.nonzero(cm)[0]
            cc = cov[pos]
            ref = self.seq_int[pos]
            scores[pos] = cache.score_arr[cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3], cc[:, 4], ref]
            scores[maxed] = TINY
            # table misses and sites zeroed by dropout last round read 0.0: computed and inserted
            missing = np.argwhere(scores == 0.0).flatten()
            if missing.shape[0] != 0:
                mp = cov[missing]
                cache.fill(mp)
                mb = self.seq_int[missing]
                scores[missing] = cache.score_arr[mp[:, 0], mp[:, 1], mp[:, 2], mp[:, 3], mp[:, 4], mb]
                entropy[missing] = cache.entropy_arr[mp[:, 0], mp[:, 1], mp[:, 2], mp[:, 3], mp[:, 4], mb]
            entropy[pos] = cache.entropy_arr[cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3], cc[:, 4], ref]

    # -- a8 ------------------------------------------------------------------------------
    def modify_scores(self):
        """reference.py:148-179; returns number of dropout rows hit (for logging)."""
        covsum = np.sum(self.coverage, axis=1)
        mean = np.mean(covsum)
        if mean > 5:
            thr = int(mean / 8)
            rows = np.where(covsum <= thr)[0]
            self.scores[rows] = 0
            return rows.shape[0]
        return 0

    # -- a9 ------------------------------------------------------------------------------
    def check_buckets(self, threshold=5.0):
        """reference.py:183-211 + utils.py:192-202."""
        w = self.bucket_size
        for b in range(self.nb):
            csum = np.sum(self.coverage[:, :, b], axis=1)
            sums = np.sum(csum[: (len(csum) // w) * w].reshape(-1, w), axis=1)
            means = adjust_length(self.bucket_switches.shape[0], np.divide(sums, w))
            sw = self.bucket_switches[:, b]
            sw[np.where(means >= threshold)] = 1
            if len(np.bincount(sw)) == 2 and not all(self.switched_on):
                self.switched_on[np.logical_not(self.switched_on)] = True

    # -- a12 -----------------------------------------------------------------------------
    def calc_smu(self, window=100, mu=400):
        """reference.py:215-237."""
        nbin = self.length // window + 1
        self.scores_ds = np.zeros((nbin, self.nb))
        for b in range(self.nb):
            np.add.at(self.scores_ds[:, b], np.arange(0, self.length) // window, self.scores[:, b])
        self.smu = smu_from_ds(self.scores_ds, window, mu)

    # -- a13 -----------------------------------------------------------------------------
    def calc_u(self, approx_ccl, window=100):
        """reference.py:241-269."""
        self.expected_benefit, self.additional_benefit = benefit_from_ds(self.scores_ds, self.smu, approx_ccl, window)


def smu_from_ds(scores_ds, window=100, mu=400):
    """The move_sum half of Contig.calc_smu (reference.py:231-236) on the bin sums `scores_ds` [nbin, nb]: usable on its own
    where only the bin sums of a contig are at hand (tests: a chromosome-length chain exported from the device)."""
    nbin, nb = scores_ds.shape
    smu = np.zeros((nbin, 2, nb))
    for b in range(nb):
        smu[:, 0, b] = move_sum(scores_ds[::-1, b], mu // window)[::-1]
        smu[:, 1, b] = move_sum(scores_ds[:, b], mu // window)
    return smu


def benefit_from_ds(scores_ds, smu, approx_ccl, window=100):
    """Contig.calc_u (reference.py:241-269) -> (expected_benefit, additional_benefit)."""
    ccl_ds = approx_ccl // window
    mult = np.arange(0.05, 1, 0.1)[::-1]
    nbin, nb = scores_ds.shape
    expected = np.zeros((nbin, 2, nb))
    for b in range(nb):
        tmp = np.zeros((nbin, 2))
        for i in range(10):
            fwd = move_sum(scores_ds[::-1, b], int(ccl_ds[i]))[::-1]
            rev = move_sum(scores_ds[:, b], int(ccl_ds[i]))
            tmp[:, 0] += (fwd * mult[i])
            tmp[:, 1] += (rev * mult[i])
        expected[:, :, b] = tmp
    additional = expected - smu
    additional[additional < 0] = 0
    return expected, additional
