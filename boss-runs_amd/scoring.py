"""Host-side (init-time) numerics: error model, genotype priors and the score/entropy table
over every coverage composition with depth < 30 (hard part H2).

Mirrors the arithmetic of /root/reference/boss/runs/sequences.py:15-326 (Priors) and
:460-549 (Scoring.calc_posterior_and_scores): same elementwise numpy operations in the same
order, so a table entry is bit-identical to what the reference's `score_arr` (or its
on-demand fill, :433-448) would hold on the same machine.  The GPU never evaluates log/pow;
it ranks the coverage pattern and gathers from this table.

The reference's 40^5 x 4 sparse LUT (:347-393) is replaced by a dense table over the
C(34,5) = 278,256 compositions (c0..c4) with c0+..+c4 <= 29; sites with depth >= 30 never
look up (sequences.py:419-430).
"""
from math import comb

import numpy as np

NCOMP = comb(34, 5)
MAXCOV = 30

_GENOTYPES_2N = ('AA', 'AC', 'AG', 'AT', 'CC', 'CG', 'CT', 'GG', 'GT', 'TT', 'A-', 'C-', 'G-', 'T-', '--')
_STATES = 'ACGT-'


def error_model(ploidy, deletion_error=0.03, err_missed_deletion=0.1, substitution_error=0.04):
    """phi[observed state, genotype] (sequences.py:70-91 haploid, :112-153 diploid)."""
    if int(ploidy) not in (1, 2):
        raise ValueError("Given ploidy is not defined")
    e_sub, e_del, e_miss = substitution_error, deletion_error, err_missed_deletion
    if int(ploidy) == 1:
        phi = np.zeros((5, 5))
        for b in range(4):
            for g in range(4):
                phi[b][g] = (1.0 - (e_sub + e_del)) if b == g else e_sub / 3
                phi[4][g] = e_del
            phi[b][4] = e_miss / 4
        phi[4][4] = 1.0 - e_miss
        return phi
    clean = 1.0 - (e_sub + e_del)
    phi = np.zeros((5, 15))
    for b in range(4):
        for g in range(10):
            copies = _GENOTYPES_2N[g].count(_STATES[b])
            phi[b][g] = (clean, clean / 2 + e_sub / (2 * 3), e_sub / 3)[2 - copies]
        for g in range(10, 14):
            copies = _GENOTYPES_2N[g].count(_STATES[b])
            if copies == 1:
                phi[b][g] = clean / 2 + e_miss / (2 * 4)
            else:
                phi[b][g] = e_sub / (2 * 3) + e_miss / (2 * 4)
        phi[b][14] = e_miss / 4
    for g in range(15):
        gaps = _GENOTYPES_2N[g].count('-')
        phi[4][g] = (e_del, (1.0 - e_miss) / 2 + e_del / 2, 1.0 - e_miss)[gaps]
    return phi


def genotype_priors(ploidy, theta=0.01, del_subs_ratio=0.4, popsize=1000):
    """priors[reference base, genotype] (sequences.py:217-237 haploid, :255-313 diploid)."""
    if int(ploidy) == 1:
        pr = np.zeros((4, 5))
        for r in range(4):
            for g in range(4):
                pr[r][g] = 1.0 - (theta * (1.0 + del_subs_ratio)) if r == g else theta / 3
        if del_subs_ratio > 0.0001:
            pr[:, -1] = theta * del_subs_ratio
        return pr
    homo = hetero = 0.0
    a_n = np.sum(1.0 / (np.arange(1, popsize + 1)))
    for i in range(popsize):
        homo += (1.0 / ((i + 1) * a_n)) * ((i + 1) * float(i + 1) / (popsize ** 2))
        hetero += (1.0 / ((i + 1) * a_n)) * 2 * ((popsize - (i + 1)) * float(i + 1) / (popsize ** 2))
    p_homo = homo / (homo + hetero)
    pr = np.zeros((4, 15))
    for r in range(4):
        for g in range(10):
            copies = _GENOTYPES_2N[g].count(_STATES[r])
            if copies == 2:
                pr[r][g] = 1 - theta * (1 + del_subs_ratio)
            elif copies == 1:
                pr[r][g] = ((1 - p_homo) * theta) / 3
            else:
                pr[r][g] = (p_homo * theta) / 3
        for g in range(10, 14):
            pr[r][g] = (1 - p_homo) * del_subs_ratio * theta
        pr[r][14] = p_homo * del_subs_ratio * theta
    return pr


def shannon_and_gain(post, phi):
    """post[n, G] -> (score[n], entropy[n]): entropy minus the expected entropy after one more
    observation (sequences.py:520-549)."""
    n, n_g = post.shape
    logs = np.zeros_like(post)
    np.log(post, where=post > 0.0, out=logs)
    entropy = np.sum(-post * logs, axis=1)
    expected = np.zeros(n)
    p_obs = np.zeros(n)
    nxt = np.zeros((n, n_g))
    for state in range(phi.shape[0]):
        np.multiply(post, phi[state], out=nxt)
        np.sum(nxt, axis=1, out=p_obs)
        p_obs[p_obs == 0] = 1e-300
        nxt /= p_obs[:, np.newaxis]
        np.log(nxt, where=nxt > 0.0, out=logs)
        for g in range(n_g):
            expected -= p_obs * nxt[:, g] * logs[:, g]
    return entropy - expected, entropy


def pattern_tables(patterns, phi, priors):
    """patterns uint[n,5] -> (entropy[4,n], score[4,n]) for all four reference bases
    (sequences.py:460-516)."""
    cov = np.array(patterns, dtype=np.int64)
    cov[cov > 990] = 990
    n = cov.shape[0]
    n_b, n_g = phi.shape
    phi_pow = np.empty((n_b, n_g, 1000))
    for i in range(n_b):
        for j in range(n_g):
            phi_pow[i, j, :] = phi[i, j] ** np.arange(1000)          # sequences.py:164-168
    post = np.repeat(priors[:, np.newaxis], repeats=n, axis=1)        # [4, n, G]
    lik = np.full(n, 1.0)
    for j in range(n_g):
        if j > 0:
            lik.fill(1.0)
        for i in range(n_b):
            lik *= phi_pow[i, j, cov[:, i]]
        for r in range(4):
            post[r, :, j] *= lik
    for r in range(4):
        z = np.sum(post[r, :, :], axis=1)
        z[z < 1e-300] = 1e-300
        post[r, :, :] /= z[:, np.newaxis]
    entropy = np.zeros((4, n))
    score = np.zeros((4, n))
    for r in range(4):
        score[r], entropy[r] = shannon_and_gain(post[r, :, :], phi)
    return entropy, score


def composition_rank(c):
    """Rank of compositions c[...,5] (depth <= 29) in the combinatorial number system the
    sweep kernel uses: sum_k C(c0+..+c(k-1) + k-1, k), k = 1..5."""
    c = np.asarray(c, dtype=np.int64)
    binom = np.array([[comb(q, k) for q in range(36)] for k in range(6)], dtype=np.int64)
    p = np.cumsum(c, axis=-1)
    rank = np.zeros(c.shape[:-1], dtype=np.int64)
    for k in range(1, 6):
        rank += binom[k][p[..., k - 1] + (k - 1)]
    return rank


def all_compositions():
    """uint16[NCOMP, 5]: row r is the composition whose rank is r."""
    grids = np.meshgrid(*[np.arange(MAXCOV)] * 5, indexing='ij', sparse=True)
    total = grids[0] + grids[1] + grids[2] + grids[3] + grids[4]
    idx = np.argwhere(total < MAXCOV)
    out = np.empty((NCOMP, 5), dtype=np.uint16)
    rank = composition_rank(idx)
    assert idx.shape[0] == NCOMP
    out[rank] = idx
    return out


class SiteScoring:
    """Holds the model of one ploidy and its dense table.  `score0`/`ent0` are the scalars the
    reference exposes as Scoring.score0 / ent0 (sequences.py:342)."""

    def __init__(self, ploidy=1):
        self.ploidy = int(ploidy)
        self.phi = error_model(self.ploidy)
        self.priors = genotype_priors(self.ploidy)
        self.len_b, self.len_g = self.phi.shape
        s, e = shannon_and_gain(np.array([self.priors[0]]), self.phi)
        self.score0, self.ent0 = s, e
        self._tables = None

    def tables(self):
        """(score[NCOMP,4], entropy[NCOMP,4]) float64, row-major = index rank*4 + ref."""
        if self._tables is None:
            pats = all_compositions()
            ent, sco = pattern_tables(pats, self.phi, self.priors)
            self._tables = (np.ascontiguousarray(sco.T), np.ascontiguousarray(ent.T))
        return self._tables

    def calc_posterior_and_scores(self, cov_patterns):
        """Same signature/return order as the reference method (entropy, score)."""
        return pattern_tables(cov_patterns, self.phi, self.priors)
