"""Error model, genotype priors and the per-site posterior / benefit score.

Follows /root/reference/boss/runs/sequences.py:15-326 (Priors) and :460-549
(Scoring.calc_posterior / calc_score / calc_posterior_and_scores).  Only the branch the
reference actually runs is restated (deletion_error=0.03 > 0, i.e. 5 observable states).
"""
import numpy as np

DEL_ERR = 0.03          # sequences.py:41  deletion_error
MISSED_DEL = 0.1        # sequences.py:42  err_missed_deletion
SUBST_ERR = 0.04        # sequences.py:43  substitution_error
THETA = 0.01            # sequences.py:188 / :244
DEL_SUBS = 0.4          # sequences.py:189 / :245
POPSIZE = 1000          # sequences.py:257

DIPLOID_GENOTYPES = ['AA', 'AC', 'AG', 'AT', 'CC', 'CG', 'CT', 'GG', 'GT', 'TT',
                     'A-', 'C-', 'G-', 'T-', '--']   # sequences.py:115-116


def make_phi(diploid):
    """P(observed state | genotype); sequences.py:70-91 (haploid), :112-153 (diploid)."""
    if not diploid:
        nb = ng = 5
        phi = np.zeros((nb, ng))
        for b in range(nb - 1):
            for g in range(ng - 1):
                phi[b][g] = 1.0 - (SUBST_ERR + DEL_ERR) if b == g else SUBST_ERR / (nb - 2)
                phi[nb - 1][g] = DEL_ERR
            phi[b][ng - 1] = MISSED_DEL / (nb - 1)
        phi[nb - 1][ng - 1] = 1.0 - MISSED_DEL
        return nb, ng, phi
    bases = 'ACGT-'
    nb, ng = 5, 15
    phi = np.zeros((nb, ng))
    ok = 1.0 - (SUBST_ERR + DEL_ERR)
    for b in range(nb - 1):
        for g in range(ng - 5):
            k = DIPLOID_GENOTYPES[g].count(bases[b])
            if k == 2:
                phi[b][g] = ok
            elif k == 1:
                phi[b][g] = ok / 2 + SUBST_ERR / (2 * (nb - 2))
            else:
                phi[b][g] = SUBST_ERR / (nb - 2)
        for g in range(10, 14):
            k = DIPLOID_GENOTYPES[g].count(bases[b])
            if k == 1:
                phi[b][g] = ok / 2 + MISSED_DEL / (2 * (nb - 1))
            elif k == 0:
                phi[b][g] = SUBST_ERR / (2 * (nb - 2)) + MISSED_DEL / (2 * (nb - 1))
        phi[b][ng - 1] = MISSED_DEL / (nb - 1)
    for g in range(ng):
        k = DIPLOID_GENOTYPES[g].count('-')
        if k == 2:
            phi[nb - 1][g] = 1.0 - MISSED_DEL
        elif k == 1:
            phi[nb - 1][g] = (1.0 - MISSED_DEL) / 2 + DEL_ERR / 2
        else:
            phi[nb - 1][g] = DEL_ERR
    return nb, ng, phi


def make_priors(diploid):
    """Genotype priors per reference base; sequences.py:217-237 (haploid), :255-313 (diploid)."""
    if not diploid:
        pr = np.zeros((4, 5))
        for i in range(4):
            for j in range(4):
                pr[i][j] = 1.0 - (THETA * (1.0 + DEL_SUBS)) if i == j else THETA / 3
        if DEL_SUBS > 0.0001:
            pr[:, -1] = THETA * DEL_SUBS
        return pr
    homo = 0.0
    hetero = 0.0
    aN = np.sum(1.0 / (np.arange(1, POPSIZE + 1)))
    for i in range(POPSIZE):
        homo += (1.0 / ((i + 1) * aN)) * ((i + 1) * float(i + 1) / (POPSIZE ** 2))
        hetero += (1.0 / ((i + 1) * aN)) * 2 * ((POPSIZE - (i + 1)) * float(i + 1) / (POPSIZE ** 2))
    p_homo = homo / (homo + hetero)
    bases = 'ACGT-'
    pr = np.zeros((4, 15))
    for b in range(4):
        for g in range(10):
            k = DIPLOID_GENOTYPES[g].count(bases[b])
            if k == 2:
                pr[b][g] = 1 - THETA * (1 + DEL_SUBS)
            elif k == 1:
                pr[b][g] = ((1 - p_homo) * THETA) / 3
            else:
                pr[b][g] = (p_homo * THETA) / 3
        for g in range(10, 14):
            pr[b][g] = (1 - p_homo) * DEL_SUBS * THETA
        pr[b][14] = p_homo * DEL_SUBS * THETA
    return pr


class SiteModel:
    """Priors + Scoring numerics for one ploidy (sequences.py:17-33, :335-342)."""

    def __init__(self, ploidy=1):
        if int(ploidy) == 1:
            self.diploid = False
        elif int(ploidy) == 2:
            self.diploid = True
        else:
            raise ValueError("Given ploidy is not defined")   # sequences.py:29
        self.len_b, self.len_g, self.phi = make_phi(self.diploid)
        # sequences.py:159-168: phi ** n for n < 1000, as one array power per (b, g)
        self.phi_pow = np.full((self.len_b, self.len_g, 1000), 1.0)
        for i in range(self.len_b):
            for j in range(self.len_g):
                self.phi_pow[i, j, :] = self.phi[i, j] ** np.arange(1000)
        self.priors = make_priors(self.diploid)
        # sequences.py:181 / :342: score of the bare prior of reference base 'A'
        s0, e0 = self.score_of_posterior(np.array([self.priors[0]]))
        self.score0, self.ent0 = s0, e0

    def posterior(self, cov):
        """cov uint[n,5] -> posterior [4, n, len_g]; sequences.py:485-516."""
        cov = np.array(cov, dtype=np.int64)
        cov[cov > 990] = 990
        n = cov.shape[0]
        post = np.repeat(self.priors[:, np.newaxis], repeats=n, axis=1)
        lik = np.full(n, 1.0)
        for j in range(self.len_g):
            if j > 0:
                lik.fill(1.0)
            for i in range(self.len_b):
                lik *= self.phi_pow[i, j, cov[:, i]]
            for h in range(4):
                post[h, :, j] *= lik
        for h in range(4):
            z = np.sum(post[h, :, :], axis=1)
            z[z < 1e-300] = 1e-300
            post[h, :, :] /= z[:, np.newaxis]
        return post

    def score_of_posterior(self, post):
        """post [n, len_g] -> (score[n], entropy[n]); sequences.py:520-549."""
        n = post.shape[0]
        logs = np.zeros_like(post)
        np.log(post, where=post > 0.0, out=logs)
        entropy = np.sum(-post * logs, axis=1)
        new_entropy = np.zeros(n)
        obs = np.zeros(n)
        new_post = np.zeros((n, self.len_g))
        for i in range(self.len_b):
            np.multiply(post, self.phi[i], out=new_post)
            np.sum(new_post, axis=1, out=obs)
            obs[obs == 0] = 1e-300
            new_post /= obs[:, np.newaxis]
            np.log(new_post, where=new_post > 0.0, out=logs)
            for j in range(self.len_g):
                new_entropy -= obs * new_post[:, j] * logs[:, j]
        return entropy - new_entropy, entropy

    def entropy_and_score(self, cov):
        """cov [n,5] -> (entropy[4,n], score[4,n]) for all reference bases; sequences.py:460-481."""
        post = self.posterior(cov)
        n = post.shape[1]
        sc = np.zeros((4, n))
        en = np.zeros((4, n))
        for r in range(4):
            sc[r], en[r] = self.score_of_posterior(post[r, :, :])
        return en, sc


class PatternCache:
    """The reference's dense lookup table `score_arr` / `entropy_arr` float64[40,40,40,40,40,4]
    (sequences.py:347-393), indexed by the five counts and the reference base, with its
    on-demand fill (:433-448): an entry that still reads 0.0 is computed from the pattern and
    inserted.  Like the reference's `np.zeros`, the two 3.3 GB arrays are virtual until touched.
    The ~137 k patterns the reference precomputes at start-up are not precomputed here: an
    entry's value does not depend on when it was inserted, and the fill below evaluates each
    distinct missing pattern once (the reference evaluates every missing SITE, same values).
    """
    MAXC = 40

    def __init__(self, model):
        self.model = model
        shape = (self.MAXC,) * 5 + (4,)
        self.score_arr = np.zeros(shape)
        self.entropy_arr = np.zeros(shape)

    @staticmethod
    def _idx(cov):
        return cov[:, 0], cov[:, 1], cov[:, 2], cov[:, 3], cov[:, 4]

    def fill(self, cov):
        """Insert every pattern of cov[n,5] that the table does not hold yet."""
        if cov.shape[0] == 0:
            return
        c = cov.astype(np.int64)
        key = (((c[:, 0] * 40 + c[:, 1]) * 40 + c[:, 2]) * 40 + c[:, 3]) * 40 + c[:, 4]
        key = np.unique(key)
        u = np.empty((key.shape[0], 5), dtype=np.uint16)
        for k in (4, 3, 2, 1, 0):
            u[:, k] = key % 40
            key = key // 40
        u = u[self.score_arr[self._idx(u) + (0,)] == 0.0]
        for lo in range(0, u.shape[0], 1 << 16):
            part = u[lo: lo + (1 << 16)]
            e, s = self.model.entropy_and_score(part)
            self.score_arr[self._idx(part)] = s.T
            self.entropy_arr[self._idx(part)] = e.T

    def lookup(self, cov):
        """cov [n,5] -> (entropy[4,n], score[4,n]) for all reference bases (fills as needed)."""
        cov = np.ascontiguousarray(cov, dtype=np.uint16)
        if cov.shape[0] == 0:
            return np.zeros((4, 0)), np.zeros((4, 0))
        self.fill(cov)
        ix = self._idx(cov)
        return self.entropy_arr[ix].T, self.score_arr[ix].T
