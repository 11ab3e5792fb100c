"""Seeded end-to-end scenarios shared by the golden-vector generator and the parity tests."""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from boss_runs_amd import synth  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")

SCENARIOS = [("p1_nb1", 1, 1), ("p2_nb1", 2, 1), ("p1_nb2", 1, 2), ("p2_nb2", 2, 2)]   # tag, ploidy, nb
E2E_LENGTHS = [150_000, 260_000, 120_000, 60_000]
E2E_NAMES = ["ctgA", "ctgB", "ctgREJ", "ctgSHORT"]
E2E_REJECT = "ctgREJ"
E2E_BATCHES = 5
E2E_READS = 420


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def e2e_reference(seed=1):
    return synth.make_reference(E2E_LENGTHS, seed=seed, names=E2E_NAMES)


def e2e_contig_strings(contigs=None):
    contigs = contigs or e2e_reference()
    return [(n, synth.codes_to_str(c)) for n, c in contigs]


def e2e_batch(contigs, b, nb):
    """Hot region on ctgB, thin coverage on ctgA (dropout); the rejected and the short contig
    still attract reads."""
    return synth.make_batch(contigs, E2E_READS, seed=10 + b, mean_len=3000.0, nbarcodes=nb,
                            start_weights=[0.6, 1.6, 0.5, 0.5])


def batch_digest(batch):
    return digest(batch["paf"].encode(), "".join(batch["seqs"].values()).encode())


def unpack_strat(g, key, shape):
    n = int(np.prod(shape))
    return np.unpackbits(g[key])[:n].reshape(shape).astype(bool)


# ---- the saturated regime (every site maxed or dropped out) -------------------------------------
SAT_DEPTH = 240          # mean/8 >= 29: no site can sit between the dropout threshold and the depth cap of 30
SAT_RAMP = 600


def saturated_coverage(seq_int, nb=1, depth=SAT_DEPTH, ramp=SAT_RAMP):
    """The coverage a long run converges to: every site `depth` reads deep, all on the reference
    base, rising from zero over the first / falling over the last `ramp` sites (reads start
    uniformly, so the ends of a contig fill last).  With depth 240 the dropout threshold
    int(mean / 8) is 29: every score is `tiny` (depth >= 30) or 0.0 (dropped out) and every
    benefit ~1e-300 — the regime in which an absolute fixed point loses ubar0."""
    seq_int = np.asarray(seq_int)
    L = seq_int.shape[0]
    d = np.full(L, depth, dtype=np.int64)
    r = np.arange(ramp, dtype=np.int64) * depth // ramp
    d[:ramp] = r
    d[L - ramp:] = r[::-1]
    cov = np.zeros((L, 5, nb), dtype=np.uint16)
    for b in range(nb):
        cov[np.arange(L), seq_int, b] = d
    return cov
