"""Threshold search over the merged benefit array and its distribution back to contigs
(SURVEY §8 a14-a16).

Follows /root/reference/boss/runs/sequences.py:553-649 (merge_benefit, find_strat_thread)
and /root/reference/boss/runs/core.py:125-186 (_distribute_strategy, update_wrapper glue).
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .contig import adjust_length


def find_strategy(benefit, smu, fhat, time_cost, detail=None):
    """sequences.py:566-649.  Returns (strat bool like benefit, threshold).
    If `detail` is a dict it receives the intermediate statistics (for parity tests)."""
    window = 100
    tbar0 = 300 // window + 300 // window + 400 // window
    tc = time_cost // window
    flat = benefit.flatten('F')
    nzi = np.nonzero(flat)
    nz = flat[nzi]
    normaliser = np.max(nz)
    _, exps = np.frexp(nz / normaliser)
    exps = np.abs(exps)
    chunks = np.array_split(exps, 12)
    with ThreadPoolExecutor(max_workers=12) as ex:
        cnts = list(ex.map(np.bincount, chunks))
    bincounts = np.zeros(np.max([c.shape[0] for c in cnts]), dtype='int')
    for c in cnts:
        bincounts[0:c.shape[0]] += c
    uniq = np.nonzero(bincounts)[0]
    counts = bincounts[uniq]
    fchunks = np.array_split(fhat.flatten('F')[nzi], 12)
    with ThreadPoolExecutor(max_workers=12) as ex:
        fgs = list(ex.map(lambda ab: np.bincount(ab[0], weights=ab[1]), zip(chunks, fchunks)))
    f_grid = np.zeros(np.max([f.shape[0] for f in fgs]), dtype='float')
    for f in fgs:
        f_grid[0:f.shape[0]] += f
    f_grid = f_grid[uniq]
    f_mean = f_grid / counts
    benefit_bin = np.power(2.0, -uniq) * normaliser
    ubar0 = np.sum(fhat * smu)
    cs_u = np.cumsum(benefit_bin * f_mean * counts) + ubar0
    cs_t = np.cumsum(tc * counts * f_mean) + tbar0
    size = np.argmax(cs_u / cs_t) + 1
    try:
        threshold = benefit_bin[size]
    except IndexError:
        threshold = benefit_bin[-1]
    if detail is not None:
        # how far the argmax is from a tie (test infrastructure only: the reference does not compute it) — the product's
        # exact sums and these 12-chunk float sums differ by ~1e-16 relative, so only a margin near that could flip the choice
        peaks = cs_u / cs_t
        rest = np.delete(peaks, size - 1)
        margin = float((peaks[size - 1] - rest.max()) / peaks[size - 1]) if rest.size and peaks[size - 1] > 0 else 1.0
        detail.update(normaliser=normaliser, exponents=uniq, counts=counts, f_grid=f_grid,
                      ubar0=ubar0, strat_size=size, threshold=threshold, argmax_margin=margin)
    return np.where(benefit >= threshold, True, False), threshold


def distribute(contigs_filt, strat, window=100):
    """core.py:125-155 (all barcodes are always looped, so one code path suffices)."""
    i = 0
    for c in contigs_filt.values():
        rep = np.repeat(c.bucket_switches, c.bucket_size // window, axis=0)
        buckets = adjust_length(c.strat.shape[0], rep)
        cs = strat[i: i + c.length // window, :]
        assert cs.shape == c.strat.shape
        for b in range(c.nb):
            c.strat[buckets[:, b], :, b] = cs[buckets[:, b], :, b]
        i += c.length // window
