"""The decision step of the reference's simulation mode: apply the CURRENT
strategy to the mu-truncated mappings of a sampled batch.

Mirrors `BossRunsSim.make_decisions` (/root/reference/boss/runs/simulation.py:37-120) for the
part that touches the strategy: per read, best truncated mapping -> start position (tstart on
'+', tend - 1 on '-') -> `strat[start // 100, rev, barcode]`; a mapping to a contig without a
strategy (unknown, rejected, or out of range) is rejected (simulation.py:81-85).  The lookup
is vectorised over the whole batch; sampling, pseudo-time accounting and the read cache
(sampler.py, batch.py) stay outside the decision-update path.
"""
import numpy as np


def make_decisions(runs, paf_trunc, read_ids, barcodes=None, window=100):
    """-> dict(read id -> bool accept) for every read with a truncated mapping, plus the
    summary arrays (`summ`) of those mappings.  `barcodes`: {read id: barcode index}."""
    summ = runs.engine.paf_summary(paf_trunc, read_ids, min_len=1)      # parse_PAF default min_len
    ids = summ["ids"]
    k = len(summ["contig_idx"])
    accept = np.zeros(k, dtype=bool)
    rev = summ["rev"].astype(np.int64)
    start = np.where(rev != 0, summ["tend"] - 1, summ["tstart"])
    row = start // window
    bc = np.zeros(k, dtype=np.int64)
    if barcodes is not None:
        bc = np.array([barcodes[ids[i]] for i in summ["read_idx"]], dtype=np.int64)
    names = runs.contig_names
    for ci in np.unique(summ["contig_idx"]):
        if ci < 0:
            continue
        cont = runs.contigs[names[ci]]
        if cont.rej or getattr(cont, "remote", False):
            continue                                                     # KeyError in contigs_filt
        sel = np.nonzero(summ["contig_idx"] == ci)[0]
        strat = cont.strat
        # numpy index semantics of strat[row, rev, bc]: negative rows wrap, rows past the end
        # raise IndexError -> reject
        r = row[sel]
        ok = (r < strat.shape[0]) & (r >= -strat.shape[0]) & (bc[sel] < strat.shape[2])
        idx = sel[ok]
        accept[idx] = strat[row[idx], rev[idx], bc[idx]]
    return {ids[summ["read_idx"][i]]: bool(accept[i]) for i in range(k)}, summ
