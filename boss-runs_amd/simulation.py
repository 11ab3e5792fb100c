"""Simulation mode of BOSS-RUNS (BASELINE.json configs[0]): `BossRunsSim`
(/root/reference/boss/runs/simulation.py:12-213) with the decision step and the strategy update
on the GPU path.

`make_decisions` (simulation.py:37-120) applies the CURRENT strategy to the mu-truncated mapping
of every sampled read — `strat[start // 100, rev, barcode]` with start = tstart on '+', tend - 1
on '-' — and keeps, per read, either the best full-length mapping (accepted) or the truncated one
(rejected; the read is cut to `seq[:mu]`); unmapped reads follow `accept_unmapped`.
`process_batch_runs_sim` (simulation.py:139-190) then updates the read-length distribution,
abundance counts and read starts from the ACCEPTED reads only, ingests both kinds of mapping, and
runs the same `update_wrapper` as the live loop.

What stays outside (SURVEY §8 "out of scope"): drawing the batch from FASTQ / PAF files
(`Sampler`, boss/sampler.py) and the pseudo-time read cache with its FASTQ dumps (`ReadCache`,
boss/batch.py).  Both are attachable: `sampler` is any object with `sample()` returning the
reference's 5-tuple (and `fq_stream.read_ids`); `read_cache`, if given, receives the same
`update_times_runs` / `fill_cache` calls the reference makes.

The per-read Python loop of the reference becomes: one native best-mapper pass over the truncated
PAF (`bossx_paf_summary`), a vectorised mask lookup, and one native line filter per PAF text
(`bossx_paf_select_lines`) whose output goes straight into the ordinary ingest path.
"""
import ctypes as C
import logging

import numpy as np

from . import _lib
from .engine import Engine
from .runs import BossRuns


def lookup_decisions(runs, summ, bc, window=100):
    """Vectorised `strat[start_pos // window, rev, barcode]` over the chosen truncated mappings
    (simulation.py:63-85).  A mapping to a contig without a strategy (unknown, rejected) or out of
    the array's range is rejected, exactly the cases the reference's `except (KeyError, IndexError)`
    catches; negative rows wrap like numpy's."""
    k = len(summ["contig_idx"])
    accept = np.zeros(k, dtype=bool)
    rev = summ["rev"].astype(np.int64)
    start = np.where(rev != 0, summ["tend"] - 1, summ["tstart"])
    row = start // window
    names = runs.contig_names
    for ci in np.unique(summ["contig_idx"]):
        if ci < 0:
            continue
        cont = runs.contigs[names[ci]]
        if cont.rej or getattr(cont, "remote", False):
            continue                                                     # KeyError in contigs_filt
        sel = np.nonzero(summ["contig_idx"] == ci)[0]
        strat = cont.strat
        r = row[sel]
        ok = (r < strat.shape[0]) & (r >= -strat.shape[0]) & (bc[sel] < strat.shape[2]) & (bc[sel] >= -strat.shape[2])
        idx = sel[ok]
        accept[idx] = strat[row[idx], rev[idx], bc[idx]]
    return accept


def make_decisions(runs, paf_trunc, read_ids, barcodes=None, window=100):
    """The decision look-up alone: -> ({read id: accept} for every read with a truncated mapping,
    summary arrays of those mappings).  `barcodes`: {read id: barcode index}."""
    summ = runs.engine.paf_summary(paf_trunc, read_ids, min_len=1)      # parse_PAF default min_len
    ids = summ["ids"]
    bc = np.zeros(len(summ["contig_idx"]), dtype=np.int64)
    if barcodes is not None:
        bc = np.array([barcodes[ids[i]] for i in summ["read_idx"]], dtype=np.int64)
    accept = lookup_decisions(runs, summ, bc, window)
    return {ids[i]: bool(a) for i, a in zip(summ["read_idx"], accept)}, summ


def _select_lines(paf_text, nptr, nlen, n, keep):
    """bossx_paf_select_lines: the lines of `paf_text` that belong to reads with keep != 0."""
    lib = _lib.load()
    paf = paf_text.encode() if isinstance(paf_text, str) else bytes(paf_text)
    out = C.create_string_buffer(len(paf) + 1)
    out_len = C.c_size_t(0)
    keep = np.ascontiguousarray(keep, dtype=np.uint8)
    rc = lib.bossx_paf_select_lines(paf, len(paf), nptr.ctypes.data, nlen.ctypes.data, n, keep.ctypes.data,
                                    C.cast(out, C.c_void_p), len(paf) + 1, C.byref(out_len))
    if rc:
        raise _lib.BossxError("bossx_paf_select_lines failed (%d)" % rc)
    return out.raw[:out_len.value]


class BossRunsSim(BossRuns):
    def init_sim(self, contigs=None, engine=None, sampler=None, read_cache=None, mapper=None) -> None:
        """simulation.py:14-33.  `sampler` / `read_cache`: see the module docstring."""
        self.init(contigs=contigs, engine=engine, mapper=mapper)
        args = self.args.simulation
        self.sampler = sampler
        self.read_cache = read_cache
        self.mu = 400
        self.accept_unmapped = args.accept_unmapped
        self.last_counts = {}

    # ---- simulation.py:37-120 -----------------------------------------------------------------
    def make_decisions(self, seqs, paf_full, paf_trunc, barcodes, window=100):
        """-> (paf_text, reads_decision, n_mapped, n_unmapped, n_accepted, n_rejected).  `paf_text`
        stands for the reference's `paf_dict`: the full-length lines of every accepted read and the
        truncated lines of every rejected one (the best mapping per read is chosen when the text is
        parsed, as `choose_best_mapper` does on the dict).  The barcode each mapping is ingested
        with is kept in `self._sim_barcodes` (index per read of `seqs`)."""
        ids = list(seqs.keys())
        n = len(ids)
        nptr, nlen = Engine._str_pointers(ids)
        summ = self.engine.paf_summary(paf_trunc, ids, min_len=1)
        ridx = summ["read_idx"]
        bc_all = np.fromiter((barcodes[i] for i in ids), dtype=np.int64, count=n) if barcodes is not None \
            else np.zeros(n, dtype=np.int64)
        accept = lookup_decisions(self, summ, bc_all[ridx], window)
        mapped = np.zeros(n, dtype=bool)
        mapped[ridx] = True
        acc = np.zeros(n, dtype=bool)
        acc[ridx] = accept
        keep_full = acc.copy()
        ingest_bc = bc_all.copy()
        if self.accept_unmapped:
            # simulation.py:102-111: the full-length mapping (if any) of an unmapped read is used,
            # and its record never gets a barcode (None -> index 0 in increment_coverage)
            keep_full |= ~mapped
            ingest_bc[~mapped] = 0
        keep_trunc = mapped & ~acc
        text = _select_lines(paf_full, nptr, nlen, n, keep_full)
        tr = _select_lines(paf_trunc, nptr, nlen, n, keep_trunc)
        paf_text = text + (b"\n" if text and tr else b"") + tr
        mu = self.mu
        take_full = acc | (~mapped & bool(self.accept_unmapped))
        reads_decision = {rid: (s if f else s[:mu]) for (rid, s), f in zip(seqs.items(), take_full.tolist())}
        n_mapped = int(mapped.sum())
        n_acc_mapped = int(acc.sum())
        n_unm = n - n_mapped
        n_accepted = n_acc_mapped + (n_unm if self.accept_unmapped else 0)
        n_rejected = (n_mapped - n_acc_mapped) + (0 if self.accept_unmapped else n_unm)
        if self.sampler is not None and hasattr(self.sampler, "fq_stream"):
            n_unmapped = len(set(self.sampler.fq_stream.read_ids) - {ids[i] for i in ridx})
        else:
            n_unmapped = n_unm
        self._sim_barcodes = ingest_bc.astype(np.int32)
        self._sim_acc = acc
        self.last_decisions = {ids[i]: bool(a) for i, a in zip(ridx, accept)}
        return paf_text, reads_decision, n_mapped, n_unmapped, n_accepted, n_rejected

    # ---- simulation.py:139-190 ----------------------------------------------------------------
    def process_batch_runs_sim(self, batch=None) -> None:
        """One simulated batch.  `batch`: the reference sampler's 5-tuple (read sequences, read
        qualities, {read id: barcode name}, full-length PAF text, truncated PAF text); by default it
        is drawn from `self.sampler.sample()`."""
        if batch is None:
            if self.sampler is None:
                raise RuntimeError("no sampler attached: pass the sampled batch to process_batch_runs_sim")
            batch = self.sampler.sample()
        read_seqs, read_quals, read_barcodes_names, paf_f, paf_t = batch
        read_barcodes = {rid: self.barcodes_index.get(bc, 0) for rid, bc in read_barcodes_names.items()}
        paf_text, reads_decision, n_mapped, n_unmapped, n_accepted, n_rejected = self.make_decisions(
            seqs=read_seqs, paf_full=paf_f, paf_trunc=paf_t, barcodes=read_barcodes)
        logging.info(f"mapped {n_mapped}, not mapped {n_unmapped}")
        logging.info(f"accepted {n_accepted}, rejected {n_rejected}")
        self.last_counts = dict(n_mapped=n_mapped, n_unmapped=n_unmapped, n_accepted=n_accepted,
                                n_rejected=n_rejected)
        eng = self.engine
        # both kinds of mapping are ingested against the FULL sequences (simulation.py:165)
        summ = eng.stage_batch(paf_text, read_seqs, barcodes=self._sim_barcodes, min_len=1)
        # an accepted read without a usable full-length mapping: the reference's
        # choose_best_mapper indexes an empty list there (simulation.py:90-91)
        qlen = summ["qlen"]
        have = np.zeros(len(read_seqs), dtype=bool)
        have[summ["read_idx"]] = True
        missing = np.nonzero(self._sim_acc & ~have)[0]
        if missing.size:
            raise IndexError("accepted read %r has no full-length mapping (choose_best_mapper on an "
                             "empty list, simulation.py:90-91)" % summ["ids"][int(missing[0])])
        eng.ingest_staged()
        if self._fused:
            eng.update_begin(self.args.optional.bucket_threshold)
        # filter_paf_dict (simulation.py:124-135): accepted reads = chosen records whose qlen != mu
        sel = qlen != self.mu
        self.rl_dist.update(np.ascontiguousarray(qlen[sel], dtype=np.int64))      # simulation.py:162
        if self._fused:
            self.launch_benefit()
        # tracker.update(n=n_accepted, paf_dict_acc) + count_read_starts(paf_dict_acc)
        self.total_reads += n_accepted
        ci = summ["contig_idx"][sel]
        for i, k in enumerate(np.bincount(ci[ci >= 0], minlength=len(self.contig_names))):
            if k:
                self.read_counts[self.contig_names[i]] += int(k)
        self.read_starts.count_starts(self.contig_names, ci, summ["rev"][sel], summ["tstart"][sel], summ["tend"][sel])
        if self.read_cache is not None:           # pseudo-time accounting and FASTQ dumps (boss/batch.py)
            fq = self.sampler.fq_stream
            self.read_cache.update_times_runs(total_bases=fq.total_bases, reads_decision=reads_decision,
                                              n_reject=n_rejected)
            if not self.args.general.barcodes:
                self.read_cache.fill_cache(read_sequences=fq.read_sequences, reads_decision=reads_decision)
            else:
                self.read_cache.fill_cache(read_sequences=fq.read_sequences, reads_decision=reads_decision,
                                           reads_barcodes=read_barcodes_names)
        self.update_wrapper()

    def cleanup(self) -> None:
        """simulation.py:194-205: flush what the read cache still holds."""
        rc = self.read_cache
        if rc is None:
            return
        for cond in ('control', 'boss'):
            cache = getattr(rc, f'cache_{cond}')
            if len(list(cache.keys())) > 0:
                rc._execute_dump(cond=cond, dump_number=getattr(rc, f'dump_n_{cond}'), cache=cache)
