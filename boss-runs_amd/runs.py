"""`BossRuns` — the reference's RUNS orchestrator (boss/runs/core.py:20-224) with its numerics
on the GPU.  Same method names, argument meaning and call order:

    init() -> process_batch_runs(new_reads, new_quals) -> update_wrapper()
           -> _write_contig_strategies(strat_dict)            (boss.npz, tmp + rename)

`process_batch_paf` is `process_batch_runs` minus the mapper call, for callers that already
hold the PAF text (simulations, tests, benchmarks).  Without the HIP extension or a GPU this
class raises at `init()`; there is no numpy fallback.
"""
import logging
import os
import time
from pathlib import Path

import numpy as np

from .config import BossConfig
from .engine import Engine
from .masks import write_mask_bits
from .readlengthdist import ReadlengthDist
from .readstartdist import ReadStartDist
from .reference import Reference
from .scoring import SiteScoring

FX_SHIFT = 100          # fixed-point scale of the histogram sums (kernels.hip.inc kFxShift)
MULT = np.arange(0.05, 1, 0.1)[::-1].copy()       # reference.py:251


def fx_to_float(lo, hi):
    """Exact 128-bit fixed-point accumulator -> correctly rounded float64."""
    return ((int(hi) << 64) + int(lo)) / (1 << FX_SHIFT)


def ubar_to_float(scaled, normaliser):
    """The engine accumulates ubar0 = sum(fhat * benefit) scaled by 2^-E, E = frexp exponent of the
    normaliser (kernels.hip.inc: ubar_scale), so that its fixed point resolves the sum at any
    magnitude of the scores; this undoes the (exact, power-of-two) scaling."""
    import math
    return math.ldexp(scaled, math.frexp(normaliser)[1]) if normaliser > 0 else scaled


def choose_threshold(normaliser, counts_all, fgrid_all, ubar0, time_cost):
    """The host tail of Scoring.find_strat_thread (sequences.py:607-646) from the binned
    statistics.  Returns (threshold, strat_size, exponents_unique)."""
    window = 100
    tbar0 = 300 // window + 300 // window + 400 // window
    tc = time_cost // window
    uniq = np.nonzero(counts_all)[0]
    counts = counts_all[uniq]
    f_grid = fgrid_all[uniq]
    f_mean = f_grid / counts
    benefit_bin = np.power(2.0, -uniq) * normaliser
    cs_u = np.cumsum(benefit_bin * f_mean * counts) + ubar0
    cs_t = np.cumsum(tc * counts * f_mean) + tbar0
    size = int(np.argmax(cs_u / cs_t)) + 1
    threshold = benefit_bin[size] if size < benefit_bin.shape[0] else benefit_bin[-1]
    return float(threshold), size, uniq


def _adjust_length(size, arr):
    """boss/utils.py:206-226: pad with the array's own tail, or trim."""
    d = size - arr.shape[0]
    if d > 0:
        return np.append(arr, arr[-d:], axis=0)
    if d < 0:
        return arr[:-abs(d)]
    return arr


def reference_order_threshold(benefit, fhat, time_cost):
    """The threshold choice of Scoring.find_strat_thread (sequences.py:566-646) with the REFERENCE's
    summation order — twelve np.array_split chunks, np.bincount per chunk, chunk results added in order,
    np.sum(fhat * benefit) pairwise — on the merged, length-adjusted arrays `benefit`, `fhat`
    [T*, 2, nb] (smu_adj := benefit, core.py:182-183).  Only used where the device's exact sums leave the
    argmax within `tie_margin` of a tie (BossRuns._resolve_near_tie): there the float rounding of the
    reference decides, and this reproduces it.  Returns (threshold, strat_size, margin)."""
    window = 100
    tbar0 = 300 // window + 300 // window + 400 // window
    tc = time_cost // window
    flat = benefit.flatten('F')
    nzi = np.nonzero(flat)
    nz = flat[nzi]
    normaliser = np.max(nz)
    exps = np.abs(np.frexp(nz / normaliser)[1])
    e_chunks = np.array_split(exps, 12)
    cnts = [np.bincount(c) for c in e_chunks]
    bincounts = np.zeros(max(c.shape[0] for c in cnts), dtype='int')
    for c in cnts:
        bincounts[0:c.shape[0]] += c
    uniq = np.nonzero(bincounts)[0]
    counts = bincounts[uniq]
    f_chunks = np.array_split(fhat.flatten('F')[nzi], 12)
    fgs = [np.bincount(e, weights=f) for e, f in zip(e_chunks, f_chunks)]
    f_grid = np.zeros(max(f.shape[0] for f in fgs), dtype='float')
    for f in fgs:
        f_grid[0:f.shape[0]] += f
    f_grid = f_grid[uniq]
    f_mean = f_grid / counts
    benefit_bin = np.power(2.0, -uniq) * normaliser
    ubar0 = np.sum(fhat * benefit)
    cs_u = np.cumsum(benefit_bin * f_mean * counts) + ubar0
    cs_t = np.cumsum(tc * counts * f_mean) + tbar0
    peaks = cs_u / cs_t
    size = int(np.argmax(peaks)) + 1
    threshold = benefit_bin[size] if size < benefit_bin.shape[0] else benefit_bin[-1]
    rest = np.delete(peaks, size - 1)
    margin = float((peaks[size - 1] - rest.max()) / peaks[size - 1]) if rest.size and peaks[size - 1] > 0 else 1.0
    return float(threshold), size, margin


class Boss:
    """Slice of boss/core.py:13-176 the RUNS path depends on: run name, output directory tree,
    the global read-length distribution and the batch counter.  Live FASTQ discovery
    (`_get_new_data`, MinKNOW, readfish) is out of scope and left to the caller."""

    def __init__(self, args: BossConfig):
        self.args = args
        self.name = args.general.name
        self.batch = 0
        assert ' ' not in self.name
        self.out_dir = f'./out_{self.name}'
        (Path(self.out_dir) / "masks").mkdir(parents=True, exist_ok=True)
        self.rl_dist = ReadlengthDist()

    data_source = None      # callable() -> (reads {id: seq}, quals {id: qual}); stands in for the FASTQ scan

    def _get_new_data(self):
        """boss/core.py:89-107.  Discovering and reading new FASTQ files (LiveRun.scan_dir,
        FastqBatch) is outside the decision-update path: the caller plugs a `data_source` in.  What
        belongs to the path is kept: the global read-length distribution is updated here, before
        the batch function runs (core.py:106)."""
        if self.data_source is None:
            raise NotImplementedError("live FASTQ scanning is outside the decision-update path: set "
                                      "`data_source` to a callable returning (reads, quals)")
        reads, quals = self.data_source()
        if not reads:
            logging.info("no new files, deferring update ")
            return {}, {}
        self.rl_dist.update(read_lengths={rid: len(seq) for rid, seq in reads.items()})
        return reads, quals

    def process_batch(self, main_processing_func):
        """boss/core.py:137-157: returns the seconds to wait until the next update."""
        logging.info(f"Next batch ---------------------------- # {self.batch}")
        tic = time.time()
        new_reads, new_quals = self._get_new_data()
        if not new_reads:
            return self.args.general.wait
        main_processing_func(new_reads=new_reads, new_quals=new_quals)
        next_update = int(self.args.general.wait - (time.time() - tic))
        self.batch += 1
        return next_update

    def process_batch_sim(self, main_processing_func):
        """boss/core.py:160-176."""
        logging.info(f"Next batch ---------------------------- # {self.batch}")
        tic = time.time()
        main_processing_func()
        next_update = int(self.args.general.wait - (time.time() - tic))
        self.batch += 1
        return next_update


class BossRuns(Boss):
    def init(self, contigs=None, engine=None, is_local=None, mapper=None) -> None:
        """boss/runs/core.py:23-55.  `contigs` optionally replaces the FASTA with an iterable
        of (name, sequence).  `mapper`: a reference-shaped mapper (boss/mapper.py:27: `mu`,
        `_mappy_batch` / `map_sequences`) — what the reference builds at core.py:41-42; without one
        only `process_batch_paf` (PAF text supplied by the caller) is available."""
        a = self.args
        if not a.general.barcodes:
            self.barcodes_index = {"": 0}
        else:
            self.barcodes_index = {int(bc.split('barcode')[1]): i for i, bc in enumerate(a.general.barcodes)}
        self.nbarcodes = len(self.barcodes_index)
        self.engine = engine or Engine(nbarcodes=self.nbarcodes, device=a.gpu.device,
                                       track_entropy=a.gpu.track_entropy)
        self.ref = Reference(ref=a.general.ref, mmi=a.general.mmi, reject_refs=a.optional.reject_refs,
                             barcodes=a.general.barcodes, engine=self.engine, contigs=contigs,
                             is_local=is_local)
        self.contigs = self.ref.contigs
        self.contigs_filt = {n: c for n, c in self.contigs.items() if not c.rej}
        self.contig_names = list(self.contigs.keys())
        # minimap2 itself is upstream of this path: the caller attaches its Mapper; without one, the reference's
        # `Mapper(ref=self.ref.mmi)` (core.py:41-43) where it can be built (mapper.default_mapper)
        if mapper is None and contigs is None:
            from .mapper import default_mapper
            mapper = default_mapper(getattr(self.ref, "mmi", None) or a.general.mmi)
        self.mapper = mapper
        self.read_counts = {n: 0 for n in self.contigs}          # AbundanceTracker
        self.total_reads = 0
        self.read_starts = ReadStartDist(contigs=self.contigs_filt)
        # Contig() is always built haploid (reference.py:314-334): haploid initial fill
        haploid = SiteScoring(ploidy=1)
        self.scoring = haploid if int(a.optional.ploidy) == 1 else SiteScoring(ploidy=a.optional.ploidy)
        self.engine.finalize(score0=haploid.score0[0], ent0=haploid.ent0[0])
        score, entropy = self.scoring.tables()
        self.engine.set_lut(score, entropy)
        if getattr(self.engine, "fhat_resident", False) and not os.environ.get("BOSSX_HOST_FHAT"):
            self.read_starts.attach_engine(self.engine)       # read-start counts mirrored in HBM (readstartdist.py)
        self.threshold = None
        self.last_stats = {}
        self.write_masks = True
        self.mask_format = getattr(a.gpu, "mask_format", "npz")
        if self.mask_format not in ("npz", "bits", "both"):
            raise ValueError("gpu.mask_format must be 'npz', 'bits' or 'both'")
        self._fused = True             # False when update_wrapper is replaced by the staged form
        self.tie_margin = float(getattr(a.gpu, "tie_margin", 1e-9))
        self.ties_resolved = 0         # updates whose threshold was re-derived in the reference's summation order
        self.keep_stats = False        # also fetch the threshold statistics (tests)
        self.log_fractions = True
        self._publish_masks(bits=None)

    def _write_contig_strategies(self, contig_strats) -> None:
        """boss/runs/core.py:59-69."""
        cpath_tmp = f'{self.out_dir}/masks/boss_tmp.npz'
        np.savez(cpath_tmp, **contig_strats)
        Path(cpath_tmp).rename(f'{self.out_dir}/masks/boss.npz')

    def _mask_entries(self):
        """(name, rows, rejected, bit offset) per contig in add order for masks.write_mask_bits."""
        out = []
        for name, c in self.contigs.items():
            off = 0 if c.rej else self.engine.strat_offset(c.index)
            out.append((name, 1 if c.rej else c.length // 100, c.rej, off))
        return out

    def _publish_masks(self, bits) -> None:
        """Write the strategy in the configured format(s).  `bits`: the packed buffer of this
        update, or None to fetch it from the device (init: all ones)."""
        if self.mask_format in ("npz", "both"):
            self._write_contig_strategies(contig_strats=self.ref.get_strategy_dict())
        if self.mask_format in ("bits", "both"):
            if bits is None:
                bits = self.engine.get_strat_bits()
            write_mask_bits(f'{self.out_dir}/masks/boss.bits', self._mask_entries(), bits, self.nbarcodes)

    # ---- batch entry points ---------------------------------------------------------------
    def process_batch_runs(self, new_reads, new_quals=None) -> None:
        """boss/runs/core.py:202-224 with the reference's own `Mapper` (boss/mapper.py:27-125):
        the raw PAF text of the batch comes from `mapper._mappy_batch(sequences=...)` (what
        `map_sequences` parses, mapper.py:63-64) and is parsed natively with the same
        `min_len = int(mapper.mu / 2)`.  A mapper that only offers `map_sequences` (a dict of PafLine
        records) or a `map_batch_paf(sequences) -> str` is accepted as well.  `new_quals` is unused,
        as in the reference (qt = 0: every base counts, sequences.py:735-736)."""
        m = self.mapper
        if m is None:
            raise RuntimeError("no mapper attached: pass one to init(mapper=...) or use "
                               "process_batch_paf(paf_text, new_reads)")
        min_len = int(getattr(m, "mu", 400) / 2)
        if hasattr(m, "_mappy_batch"):
            paf_text = m._mappy_batch(sequences=new_reads)
        elif hasattr(m, "map_batch_paf"):
            paf_text = m.map_batch_paf(sequences=new_reads)
        elif hasattr(m, "map_sequences"):
            from .mapper import paf_dict_to_text
            paf_text = paf_dict_to_text(m.map_sequences(sequences=new_reads))
        else:
            raise TypeError("mapper must provide _mappy_batch, map_batch_paf or map_sequences")
        self.process_batch_paf(paf_text, new_reads, min_len=min_len)

    def process_batch_paf(self, paf_text, new_reads, barcodes=None, min_len=200, starts_filter=None,
                          n_reads_total=None, lookahead=None) -> None:
        """convert_records + _effect_increments + tracker + count_read_starts + update_wrapper
        (core.py:214-224).  `barcodes`: {read id: barcode index} as in simulation.py:148.

        `lookahead`: the NEXT batch, if the caller already holds it (a replay, `process_batch_sim`,
        a mapper that runs ahead) — `(paf_text, new_reads[, barcodes[, min_len]])` or a dict with those
        keys.  It is parsed, uploaded and walked (Engine.stage_batch, into the other slot, on the
        engine's staging stream) while the GPU runs this batch's sweep and chain; nothing of it
        touches the site state before the call that processes it — passing the same objects as
        `paf_text` / `new_reads` then — and results are bit-identical with and without.  A batch the
        reference would reject raises when IT is processed, not while it is staged ahead."""
        summ = self._ingest_batch(paf_text, new_reads, barcodes, min_len)
        if self._fused:       # the GPU sweeps / runs the chain while the host does its bookkeeping
            self.engine.update_begin(self.args.optional.bucket_threshold)
            self.launch_benefit()
        self._account_reads(summ, len(new_reads) if n_reads_total is None else n_reads_total, starts_filter)
        if self._fused and lookahead is not None:
            # the whole update is in the GPU's queue (masks and results included) before the next batch is staged
            self.update_wrapper(between=lambda: self._stage_ahead(lookahead))
        else:
            self._stage_ahead(lookahead)
            self.update_wrapper()

    _ahead = None        # the batch staged ahead by the previous call (dict), or None
    _cur_slot = 0        # engine slot of the batch being processed

    def _ingest_batch(self, paf_text, new_reads, barcodes, min_len):
        """This batch's increments become pending (Engine.ingest_paf) — taken from the slot it was
        staged ahead into when the previous call's `lookahead` named exactly these objects."""
        a, self._ahead = self._ahead, None
        eng = self.engine
        if a is not None and a["paf_text"] is paf_text and a["new_reads"] is new_reads and \
                a["barcodes"] is barcodes and a["min_len"] == min_len and a["n_reads"] == len(new_reads):
            # (the objects staged ahead are frozen until this call: same objects, same number of reads)
            if a["error"] is not None:
                raise a["error"]
            eng.ingest_staged(slot=a["slot"])
            self._cur_slot = a["slot"]
            return a["summ"]
        return eng.ingest_paf(paf_text, new_reads, barcodes=barcodes, min_len=min_len)

    def _stage_ahead(self, lookahead) -> None:
        if lookahead is None or not hasattr(self.engine, "select_batch"):
            return
        if isinstance(lookahead, dict):
            a = dict(paf_text=lookahead["paf_text"], new_reads=lookahead["new_reads"],
                     barcodes=lookahead.get("barcodes"), min_len=lookahead.get("min_len", 200))
        else:
            t = tuple(lookahead)
            a = dict(paf_text=t[0], new_reads=t[1], barcodes=t[2] if len(t) > 2 else None,
                     min_len=t[3] if len(t) > 3 else 200)
        a.update(slot=1 - self._cur_slot, summ=None, error=None, n_reads=len(a["new_reads"]))
        eng = self.engine
        try:
            eng.select_batch(a["slot"])
            a["summ"] = eng.stage_batch(a["paf_text"], a["new_reads"], barcodes=a["barcodes"], min_len=a["min_len"])
        except (ValueError, KeyError, IndexError, TypeError, AssertionError, OverflowError) as e:     # the reference's rejects
            a["error"] = e
        finally:
            eng.select_batch(self._cur_slot)
        self._ahead = a

    def _account_reads(self, summ, n_reads, starts_filter=None):
        # AbundanceTracker.update (abundance_tracker.py:58-69)
        self.total_reads += n_reads
        ci = summ["contig_idx"]
        for i, n in enumerate(np.bincount(ci[ci >= 0], minlength=len(self.contig_names))):
            if n:
                self.read_counts[self.contig_names[i]] += int(n)
        sel = slice(None)
        if starts_filter is not None:
            sel = np.array([starts_filter(summ["ids"][i]) for i in summ["read_idx"]], dtype=bool)
        self.read_starts.count_starts(self.contig_names, summ["contig_idx"][sel], summ["rev"][sel],
                                      summ["tstart"][sel], summ["tend"][sel])

    def launch_benefit(self):
        """Enqueue calc_smu/calc_u as soon as the read-length windows are known (the caller has
        already updated `rl_dist`, as boss/core.py:106 does before process_batch_runs)."""
        if hasattr(self.rl_dist, "time_cost"):
            windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
            self.engine.update_benefit(windows, MULT)

    # ---- the strategy update --------------------------------------------------------------
    def _update_scores_contigs(self) -> None:
        """core.py:90-99 (+ the covsum passes of check_buckets and the bin sums of calc_smu):
        one fused device sweep."""
        self.engine.sweep()

    def _check_buckets_contigs(self) -> bool:
        """core.py:102-111."""
        thr = self.args.optional.bucket_threshold
        for cont in self.contigs_filt.values():
            cont.check_buckets(self.engine.bucket_sums(cont.index), threshold=thr)
        return any(any(c.switched_on) for c in self.contigs.values())

    def update_wrapper(self, between=None) -> None:
        """core.py:160-198, enqueued as one fused device update (bossx_update): scores +
        dropout + bucket sums + bin sums, bucket switches, and — once any strategy is switched
        on — benefits, threshold search, masks."""
        eng = self.engine
        thr = self.args.optional.bucket_threshold
        armed = any(any(c.switched_on) for c in self.contigs.values())
        # Before anything is switched on the strategy stages are skipped on the device as well;
        # they are still enqueued (gated) so that the update in which the first bucket flips
        # produces masks, as in the reference.
        have_rl = hasattr(self.rl_dist, "time_cost")
        use_bits = self._fused and self.mask_format != "npz"   # masks come back packed 8:1 (masks.py)
        if have_rl:
            windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
            if self.read_starts._engine is not None:
                # update_f_pointmass on the device, from the counts resident there
                res = eng.update(thr, windows, MULT, tc=self.rl_dist.time_cost // 100,
                                 fhat_model=self.read_starts.fhat_model(), want_stats=self.keep_stats, bits=use_bits, between=between)
            else:
                fhat_c, target_rs = self.read_starts.fhat_compact()
                res = eng.update(thr, windows, MULT, tc=self.rl_dist.time_cost // 100, fhat_c=fhat_c,
                                 target_rs=target_rs, want_stats=self.keep_stats, bits=use_bits, between=between)
        else:
            res = eng.update(thr, between=between)
        for cname, cont in self.contigs_filt.items():
            if res["contig_on"][cont.index] and not all(cont.switched_on):
                cont.switched_on[:] = True
                logging.info(f"Activated strategy for: {cname}")
        switched_on = armed or res["any_on"]
        if not switched_on:
            if res.get("between_error") is not None:
                raise res["between_error"]
            return
        if not have_rl:
            # the reference raises AttributeError here (readlengthdist.py:68)
            raise AttributeError("'ReadlengthDist' object has no attribute 'time_cost'")
        self.threshold = res["threshold"]
        self.last_stats = dict(normaliser=res["normaliser"], ubar0=res["ubar0"],
                               strat_size=res["strat_size"], n_bins=res["n_bins"],
                               argmax_margin=res.get("argmax_margin"), thr_code=res.get("thr_code"))
        tie = res["updated"] and res.get("argmax_margin") is not None and res["argmax_margin"] < self.tie_margin
        if tie:
            self._resolve_near_tie()
            use_bits = False           # (the masks were re-formed: take them from the device, as bytes)
        if self.keep_stats:
            counts = res["counts"]
            uniq = np.nonzero(counts)[0]
            self.last_stats.update(exponents=uniq, counts=counts[uniq],
                                   f_grid=np.array([fx_to_float(*res["fgrid_fx"][e]) for e in uniq]))
        for cname, cont in self.contigs_filt.items():
            if tie:
                cont.strat = eng.get_strat(cont.index)
            elif use_bits:
                cont.strat = self._lazy_strat(cont)          # unpacked on first access
            else:
                cont.strat = eng.strat_view(cont.index)
            if self.log_fractions:
                f_perc = np.count_nonzero(cont.strat[:, 0]) / cont.strat.shape[0]
                r_perc = np.count_nonzero(cont.strat[:, 1]) / cont.strat.shape[0]
                logging.info(f'{cname}: {f_perc}, {r_perc}')
        if self.write_masks:
            self._publish_masks(bits=eng.strat_bits if use_bits else None)
        if res.get("between_error") is not None:
            raise res["between_error"]

    def _resolve_near_tie(self):
        """The device chose the threshold bin from exact sums and reports that the best and the second-best
        cs_u / cs_t lie within `tie_margin` of each other: the reference's float sums (sequences.py:609-636)
        may round the other way, and a flipped argmax halves or doubles the threshold.  Re-derive the choice
        on the host in the reference's own summation order from the exported benefits and the expanded
        read-start posterior; if it differs, re-form the masks with that threshold (double compare)."""
        eng = self.engine
        benefit = np.concatenate([eng.export(c.index, "benefit") for c in self.contigs_filt.values()])
        target = self.ref.n_sites // 100
        fhat = self.read_starts.fhat_expanded_reference_order()            # (normalised as the reference normalises it: to the last bit)
        fhat = np.repeat(fhat[:, :, np.newaxis], self.nbarcodes, axis=2)
        threshold, size, margin = reference_order_threshold(_adjust_length(target, benefit), _adjust_length(target, fhat), self.rl_dist.time_cost)
        self.ties_resolved += 1
        self.last_stats.update(tie_resolved=True, device_threshold=self.threshold, device_strat_size=self.last_stats["strat_size"],
                               reference_order_margin=margin)
        logging.info(f"argmax within {self.last_stats['argmax_margin']:.3g} of a tie: threshold re-derived in the reference's "
                     f"summation order ({self.threshold} -> {threshold})")
        self.threshold = threshold
        self.last_stats["strat_size"] = size
        eng.apply_threshold(threshold)       # (same rows, same bucket gate: overwrites what the device-picked threshold wrote)

    def _lazy_strat(self, cont):
        bits, off = self.engine.strat_bits, self.engine.strat_offset(cont.index)
        rows, nb = cont.length // 100, self.nbarcodes

        def unpack():
            n = rows * 2 * nb
            flat = np.unpackbits(bits[off >> 3: (off + n + 7) >> 3])[(off & 7):(off & 7) + n]
            return flat.view(np.bool_).reshape(rows, 2, nb)
        return unpack

    def update_wrapper_staged(self) -> None:
        """The same update through the stage-wise C-ABI (sweep / bucket sums / benefit /
        histogram / apply_threshold) with the bucket and threshold decisions on the host —
        the form the multi-GPU protocol interleaves with collectives."""
        self._update_scores_contigs()
        switched_on = self._check_buckets_contigs()
        if not switched_on:
            return
        fhat_c, target_rs = self.read_starts.fhat_compact()
        windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
        normaliser = self.engine.benefit(windows, MULT)              # _update_benefits
        target = self.ref.n_sites // 100
        counts, fg, ub = self.engine.histogram(normaliser, fhat_c, target_rs, target)
        fgrid = np.array([fx_to_float(lo, hi) for lo, hi in fg])
        ubar0 = ubar_to_float(fx_to_float(ub[0], ub[1]), normaliser)
        threshold, size, uniq = choose_threshold(normaliser, counts, fgrid, ubar0, self.rl_dist.time_cost)
        self.threshold = threshold
        self.last_stats = dict(normaliser=normaliser, exponents=uniq, counts=counts[uniq],
                               f_grid=fgrid[uniq], ubar0=ubar0, strat_size=size)
        self.engine.apply_threshold(threshold)                       # find_strat + _distribute_strategy
        for cname, cont in self.contigs_filt.items():
            cont.strat = self.engine.get_strat(cont.index)
        if self.write_masks:
            self._publish_masks(bits=None)

    update_strategy = update_wrapper      # name used by BASELINE.json's north_star

    # ---- checkpoint / resume (absent in the reference: all its state lives in memory)
    def save_state(self, path) -> None:
        """Everything the next update depends on, as one .npz: per-contig coverage, site state,
        entropy, bucket switches and masks (exported from the device), read-start counts, the
        read-length histogram and the counters."""
        d = dict(batch=np.array(self.batch), total_reads=np.array(self.total_reads),
                 threshold=np.array(np.nan if self.threshold is None else self.threshold),
                 read_lengths=self.rl_dist.read_lengths, rl_hi=np.array(getattr(self.rl_dist, "_hi", 0)),
                 read_counts=np.array([self.read_counts[n] for n in self.contig_names], dtype=np.int64),
                 contig_names=np.array(self.contig_names), nbarcodes=np.array(self.nbarcodes))
        for n, c in self.contigs_filt.items():
            d["cov_" + n] = c.coverage
            d["state_" + n] = self.engine.export(c.index, "state")
            if self.args.gpu.track_entropy:
                d["ent_" + n] = c.entropy
            d["buckets_" + n] = c.bucket_switches
            d["strat_" + n] = np.ascontiguousarray(c.strat)
            d["on_" + n] = c.switched_on
            d["starts_" + n] = self.read_starts.read_starts[n]
        np.savez_compressed(path, **d)

    def load_state(self, path) -> None:
        """Inverse of save_state on a freshly `init()`-ed object with the same reference."""
        z = np.load(path)
        if list(z["contig_names"]) != self.contig_names or int(z["nbarcodes"]) != self.nbarcodes:
            raise ValueError("checkpoint does not match this reference / barcode set")
        self.batch = int(z["batch"])
        self.total_reads = int(z["total_reads"])
        thr = float(z["threshold"])
        self.threshold = None if np.isnan(thr) else thr
        for n, v in zip(self.contig_names, z["read_counts"]):
            self.read_counts[n] = int(v)
        self.rl_dist.read_lengths[:] = z["read_lengths"]
        self.rl_dist._hi = int(z["rl_hi"])
        self.rl_dist.update(np.zeros(0, dtype=np.int64))          # recompute lam / approx_ccl / time_cost
        for n, c in self.contigs_filt.items():
            self.engine.import_state(c.index, "coverage", z["cov_" + n])
            self.engine.import_state(c.index, "state", z["state_" + n])
            if self.args.gpu.track_entropy and "ent_" + n in z.files:
                self.engine.import_state(c.index, "entropy", z["ent_" + n])
            self.engine.import_state(c.index, "bucket_switches", z["buckets_" + n])
            self.engine.import_state(c.index, "strat", z["strat_" + n])
            c.strat = z["strat_" + n].astype(bool)
            c.switched_on[:] = z["on_" + n]
            self.read_starts.read_starts[n][:] = z["starts_" + n]
        self.read_starts.resync_engine()

