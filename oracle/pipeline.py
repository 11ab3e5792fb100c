"""End-to-end oracle of `BossRuns.process_batch_runs` minus the mapper
(/root/reference/boss/runs/core.py:23-224, reference.py:276-373)."""
import time
from contextlib import contextmanager

import numpy as np

from .contig import OContig, adjust_length
from .dists import OReadlengthDist, OReadStartDist
from .model import SiteModel, PatternCache
from .pafcigar import parse_paf, convert_records
from .strategy import find_strategy, distribute


class OracleRuns:
    def __init__(self, contigs, ploidy=1, reject_refs=(), nbarcodes=1, bucket_threshold=5,
                 min_len=int(1e5)):
        """`contigs`: iterable of (name, sequence) in FASTA order.  reference.py:305-338:
        sequences shorter than min_len are dropped before reject_refs is consulted."""
        self.nb = nbarcodes
        self.bucket_threshold = bucket_threshold
        self.contigs = {}
        for name, seq in contigs:
            if len(seq) < min_len:
                continue
            if name not in reject_refs:
                self.contigs[name] = OContig(name, seq, nbarcodes=nbarcodes)
            else:
                self.contigs[name] = OContig(name, "ACGT", rej=True)
        self.contigs_filt = {n: c for n, c in self.contigs.items() if not c.rej}
        self.n_sites = int(np.sum([c.length for c in self.contigs.values()]))   # reference.py:343-347
        self.rl_dist = OReadlengthDist()
        self.read_starts = OReadStartDist(self.contigs_filt)
        self.cache = PatternCache(SiteModel(ploidy))
        self.threshold = None
        self.detail = {}
        self.timings = None          # dict: per-stage wall-clock seconds of the last update (bench.py)

    @contextmanager
    def _stage(self, key):
        t0 = time.perf_counter()
        yield
        if self.timings is not None:
            self.timings[key] = self.timings.get(key, 0.0) + time.perf_counter() - t0

    def strategies(self):
        return {n: c.strat for n, c in self.contigs.items()}

    def process_batch(self, paf_text, seqs, read_lengths=None, barcodes=None, min_len=200,
                      starts_paf_filter=None):
        """core.py:202-224 with the PAF text standing in for mapper output (mapper.py:64,
        min_len = int(mu/2) = 200).  `read_lengths` feeds rl_dist.update (core.py:106)."""
        if read_lengths is None:
            read_lengths = {k: len(v) for k, v in seqs.items()}
        with self._stage("ReadlengthDist.update"):
            self.rl_dist.update(read_lengths)
        with self._stage("Paf.parse_PAF"):
            paf = parse_paf(paf_text, min_len=min_len)
            if barcodes is not None:
                for recs in paf.values():
                    for r in recs:
                        r.barcode = barcodes[r.qname]
        with self._stage("convert_records"):
            inc = convert_records(paf, seqs)
        with self._stage("increment_coverage"):
            for name, c in self.contigs_filt.items():
                c.increment_coverage(inc.get(name, []))
        with self._stage("count_read_starts"):
            paf_starts = paf if starts_paf_filter is None else {k: v for k, v in paf.items() if starts_paf_filter(k)}
            self.read_starts.count_read_starts(paf_starts)
        self.update_wrapper()

    def update_wrapper(self):
        """core.py:160-198."""
        with self._stage("update_scores+modify_scores"):
            for c in self.contigs_filt.values():
                c.update_scores(self.cache)
                c.modify_scores()
        with self._stage("check_buckets"):
            for c in self.contigs_filt.values():
                c.check_buckets(threshold=self.bucket_threshold)
        if not any(any(c.switched_on) for c in self.contigs.values()):
            return False
        with self._stage("update_f_pointmass"):
            fhat = self.read_starts.update_f_pointmass()
            fhat = np.repeat(fhat[:, :, np.newaxis], self.nb, axis=2)
        for c in self.contigs_filt.values():
            with self._stage("calc_smu"):
                c.calc_smu()
            with self._stage("calc_u"):
                c.calc_u(self.rl_dist.approx_ccl)
        with self._stage("merge+find_strat_thread"):
            benefit = np.concatenate([c.additional_benefit for c in self.contigs_filt.values()])
            target = self.n_sites // 100
            benefit_adj = adjust_length(target, benefit)
            smu_adj = adjust_length(target, benefit)          # core.py:182-183: built from benefit
            fhat_adj = adjust_length(target, fhat)
            assert fhat_adj.shape == benefit_adj.shape == smu_adj.shape
            self.detail = {}
            strat, self.threshold = find_strategy(benefit_adj, smu_adj, fhat_adj,
                                                  self.rl_dist.time_cost, detail=self.detail)
        with self._stage("_distribute_strategy"):
            distribute(self.contigs_filt, strat)
        return True
