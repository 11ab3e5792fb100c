"""Test double for `boss_runs_amd.engine.Engine` with the oracle's numerics (TEST
INFRASTRUCTURE).  It lets the multi-rank protocol in `boss_runs_amd.parallel` run on CPUs
(gloo, world_size 2) where no HIP device exists: same stage-wise interface, same ownership
rules for remote contigs, same exact 128-bit statistics."""
import math

import numpy as np

from oracle.contig import OContig
from oracle.model import SiteModel, PatternCache
from oracle.pafcigar import parse_paf, convert_records, best_mapper

FX = 100


def _fx(v):
    """float64 >= 0 -> integer floor(v * 2^100) exactly as the device truncates."""
    if v <= 0.0:
        return 0
    m, e = np.frexp(v)
    mant = int(m * (1 << 53))
    sh = int(e) - 53 + FX
    return mant << sh if sh >= 0 else mant >> (-sh)


class FakeEngine:
    def __init__(self, nbarcodes=1, ploidy=1, **kw):
        self.nb = nbarcodes
        self.ploidy = ploidy
        self.names, self.lengths, self.rejected, self.remote = [], [], [], []
        self.contigs = []          # OContig or None (rejected / remote)

    def add_contig(self, name, seq, rejected=False, remote_length=None):
        self.names.append(name)
        self.rejected.append(bool(rejected))
        self.remote.append(remote_length is not None and not rejected)
        if rejected:
            self.lengths.append(4); self.contigs.append(None)
        elif remote_length is not None:
            self.lengths.append(int(remote_length)); self.contigs.append(None)
        else:
            self.lengths.append(len(seq)); self.contigs.append(OContig(name, seq, nbarcodes=self.nb))
        return len(self.names) - 1

    def finalize(self, score0, ent0):
        self.cache = PatternCache(SiteModel(self.ploidy))
        self.filt = [i for i, r in enumerate(self.rejected) if not r]
        T = [self.lengths[i] // 100 for i in self.filt]
        self.row_off = np.concatenate(([0], np.cumsum(T))).astype(np.int64)
        self.bin_off = np.concatenate(([0], np.cumsum(np.array(T) + 1))).astype(np.int64)
        self.B = int(self.bin_off[-1])
        self.n_sites = int(sum(self.lengths))
        self.pending = {i: [] for i in self.filt}

    def set_lut(self, score, entropy):
        pass

    def ingest_paf(self, paf_text, seqs, barcodes=None, min_len=200):
        paf = parse_paf(paf_text, min_len=min_len)
        ids = list(seqs.keys())
        pos = {r: i for i, r in enumerate(ids)}
        name_idx = {n: i for i, n in enumerate(self.names)}
        rec_list = []
        for rid, recs in paf.items():
            rec = best_mapper(recs) if len(recs) > 1 else recs[0]
            if barcodes is not None:
                rec.barcode = barcodes[rec.qname]
            rec_list.append(rec)
        local_names = {self.names[i] for i in self.filt if not self.remote[i]}
        sub = {r.qname: [r] for r in rec_list if r.tname in local_names}
        inc = convert_records(sub, seqs)
        for tname, lst in inc.items():
            self.pending[name_idx[tname]].extend(lst)
        return dict(read_idx=np.array([pos[r.qname] for r in rec_list], dtype=np.int32),
                    contig_idx=np.array([name_idx.get(r.tname, -1) for r in rec_list], dtype=np.int32),
                    rev=np.array([r.rev for r in rec_list], dtype=np.uint8),
                    tstart=np.array([r.tstart for r in rec_list], dtype=np.int64),
                    tend=np.array([r.tend for r in rec_list], dtype=np.int64),
                    qlen=np.array([r.qlen for r in rec_list], dtype=np.int64), ids=ids, aligned=0)

    # ---- the pieces simulation.py uses: mapping choice only, and stage / ingest in two steps ----
    def paf_summary(self, paf_text, read_ids, min_len=1):
        if isinstance(paf_text, (bytes, bytearray)):
            paf_text = paf_text.decode()
        paf = parse_paf(paf_text, min_len=min_len)
        ids = list(read_ids)
        pos = {r: i for i, r in enumerate(ids)}
        name_idx = {n: i for i, n in enumerate(self.names)}
        recs = [best_mapper(r) if len(r) > 1 else r[0] for r in paf.values()]
        return dict(read_idx=np.array([pos[r.qname] for r in recs], dtype=np.int32),
                    contig_idx=np.array([name_idx.get(r.tname, -1) for r in recs], dtype=np.int32),
                    rev=np.array([r.rev for r in recs], dtype=np.uint8),
                    tstart=np.array([r.tstart for r in recs], dtype=np.int64),
                    tend=np.array([r.tend for r in recs], dtype=np.int64),
                    qlen=np.array([r.qlen for r in recs], dtype=np.int64), ids=ids)

    def stage_batch(self, paf_text, seqs, barcodes=None, min_len=200, **kw):
        if isinstance(paf_text, (bytes, bytearray)):
            paf_text = paf_text.decode()
        if barcodes is not None and not isinstance(barcodes, dict):
            barcodes = dict(zip(seqs.keys(), [int(b) for b in barcodes]))
        saved = self.pending
        self.pending = {i: [] for i in self.filt}
        summ = self.ingest_paf(paf_text, seqs, barcodes=barcodes, min_len=min_len)
        self._staged, self.pending = self.pending, saved
        return summ

    def ingest_staged(self, slot=None):
        for i, lst in self._staged.items():
            self.pending[i].extend(lst)
        self._staged = {i: [] for i in self.filt}

    def _local(self):
        return [(i, self.contigs[i]) for i in self.filt if not self.remote[i]]

    def sweep(self):
        for i, c in self._local():
            c.increment_coverage(self.pending[i])
            self.pending[i] = []
            c.update_scores(self.cache)
            c.modify_scores()

    def bucket_sums(self, contig):
        c = self.contigs[contig]
        n = c.length // 20000
        out = np.zeros((self.nb, n), dtype=np.uint64)
        for b in range(self.nb):
            cs = np.sum(c.coverage[:, :, b], axis=1)
            out[b] = cs[: n * 20000].reshape(-1, 20000).sum(axis=1)
        return out

    def set_bucket_switches(self, contig, sw):
        self.contigs[contig].bucket_switches = np.asarray(sw, dtype=bool).copy()

    def benefit(self, windows, mult):
        windows = np.asarray(windows)
        assert windows[0] == 4
        for i in self.filt:
            n = self.lengths[i] // 100 + 1
            for w in windows:
                if w < 1 or w > n:
                    raise ValueError("Moving window (=%d) must between 1 and %d, inclusive" % (w, n))
        limit = min(self.B, self.n_sites // 100)
        mx = 0.0
        for k, i in enumerate(self.filt):
            if self.remote[i]:
                continue
            c = self.contigs[i]
            c.calc_smu()
            c.calc_u(windows[1:] * 100)
            g = self.bin_off[k] + np.arange(c.additional_benefit.shape[0])
            v = c.additional_benefit[g < limit]
            if v.size:
                mx = max(mx, float(v.max()))
        return mx

    def histogram(self, normaliser, fhat_c, target_rs, target):
        if not normaliser > 0:
            raise ValueError("no non-zero benefit")
        counts = np.zeros(1088, dtype=np.int64)
        fg = [0] * 1088
        ub = 0
        norm_e = math.frexp(float(normaliser))[1]
        fexp = fhat_c.shape[0] * 20
        d1 = max(target_rs - fexp, 0)
        d2 = max(target - target_rs, 0)
        dpad = max(target - self.B, 0)
        owner = np.full(self.B, -1, dtype=np.int64)
        for k, i in enumerate(self.filt):
            if not self.remote[i]:
                owner[self.bin_off[k]:self.bin_off[k + 1]] = k
        for g in range(target):
            src = g if g < self.B else g - dpad
            k = owner[src]
            if k < 0:
                continue
            c = self.contigs[self.filt[k]]
            g1 = g - d2 if g >= target_rs else g
            g2 = g1 - d1 if g1 >= fexp else g1
            for s in (0, 1):
                fh = float(fhat_c[g2 // 20, s])
                for b in range(self.nb):
                    x = float(c.additional_benefit[src - self.bin_off[k], s, b])
                    ub += _fx(math.ldexp(fh * x, -norm_e))      # the engine's convention: relative to the normaliser's binade
                    if x != 0.0:
                        e = abs(int(np.frexp(x / normaliser)[1]))
                        counts[e] += 1
                        fg[e] += _fx(fh)
        m = (1 << 64) - 1
        fga = np.array([[v & m, v >> 64] for v in fg], dtype=np.uint64)
        return counts, fga, np.array([ub & m, ub >> 64], dtype=np.uint64)

    def apply_threshold(self, threshold):
        self.threshold = threshold
        for k, i in enumerate(self.filt):
            if self.remote[i]:
                continue
            c = self.contigs[i]
            T = c.length // 100
            sw = c.bucket_switches
            for r in range(T):
                g = self.row_off[k] + r
                j = int(np.searchsorted(self.bin_off, g, side="right") - 1)
                if self.remote[self.filt[j]]:
                    continue
                row = self.contigs[self.filt[j]].additional_benefit[g - self.bin_off[j]] >= threshold
                on = sw[r // 200]
                c.strat[r][:, on] = row[:, on]

    def get_strat(self, contig, out=None):
        if self.rejected[contig]:
            return np.zeros(1, dtype=bool)
        return self.contigs[contig].strat.copy()

    def export(self, contig, which):
        c = self.contigs[contig]
        if which == "benefit_tail":
            k = min(c.additional_benefit.shape[0], len(self.filt))
            return c.additional_benefit[-k:].copy()
        if which == "bucket_switches":
            return c.bucket_switches.copy()
        return {"coverage": c.coverage, "scores": c.scores, "entropy": c.entropy}[which].copy()
