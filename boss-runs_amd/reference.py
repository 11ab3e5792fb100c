"""Host-side mirrors of `Reference` and `Contig` (boss/runs/reference.py:18-373).

A `Contig` here is a thin view: the per-site arrays live in HBM inside the engine; the
attributes the reference exposes (`coverage`, `scores`, `entropy`, `scores_ds`,
`additional_benefit`) are materialised on access in the reference's layout.  `strat`
and `switched_on` are real numpy arrays kept current by `BossRuns`.
"""
import gzip
import logging
from pathlib import Path

import numpy as np


def read_fasta(path):
    """Yield (name_first_token, sequence) from a (gzipped) FASTA; stands in for
    mappy.fastx_read at reference.py:326."""
    opener = gzip.open if str(path).endswith(".gz") else open
    name, chunks = None, []
    with opener(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    yield name, "".join(chunks)
                header = line[1:].split()
                name, chunks = (header[0] if header else ""), []
            elif line:
                chunks.append(line)
    if name is not None:
        yield name, "".join(chunks)


class Contig:
    def __init__(self, name, length, rej=False, barcodes=None, engine=None, index=-1, remote=False):
        self.name = name.strip().split(" ")[0]
        self.length = int(length)
        self.rej = rej
        self.barcodes = barcodes
        self.nbarcodes = len(barcodes) if barcodes is not None else 1
        self.bucket_size = 20_000
        self._engine = engine
        self.index = index
        self.remote = remote          # multi-GPU: per-site state lives on another rank
        nb = self.nbarcodes
        self.switched_on = np.zeros(nb, dtype="bool")
        if rej:
            self._strat = np.zeros(1, dtype="bool")                       # reference.py:116
        else:
            self._strat = np.ones((self.length // 100, 2, nb), dtype="bool")   # reference.py:118

    @property
    def strat(self):
        """bool[length//100, 2, nb]; with the bit-packed mask path the array is unpacked from
        the packed buffer on first access after an update."""
        if callable(self._strat):
            self._strat = self._strat()
        return self._strat

    @strat.setter
    def strat(self, value):
        self._strat = value

    # device-resident state, reference layout ------------------------------------------------
    def _export(self, which):
        if self.rej:
            raise AttributeError("rejected contigs hold no per-site state")
        if self.remote:
            raise AttributeError("contig %s is owned by another rank" % self.name)
        return self._engine.export(self.index, which)

    @property
    def bucket_switches(self):
        """bool[length // 20000 + 1, nb]; the switches live on the device (they gate the mask
        kernel), this reads them back."""
        if self.rej or self.remote or self._engine is None:
            return np.zeros((self.length // self.bucket_size + 1, self.nbarcodes), dtype="bool")
        return self._engine.export(self.index, "bucket_switches")

    @property
    def coverage(self):
        return self._export("coverage")

    @property
    def scores(self):
        return self._export("scores")

    @property
    def entropy(self):
        return self._export("entropy")

    @property
    def scores_ds(self):
        return self._export("scores_ds")

    @property
    def additional_benefit(self):
        return self._export("benefit")

    def check_buckets(self, bucket_sums, threshold=5.0):
        """Contig.check_buckets (reference.py:183-211) on the host, from the engine's 20-kb
        covsum totals (`bucket_sums` uint64[nb, length // 20000]); used by the stage-wise
        update.  The fused update takes the same decision on the device."""
        old = self.bucket_switches
        new = old.copy()
        for b in range(self.nbarcodes):
            means = np.divide(bucket_sums[b], self.bucket_size)
            d = new.shape[0] - means.shape[0]                       # utils.adjust_length
            if d > 0:
                means = np.append(means, means[-d:], axis=0)
            new[np.where(means >= threshold)[0], b] = True
            if new[:, b].any() and not all(self.switched_on):
                self.switched_on[np.logical_not(self.switched_on)] = True
                logging.info(f"Activated strategy for: {self.name}")
        if not np.array_equal(old, new):
            self._engine.set_bucket_switches(self.index, new)


class Reference:
    def __init__(self, ref, mmi=None, reject_refs=None, barcodes=None, engine=None, contigs=None,
                 is_local=None):
        """Loads the FASTA (or takes `contigs`: iterable of (name, sequence-or-length)),
        registers every contig >= 100 kb with the engine in file order (reference.py:305-338).
        `is_local(name, index_among_non_rejected) -> bool` marks contigs owned by another rank
        (multi-GPU); those may be given as a bare length."""
        self.ref, self.mmi, self.barcodes = ref, mmi, barcodes
        if contigs is None:
            if not Path(ref).is_file():
                raise FileNotFoundError("Reference file not found")
            if not any(r in {".fa", ".fasta"} for r in Path(ref).suffixes):
                raise ValueError("Reference needs to be either .fa or .fasta (optionally gzipped).")
            if self.mmi and not Path(self.mmi).is_file():
                raise FileNotFoundError("Given mmi file not found")
            contigs = read_fasta(ref)
        self.reject_refs = set(reject_refs.split(',')) if reject_refs else set()
        logging.info("Reading reference file")
        self.contigs = {}
        min_len = int(1e5)
        k = 0
        for cname, cseq in contigs:
            clen = int(cseq) if isinstance(cseq, (int, np.integer)) else len(cseq)
            if clen < min_len:
                continue                                            # reference.py:330-331
            if cname not in self.reject_refs:
                if is_local is not None and not is_local(cname, k):
                    idx = engine.add_contig(cname, None, remote_length=clen)
                    self.contigs[cname] = Contig(cname, clen, barcodes=barcodes, engine=engine, index=idx, remote=True)
                else:
                    if isinstance(cseq, (int, np.integer)):
                        raise ValueError("local contig %s needs its sequence" % cname)
                    idx = engine.add_contig(cname, cseq, rejected=False)
                    self.contigs[cname] = Contig(cname, clen, barcodes=barcodes, engine=engine, index=idx)
                k += 1
            else:
                idx = engine.add_contig(cname, None, rejected=True)
                self.contigs[cname] = Contig(cname, 4, rej=True, engine=engine, index=idx)
        self.n_sites = int(np.sum([c.length for c in self.contigs.values()]))

    def contig_lengths(self):
        return {c.name: c.length for c in self.contigs.values()}

    def get_strategy_dict(self):
        return {cname: cont.strat for cname, cont in self.contigs.items()}
