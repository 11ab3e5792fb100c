"""Seeded synthetic references and PAF read batches (SURVEY.md §8d "Synthetic inputs").

Used by the parity tests, `bench.py` and the golden-vector generator; the reference has no
counterpart (its tests use real data from an un-vendored submodule).
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP_CODE = np.array([3, 2, 1, 0], dtype=np.uint8)

ECOLI_LEN = 4_641_652
CHR20_LEN, CHR21_LEN, MT_LEN = 64_444_167, 46_709_983, 16_569
GRCH38_LENS = [248_956_422, 242_193_529, 198_295_559, 190_214_555, 181_538_259, 170_805_979,
               159_345_973, 145_138_636, 138_394_717, 133_797_422, 135_086_622, 133_275_309,
               114_364_328, 107_043_718, 101_991_189, 90_338_345, 83_257_441, 80_373_285,
               58_617_616, 64_444_167, 46_709_983, 50_818_468, 156_040_895, 57_227_415]


def random_codes(length, rng):
    """uint8[length] of base codes 0..3 (uniform)."""
    return rng.integers(0, 4, size=length, dtype=np.uint8)


def codes_to_str(codes):
    return _ACGT[codes].tobytes().decode("ascii")


def make_reference(lengths, seed=1, names=None):
    """-> list of (name, uint8 code array).  i.i.d. uniform ACGT, default_rng(seed)."""
    rng = np.random.default_rng(seed)
    names = names or ["ctg%d" % (i + 1) for i in range(len(lengths))]
    return [(n, random_codes(L, rng)) for n, L in zip(names, lengths)]


def write_fasta(path, contigs, width=0):
    with open(path, "w") as fh:
        for name, codes in contigs:
            fh.write(">%s\n" % name)
            fh.write(codes_to_str(codes))
            fh.write("\n")


def _rle_cigar(ops):
    """ops: uint8 array of op codes (0=M,1=I,2=D) in alignment order -> 'cg' string."""
    if ops.shape[0] == 0:
        return ""
    brk = np.flatnonzero(np.diff(ops)) + 1
    starts = np.concatenate(([0], brk))
    lens = np.diff(np.concatenate((starts, [ops.shape[0]])))
    sym = "MID"
    return "".join("%d%s" % (l, sym[o]) for l, o in zip(lens.tolist(), ops[starts].tolist()))


def _truncated_record(ops, tstart, tend, fl, fr, rev, mu):
    """The mapping of the first `mu` bases of a read, cut out of its full alignment (columns
    `ops` in target order: 0 M, 1 I, 2 D).  '+': a prefix of the columns; '-': the read as sequenced
    is the reverse complement, so its first bases align to the END of the target span.  Returns
    (qstart, qend, tstart, tend, ops) or None when fewer than 100 aligned bases fall inside."""
    lead = fr if rev else fl
    need = mu - lead
    if need < 100:
        return None
    o = ops[::-1] if rev else ops
    qcons = np.cumsum(o != 2)
    if qcons[-1] < need:
        return None
    k = int(np.searchsorted(qcons, need, side="left"))
    part = o[:k + 1]
    while part.shape[0] and part[-1] != 0:          # end on a match column
        part = part[:-1]
    if part.shape[0] < 100:
        return None
    qc, tc = int((part != 2).sum()), int((part != 1).sum())
    if rev:
        return lead, lead + qc, tend - tc, tend, part[::-1]
    return lead, lead + qc, tstart, tstart + tc, part


def make_batch(contigs, n_reads, seed, mean_len=6000.0, min_len=900, max_len=60000,
               sub=0.04, dele=0.03, ins=0.02, nbarcodes=1, start_weights=None, flank=True,
               extras=True, prefix="r", trunc_mu=None):
    """One synthetic read batch.

    contigs: list of (name, codes); reads start uniformly over the concatenated genome
    (or with per-contig relative `start_weights`), length clip(Gamma(2, mean_len/2)),
    strand Bernoulli(.5), barcode uniform.  Substitutions / deletions / insertions are
    encoded in `cg:Z:`.  With `extras`, a few records exercise the filters: secondary
    (`tp:A:S`) lines, short (< 200) alignments and two-way multi-mappers.

    Returns dict(paf=str, seqs={id: str}, barcodes={id: int}, read_lengths={id: int},
    aligned=int) where `aligned` counts emitted reference bases of the chosen mappings.

    With `trunc_mu` (simulation mode, boss/runs/simulation.py) the batch also carries `paf_trunc`:
    the mappings of every read's first `trunc_mu` bases (qlen = trunc_mu), cut out of the full
    alignments; every 29th read has no truncated mapping, every 31st none at all.
    """
    rng = np.random.default_rng(seed)
    lens = np.array([c[1].shape[0] for c in contigs], dtype=np.int64)
    w = lens.astype(np.float64) if start_weights is None else np.asarray(start_weights, float) * lens
    w = w / w.sum()
    lines, seqs, bcs, rls = [], {}, {}, {}
    tlines = []
    aligned = 0
    for i in range(n_reads):
        rid = "%s%d_%d" % (prefix, seed, i)
        ci = int(rng.choice(len(contigs), p=w))
        name, ref = contigs[ci]
        L = int(lens[ci])
        rl = int(np.clip(rng.gamma(2.0, mean_len / 2.0), min_len, max_len))
        tstart = int(rng.integers(0, L - 1))
        tend = min(L, tstart + rl)
        if extras and i % 97 == 13:
            tend = min(tend, tstart + 150)            # short alignment: dropped by min_len=200
        n = tend - tstart
        if n < 2:
            continue
        seg = ref[tstart:tend]
        is_del = rng.random(n) < dele
        is_del[0] = is_del[-1] = False
        is_sub = (rng.random(n) < sub) & ~is_del
        has_ins = rng.random(n) < ins
        has_ins[-1] = False
        has_ins &= ~is_del
        obs = seg.copy()
        obs[is_sub] = (obs[is_sub] + rng.integers(1, 4, size=int(is_sub.sum()), dtype=np.uint8)) % 4
        # alignment columns: each ref position emits M or D, optionally followed by one I
        ncol = n + int(has_ins.sum())
        col_of_ref = np.arange(n) + np.concatenate(([0], np.cumsum(has_ins)[:-1]))
        ops = np.ones(ncol, dtype=np.uint8)            # default I
        ops[col_of_ref] = np.where(is_del, 2, 0)
        qcols = ops != 2
        qbases = np.empty(ncol, dtype=np.uint8)
        qbases[col_of_ref] = obs
        ins_cols = np.flatnonzero(ops == 1)
        qbases[ins_cols] = rng.integers(0, 4, size=ins_cols.shape[0], dtype=np.uint8)
        query = qbases[qcols]
        fl = int(rng.integers(0, 30)) if flank else 0
        fr = int(rng.integers(0, 30)) if flank else 0
        full = np.concatenate((random_codes(fl, rng), query, random_codes(fr, rng)))
        qlen = int(full.shape[0])
        rev = bool(rng.random() < 0.5)
        if rev:
            read = _COMP_CODE[full][::-1]
            qstart, qend = fr, qlen - fl
        else:
            read = full
            qstart, qend = fl, qlen - fr
        cigar = _rle_cigar(ops)
        nmatch = int(n - is_del.sum() - is_sub.sum())
        tp = "P"
        if extras and i % 101 == 7:
            tp = "S"                                   # secondary only: read is dropped
        bc = int(rng.integers(0, nbarcodes)) if nbarcodes > 1 else 0
        seqs[rid] = codes_to_str(read)
        bcs[rid] = bc
        rls[rid] = qlen
        core = "%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d" % (
            rid, qlen, qstart, qend, "-" if rev else "+", name, L, tstart, tend, nmatch, ncol, 60)
        unmapped = trunc_mu is not None and i % 31 == 5          # simulation: no mapping at all
        if not unmapped:
            lines.append("%s\ttp:A:%s\tcm:i:%d\ts1:i:%d\tdv:f:0.0700\tcg:Z:%s\tAS:i:%d" %
                         (core, tp, n // 10, n, cigar, n))
        if trunc_mu is not None and not unmapped and i % 29 != 3:
            tr = _truncated_record(ops, tstart, tend, fl, fr, rev, int(trunc_mu))
            if tr is not None:
                tq0, tq1, tt0, tt1, tops = tr
                tn = tt1 - tt0
                tlines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\ttp:A:%s\tcm:i:%d\ts1:i:%d\tcg:Z:%s\tAS:i:%d" %
                              (rid, int(trunc_mu), tq0, tq1, "-" if rev else "+", name, L, tt0, tt1,
                               int((tops == 0).sum()), tops.shape[0], 60, tp, tn // 10, tn, _rle_cigar(tops), tn))
        chosen = tp == "P" and ncol >= 200 and not unmapped
        if extras and i % 53 == 5 and n > 400 and not unmapped:
            # a second, worse primary-tagged mapping (lower mapq): best_mapper must skip it
            other = contigs[(ci + 1) % len(contigs)]
            o_start = int(rng.integers(0, other[1].shape[0] - n))
            lines.append("%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\ttp:A:P\tcg:Z:%dM\tAS:i:%d" %
                         (rid, qlen, qstart, qstart + 250, "+", other[0], other[1].shape[0],
                          o_start, o_start + 250, 200, 250, 3, 250, 100))
        if chosen:
            aligned += n
    out = dict(paf="\n".join(lines), seqs=seqs, barcodes=bcs, read_lengths=rls, aligned=aligned)
    if trunc_mu is not None:
        out["paf_trunc"] = "\n".join(tlines)
    return out
