"""PAF records, best-mapper choice and CIGAR -> per-reference-base observations.

Follows /root/reference/boss/paf.py:12-75 (PafLine), :631-672 (Paf.parse_PAF /
_parse_content), :709-722 (choose_best_mapper) and
/root/reference/boss/runs/sequences.py:678-794 (CoverageConverter).
"""
import re
from collections import defaultdict
from io import StringIO

import numpy as np

_INT_FIELDS = ('qlen', 'qstart', 'qend', None, None, 'tlen', 'tstart', 'tend',
               'num_matches', 'alignment_block_length', 'mapq')


def _maybe_int(s):
    try:
        return int(s)
    except ValueError:
        return s


class PafRec:
    """One PAF line: 12 core columns + tags AS, cg, s1, tp (paf.py:18-75)."""

    def __init__(self, line):
        rec = line.strip().split("\t")
        self.qname = str(_maybe_int(rec[0]))
        self.qlen = _maybe_int(rec[1])
        self.qstart = _maybe_int(rec[2])
        self.qend = _maybe_int(rec[3])
        self.strand = rec[4]
        self.tname = str(_maybe_int(rec[5]))
        self.tlen = _maybe_int(rec[6])
        self.tstart = _maybe_int(rec[7])
        self.tend = _maybe_int(rec[8])
        self.num_matches = _maybe_int(rec[9])
        self.alignment_block_length = _maybe_int(rec[10])
        self.mapq = _maybe_int(rec[11])
        self.rev = 0 if self.strand == '+' else 1
        tags = {}
        for x in rec[12:]:
            key, typ, val = x.split(":")
            conv = {"i": int, "A": str, "f": float, "Z": str}[typ]
            try:
                tags[key] = conv(val)
            except ValueError:
                tags[key] = val
        self.align_score = int(tags.get("AS", 0))
        self.cigar = tags.get("cg", None)
        self.s1 = tags.get("s1", 0)
        self.primary = 1 if tags.get("tp", None) == 'P' else 0
        self.barcode = None


def parse_paf(text, min_len=1):
    """PAF text -> {qname: [PafRec]} keeping primary records with block length >= min_len
    (paf.py:654-672)."""
    out = defaultdict(list)
    for line in StringIO(text):          # as the reference reads it (mapper.py:64): lines end at '\n' only
        rec = PafRec(line)               # (a blank line is an IndexError there, paf.py:50-51)
        if rec.alignment_block_length < min_len:
            continue
        if not rec.primary:
            continue
        out[rec.qname].append(rec)
    return out


def best_mapper(records):
    """paf.py:709-722: argsort of (mapq, AS) structured array, last one wins."""
    if len(records) == 1:
        return records[0]
    mq = np.array([(r.mapq, r.align_score) for r in records], dtype=[('q', int), ('dp', int)])
    order = np.argsort(mq, order=["q", "dp"])
    return records[order[-1]]


_COMP = str.maketrans('ATGC', 'TACG')           # utils.py:93
_BASE2INT = str.maketrans({'A': '0', 'C': '1', 'G': '2', 'T': '3'})   # sequences.py:666-667
_CIG2INT = str.maketrans({'M': '6', 'D': '7', 'I': '8', 'S': '9'})    # sequences.py:669-670
_CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")                        # sequences.py:672


def expand_cigar(cigar, read, start, end):
    """sequences.py:744-794 without the quality track (qt=0 => `addition` is all ones,
    sequences.py:735-736): returns uint8[tend-tstart] with codes 0..3 = ACGT, 4 = deletion."""
    int_seq = np.frombuffer(read.translate(_BASE2INT).encode(), 'u1') - ord('0')
    parts = _CIGAR_RE.findall(cigar)
    lengths, ops = zip(*parts)
    lengths = np.array(lengths, dtype=np.uint32)
    ops_arr = np.frombuffer(''.join(ops).translate(_CIG2INT).encode(), 'u1') - ord('0')
    rep = np.repeat(ops_arr, lengths)
    notins = np.where(rep != 8)
    notdel = np.where(rep != 7)
    rep[notdel] = int_seq[start:end]
    q = rep[notins]
    q[q == 7] = 4
    return q


def convert_records(paf_dict, seqs):
    """sequences.py:678-739: {tname: [(tstart, tend, query_codes, barcode)]} in read order."""
    inc = defaultdict(list)
    for rid in list(paf_dict.keys()):
        recs = paf_dict[rid]
        rec = best_mapper(recs) if len(recs) > 1 else recs[0]
        if rec.rev:
            seq = seqs[rec.qname].translate(_COMP)[::-1]
            qs, qe = rec.qlen - rec.qend, rec.qlen - rec.qstart
        else:
            seq = seqs[rec.qname]
            qs, qe = rec.qstart, rec.qend
        assert rec.cigar is not None
        q = expand_cigar(rec.cigar, seq, qs, qe)
        start = min(rec.tstart, rec.tend)
        end = max(rec.tstart, rec.tend)
        assert (end - start) == q.shape[0]
        inc[rec.tname].append((start, end, q, rec.barcode))
    return inc
