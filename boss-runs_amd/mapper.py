"""Upstream of the decision-update path.  The minimap2 wrapper (`Mapper`, /root/reference/boss/mapper.py:27-125)
is out of scope: `BossRuns.init(mapper=...)` takes the reference's own object (or anything
with its surface: `mu` and `_mappy_batch` / `map_sequences`).  So that the documented flow — `exp.init();
exp.process_batch(exp.process_batch_runs)` — works where nothing is passed, `default_mapper` tries, in this
order, the reference's class (when the `boss` package is importable) and a few lines around `mappy.Aligner`
that produce the same PAF text (`MappyText`); without mappy it returns None and `init()` says so.
Besides: the adapter for mappers that only hand out the reference's dict of PafLine records."""
import logging
from pathlib import Path


class MappyText:
    """mappy.Aligner(preset="map-ont") behind the two members the path reads: `mu` and
    `_mappy_batch(sequences) -> PAF text` (one line per hit: read id, read length, str(hit) —
    what boss/mapper.py:100-127 joins)."""

    def __init__(self, ref, mu=400):
        import mappy
        if not Path(ref).is_file():
            raise FileNotFoundError("Given reference file does not exist")
        self.mu = mu
        self._mappy = mappy
        self.aligner = mappy.Aligner(fn_idx_in=ref, preset="map-ont")

    def _mappy_batch(self, sequences, out=None, log=True):
        buf = self._mappy.ThreadBuffer()
        lines = ["%s\t%d\t%s" % (rid, len(seq), hit) for rid, seq in sequences.items() for hit in self.aligner.map(seq, buf=buf)]
        return "\n".join(lines)


def default_mapper(mmi, mu=400):
    """What BossRuns.init() attaches when the caller passes no mapper (boss/runs/core.py:41-43 builds
    `Mapper(ref=self.ref.mmi)` unconditionally): None when there is no index path or no mappy."""
    if not mmi:
        return None
    try:
        from boss.mapper import Mapper          # the reference's own class, where its package is installed
        return Mapper(ref=mmi, mu=mu)
    except ImportError:
        pass
    try:
        return MappyText(mmi, mu=mu)
    except ImportError:
        logging.warning("mappy is not installed: no mapper attached — pass one to init(mapper=...) "
                        "or feed PAF text to process_batch_paf()")
        return None


def paf_dict_to_text(paf_dict):
    """{read id: [PafLine-like]} (what `Mapper.map_sequences` returns, mapper.py:52-65) back to
    PAF text with the tags the path reads (tp, cg, AS, s1).  The records were already filtered by
    parse_PAF (primary, min_len); re-parsing the text picks the same best mapping per read."""
    lines = []
    for recs in paf_dict.values():
        for r in recs:
            cols = [r.qname, r.qlen, r.qstart, r.qend, r.strand, r.tname, r.tlen, r.tstart, r.tend,
                    r.num_matches, r.alignment_block_length, r.mapq]
            tags = ["tp:A:%s" % ("P" if getattr(r, "primary", 1) else "S")]
            if getattr(r, "cigar", None) is not None:
                tags.append("cg:Z:%s" % r.cigar)
            tags.append("AS:i:%d" % int(getattr(r, "align_score", 0)))
            tags.append("s1:i:%d" % int(getattr(r, "s1", 0) or 0))
            lines.append("\t".join(str(c) for c in cols) + "\t" + "\t".join(tags))
    return "\n".join(lines)
