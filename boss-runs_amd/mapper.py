"""Upstream of the decision-update path: the minimap2 wrapper (`Mapper`, /root/reference/boss/mapper.py:27-125)
is out of scope and is NOT rebuilt here — `BossRuns.init(mapper=...)` takes the
reference's own object (or anything with its surface: `mu` and `_mappy_batch` / `map_sequences`).
What remains is the adapter for mappers that only hand out the reference's dict of PafLine records."""


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
