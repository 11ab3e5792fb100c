"""Upstream of the decision-update path: the minimap2 wrapper the reference's loop calls first
(`Mapper`, /root/reference/boss/mapper.py:27-125).  Mapping itself is out of scope;
this thin class only exists so that `BossRuns.init()` can build the same object the reference
builds (core.py:41-42) when `mappy` is installed, and so that `process_batch_runs` has a
reference-shaped mapper to talk to.  What the GPU path needs from a mapper is the raw PAF text of a
batch (`_mappy_batch`) and `mu` (min_len = int(mu / 2), mapper.py:64): any object with those works.
"""
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path


class Mapper:
    def __init__(self, ref, mu=400, workers=4, default=True):
        try:
            import mappy
        except ImportError as e:       # pragma: no cover - mappy is absent from the build image
            raise ImportError("the minimap2 bindings (mappy) are needed to map reads; pass PAF text to "
                              "BossRuns.process_batch_paf or attach your own mapper instead") from e
        self.mu, self.workers = mu, workers
        if not Path(ref).is_file():
            raise FileNotFoundError("Given reference file does not exist")
        if default:
            self.aligner = mappy.Aligner(fn_idx_in=ref, preset="map-ont")
        else:
            self.aligner = mappy.Aligner(fn_idx_in=ref, fn_idx_out=f'{ref}.mmi', preset="map-ont",
                                         k=13, w=5, min_cnt=2, min_chain_score=20)

    def _map_query(self, query):
        """PAF lines of one read (mapper.py:112-125: `str(hit)` of a mappy alignment is the PAF
        record from column 3 on; id and length are prepended)."""
        read_id, seq = query
        import mappy
        thrbuf = mappy.ThreadBuffer()
        return [f"{read_id}\t{len(seq)}\t{hit}" for hit in self.aligner.map(seq, buf=thrbuf)]

    def _mappy_batch(self, sequences, out=None, log=True):
        """PAF text of a whole batch (mapper.py:68-108)."""
        with ThreadPoolExecutor(max_workers=self.workers) as ex:
            results = list(ex.map(self._map_query, sequences.items()))
        hits = ['\n'.join(r) for r in results if r]
        self.mapped_count, self.unmapped_count = len(hits), len(results) - len(hits)
        text = '\n'.join(hits)
        if out:
            with open(out, 'w') as fh:
                fh.write(text)
        return text


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
