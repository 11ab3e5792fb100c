"""Seeded end-to-end scenarios shared by the golden-vector generator and the parity tests."""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from boss_runs_amd import synth  # noqa: E402

GOLDEN = os.path.join(REPO, "tests", "golden")

SCENARIOS = [("p1_nb1", 1, 1), ("p2_nb1", 2, 1), ("p1_nb2", 1, 2), ("p2_nb2", 2, 2)]   # tag, ploidy, nb
E2E_LENGTHS = [150_000, 260_000, 120_000, 60_000]
E2E_NAMES = ["ctgA", "ctgB", "ctgREJ", "ctgSHORT"]
E2E_REJECT = "ctgREJ"
E2E_BATCHES = 5
E2E_READS = 420


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(a if isinstance(a, (bytes, bytearray)) else np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def e2e_reference(seed=1):
    return synth.make_reference(E2E_LENGTHS, seed=seed, names=E2E_NAMES)


def e2e_contig_strings(contigs=None):
    contigs = contigs or e2e_reference()
    return [(n, synth.codes_to_str(c)) for n, c in contigs]


def e2e_batch(contigs, b, nb):
    """Hot region on ctgB, thin coverage on ctgA (dropout); the rejected and the short contig
    still attract reads."""
    return synth.make_batch(contigs, E2E_READS, seed=10 + b, mean_len=3000.0, nbarcodes=nb,
                            start_weights=[0.6, 1.6, 0.5, 0.5])


def batch_digest(batch):
    return digest(batch["paf"].encode(), "".join(batch["seqs"].values()).encode())


def unpack_strat(g, key, shape):
    n = int(np.prod(shape))
    return np.unpackbits(g[key])[:n].reshape(shape).astype(bool)


# ---- the saturated regime (every site maxed or dropped out) -------------------------------------
SAT_DEPTH = 240          # mean/8 >= 29: no site can sit between the dropout threshold and the depth cap of 30
SAT_RAMP = 600


def saturated_coverage(seq_int, nb=1, depth=SAT_DEPTH, ramp=SAT_RAMP):
    """The coverage a long run converges to: every site `depth` reads deep, all on the reference
    base, rising from zero over the first / falling over the last `ramp` sites (reads start
    uniformly, so the ends of a contig fill last).  With depth 240 the dropout threshold
    int(mean / 8) is 29: every score is `tiny` (depth >= 30) or 0.0 (dropped out) and every
    benefit ~1e-300 — the regime in which an absolute fixed point loses ubar0."""
    seq_int = np.asarray(seq_int)
    L = seq_int.shape[0]
    d = np.full(L, depth, dtype=np.int64)
    r = np.arange(ramp, dtype=np.int64) * depth // ramp
    d[:ramp] = r
    d[L - ramp:] = r[::-1]
    cov = np.zeros((L, 5, nb), dtype=np.uint16)
    for b in range(nb):
        cov[np.arange(L), seq_int, b] = d
    return cov


# ---- malformed / unusual PAF batches: what the reference does with them ---------------------------
ERR_LENGTHS = [6000, 5000]
ERR_NAMES = ["e1", "e2"]


def error_cases():
    """Named mutations of one small seeded batch: list of (name, paf_text, seqs).  The golden fixture
    g_errors.json holds what the REFERENCE does with each (exception class of Paf.parse_PAF ->
    CoverageConverter.convert_records -> Contig.increment_coverage, or a digest of the resulting
    coverage); tests hold the oracle and the native front end to it."""
    contigs = synth.make_reference(ERR_LENGTHS, seed=5, names=ERR_NAMES)
    b = synth.make_batch(contigs, 40, seed=91, mean_len=900.0, min_len=300, max_len=2500, extras=False)
    lines = b["paf"].split("\n")
    seqs = b["seqs"]
    cases = []

    def fields(i):
        return lines[i].split("\t")

    def with_line(i, f):
        out = list(lines)
        out[i] = "\t".join(f) if isinstance(f, list) else f
        return "\n".join(out)

    def cg_index(f):
        return [j for j, x in enumerate(f) if x.startswith("cg:Z:")][0]

    def with_cigar(i, fn):
        f = fields(i)
        k = cg_index(f)
        f[k] = "cg:Z:" + fn(f[k][5:])
        return with_line(i, f)

    def set_col(i, col, val):
        f = fields(i)
        f[col] = val
        return with_line(i, f)

    plus = [i for i in range(len(lines)) if fields(i)[4] == "+"][0]
    minus = [i for i in range(len(lines)) if fields(i)[4] == "-"][0]
    cases.append(("base", b["paf"], seqs))
    cases.append(("truncated_line", with_line(7, fields(7)[:7]), seqs))
    cases.append(("eleven_columns", with_line(7, fields(7)[:11]), seqs))
    cases.append(("twelve_columns_no_tags", with_line(7, fields(7)[:12]), seqs))
    cases.append(("blank_line_inside", "\n".join(lines[:5] + [""] + lines[5:]), seqs))
    cases.append(("trailing_newline", b["paf"] + "\n", seqs))
    cases.append(("huge_tlen", set_col(3, 6, "9" * 30), seqs))
    cases.append(("huge_qend", set_col(plus, 3, "9" * 30), seqs))
    cases.append(("nonint_alnlen", set_col(3, 10, "12x"), seqs))
    cases.append(("nonint_qstart", set_col(plus, 2, "abc"), seqs))
    cases.append(("nonint_tstart", set_col(plus, 7, "abc"), seqs))
    cases.append(("nonint_nmatch", set_col(3, 9, "abc"), seqs))
    cases.append(("nonint_mapq_single", set_col(3, 11, "abc"), seqs))
    cases.append(("nonint_mapq_multi", "\n".join(lines[:4] + [set_col(3, 11, "abc").split("\n")[3]] + lines[4:]), seqs))
    cases.append(("negative_mapq", set_col(3, 11, "-5"), seqs))
    f = fields(9)
    cases.append(("missing_tp", with_line(9, [x for x in f if not x.startswith("tp:")]), seqs))
    cases.append(("secondary_only", with_line(9, [x.replace("tp:A:P", "tp:A:S") for x in f]), seqs))
    cases.append(("missing_cg", with_line(9, [x for x in f if not x.startswith("cg:")]), seqs))
    cases.append(("tag_four_parts", with_line(9, f + ["zz:Z:a:b"]), seqs))
    cases.append(("tag_two_parts", with_line(9, f + ["zz:1"]), seqs))
    cases.append(("tag_unknown_type", with_line(9, f + ["zz:q:1"]), seqs))
    cases.append(("as_not_int", with_line(9, [("AS:i:abc" if x.startswith("AS:") else x) for x in f]), seqs))
    cases.append(("float_tag_bad", with_line(9, f + ["de:f:abc"]), seqs))
    for tag, i in (("plus", plus), ("minus", minus)):
        cases.append(("cigar_consumes_more_" + tag, with_cigar(i, lambda c: "7M" + c), seqs))
        cases.append(("cigar_spans_more_" + tag, with_cigar(i, lambda c: "3D" + c), seqs))
        cases.append(("cigar_trailing_digits_" + tag, with_cigar(i, lambda c: c + "12"), seqs))
        cases.append(("cigar_op_without_length_" + tag, with_cigar(i, lambda c: "M" + c), seqs))
        cases.append(("cigar_unknown_op_" + tag, with_cigar(i, lambda c: c.replace("M", "Z", 1)), seqs))

        def garbage(c):        # "?!" between two operations
            k = len(c) // 2
            while c[k - 1] not in "MIDNSHP=XB":
                k += 1
            return c[:k] + "?!" + c[k:]
        cases.append(("cigar_garbage_between_" + tag, with_cigar(i, garbage), seqs))
        cases.append(("cigar_digits_then_garbage_" + tag, with_cigar(i, lambda c: "5Q" + c), seqs))
        cases.append(("cigar_zero_length_ops_" + tag, with_cigar(i, lambda c: "0M0D0I" + c), seqs))
        cases.append(("cigar_leading_zeros_" + tag, with_cigar(i, lambda c: "000" + c), seqs))
        cases.append(("cigar_soft_clip_ops_" + tag, with_cigar(i, lambda c: c.replace("M", "=", 1).replace("M", "X", 1)), seqs))
        rid = fields(i)[0]
        qs, qe = int(fields(i)[2]), int(fields(i)[3])
        mid = (qs + qe) // 2
        cases.append(("read_with_N_" + tag, b["paf"], {**seqs, rid: seqs[rid][:mid] + "N" + seqs[rid][mid + 1:]}))
        cases.append(("read_lowercase_" + tag, b["paf"], {**seqs, rid: seqs[rid].lower()}))
        cases.append(("read_N_in_flank_" + tag, b["paf"], {**seqs, rid: ("N" * qs + seqs[rid][qs:]) if qs else seqs[rid]}))
        cases.append(("read_too_short_" + tag, b["paf"], {**seqs, rid: seqs[rid][: qe - 40]}))
        cases.append(("read_empty_" + tag, b["paf"], {**seqs, rid: ""}))
        cases.append(("read_missing_" + tag, b["paf"], {k: v for k, v in seqs.items() if k != rid}))
        g = fields(i)
        span = int(g[8]) - int(g[7])
        L = int(g[6])
        g[7], g[8] = str(L - span + 50), str(L + 50)
        cases.append(("past_contig_end_" + tag, with_line(i, g), seqs))
    cases.append(("empty_cigar", with_cigar(plus, lambda c: ""), seqs))
    cases.append(("cigar_only_garbage", with_cigar(plus, lambda c: "hello"), seqs))
    cases.append(("nul_in_cigar", with_cigar(plus, lambda c: c[:7] + "\x00" + c[7:]), seqs))
    cases.append(("nul_in_name", set_col(5, 0, "x\x00y"), seqs))
    cases.append(("unknown_target", set_col(5, 5, "nowhere"), seqs))
    cases.append(("strand_garbage", set_col(plus, 4, "?"), seqs))
    cases.append(("empty_batch", "", {}))
    return contigs, cases
