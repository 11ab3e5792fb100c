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


# ---- differential fuzz of the PAF / CIGAR front end against the reference itself ---------------------------------
FUZZ_N = 2000
FUZZ_SEED = 20260402


def fuzz_error_cases(n=FUZZ_N, seed=FUZZ_SEED):
    """`n` seeded random mutations of small consistent batches (the mutation set of scripts/fuzz/fuzz_paf.cpp, in Python so
    that the REFERENCE can be run on the very same inputs): list of (name, paf_text, seqs).  tests/golden/g_errors_fuzz.json
    holds what the reference does with each (exception class, or a digest of the coverage it ends with;
    tests/golden/make_golden.py, GOLDEN_ONLY=errors_fuzz); the oracle, the native host front end and the device walk are
    held to it case by case."""
    contigs = synth.make_reference(ERR_LENGTHS, seed=5, names=ERR_NAMES)
    rng = np.random.default_rng(seed)
    cases = []
    ints = ["0", "-1", "-7", "1", "9" * 12, "9" * 30, "12x", "abc", "", " 5", "5 ", "+3", "0x10", "1e3", "3.0", "١٢",
            "1_0", "_1", "٣", "2147483648", "4294967296", "-0", "6000", "5000", "-5000", "-6000", "9223372036854775808", "\u20035"]
    tags_bad = ["zz:Z:a:b", "zz:1", "zz:q:1", "de:f:abc", "AS:i:abc", "tp:A:S", "cg:Z:", "cg:Z:10", "cg:Z:M", "NM:i:", "s1:i:x", ":::", "cg:Z:5Q",
                "AS:f:inf", "AS:f:-Infinity", "AS:f:nan", "AS:f:1e3", "AS:f:2.7", "AS:f:1_0.5", "AS:f:0x10", "AS:Z:12", "AS:A:7", "AS:Z: 8 ", "AS:i:" + "9" * 25,
                "AS:f:1e300", "cg:i:5", "cg:f:1.5", "cg:i:x5M", "tp:i:P", "tp:Z:P", "tp:A:p", "AS:i:", "AS:f:"]
    ops = "MIDNSHP=XB"

    def mutate(lines, seqs, k):
        lines = list(lines)
        seqs = dict(seqs)
        if not lines:
            return lines, seqs
        i = int(rng.integers(0, len(lines)))
        f = lines[i].split("\t")
        kind = k
        if kind == 0:       # truncate a line to j columns
            lines[i] = "\t".join(f[:int(rng.integers(0, len(f)))])
        elif kind == 1:     # delete / duplicate / swap fields
            j = int(rng.integers(0, len(f)))
            w = int(rng.integers(0, 3))
            if w == 0 and len(f) > 1:
                del f[j]
            elif w == 1:
                f.insert(j, f[j])
            else:
                j2 = int(rng.integers(0, len(f)))
                f[j], f[j2] = f[j2], f[j]
            lines[i] = "\t".join(f)
        elif kind == 2:     # delete / duplicate / swap lines, blank line, trailing newline
            w = int(rng.integers(0, 5))
            if w == 0:
                del lines[i]
            elif w == 1:
                lines.insert(i, lines[i])
            elif w == 2:
                j2 = int(rng.integers(0, len(lines)))
                lines[i], lines[j2] = lines[j2], lines[i]
            elif w == 3:
                lines.insert(i, "")
            else:
                lines.append("")
        elif kind == 3:     # an integer column replaced
            col = int(rng.choice([1, 2, 3, 6, 7, 8, 9, 10, 11]))
            if col < len(f):
                f[col] = ints[int(rng.integers(0, len(ints)))]
                lines[i] = "\t".join(f)
        elif kind == 4:     # strand / contig name / read name replaced
            col = int(rng.choice([0, 4, 5]))
            if col < len(f):
                f[col] = [["rX", "", "7", "007", f[0] + "x"], ["+", "-", "*", "", "++"], ["e1", "e2", "nope", "", "E1"]][[0, 4, 5].index(col)][int(rng.integers(0, 5))]
                lines[i] = "\t".join(f)
        elif kind == 5:     # tags: missing / malformed / unknown type / appended junk
            w = int(rng.integers(0, 3))
            tag_idx = [j for j in range(12, len(f))]
            if w == 0 and tag_idx:
                del f[int(rng.choice(tag_idx))]
            elif w == 1 and tag_idx:
                f[int(rng.choice(tag_idx))] = tags_bad[int(rng.integers(0, len(tags_bad)))]
            else:
                f.append(tags_bad[int(rng.integers(0, len(tags_bad)))])
            lines[i] = "\t".join(f)
        elif kind == 6:     # CIGAR edits: prepend / append / drop an operation, other letters, zero lengths, huge lengths
            cg = [j for j, x in enumerate(f) if x.startswith("cg:Z:")]
            if cg:
                c = f[cg[0]][5:]
                w = int(rng.integers(0, 7))
                op = ops[int(rng.integers(0, len(ops)))]
                ln = str(int(rng.integers(0, 40)))
                if w == 0:
                    c = ln + op + c
                elif w == 1:
                    c = c + ln + op
                elif w == 2:
                    c = c[:max(0, len(c) - int(rng.integers(1, 6)))]
                elif w == 3:
                    c = c.replace("M", op, 1)
                elif w == 4:
                    c = c + "0M" + "0D"
                elif w == 5:
                    c = c + "9" * 11 + "M"
                else:
                    c = "x" + c[:len(c) // 2] + "?" + c[len(c) // 2:]
                f[cg[0]] = "cg:Z:" + c
                lines[i] = "\t".join(f)
        elif kind == 7:     # the read: cut short, emptied, other letters, lower case, dropped from the dict
            name = f[0]
            if name in seqs:
                sq = seqs[name]
                w = int(rng.integers(0, 5))
                if w == 0:
                    seqs[name] = sq[:int(rng.integers(0, len(sq) + 1))]
                elif w == 1:
                    seqs[name] = ""
                elif w == 2 and sq:
                    p = int(rng.integers(0, len(sq)))
                    seqs[name] = sq[:p] + "NRYacgt-*"[int(rng.integers(0, 9))] + sq[p + 1:]
                elif w == 3:
                    seqs[name] = sq.lower()
                else:
                    del seqs[name]
        elif kind == 8:     # garbage bytes inside a line
            p = int(rng.integers(0, len(lines[i]) + 1))
            lines[i] = lines[i][:p] + ["\x00", "\t\t", " ", "\r", "\x7f", "é"][int(rng.integers(0, 6))] + lines[i][p:]
        elif kind == 10:    # a second mapping of the same read: the line again with another mapq / AS / strand / target
            g = list(f)
            w = int(rng.integers(0, 4))
            if w == 0 and len(g) > 11:
                g[11] = ["0", "60", "61", "7", "abc", "9" * 20, "-3"][int(rng.integers(0, 7))]
            elif w == 1:
                g = [x if not x.startswith("AS:") else "AS:i:" + ["0", "5000", "abc", "9" * 20, "-1"][int(rng.integers(0, 5))] for x in g]
            elif w == 2 and len(g) > 4:
                g[4] = "-" if g[4] == "+" else "+"
            elif len(g) > 5:
                g[5] = ["e1", "e2", "nope"][int(rng.integers(0, 3))]
            lines.insert(i + int(rng.integers(0, 2)), "\t".join(g))
        elif kind == 11:    # the query columns select ONE base of the read (numpy broadcasts it over the whole CIGAR) or none
            if len(f) > 3:
                try:
                    ql = max(int(f[1]), 2)
                    v = int(rng.integers(0, ql - 1))
                    f[2] = str(v)
                    f[3] = str(v + int(rng.integers(0, 2)))
                    lines[i] = "\t".join(f)
                except ValueError:
                    pass
        else:               # coordinates shifted consistently (a mapping that runs past the contig end / before its start)
            if len(f) > 8:
                try:
                    sh = int(rng.choice([-10 ** 6, -50, 50, 5000, 10 ** 6, -5000, -6000, -4000]))
                    f[7] = str(int(f[7]) + sh)
                    f[8] = str(int(f[8]) + sh)
                    lines[i] = "\t".join(f)
                except ValueError:
                    pass
        return lines, seqs

    for c in range(n):
        k_reads = int(rng.integers(1, 6))
        b = synth.make_batch(contigs, k_reads, seed=int(rng.integers(0, 2 ** 31)), mean_len=600.0, min_len=250, max_len=1500,
                             extras=bool(rng.integers(0, 4) == 0))
        lines, seqs = b["paf"].split("\n"), b["seqs"]
        for _ in range(int(rng.integers(0, 4))):
            lines, seqs = mutate(lines, seqs, int(rng.integers(0, 12)))
        cases.append(("f%04d" % c, "\n".join(lines), seqs))
    # ---- round 5: DIGITS inside reads.  The reference turns A C G T into '0'..'3' and takes every byte of the read minus '0' as the column
    # of np.add.at (sequences.py:666-667, 790): '0'..'3' in a read count as A C G T, '4' as a deletion, '7' becomes one with the D
    # operations (:803), '5' '6' '8' '9' are out of range (IndexError) — on either strand (utils.py:93 complements letters only).
    # Appended behind the cases above (their own generator: the first `n` cases stay what they were).
    rng = np.random.default_rng(seed + 1)
    for c in range(max(n // 8, 40) if n >= FUZZ_N else max(n // 8, 4)):
        k_reads = int(rng.integers(1, 5))
        b = synth.make_batch(contigs, k_reads, seed=int(rng.integers(0, 2 ** 31)), mean_len=600.0, min_len=250, max_len=1500,
                             extras=bool(rng.integers(0, 4) == 0))
        lines, seqs = b["paf"].split("\n"), dict(b["seqs"])
        accepted_only = bool(rng.integers(0, 3))             # two cases in three hold only digits the reference goes on with
        for name in list(seqs):
            if rng.integers(0, 3) == 0:
                continue
            sq = list(seqs[name])
            for _ in range(int(rng.integers(1, 5))):
                sq[int(rng.integers(0, len(sq)))] = ("012347" if accepted_only else "0123456789")[int(rng.integers(0, 6 if accepted_only else 10))]
            seqs[name] = "".join(sq)
        for _ in range(int(rng.integers(0, 2))):
            lines, seqs = mutate(lines, seqs, int(rng.integers(0, 12)))
        cases.append(("d%04d" % c, "\n".join(lines), seqs))
    return contigs, cases
