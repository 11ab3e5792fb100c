"""The oracle against the reference: data-free known answers of the reference's own tests and
the golden vectors captured by tests/golden/make_golden.py (reference imported in the build
container).  CPU only."""
import os

import numpy as np
import pytest

from scenarios import (GOLDEN, SCENARIOS, E2E_BATCHES, E2E_REJECT, batch_digest, digest, e2e_batch,
                       e2e_contig_strings, e2e_reference, unpack_strat)

from oracle.model import SiteModel
from oracle.contig import OContig
from oracle.dists import OReadlengthDist
from oracle.pafcigar import parse_paf, convert_records
from oracle.pipeline import OracleRuns
from oracle.movesum import move_sum


def test_reference_known_answers_scoring():
    # /root/reference/tests/base/test_runs_sequences.py:113-126
    m = SiteModel(1)
    assert np.isclose(m.score0, 0.04969294)
    assert np.isclose(m.ent0, 0.09302521)
    pats = np.array([[28, 0, 0, 0, 0], [2, 0, 0, 0, 0]], dtype=np.uint16)
    ent, sco = m.entropy_and_score(pats)
    assert np.isclose(sco[3, 0], 3.834200141940696e-44)
    assert np.isclose(ent[3, 0], 3.834200141940696e-44)
    assert np.isclose(sco[3, 1], 0.17253973305650225)
    assert np.isclose(ent[3, 1], 0.22957118271635163)


@pytest.mark.parametrize("ploidy,b,g", [(1, 4, 5), (2, 4, 15)])
def test_reference_known_answers_priors(ploidy, b, g):
    # /root/reference/tests/base/test_runs_sequences.py:9-19
    m = SiteModel(ploidy)
    assert m.len_b == b + 1 and m.len_g == g
    assert m.phi_pow.shape == (b + 1, g, 1000) and m.priors.shape == (b, g)
    with pytest.raises(ValueError):
        SiteModel(3)


def test_reference_known_answers_readlengthdist():
    # /root/reference/tests/base/test_readlengthdist.py:21-32
    r = OReadlengthDist()
    r.update({'a': 1, 'b': 2, 'c': 3})
    assert int(r.lam) == 6000 and not hasattr(r, 'longest_read')
    assert np.array_equal(r.approx_ccl, [1167, 2729, 3903, 4918, 5866, 6808, 7797, 8912, 10321, 12713])


@pytest.mark.parametrize("name,seq,nb", [("ch1", "ACGTACGT", 1), ("ch1", "ACGTACGTNnWwIi", 1),
                                         ("ch2", "ACgtacGT", 1), ("ch3  r", "ACGTACGT", 1),
                                         ("ch1_bc", "ACGTACGT", 2)])
def test_reference_known_answers_contig(name, seq, nb):
    # /root/reference/tests/base/test_reference.py:10-36
    c = OContig(name, seq, nbarcodes=nb)
    assert c.length == len(seq) == len(c.seq_int) and " " not in c.name
    assert c.coverage.shape == (len(seq), 5, nb) and c.coverage.sum() == 0
    assert c.bucket_switches.shape == ((len(seq) // 20_000) + 1, nb)
    assert np.all(c.scores[0] == SiteModel(1).score0)


def test_golden_tables():
    g = np.load(os.path.join(GOLDEN, "g_tables.npz"))
    for pl in (1, 2):
        m = SiteModel(pl)
        assert np.array_equal(g["phi_p%d" % pl], m.phi)
        assert np.array_equal(g["priors_p%d" % pl], m.priors)
        assert g["score0_p%d" % pl][0] == m.score0[0] and g["ent0_p%d" % pl][0] == m.ent0[0]
        ent, sco = m.entropy_and_score(g["patterns"])
        # same machine family -> bit-equal; across CPUs numpy's SIMD log/pow may differ by ulps
        assert np.allclose(ent, g["entropy_p%d" % pl], rtol=1e-12, atol=0)
        assert np.allclose(sco, g["score_p%d" % pl], rtol=1e-9, atol=1e-30)


def test_golden_cigar():
    g = np.load(os.path.join(GOLDEN, "g_cigar.npz"))
    paf_text = g["paf"].tobytes().decode()
    seqs = dict(zip(g["read_ids"].tolist(), g["read_seqs"].tolist()))
    inc = convert_records(parse_paf(paf_text, min_len=200), seqs)
    flat = [(t, s, e, q) for t, lst in inc.items() for (s, e, q, bc) in lst]
    assert len(flat) == int(g["n_inc"]) > 20
    for k, (t, s, e, q) in enumerate(flat):
        assert t == str(g["inc%03d_tname" % k])
        assert [s, e] == g["inc%03d_range" % k].tolist()
        assert np.array_equal(q, g["inc%03d_codes" % k])


def test_golden_dists():
    g = np.load(os.path.join(GOLDEN, "g_dists.npz"))
    r = OReadlengthDist()
    assert np.array_equal(r.approx_ccl, g["default_approx_ccl"])
    for k in range(3):
        r.update({"x%d" % i: int(v) for i, v in enumerate(g["lens%d" % k])})
        assert np.array_equal(r.approx_ccl, g["approx_ccl%d" % k])
        assert r.lam == float(g["lam%d" % k]) and r.time_cost == float(g["time_cost%d" % k])


def test_move_sum_semantics():
    a = np.array([1.0, 2.0, 4.0, 8.0, 16.0, 32.0])
    assert np.array_equal(move_sum(a, 3), [1, 3, 7, 14, 28, 56])
    assert np.array_equal(move_sum(a[::-1], 2)[::-1], [3, 6, 12, 24, 48, 32])
    with pytest.raises(ValueError):
        move_sum(a, 7)
    with pytest.raises(ValueError):
        move_sum(a, 0)
    # the running recurrence (asum += a[i] - a[i-w]), not a fresh window sum
    rng = np.random.default_rng(0)
    x = rng.random(500)
    y = move_sum(x, 7)
    asum, ref = 0.0, []
    for i in range(500):
        asum = asum + x[i] if i < 7 else asum + (x[i] - x[i - 7])
        ref.append(asum)
    assert np.array_equal(y, ref)
    fresh = np.array([x[max(0, i - 6): i + 1].sum() for i in range(500)])
    assert np.allclose(y, fresh) and not np.array_equal(y, fresh)


@pytest.mark.parametrize("tag,ploidy,nb", SCENARIOS)
def test_golden_end_to_end(tag, ploidy, nb):
    """Five batches through the oracle reproduce the reference's masks, thresholds, coverage,
    scores, entropy, bucket switches, downsampled scores and benefits bit-for-bit."""
    g = np.load(os.path.join(GOLDEN, "g_e2e_%s.npz" % tag))
    contigs = e2e_reference()
    assert str(g["ref_digest"]) == digest(*[c[1] for c in contigs])
    o = OracleRuns(e2e_contig_strings(contigs), ploidy=ploidy, reject_refs={E2E_REJECT}, nbarcodes=nb)
    cross_cpu = False
    for b in range(E2E_BATCHES):
        batch = e2e_batch(contigs, b, nb)
        assert str(g["b%d_input_digest" % b]) == batch_digest(batch), "synthetic inputs drifted"
        o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"],
                        barcodes=batch["barcodes"] if nb > 1 else None)
        updated = bool(int(g["b%d_updated" % b]))
        assert np.array_equal(o.rl_dist.approx_ccl, g["b%d_approx_ccl" % b])
        for cname, c in o.contigs.items():
            key = "b%d_%s_" % (b, cname)
            if not c.rej:
                assert str(g[key + "cov_digest"]) == digest(c.coverage)
                assert int(g[key + "change_count"]) == int(c.change_mask.sum())
                assert np.array_equal(g[key + "bucket_switches"], c.bucket_switches)
                assert np.array_equal(g[key + "switched_on"], c.switched_on)
                assert int(g[key + "n_zero_scores"]) == int((c.scores == 0).sum())
                assert int(g[key + "n_tiny_scores"]) == int((c.scores == np.finfo(float).tiny).sum())
                if str(g[key + "scores_digest"]) != digest(c.scores):
                    cross_cpu = True      # numpy log/pow differ by ulps between CPU generations
                if updated:
                    assert np.allclose(g[key + "scores_ds"], c.scores_ds, rtol=1e-9, atol=1e-300)
                    assert np.allclose(g[key + "additional_benefit"], c.additional_benefit, rtol=1e-6, atol=1e-12)
                    if not cross_cpu:
                        assert np.array_equal(g[key + "scores_ds"], c.scores_ds)
                        assert np.array_equal(g[key + "smu"], c.smu)
                        assert np.array_equal(g[key + "additional_benefit"], c.additional_benefit)
            assert np.array_equal(unpack_strat(g, key + "strat", c.strat.shape), c.strat), (b, cname)
        if updated:
            assert np.isclose(float(g["b%d_threshold" % b]), o.threshold, rtol=1e-9)
            if not cross_cpu:
                assert float(g["b%d_threshold" % b]) == o.threshold
    assert np.allclose(g["final_ctgA_scores"], o.contigs["ctgA"].scores, rtol=1e-6, atol=1e-300)
    assert np.array_equal(g["final_ctgA_coverage"], o.contigs["ctgA"].coverage)
    assert np.array_equal(g["final_read_starts"], o.read_starts.merge())


def test_golden_saturated_regime():
    """The state a long run converges to — every score `tiny` (depth >= 30) or 0.0 (dropped out), every
    benefit ~1e-300 (scenarios.saturated_coverage) — through the reference's own update_wrapper, and one
    more ordinary batch on top of it: thresholds of 1e-304, masks, bin sums and benefits of the oracle
    equal the reference's.  (An absolute fixed point for ubar0 lost the whole sum here and put the
    threshold one exponent bin low: DESIGN 4.4.  The GPU is held to the oracle in this regime by
    test_saturated_coverage_vs_oracle.)"""
    from scenarios import saturated_coverage
    g = np.load(os.path.join(GOLDEN, "g_sat_p1_nb1.npz"))
    contigs = e2e_reference()
    assert str(g["ref_digest"]) == digest(*[c[1] for c in contigs])
    o = OracleRuns(e2e_contig_strings(contigs), ploidy=1, reject_refs={E2E_REJECT}, nbarcodes=1)

    def ingest(b):
        batch = e2e_batch(contigs, b, 1)
        assert str(g["b%d_input_digest" % b]) == batch_digest(batch), "synthetic inputs drifted"
        o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"])

    def check(tag):
        tiny = np.finfo(float).tiny
        for cname, c in o.contigs.items():
            key = "%s_%s_" % (tag, cname)
            assert np.array_equal(unpack_strat(g, key + "strat", c.strat.shape), c.strat), (tag, cname)
            if c.rej:
                continue
            assert int(g[key + "n_zero_scores"]) == int((c.scores == 0).sum())
            assert int(g[key + "n_tiny_scores"]) == int((c.scores == tiny).sum())
            assert np.array_equal(g[key + "scores_ds"], c.scores_ds)
            assert np.array_equal(g[key + "additional_benefit"], c.additional_benefit)
        assert float(g[tag + "_threshold"]) == o.threshold
        assert 0.0 < o.threshold < 1e-300

    ingest(0)                                    # (process_batch includes the update of batch 0)
    for c in o.contigs_filt.values():
        c.coverage[:] = saturated_coverage(c.seq_int, nb=1)
        c.change_mask[:] = True
    o.update_wrapper()
    # the regime: nothing but `tiny` and 0.0 (no LUT value involved, hence bit-equality across CPUs)
    assert all(int(g["sat_%s_n_other_scores" % n]) == 0 for n in o.contigs_filt)
    check("sat")
    ingest(1)
    check("sat1")



def _error_fixture():
    import json
    return json.load(open(os.path.join(GOLDEN, "g_errors.json")))


def test_oracle_error_classes_equal_the_reference():
    """tests/golden/g_errors.json: what the REFERENCE does with 60-odd malformed / unusual batches
    (scenarios.error_cases) — the exception class of Paf.parse_PAF / convert_records /
    increment_coverage (paf.py:18-75, 631-672; sequences.py:678-794; reference.py:122-145) or the
    coverage it ends with.  The oracle must do the same, case by case."""
    from scenarios import digest, error_cases
    from oracle.contig import OContig
    from oracle.pafcigar import parse_paf, convert_records
    from boss_runs_amd import synth
    contigs, cases = error_cases()
    gold = _error_fixture()
    assert set(gold) == {n for n, _, _ in cases}
    for name, paf_text, seqs in cases:
        conts = {n: OContig(n, synth.codes_to_str(c)) for n, c in contigs}
        try:
            paf = parse_paf(paf_text, min_len=200)
            inc = convert_records(paf, seqs)
            for n, c in conts.items():
                c.increment_coverage(inc[n])
            got = {"ok": digest(*[conts[n].coverage for n in sorted(conts)])}
        except Exception as e:          # noqa: BLE001
            got = {"error": type(e).__name__}
        want = gold[name]
        assert got.get("error") == want.get("error") and got.get("ok") == want.get("ok"), (name, got, want)


def test_oracle_fuzz_error_classes_equal_the_reference():
    """Differential fuzz against the reference itself: tests/golden/g_errors_fuzz.json holds what the REFERENCE does
    with 2,000 seeded random mutations of small batches (scenarios.fuzz_error_cases: truncated / swapped / garbled
    columns and lines, integers int() takes and does not take, repeated and malformed tags, CIGAR edits, reads cut
    short or dropped, coordinates shifted off either end of the contig, second mappings of a read) — exception
    class or a digest of the coverage it ends with.  The oracle must do the same with every one of them."""
    import json
    from scenarios import digest, fuzz_error_cases
    from oracle.contig import OContig
    from oracle.pafcigar import parse_paf, convert_records
    from boss_runs_amd import synth
    contigs, cases = fuzz_error_cases()
    gold = json.load(open(os.path.join(GOLDEN, "g_errors_fuzz.json")))
    assert set(gold) == {n for n, _, _ in cases} and len(gold) >= 2000
    bad = []
    for name, paf_text, seqs in cases:
        conts = {n: OContig(n, synth.codes_to_str(c)) for n, c in contigs}
        try:
            paf = parse_paf(paf_text, min_len=200)
            inc = convert_records(paf, seqs)
            for n, c in conts.items():
                c.increment_coverage(inc[n])
            got = {"ok": digest(*[conts[n].coverage for n in sorted(conts)])[:16]}
        except Exception as e:          # noqa: BLE001
            got = {"error": type(e).__name__}
        if got != gold[name]:
            bad.append((name, got, gold[name]))
    assert not bad, (len(bad), bad[:10])
    classes = {v.get("error", "ok") for v in gold.values()}
    assert classes >= {"ok", "ValueError", "IndexError", "KeyError", "AssertionError", "TypeError", "OverflowError"}


# ---- move_sum pinned to Bottleneck itself (tests/golden/make_movesum_golden.py, real Bottleneck 1.3.2) --------------
def _movesum_golden():
    import hashlib
    import json
    g = np.load(os.path.join(GOLDEN, "g_movesum.npz"))
    meta = json.loads(str(g["meta"]))

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a, dtype="<f8").tobytes()).hexdigest()

    def case_input(c):
        if c["source"] == "inline":
            return g["in_" + c["name"]]
        fname, key, b = c["name"].split(":")
        return np.ascontiguousarray(np.load(os.path.join(GOLDEN, fname))[key][:, int(b)])
    return g, meta, sha, case_input


def _shim_move_sum():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bn_shim", os.path.join(GOLDEN, "_shims", "bottleneck.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.move_sum


def test_move_sum_equals_bottleneck():
    """oracle/movesum.c (and the import shim the golden runs went through) against the outputs of the REAL
    Bottleneck 1.3.2 — reference call sites boss/runs/reference.py:233-234, 259-260: every `scores_ds` column the
    reference handed to move_sum in the e2e / saturated golden runs at the windows those runs used, and random arrays
    over forty decades with runs of `tiny` and zeros, forward (double reversal) and reverse, bit for bit."""
    g, meta, sha, case_input = _movesum_golden()
    assert meta["bottleneck"].startswith("1.3.") and meta["n_fixture_columns_equal_to_bottleneck"] >= 52
    shim = _shim_move_sum()
    n = 0
    for ci, c in enumerate(meta["cases"]):
        a = case_input(c)
        assert a.shape[0] == c["n"]
        for w in sorted(set(c["windows"])):
            assert sha(move_sum(a, w)) == c["rev"][str(w)], (c["name"], w)
            assert sha(move_sum(a[::-1], w)[::-1]) == c["fwd"][str(w)], (c["name"], w)
            n += 2
            if ci % 6 == 0 and w == c["windows"][3]:          # the (slow, pure-Python) shim on a sample of the arrays
                assert sha(shim(a, w, min_count=1)) == c["rev"][str(w)], (c["name"], w)
                assert sha(shim(a[::-1], w, min_count=1)[::-1]) == c["fwd"][str(w)], (c["name"], w)
    assert n >= 1500
    # full outputs: short arrays at every window 1..n (the min_count=1 head; w == n), and four long ones
    for s, ns in enumerate(meta["short_n"]):
        a = g["short%d_in" % s]
        for w in range(1, ns + 1):
            assert np.array_equal(move_sum(a, w), g["short%d_rev" % s][w - 1]), (s, w)
            assert np.array_equal(move_sum(a[::-1], w)[::-1], g["short%d_fwd" % s][w - 1]), (s, w)
            assert np.array_equal(shim(a, w, min_count=1), g["short%d_rev" % s][w - 1]), (s, w)
    for k in g.files:
        if k.startswith("rev_w") or k.startswith("fwd_w"):
            d, w, name = k.split("_")
            a = g["in_" + name]
            got = move_sum(a, int(w[1:])) if d == "rev" else move_sum(a[::-1], int(w[1:]))[::-1]
            assert np.array_equal(got, g[k]), k
    # the window edges: Bottleneck raises ValueError for w < 1 and w > n, and so do the oracle and the shim
    a = np.arange(10, dtype=np.float64)
    for w, what in meta["edges_n10"].items():
        for f in (lambda: move_sum(a, int(w)), lambda: shim(a, int(w), min_count=1)):
            if what == "ok":
                f()
            else:
                assert what == "ValueError"
                with pytest.raises(ValueError):
                    f()


def test_calc_u_equals_bottleneck():
    """Contig.calc_smu + calc_u (reference.py:215-269) of the oracle on the same arrays: `additional_benefit`
    equals the one formed from Bottleneck's own sums."""
    g, meta, sha, case_input = _movesum_golden()
    for c in meta["cases"]:
        a = case_input(c)
        oc = OContig.__new__(OContig)
        oc.nb, oc.length = 1, (a.shape[0] - 1) * 100
        oc.scores_ds = a.reshape(-1, 1).copy()
        oc.smu = np.zeros((a.shape[0], 2, 1))
        oc.smu[:, 0, 0] = move_sum(oc.scores_ds[::-1, 0], c["windows"][0])[::-1]
        oc.smu[:, 1, 0] = move_sum(oc.scores_ds[:, 0], c["windows"][0])
        oc.calc_u(np.array(c["windows"][1:]) * 100)
        assert sha(oc.additional_benefit[:, :, 0]) == c["benefit_sha"], c["name"]
        if "benefit_" + c["name"] in g.files:
            assert np.array_equal(oc.additional_benefit[:, :, 0], g["benefit_" + c["name"]])
