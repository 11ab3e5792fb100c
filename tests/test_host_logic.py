"""Host-side product logic on the CPU (no GPU needed): the score table builder, the
read-length / read-start distributions, the threshold choice from binned statistics and the
composition ranking — against the golden vectors and the oracle."""
import os

import numpy as np
import pytest

from scenarios import GOLDEN, e2e_batch, e2e_contig_strings, e2e_reference

from boss_runs_amd.readlengthdist import ReadlengthDist
from boss_runs_amd.readstartdist import ReadStartDist
from boss_runs_amd.runs import choose_threshold, fx_to_float, FX_SHIFT
from boss_runs_amd.scoring import SiteScoring, all_compositions, composition_rank, NCOMP


def test_reference_known_answers_product_scoring():
    # /root/reference/tests/base/test_runs_sequences.py:113-126 through the product's builder
    m = SiteScoring(1)
    assert np.isclose(m.score0, 0.04969294) and np.isclose(m.ent0, 0.09302521)
    sc, en = m.tables()
    assert sc.shape == (NCOMP, 4) and en.shape == (NCOMP, 4)
    r = composition_rank(np.array([[28, 0, 0, 0, 0], [2, 0, 0, 0, 0]]))
    assert np.isclose(sc[r[0], 3], 3.834200141940696e-44) and np.isclose(en[r[0], 3], 3.834200141940696e-44)
    assert np.isclose(sc[r[1], 3], 0.17253973305650225) and np.isclose(en[r[1], 3], 0.22957118271635163)
    with pytest.raises(ValueError):
        SiteScoring(3)


def test_composition_rank_is_a_bijection():
    pats = all_compositions()
    assert pats.shape == (NCOMP, 5) and int(pats.sum(axis=1).max()) == 29
    assert np.array_equal(composition_rank(pats), np.arange(NCOMP))
    assert composition_rank(np.array([0, 0, 0, 0, 0])) == 0 and composition_rank(np.array([29, 0, 0, 0, 0])) == NCOMP - 1


@pytest.mark.parametrize("ploidy", [1, 2])
def test_product_table_vs_golden(ploidy):
    g = np.load(os.path.join(GOLDEN, "g_tables.npz"))
    m = SiteScoring(ploidy)
    assert np.array_equal(g["phi_p%d" % ploidy], m.phi) and np.array_equal(g["priors_p%d" % ploidy], m.priors)
    sc, en = m.tables()
    pats = g["patterns"]
    r = composition_rank(pats)
    # numpy's SIMD log/pow may differ by ulps between CPU generations: 1e-9 relative
    assert np.allclose(sc[r].T, g["score_p%d" % ploidy], rtol=1e-9, atol=1e-30)
    assert np.allclose(en[r].T, g["entropy_p%d" % ploidy], rtol=1e-12, atol=0)
    # and bit-identical to the oracle on this host
    from oracle.model import SiteModel
    e2, s2 = SiteModel(ploidy).entropy_and_score(pats)
    assert np.array_equal(sc[r].T, s2) and np.array_equal(en[r].T, e2)


def test_product_readlengthdist_vs_golden():
    g = np.load(os.path.join(GOLDEN, "g_dists.npz"))
    r = ReadlengthDist()
    assert np.array_equal(r.approx_ccl, g["default_approx_ccl"])
    r.update({'a': 1, 'b': 2, 'c': 3})                       # test_readlengthdist.py:21-32
    assert int(r.lam) == 6000 and not hasattr(r, 'longest_read') and not hasattr(r, 'time_cost')
    for k in range(3):
        r.update({"x%d" % i: int(v) for i, v in enumerate(g["lens%d" % k])})
        assert np.array_equal(r.approx_ccl, g["approx_ccl%d" % k])
        assert r.lam == float(g["lam%d" % k]) and r.time_cost == float(g["time_cost%d" % k])


def test_product_readstartdist_vs_oracle():
    from oracle.dists import OReadStartDist
    from oracle.pafcigar import parse_paf, best_mapper

    class C:
        def __init__(self, L):
            self.length = L
    contigs = e2e_reference()
    cs = {"ctgA": C(150_000), "ctgB": C(260_000)}
    names = ["ctgA", "ctgB", "ctgREJ"]
    o = OReadStartDist(cs)
    p = ReadStartDist(cs)
    for b in range(3):
        batch = e2e_batch(contigs, b, 1)
        paf = parse_paf(batch["paf"], min_len=200)
        o.count_read_starts(paf)
        recs = [best_mapper(v) for v in paf.values()]
        ci = np.array([names.index(r.tname) if r.tname in names else -1 for r in recs])
        p.count_starts(names, ci, np.array([r.rev for r in recs]), np.array([r.tstart for r in recs]),
                       np.array([r.tend for r in recs]))
        assert np.array_equal(o.merge(), p.merge())
        fh_c, target_rs = p.fhat_compact()
        want = o.update_f_pointmass()
        got = p.expand(fh_c, want.shape[0])
        assert target_rs == o.target_size and got.shape == want.shape
        assert np.allclose(got, want, rtol=1e-13, atol=0)      # only the normaliser's rounding differs
        assert np.array_equal(p.fhat_expanded_reference_order(), want)      # the near-tie fallback's form: bit for bit the reference's


def test_choose_threshold_vs_oracle_find_strategy():
    from oracle.strategy import find_strategy
    rng = np.random.default_rng(3)
    for trial in range(5):
        T, nb = 4000, 2
        benefit = rng.gamma(0.3, 2.0, size=(T, 2, nb)) * (rng.random((T, 2, nb)) > 0.2)
        fhat = rng.random((T, 2))
        fhat /= fhat.sum()
        fh3 = np.repeat(fhat[:, :, None], nb, axis=2)
        detail = {}
        strat, thr = find_strategy(benefit, benefit, fh3, 5000.0, detail=detail)
        # binned statistics as the device produces them (exact sums, here via Python ints)
        norm = benefit.max()
        exps = np.abs(np.frexp(benefit[benefit != 0] / norm)[1])
        counts = np.bincount(exps, minlength=1088).astype(np.int64)
        fgrid = np.zeros(1088)
        for e in np.nonzero(counts)[0]:
            fgrid[e] = float(np.sum(fh3[benefit != 0][exps == e]))
        ubar0 = float(np.sum(fh3 * benefit))
        t, size, uniq = choose_threshold(norm, counts, fgrid, ubar0, 5000.0)
        assert t == thr and size == detail["strat_size"] and np.array_equal(uniq, detail["exponents"])


def test_reference_order_threshold_vs_oracle_find_strategy():
    """runs.reference_order_threshold (the host fallback of a near-tie) is find_strat_thread's own arithmetic."""
    from oracle.strategy import find_strategy
    from boss_runs_amd.runs import reference_order_threshold
    rng = np.random.default_rng(4)
    for trial in range(5):
        T, nb = 3000, 1 + trial % 2
        benefit = rng.gamma(0.3, 2.0, size=(T, 2, nb)) * np.power(2.0, -rng.integers(0, 40, (T, 2, nb))) * (rng.random((T, 2, nb)) > 0.2)
        fhat = rng.random((T, 2))
        fhat /= fhat.sum()
        fh3 = np.repeat(fhat[:, :, None], nb, axis=2)
        d = {}
        _, thr = find_strategy(benefit, benefit, fh3, 4100.0, detail=d)
        t, size, margin = reference_order_threshold(benefit, fh3, 4100.0)
        assert t == thr and size == d["strat_size"] and margin == d["argmax_margin"]


def test_near_tie_exact_sums_and_reference_order_can_disagree_and_the_fallback_follows_the_reference():
    """VERDICT r5 weak 1a, constructed.  The device accumulates f_grid / ubar0 EXACTLY, the reference in a 12-chunk float
    order; both only feed argmax(cs_u / cs_t).  Scaling the read-start posterior moves that argmax from bin to bin; at a
    crossing the two leading ratios agree to ~1e-16 and the two summation orders may pick DIFFERENT bins — thresholds a
    factor of two apart.  What the product does there: the device reports the margin (bossx_update_result.argmax_margin),
    BossRuns falls back to runs.reference_order_threshold below `tie_margin` (1e-9), and that function is the reference's
    arithmetic — so the masks are the reference's on either side of the crossing."""
    from fractions import Fraction
    from oracle.strategy import find_strategy
    from boss_runs_amd.config import GpuConfig
    from boss_runs_amd.runs import reference_order_threshold

    def device_choice(benefit, fh3, time_cost):
        """exact sums (as the 128-bit fixed point gives them), rounded once per statistic, then the float tail"""
        norm = benefit.max()
        nzm = benefit != 0
        exps = np.abs(np.frexp(benefit[nzm] / norm)[1])
        counts = np.bincount(exps, minlength=1088).astype(np.int64)
        fgrid = np.zeros(1088)
        fv = fh3[nzm]
        for e in np.nonzero(counts)[0]:
            fgrid[e] = float(sum(Fraction(float(v)) for v in fv[exps == e]))
        ubar0 = float(sum(Fraction(float(a)) * Fraction(float(b)) for a, b in zip(fh3.ravel(), benefit.ravel())))
        thr, size, uniq = choose_threshold(norm, counts, fgrid, ubar0, time_cost)
        f_mean = fgrid[uniq] / counts[uniq]
        bb = np.power(2.0, -uniq) * norm
        peaks = (np.cumsum(bb * f_mean * counts[uniq]) + ubar0) / (np.cumsum((time_cost // 100) * counts[uniq] * f_mean) + 10)
        rest = np.delete(peaks, size - 1)
        return thr, size, float((peaks[size - 1] - rest.max()) / peaks[size - 1])

    rng = np.random.default_rng(5)
    T, disagreements = 400, 0
    for trial in range(40):
        benefit = np.zeros((T, 2, 1))
        benefit[:, 0, 0] = rng.uniform(0.5, 1.0, T)
        benefit[:, 1, 0] = rng.uniform(0.125, 0.25, T) * np.power(2.0, -rng.integers(0, 6, T))
        benefit[0, 0, 0] = 1.0
        fhat = rng.random((T, 2))
        fhat /= fhat.sum()
        tc = float(rng.choice([900, 2000, 5000, 20000]))
        choice = lambda s: reference_order_threshold(benefit, (fhat * s)[:, :, None], tc)
        lo, hi = 1e-9, 1e9
        size_hi = choice(hi)[1]
        if choice(lo)[1] == size_hi:
            continue
        for _ in range(200):            # the crossing into the bin the largest scale chooses
            mid = (lo * hi) ** 0.5
            if mid == lo or mid == hi:
                break
            if choice(mid)[1] == size_hi:
                hi = mid
            else:
                lo = mid
        for s in (lo, hi):
            fh3 = (fhat * s)[:, :, None]
            t_ref, s_ref, m_ref = choice(s)
            d = {}
            _, t_or = find_strategy(benefit, benefit, fh3, tc, detail=d)
            assert t_ref == t_or and s_ref == d["strat_size"]            # the fallback IS the reference's choice
            t_dev, s_dev, m_dev = device_choice(benefit, fh3, tc)
            assert m_dev < 1e-12 < GpuConfig().tie_margin and m_ref < 1e-12     # ... and the device's margin sends the update there
            if t_dev != t_ref:
                disagreements += 1
                assert t_dev in (2.0 * t_ref, 0.5 * t_ref)
        if disagreements >= 3:
            break
    assert disagreements >= 3      # the hazard is real: exact and 12-chunk sums do pick different thresholds at a crossing


def test_fixed_point_conversion():
    for v in (0.0, 1.0, 1e-6, 0.123456789, 3.5e-12):
        n = int(v * (1 << FX_SHIFT))
        assert fx_to_float(n & ((1 << 64) - 1), n >> 64) == n / (1 << FX_SHIFT)


def test_mask_bits_file_roundtrip_and_lookup(tmp_path):
    """masks.py: the bit-packed mask file gives back exactly the arrays of boss.npz and the
    reader answers like dynamic_readfish.py:169-210 (`arr[:, rev, b][pos // 100]`, rejected -> 0,
    unknown contig / out of range -> 1, negative rows wrap like numpy)."""
    import os
    from boss_runs_amd.masks import MaskFile, MaskReader, write_mask_bits
    rng = np.random.default_rng(5)
    nb = 3
    strats = {"c1": rng.random((1501, 2, nb)) < 0.3, "rejX": np.zeros(1, dtype=bool),
              "c2": rng.random((1000, 2, nb)) < 0.7, "c3": rng.random((7, 2, nb)) < 0.5}
    entries, flat, off = [], [], 0
    for name, a in strats.items():
        rej = a.shape[0] == 1 and a.ndim == 1
        entries.append((name, 1 if rej else a.shape[0], rej, 0 if rej else off))
        if not rej:
            flat.append(a.reshape(-1).astype(np.uint8))
            off += a.size
    bits = np.packbits(np.concatenate(flat))
    path = tmp_path / "boss.bits"
    write_mask_bits(path, entries, bits, nb)
    mf = MaskFile(path)
    got = mf.to_dict()
    assert list(got) == list(strats)
    for name, a in strats.items():
        assert got[name].dtype == np.bool_ and np.array_equal(got[name], a), name
    index = {1: 0, 2: 1, 7: 2}
    rd = MaskReader(path, barcodes_index=index)
    assert rd.check_coord("c1", 0, 0, "barcode01") == 1          # nothing loaded: accept all
    assert rd.reload() == 1 and rd.reload() == 0                 # mtime gate (dynamic_readfish.py:101)

    def ref_check(contig, pos, rev, barcode):
        if contig not in strats:
            return 1
        arr = strats[contig]
        if arr.shape[0] == 1:
            return 0
        try:
            b = index[int(barcode.split('barcode')[1])]
            return int(arr[:, int(rev), b][pos // 100])
        except Exception:
            return 1
    for _ in range(3000):
        contig = ["c1", "c2", "c3", "rejX", "nope"][rng.integers(5)]
        pos = int(rng.integers(-900, 160_000))
        rev = bool(rng.integers(2))
        bc = ["barcode01", "barcode02", "barcode07", "barcode09"][rng.integers(4)]
        assert rd.check_coord(contig, pos, rev, bc) == ref_check(contig, pos, rev, bc), (contig, pos, rev, bc)
    # a rewritten file is picked up; an unreadable one accepts everything
    strats["c3"][:] = True
    flat[2] = strats["c3"].reshape(-1).astype(np.uint8)
    write_mask_bits(path, entries, np.packbits(np.concatenate(flat)), nb)
    os.utime(path, (rd.last_mask_mtime + 5, rd.last_mask_mtime + 5))
    assert rd.reload() == 1 and rd.check_coord("c3", 650, 1, "barcode02") == 1
    path.write_bytes(b"garbage")
    os.utime(path, (rd.last_mask_mtime + 5, rd.last_mask_mtime + 5))
    assert rd.reload() == 1 and rd.check_coord("rejX", 0, 0, "barcode01") == 1


def _oracle_expansion(paf_text, seqs, contig_names, barcodes=None, min_len=200):
    """Per aligned base (contig index, position, code, barcode) from the oracle's restatement of
    convert_records, in record order."""
    from oracle.pafcigar import parse_paf, best_mapper, expand_cigar, _COMP
    cidx = {n: i for i, n in enumerate(contig_names)}
    oc, op, ocode, obc = [], [], [], []
    for rid, recs in parse_paf(paf_text, min_len=min_len).items():
        rec = best_mapper(recs) if len(recs) > 1 else recs[0]
        if rec.tname not in cidx:
            continue
        if rec.rev:
            seq = seqs[rec.qname].translate(_COMP)[::-1]
            qs, qe = rec.qlen - rec.qend, rec.qlen - rec.qstart
        else:
            seq, qs, qe = seqs[rec.qname], rec.qstart, rec.qend
        q = expand_cigar(rec.cigar, seq, qs, qe)
        start = min(rec.tstart, rec.tend)
        oc.append(np.full(q.size, cidx[rec.tname], np.int32))
        op.append(np.arange(start, start + q.size, dtype=np.int64))
        ocode.append(q.astype(np.uint8))
        obc.append(np.full(q.size, 0 if barcodes is None else barcodes[rec.qname], np.uint8))
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
    return cat(oc, np.int32), cat(op, np.int64), cat(ocode, np.uint8), cat(obc, np.uint8)


@pytest.mark.parametrize("threads", [1, 3, 8, -3])
def test_native_parser_vs_oracle_expansion(threads, monkeypatch):
    """The C++ PAF/CIGAR front end (threaded line parse + CIGAR walk, tile segments) against the
    oracle's restatement of Paf.parse_PAF / choose_best_mapper / _parse_cigar, base by base, on
    the golden CIGAR fixture and on a synthetic 1500-read batch.  No device involved."""
    from boss_runs_amd import synth
    from boss_runs_amd.engine import host_parse
    if threads < 0:         # (three threads, the pre-pass over the chosen mappings in ranges of 50 records: its parallel form)
        threads = -threads
        monkeypatch.setenv("BOSSX_PLAN_RANGE", "50")
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g_cigar.npz"))
    paf_text = g["paf"].tobytes().decode()
    seqs = dict(zip(g["read_ids"].tolist(), g["read_seqs"].tolist()))
    tnames = sorted({str(g["inc%03d_tname" % k]) for k in range(int(g["n_inc"]))})
    contigs = [(n, 400_000, 0) for n in tnames]
    got = host_parse(contigs, paf_text, seqs, n_threads=threads)
    want = _oracle_expansion(paf_text, seqs, tnames)
    for key, w in zip(("contig", "pos", "code", "barcode"), want):
        assert np.array_equal(got[key], w), key
    assert got["aligned"] == want[0].size > 0

    ref = synth.make_reference([150_300, 260_100, 120_000], seed=3, names=["a", "b", "rej"])
    batch = synth.make_batch(ref, 1500, seed=77, mean_len=5000.0, nbarcodes=4)
    contigs = [("a", 150_300, 0), ("b", 260_100, 0), ("rej", 4, 1)]
    got = host_parse(contigs, batch["paf"], batch["seqs"], barcodes=batch["barcodes"], nbarcodes=4, n_threads=threads)
    want = _oracle_expansion(batch["paf"], batch["seqs"], ["a", "b"], barcodes=batch["barcodes"])
    for key, w in zip(("contig", "pos", "code", "barcode"), want):
        assert np.array_equal(got[key], w), key
    # summary: one entry per chosen mapping, in first-appearance order of the read ids
    from oracle.pafcigar import parse_paf, best_mapper
    recs = [best_mapper(r) if len(r) > 1 else r[0] for r in parse_paf(batch["paf"], min_len=200).values()]
    assert [got["ids"][i] for i in got["read_idx"]] == [r.qname for r in recs]
    assert np.array_equal(got["tstart"], [r.tstart for r in recs])
    assert np.array_equal(got["rev"], [int(r.rev) for r in recs])


def test_read_packing_two_bits_per_base(monkeypatch):
    """include/bossx.h (bossx_pack_reads2): a batch of nothing but A C G T crosses PCIe with two bits per base.  Vector packer
    against scalar packer against numpy at every length 0..140 and two long ones; a read with any other byte is reported dirty
    by both (its bytes are void: the staging packs the batch again as nibbles)."""
    import ctypes as C
    from boss_runs_amd import _lib
    lib = _lib.load()
    table = np.full(256, 255, np.uint8)
    for ch, v in zip("ACGT", range(4)):
        table[ord(ch)] = v
    rng = np.random.default_rng(23)

    def pack(raw):
        nbytes = (len(raw) + 3) // 4
        dst = np.full(nbytes + 16, 0xEE, np.uint8)
        dirty = C.c_int32(-1)
        assert lib.bossx_pack_reads2(raw, len(raw), dst.ctypes.data, C.byref(dirty)) == 0
        assert np.all(dst[nbytes:] == 0xEE)           # nothing past the read's own bytes
        return dst[:nbytes].copy(), dirty.value

    for n in list(range(0, 141)) + [1000, 4099]:
        clean = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
        c = table[np.frombuffer(clean, np.uint8)]
        c = np.append(c, np.zeros((-n) % 4, np.uint8))
        want = (c[0::4] | (c[1::4] << 2) | (c[2::4] << 4) | (c[3::4] << 6)).astype(np.uint8)
        for scalar in (False, True):
            if scalar:
                monkeypatch.setenv("BOSSX_PACK_SCALAR", "1")
            else:
                monkeypatch.delenv("BOSSX_PACK_SCALAR", raising=False)
            got, d = pack(clean)
            assert d == 0 and np.array_equal(got, want), (n, scalar)
            if n:
                for alphabet in (b"0123456789", b"acgtN", bytes(range(1, 65)) + bytes(range(0x55, 256))):
                    x = bytearray(clean)
                    x[int(rng.integers(0, n))] = alphabet[int(rng.integers(0, len(alphabet)))]
                    assert pack(bytes(x))[1] == 1, (n, scalar)


def test_read_packing_four_bits_per_base(monkeypatch):
    """include/bossx.h (bossx_pack_reads): the reads cross PCIe as four bits per base.  The vector packer against the
    scalar table and both against a numpy restatement of the code table, at every length 0..130 (tails, odd lengths), with
    clean reads, digits (what sequences.py:766's np.fromstring reads as an index), lower case, N and arbitrary bytes."""
    import ctypes as C
    from boss_runs_amd import _lib
    lib = _lib.load()
    table = np.full(256, 15, np.uint8)
    for ch, v in zip("ACGT012347", range(10)):
        table[ord(ch)] = v
    rng = np.random.default_rng(17)

    def pack(raw):
        dst = np.full((len(raw) + 1) // 2 + 8, 0xEE, np.uint8)
        dirty = C.c_int32(-1)
        assert lib.bossx_pack_reads(raw, len(raw), dst.ctypes.data, C.byref(dirty)) == 0
        assert np.all(dst[(len(raw) + 1) // 2:] == 0xEE)          # nothing past the read's own bytes
        return dst[:(len(raw) + 1) // 2].copy(), dirty.value

    def want(raw):
        c = table[np.frombuffer(raw, np.uint8)]
        if c.size & 1:
            c = np.append(c, np.uint8(15))
        return (c[0::2] | (c[1::2] << 4)).astype(np.uint8), int(any(ch not in b"ACGT" for ch in raw))

    for n in list(range(0, 131)) + [1000, 4097]:
        clean = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
        kinds = [clean]
        if n:
            for alphabet in (b"ACGT0123456789", b"ACGTacgtN", bytes(range(1, 256))):
                x = bytearray(clean)
                for k in rng.integers(0, n, 1 + n // 37):
                    x[k] = alphabet[int(rng.integers(0, len(alphabet)))]
                kinds.append(bytes(x))
            kinds.append(bytes(rng.integers(1, 256, n, dtype=np.uint8)))
        for raw in kinds:
            w, wd = want(raw)
            for scalar in (False, True):
                if scalar:
                    monkeypatch.setenv("BOSSX_PACK_SCALAR", "1")
                else:
                    monkeypatch.delenv("BOSSX_PACK_SCALAR", raising=False)
                got, d = pack(raw)
                assert np.array_equal(got, w) and d == wd, (n, scalar)


def test_dirty_reads_are_flagged_after_the_two_phase_job():
    """The host half of staging runs as a two-phase job (paf_host.cpp, WorkPool::start / JobGuard): the
    line parse + read index first, the caller's tasks — here, as in bossx_stage_batch_ptrs, slices of
    reads looked at for bytes other than A/C/G/T — finish while the calling thread groups and plans,
    and only after their collection are the plans of such reads flagged for the device walk's base
    check.  bossx_host_parse drives exactly that and fails if the tasks were not all done at the
    collection or if a flag does not follow its read.  A byte other than A/C/G/T in the soft-clipped
    flank makes a read dirty without making the batch fail."""
    from boss_runs_amd import synth
    from boss_runs_amd.engine import host_parse
    ref = synth.make_reference([150_300, 260_100], seed=5, names=["a", "b"])
    batch = synth.make_batch(ref, 900, seed=91, mean_len=4000.0, nbarcodes=1, extras=False)
    contigs = [("a", 150_300, 0), ("b", 260_100, 0)]
    clean = host_parse(contigs, batch["paf"], batch["seqs"], n_threads=4)
    seqs = dict(batch["seqs"])
    n_dirty = 0
    for line in batch["paf"].split("\n"):
        f = line.split("\t")
        if len(f) > 3 and int(f[2]) >= 1 and n_dirty < 200:       # qstart >= 1: base 0 is outside the alignment
            seqs[f[0]] = "N" + seqs[f[0]][1:]
            n_dirty += 1
    assert n_dirty >= 50
    for threads in (1, 4, 8):
        got = host_parse(contigs, batch["paf"], seqs, n_threads=threads)
        for key in ("contig", "pos", "code", "barcode"):
            assert np.array_equal(got[key], clean[key]), key


def test_native_parser_errors_are_first_in_record_order():
    """With several faulty records the failure reported is the first one in record order,
    whatever the thread partition (the reference raises inside its per-record loop)."""
    from boss_runs_amd import synth
    from boss_runs_amd.engine import host_parse
    ref = synth.make_reference([200_000], seed=9, names=["c"])
    batch = synth.make_batch(ref, 600, seed=5, mean_len=3000.0)
    lines = batch["paf"].strip().split("\n")
    contigs = [("c", 200_000, 0)]

    def corrupt(i, how):
        f = lines[i].split("\t")
        if how == "cigar":
            f = [x.replace("cg:Z:", "cg:Z:7M", 1) if x.startswith("cg:Z:") else x for x in f]
        elif how == "name":
            f[0] = "ghost_read"
        lines_c = list(lines)
        lines_c[i] = "\t".join(f)
        return lines_c
    for threads in (1, 4):
        bad = corrupt(400, "cigar")
        with pytest.raises(ValueError, match="CIGAR"):
            host_parse(contigs, "\n".join(bad), batch["seqs"], n_threads=threads, expand=False)
        bad = corrupt(400, "cigar")
        bad[100] = corrupt(100, "name")[100]
        with pytest.raises(KeyError):
            host_parse(contigs, "\n".join(bad), batch["seqs"], n_threads=threads, expand=False)
        bad = corrupt(100, "cigar")
        bad[400] = corrupt(400, "name")[400]
        with pytest.raises(ValueError, match="CIGAR"):
            host_parse(contigs, "\n".join(bad), batch["seqs"], n_threads=threads, expand=False)
        bad = list(lines)
        bad[300] = "\t".join(bad[300].split("\t")[:7])
        with pytest.raises(IndexError, match="PAF line 301"):          # f[i] on a short line (paf.py:50-51)
            host_parse(contigs, "\n".join(bad), batch["seqs"], n_threads=threads, expand=False)


def test_native_parser_index_error_class():
    """IndexError parity (reference.py:138: np.add.at with an out-of-range column / a slice cut
    short by the contig end).  A base other than A/C/G/T (any case) inside an M run, or a mapping
    that runs past its contig, fails the whole batch BEFORE anything is staged; the same byte in
    a clipped flank or inside an insertion is harmless, as in the reference.  Such a failure never
    hides a ValueError / KeyError of another read (the reference raises those first, in
    convert_records)."""
    from boss_runs_amd import synth
    from boss_runs_amd.engine import host_parse
    from oracle.pafcigar import parse_paf, convert_records
    ref = synth.make_reference([200_000], seed=9, names=["c"])
    batch = synth.make_batch(ref, 400, seed=6, mean_len=3000.0, extras=False)
    contigs = [("c", 200_000, 0)]
    lines = batch["paf"].strip().split("\n")
    seqs = batch["seqs"]
    good = host_parse(contigs, batch["paf"], seqs, expand=False)["aligned"]

    def first_op_read(strand, need_flank):
        for i, l in enumerate(lines):
            f = l.split("\t")
            if f[4] == strand and (not need_flank or (int(f[2]) > 2 and int(f[1]) - int(f[3]) > 2)) \
                    and "I" in f[-2 if f[-1].startswith("AS") else -1]:
                return i, f
        raise AssertionError("no suitable read")

    for strand in "+-":
        i, f = first_op_read(strand, True)
        rid, qlen, qs, qe = f[0], int(f[1]), int(f[2]), int(f[3])
        s = seqs[rid]
        # position (in read coordinates) of the first aligned base as the walk reads it
        first = qs if strand == "+" else qe - 1
        for ch in "NnaR":
            bad = dict(seqs)
            bad[rid] = s[:first] + ch + s[first + 1:]
            for threads in (1, 4):
                with pytest.raises(IndexError, match="A/C/G/T"):
                    host_parse(contigs, batch["paf"], bad, n_threads=threads, expand=False)
            # the oracle (numpy) raises IndexError on the same input
            with pytest.raises(IndexError):
                inc = convert_records(parse_paf(batch["paf"], min_len=200), bad)
                tmp = np.zeros((200_000, 5, 1), dtype="uint16")
                for (st, en, q, bc) in inc["c"]:
                    np.add.at(tmp[st:en], (np.arange(q.shape[0]), q, 0), 1)
        # the same byte in a clipped flank: ignored
        ok = dict(seqs)
        ok[rid] = "N" + s[1:-1] + "N"
        assert host_parse(contigs, batch["paf"], ok, expand=False)["aligned"] == good
        # ... and inside an insertion: consumed, never counted (sequences.py:781)
        cg = [x for x in f if x.startswith("cg:Z:")][0][5:]
        import re
        q = 0
        ins_at = None
        for n, op in re.findall(r"(\d+)([MIDNSHP=XB])", cg):
            n = int(n)
            if op == "I":
                ins_at = q
                break
            if op != "D":
                q += n
        rd = s if strand == "+" else None
        if strand == "+":
            ok = dict(seqs)
            ok[rid] = s[:qs + ins_at] + "N" + s[qs + ins_at + 1:]
            assert host_parse(contigs, batch["paf"], ok, expand=False)["aligned"] == good
        else:       # the walk reads the reverse complement: alignment column q is read base qe-1-q
            ok = dict(seqs)
            p = qe - 1 - ins_at
            ok[rid] = s[:p] + "N" + s[p + 1:]
            assert host_parse(contigs, batch["paf"], ok, expand=False)["aligned"] == good
    # a mapping that runs past the end of its contig
    short = [("c", 150_000, 0)]
    with pytest.raises(IndexError, match="past the end"):
        host_parse(short, batch["paf"], seqs, expand=False)
    # precedence: a malformed CIGAR in a LATER read wins over the IndexError of an earlier one
    i, f = first_op_read("+", True)
    bad = dict(seqs)
    rid, qs = f[0], int(f[2])
    bad[rid] = seqs[rid][:qs] + "N" + seqs[rid][qs + 1:]
    j = len(lines) - 5
    assert j > i
    g = lines[j].split("\t")
    g = [x.replace("cg:Z:", "cg:Z:7M", 1) if x.startswith("cg:Z:") else x for x in g]
    lines_c = list(lines)
    lines_c[j] = "\t".join(g)
    for threads in (1, 4):
        with pytest.raises(ValueError, match="CIGAR"):
            host_parse(contigs, "\n".join(lines_c), bad, n_threads=threads, expand=False)
    # barcode index out of range: IndexError class as well
    with pytest.raises(IndexError, match="barcode"):
        host_parse(contigs, batch["paf"], seqs, barcodes=[3] * len(seqs), nbarcodes=2, expand=False)


def test_readlength_fast_path_equals_float_scan():
    """ReadlengthDist.update decides approx_ccl on the exact integer histogram; it must agree
    with the reference's float cumsum scan (ccl_approx_constant) and with the oracle on every
    distribution, including degenerate ones and uint16 wrap-around."""
    from oracle.dists import OReadlengthDist
    rng = np.random.default_rng(12)
    cases = [np.array([5000]), np.array([5000] * 10), np.array([900, 60000]), np.array([801, 802, 803]),
             np.array([1_500_000, 2000, 2000]), np.arange(801, 1801), np.array([100, 200])]
    for _ in range(40):
        n = int(rng.integers(1, 5000))
        cases.append(np.clip(rng.gamma(2, rng.uniform(500, 6000), n), 1, 2_000_000).astype(np.int64))
    p, o, q = ReadlengthDist(), OReadlengthDist(), ReadlengthDist()
    for lens in cases:
        p.update(lens)                                   # native (bossx_rl_update) when the library is built
        q._update_numpy(np.asarray(lens, dtype=np.int64))   # the numpy form of the same step
        o.update({"r%d" % i: int(v) for i, v in enumerate(lens)})
        assert np.array_equal(p.approx_ccl, o.approx_ccl)
        assert np.array_equal(q.approx_ccl, o.approx_ccl)
        assert np.array_equal(p.approx_ccl, p.ccl_approx_constant())
        if hasattr(o, "time_cost"):
            assert q.lam == o.lam and q.time_cost == o.time_cost and q.longest_read == o.longest_read
            assert p.lam == o.lam and p.time_cost == o.time_cost and p.longest_read == o.longest_read
            assert np.array_equal(p.L, o.L) and np.array_equal(p.ccl, o.ccl)
    # fresh objects on single batches (the first update matters: few reads, coarse steps)
    for lens in cases:
        p, o = ReadlengthDist(), OReadlengthDist()
        p.update(lens)
        o.update({"r%d" % i: int(v) for i, v in enumerate(lens)})
        assert np.array_equal(p.approx_ccl, o.approx_ccl)
        assert hasattr(p, "time_cost") == hasattr(o, "time_cost")
    # 65536 reads of one length wrap its uint16 counter to zero (readlengthdist.py:23)
    p, o = ReadlengthDist(), OReadlengthDist()
    lens = np.concatenate([np.full(65536, 7000), np.full(10, 3000)])
    p.update(lens)
    o.update({"r%d" % i: int(v) for i, v in enumerate(lens)})
    assert np.array_equal(p.approx_ccl, o.approx_ccl) and p.lam == o.lam and p.longest_read == o.longest_read == 3000


@pytest.mark.parametrize("plan_range", [None, "1"])
def test_native_front_end_error_classes_equal_the_reference(plan_range, monkeypatch):
    """The native PAF / CIGAR front end against tests/golden/g_errors.json (what the reference itself does
    with 60-odd malformed / unusual batches, scenarios.error_cases): the same exception class where the
    reference raises, the same coverage where it does not — truncated lines (IndexError), non-integer
    columns (TypeError only where the reference ever looks at them), tag syntax (ValueError / KeyError),
    CIGAR text the reference's regex skips over, shape mismatches (ValueError) vs span assertions
    (AssertionError), reads with other letters (IndexError) ..."""
    if plan_range:          # (the pre-pass in ranges of one record on the pool's threads: see the fuzz test below)
        monkeypatch.setenv("BOSSX_PLAN_RANGE", plan_range)
    import json
    from scenarios import GOLDEN, digest, error_cases
    from boss_runs_amd.engine import host_parse
    contigs, cases = error_cases()
    gold = json.load(open(os.path.join(GOLDEN, "g_errors.json")))
    clist = [(n, c.shape[0], 0) for n, c in contigs]
    bad = []
    for name, paf_text, seqs in cases:
        for threads in (1, 3):
            try:
                out = host_parse(clist, paf_text, seqs, n_threads=threads, min_len=200)
                cov = {n: np.zeros((c.shape[0], 5, 1), dtype=np.uint16) for n, c in contigs}
                for k, (n, _) in enumerate(contigs):
                    sel = out["contig"] == k
                    np.add.at(cov[n], (out["pos"][sel], out["code"][sel].astype(np.int64), 0), 1)
                got = {"ok": digest(*[cov[n] for n in sorted(cov)])}
            except Exception as e:          # noqa: BLE001
                got = {"error": type(e).__name__}
            want = gold[name]
            if got.get("error") != want.get("error") or got.get("ok") != want.get("ok"):
                bad.append((name, threads, got, {k: want[k] for k in want if k in ("ok", "error")}))
    assert not bad, bad


def test_line_scanner_vector_equals_scalar_on_long_and_odd_lines(monkeypatch):
    """paf_host.cpp: scan_line_avx2 / scan_line_scalar (round 6: ONE pass over a line for its tabs and colons instead of a memchr for
    the newline, a find per tab and finds for every tag's colons).  Realistic batches (CIGAR tags of 1-6 KB: tabs, colons and newlines at
    every offset inside the 32-byte blocks) plus lines bent on purpose — extra colons in tags and in names, colons inside the CIGAR
    value, tags and tabs by the hundred (more specials than the scanner's table holds: the old splitting takes over), blank lines,
    whitespace at the ends, no newline at the end — give the same summary, the same expansion or the same exception with either form."""
    from boss_runs_amd import synth
    from boss_runs_amd.engine import host_parse
    contigs = synth.make_reference([120_000, 60_000], seed=5, names=["s1", "s2"])
    clist = [(n, c.shape[0], 0) for n, c in contigs]
    rng = np.random.default_rng(99)
    batch = synth.make_batch(contigs, 60, seed=77, mean_len=6000.0)
    lines = batch["paf"].split("\n")
    if lines and not lines[-1]:
        lines.pop()

    def variants():
        yield "plain", "\n".join(lines) + "\n"
        yield "no final newline", "\n".join(lines)
        for trial in range(40):
            ls = list(lines)
            k = int(rng.integers(0, len(ls)))
            f = ls[k].split("\t")
            kind = trial % 8
            if kind == 0:
                f.append("zz:Z:" + ":".join("x" * int(rng.integers(0, 40)) for _ in range(int(rng.integers(2, 5)))))      # too many colons
            elif kind == 1:
                f[0] = f[0] + ":a:b"                                                                          # colons in a name: nothing
            elif kind == 2:
                f = [x.replace("cg:Z:", "cg:Z::") if x.startswith("cg:Z:") else x for x in f]                  # a colon inside the value
            elif kind == 3:
                f += ["t%d:i:%d" % (j, j) for j in range(int(rng.integers(60, 120)))]                        # > 192 specials
            elif kind == 4:
                f += [""] * int(rng.integers(200, 260))                                                        # tabs by the hundred: empty tags
            elif kind == 5:
                ls.insert(k, "")                                                                               # a blank line
                yield "blank %d" % trial, "\n".join(ls) + "\n"
                continue
            elif kind == 6:
                ls[k] = " \t" + ls[k] + "\t \r"                                                                 # stripped like str.strip()
                yield "ws %d" % trial, "\n".join(ls) + "\n"
                continue
            else:
                f.append("AS:i:" + "9" * int(rng.integers(1, 25)))                                             # 1-24 digits: fast path and beyond
            ls[k] = "\t".join(f)
            yield "kind%d %d" % (kind, trial), "\n".join(ls) + "\n"

    def run(text):
        try:
            out = host_parse(clist, text, batch["seqs"], n_threads=2, min_len=200)
            return ("ok", out["read_idx"].tobytes(), out["tstart"].tobytes(), out["pos"].tobytes(), out["code"].tobytes())
        except Exception as e:          # noqa: BLE001
            return ("error", type(e).__name__)

    seen_err = seen_ok = 0
    for name, text in variants():
        monkeypatch.delenv("BOSSX_PARSE_SCALAR", raising=False)
        a = run(text)
        monkeypatch.setenv("BOSSX_PARSE_SCALAR", "1")
        b = run(text)
        assert a == b, name
        seen_err += a[0] == "error"
        seen_ok += a[0] == "ok"
    assert seen_err >= 5 and seen_ok >= 10, (seen_err, seen_ok)


@pytest.mark.parametrize("plan_range", [None, "1", "scalar_scan"])
def test_native_front_end_fuzz_error_classes_equal_the_reference(plan_range, monkeypatch):
    """(`plan_range` = "scalar_scan": BOSSX_PARSE_SCALAR=1 — the line scanner's byte loop instead of its AVX2 form: one pass over a line
    for its tabs and colons, round 6.)  (`plan_range` = "1": the pre-pass over the chosen mappings split into ranges of ONE record on the pool's threads — the form a
    4000-read batch takes with ranges of 256+, forced onto these small batches: the first failure in record order, the summary, the
    running sums and the bitmap of touched tiles must come out as from the single range.)
    The native PAF / CIGAR front end (bossx_host_parse: line parser, grouping, pre-pass, host walk, and the plans
    of the device walk) against tests/golden/g_errors_fuzz.json — what the reference itself does with 2,000 seeded
    random mutations (scenarios.fuzz_error_cases), case by case: the same exception class where it raises, the same
    coverage where it goes on (Python slicing of the read and of the coverage array included: a qlen column of 0
    or -7 on a '-' mapping still selects bases, two negative target coordinates count from the contig's end)."""
    import json
    from scenarios import GOLDEN, digest, fuzz_error_cases
    from boss_runs_amd.engine import host_parse
    if plan_range == "scalar_scan":
        monkeypatch.setenv("BOSSX_PARSE_SCALAR", "1")
    elif plan_range:
        monkeypatch.setenv("BOSSX_PLAN_RANGE", plan_range)
    contigs, cases = fuzz_error_cases()
    gold = json.load(open(os.path.join(GOLDEN, "g_errors_fuzz.json")))
    clist = [(n, c.shape[0], 0) for n, c in contigs]
    bad = []
    for name, paf_text, seqs in cases:
        for threads in (1, 3):
            try:
                out = host_parse(clist, paf_text, seqs, n_threads=threads, min_len=200)
                cov = {n: np.zeros((c.shape[0], 5, 1), dtype=np.uint16) for n, c in contigs}
                for k, (n, _) in enumerate(contigs):
                    sel = out["contig"] == k
                    np.add.at(cov[n], (out["pos"][sel], out["code"][sel].astype(np.int64), 0), 1)
                got = {"ok": digest(*[cov[n] for n in sorted(cov)])[:16]}
            except Exception as e:          # noqa: BLE001
                got = {"error": type(e).__name__}
            if got != gold[name]:
                bad.append((name, threads, got, gold[name]))
    assert not bad, (len(bad), bad[:10])
