"""Full-size parity of the HIP path against the oracle for the BASELINE.json configurations that
fit one GPU, plus the geometry / error cases the small scenarios do not reach:

* chr20 + chr21 (+ a 16.5 kb "MT"), ploidy 2, reject_refs=MT, ~8x preloaded coverage — the
  whole oracle at 111 Mb;
* the barcoded panel, 10 x 5 Mb x 8 barcodes (split ingest-only + plain sweep) — the oracle's
  per-contig stages on two of the ten contigs (per-site and per-bin work is contig-local) and
  the oracle's strategy stage (find_strat_thread + _distribute_strategy) over all of them;
* 40 contigs (more dropout thresholds than fit the sweep's launch arguments);
* IndexError-class batches, which must leave the state untouched.
Everything is compared bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _poisson_coverage(ref_codes, rng, depth=8.0, hot=None):
    """uint16[L, 5] synthetic pile-up: ~93 % reference base, substitutions, deletions; `hot` =
    (start, end, extra) adds depth so that some sites reach the 30-observation cap."""
    L = ref_codes.shape[0]
    cov = np.zeros((L, 5), dtype=np.uint16)
    idx = np.arange(L)
    main = rng.poisson(depth * 0.93, L).astype(np.uint16)
    if hot is not None:
        main[hot[0]:hot[1]] += np.uint16(hot[2])
    cov[idx, ref_codes] = main
    for k in (1, 2, 3):
        cov[idx, (ref_codes + k) & 3] = rng.poisson(depth * 0.04 / 3, L).astype(np.uint16)
    cov[:, 4] = rng.poisson(depth * 0.03, L).astype(np.uint16)
    return cov


def _compare_contig(pc, oc, updated, tag):
    assert np.array_equal(pc.strat, oc.strat), tag
    assert np.array_equal(pc.coverage, oc.coverage), tag
    assert np.array_equal(pc.bucket_switches, oc.bucket_switches), tag
    assert np.array_equal(pc.switched_on, oc.switched_on), tag
    assert np.array_equal(pc.scores, oc.scores), tag
    assert np.array_equal(pc.entropy, oc.entropy), tag
    if updated:
        assert np.array_equal(pc.scores_ds, oc.scores_ds), tag
        assert np.array_equal(pc.additional_benefit, oc.additional_benefit), tag


def test_chr20_21_full_size_vs_oracle(in_tmp):
    """BASELINE configs[2]: 64,444,167 + 46,709,983 bp, ploidy 2, reject_refs=MT (the 16,569-bp MT
    is dropped by the 100-kb filter before reject_refs is consulted, reference.py:330-331).  Both
    sides start from the same ~8x pile-up (bossx_import), every site is scored once, then two
    4000-read batches: masks, threshold, statistics, bucket switches, coverage, scores, entropy,
    bin sums and benefits of BOTH contigs equal the oracle's, bit for bit."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from oracle.pipeline import OracleRuns
    lens = [synth.CHR20_LEN, synth.CHR21_LEN, synth.MT_LEN]
    names = ["chr20", "chr21", "MT"]
    contigs = synth.make_reference(lens, seed=1, names=names)
    strs = [(n, synth.codes_to_str(c)) for n, c in contigs]
    args = BossConfig()
    args.general.name = "chr2021"
    args.optional.ploidy = 2
    args.optional.reject_refs = "MT"
    args.gpu.track_entropy = True
    runs = BossRuns(args)
    runs.init(contigs=strs)
    runs.keep_stats = True
    runs.write_masks = False
    runs.log_fractions = False
    o = OracleRuns(strs, ploidy=2, reject_refs={"MT"})
    assert list(runs.contigs) == list(o.contigs) == ["chr20", "chr21"]
    assert runs.ref.n_sites == o.n_sites == lens[0] + lens[1]
    rng = np.random.default_rng(99)
    for name, codes in contigs[:2]:
        cov = _poisson_coverage(codes, rng, hot=(1_000_000, 1_300_000, 24) if name == "chr21" else None)
        oc, pc = o.contigs[name], runs.contigs[name]
        oc.coverage[:, :, 0] = cov
        oc.change_mask[:] = True
        runs.engine.import_state(pc.index, "coverage", cov[:, :, None])
        runs.engine.import_state(pc.index, "touched", np.ones(codes.shape[0], dtype=np.uint8))
        del cov
    prime_rl = {"p%d" % i: 3000 + 41 * i for i in range(300)}
    runs.rl_dist.update(prime_rl)
    o.rl_dist.update(prime_rl)

    def compare(step):
        assert runs.threshold == o.threshold and o.threshold is not None, step
        d = o.detail
        assert runs.last_stats["normaliser"] == d["normaliser"], step
        assert np.array_equal(runs.last_stats["exponents"], d["exponents"]), step
        assert np.array_equal(runs.last_stats["counts"], d["counts"]), step
        assert runs.last_stats["strat_size"] == d["strat_size"], step
        assert np.allclose(runs.last_stats["f_grid"], d["f_grid"], rtol=1e-11)
        for name in ("chr20", "chr21"):
            _compare_contig(runs.contigs[name], o.contigs[name], True, (step, name))
        assert np.array_equal(runs.read_starts.merge(), o.read_starts.merge())

    # every site scored once (change_mask all true / touched all set), no new reads
    runs.update_wrapper()
    o.update_wrapper()
    compare("prime")
    c21 = o.contigs["chr21"]
    assert (c21.coverage[1_000_000:1_300_000].sum(axis=(1, 2)) >= 30).any()      # capped sites exist
    assert (c21.scores == 0.0).any()                                              # dropout is active
    for b in range(2):
        batch = synth.make_batch(contigs[:2], 4000, seed=500 + b, extras=True)
        if b == 0:      # a read mapped to the dropped MT contig: ignored by both sides
            first = batch["paf"].split("\n")[0].split("\t")
            first[5], first[6] = "MT", str(synth.MT_LEN)
            first[7], first[8] = "100", str(100 + int(first[8]) - int(first[7]))
            batch["paf"] = "\t".join(first) + "\n" + batch["paf"].split("\n", 1)[1]
        o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"])
        runs.rl_dist.update(batch["read_lengths"])
        runs.process_batch_paf(batch["paf"], batch["seqs"])
        assert np.array_equal(runs.rl_dist.approx_ccl, o.rl_dist.approx_ccl)
        compare(b)
    frac = runs.contigs["chr20"].strat.mean()
    assert 0.0 < frac < 1.0


class _StandIn:
    """What oracle.strategy.distribute needs of a contig."""

    def __init__(self, length, nb, strat, switches):
        self.length, self.nb, self.bucket_size = length, nb, 20_000
        self.strat, self.bucket_switches = strat, switches


def test_barcoded_panel_full_size_vs_oracle(in_tmp):
    """BASELINE configs[4] at full size: 10 x 5 Mb, 8 barcodes co-resident (split sweep: ingest-only
    launch + plain launch), ~6x preloaded depth per barcode so dropout acts on whole rows across
    barcodes.  Oracle: Contig stages (increment_coverage, update_scores, modify_scores,
    check_buckets, calc_smu, calc_u) on two of the ten contigs; find_strat_thread and
    _distribute_strategy on the merged benefit of all ten (the other eight blocks come from the
    device, whose per-contig stages are the ones just verified).  Plus conservation and
    idempotence at full size."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from oracle.contig import OContig, adjust_length
    from oracle.dists import OReadlengthDist, OReadStartDist
    from oracle.model import SiteModel, PatternCache
    from oracle.pafcigar import parse_paf, convert_records
    from oracle.strategy import find_strategy, distribute
    nb = 8
    lens = [5_000_000] * 10
    names = ["bac%02d" % i for i in range(10)]
    contigs = synth.make_reference(lens, seed=4, names=names)
    strs = [(n, synth.codes_to_str(c)) for n, c in contigs]
    args = BossConfig()
    args.general.name = "panel"
    args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = BossRuns(args)
    runs.init(contigs=strs)
    runs.write_masks = False
    runs.log_fractions = False
    eng = runs.engine
    eng.preload_coverage(6.0, seed=11)
    subset = ["bac02", "bac07"]
    cache = PatternCache(SiteModel(1))
    ocs = {}
    for n in subset:
        oc = OContig(n, dict(strs)[n], nbarcodes=nb)
        pc = runs.contigs[n]
        oc.coverage[:] = pc.coverage
        oc.change_mask[:] = eng.export(pc.index, "touched")[:, None].astype(bool)
        ocs[n] = oc
    total0 = sum(int(runs.contigs[n].coverage.sum(dtype=np.uint64)) for n in subset)
    o_rl = OReadlengthDist()
    o_rs = OReadStartDist({n: runs.contigs[n] for n in names})
    expect = {n: np.ones((L // 100, 2, nb), dtype=bool) for n, L in zip(names, lens)}
    aligned_subset = 0

    def oracle_update(step, inc):
        for n, oc in ocs.items():
            if inc is not None:
                oc.increment_coverage(inc.get(n, []))
            oc.update_scores(cache)
            oc.modify_scores()
            oc.check_buckets(threshold=args.optional.bucket_threshold)
            oc.calc_smu()
            oc.calc_u(o_rl.approx_ccl)
            pc = runs.contigs[n]
            assert np.array_equal(pc.coverage, oc.coverage), (step, n)
            assert np.array_equal(pc.scores, oc.scores), (step, n)
            assert np.array_equal(pc.entropy, oc.entropy), (step, n)
            assert np.array_equal(pc.bucket_switches, oc.bucket_switches), (step, n)
            assert np.array_equal(pc.scores_ds, oc.scores_ds), (step, n)
            assert np.array_equal(pc.additional_benefit, oc.additional_benefit), (step, n)
            assert (oc.scores == 0.0).any(axis=1).sum() > 1000            # row-wide dropout is active
        # strategy stage over all ten contigs
        benefit = np.concatenate([runs.contigs[n].additional_benefit for n in names])
        fhat = np.repeat(o_rs.update_f_pointmass()[:, :, np.newaxis], nb, axis=2)
        target = sum(lens) // 100
        detail = {}
        strat, thr = find_strategy(adjust_length(target, benefit), adjust_length(target, benefit),
                                   adjust_length(target, fhat), o_rl.time_cost, detail=detail)
        assert runs.threshold == thr, step
        assert runs.last_stats["strat_size"] == detail["strat_size"], step
        assert runs.last_stats["normaliser"] == detail["normaliser"], step
        stand = {n: _StandIn(L, nb, expect[n], runs.contigs[n].bucket_switches) for n, L in zip(names, lens)}
        distribute(stand, strat)
        for n in names:
            assert np.array_equal(runs.contigs[n].strat, expect[n]), (step, n)

    prime_rl = {"p%d" % i: 2500 + 53 * i for i in range(300)}
    runs.rl_dist.update(prime_rl)
    o_rl.update(prime_rl)
    runs.update_wrapper()
    oracle_update("prime", None)
    for b in range(2):
        batch = synth.make_batch(contigs, 4000, seed=800 + b, nbarcodes=nb, extras=True)
        paf = parse_paf(batch["paf"], min_len=200)
        for recs in paf.values():
            for r in recs:
                r.barcode = batch["barcodes"][r.qname]
        o_rl.update(batch["read_lengths"])
        o_rs.count_read_starts(paf)
        inc = convert_records({k: v for k, v in paf.items() if any(r.tname in subset for r in v)}, batch["seqs"])
        aligned_subset += sum(q.shape[0] for n in subset for (_, _, q, _) in inc.get(n, []))
        runs.rl_dist.update(batch["read_lengths"])
        runs.process_batch_paf(batch["paf"], batch["seqs"], barcodes=batch["barcodes"])
        assert np.array_equal(runs.rl_dist.approx_ccl, o_rl.approx_ccl)
        assert np.array_equal(runs.read_starts.merge(), o_rs.merge())
        oracle_update(b, inc)
    # conservation: every aligned base of the subset's reads counted exactly once
    total1 = sum(int(runs.contigs[n].coverage.sum(dtype=np.uint64)) for n in subset)
    assert total1 - total0 == aligned_subset > 0
    # bucket sums = coverage sums; an update without new reads changes nothing
    c = runs.contigs["bac07"]
    depth = c.coverage.sum(axis=1, dtype=np.uint64)                       # [L, nb]
    bs = eng.bucket_sums(c.index)
    for b in range(nb):
        assert np.array_equal(bs[b], depth[:, b].reshape(-1, 20000).sum(axis=1))
    before = {n: runs.contigs[n].strat.copy() for n in names}
    thr1, sc1 = runs.threshold, c.scores
    runs.update_wrapper()
    assert runs.threshold == thr1 and np.array_equal(c.scores, sc1)
    for n in names:
        assert np.array_equal(runs.contigs[n].strat, before[n])
    assert 0.0 < np.mean([before[n].mean() for n in names]) < 1.0


def test_forty_contigs_vs_oracle(in_tmp):
    """More than 32 non-rejected contigs: the dropout thresholds no longer travel in the sweep's
    launch arguments (bossx.hip launch_sweep), the row drift of _distribute_strategy reaches 39
    rows, and one contig in the middle is rejected.  Whole oracle, every array, every update."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from oracle.pipeline import OracleRuns
    rng = np.random.default_rng(40)
    lens = [int(x) for x in rng.integers(100_000, 300_000, size=41)]
    names = ["sc%02d" % i for i in range(41)]
    contigs = synth.make_reference(lens, seed=8, names=names)
    strs = [(n, synth.codes_to_str(c)) for n, c in contigs]
    args = BossConfig()
    args.general.name = "forty"
    args.optional.ploidy = 2
    args.optional.bucket_threshold = 1
    args.optional.reject_refs = "sc17"
    runs = BossRuns(args)
    runs.init(contigs=strs)
    runs.log_fractions = False
    assert len(runs.contigs_filt) == 40
    o = OracleRuns(strs, ploidy=2, bucket_threshold=1, reject_refs={"sc17"})
    # uneven depth: a few contigs deep enough for dropout (mean > 5), most not
    w = np.ones(41)
    w[[3, 20, 33]] = 14.0
    for b in range(4):
        batch = synth.make_batch(contigs, 3000, seed=4000 + b, mean_len=5000.0, start_weights=w)
        o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"])
        runs.rl_dist.update(batch["read_lengths"])
        runs.process_batch_paf(batch["paf"], batch["seqs"])
        assert runs.threshold == o.threshold, b
        for n, oc in o.contigs.items():
            pc = runs.contigs[n]
            assert np.array_equal(pc.strat, oc.strat), (b, n)
            if oc.rej:
                continue
            _compare_contig(pc, oc, o.threshold is not None, (b, n))
    assert o.threshold is not None
    assert any((c.scores == 0.0).any() for c in o.contigs_filt.values())      # dropout reached
    assert not all((c.scores == 0.0).any() for c in o.contigs_filt.values())


def test_index_error_batches_ingest_nothing(in_tmp):
    """reference.py:138: a base other than A/C/G/T inside an aligned run, or a mapping that runs
    past its contig, raises IndexError.  Here the whole batch is refused before anything reaches
    the device: coverage, the dropout bookkeeping and the masks stay as they were, and the run
    continues bit-identically to an oracle that never saw the bad batch."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from oracle.pipeline import OracleRuns
    contigs = synth.make_reference([180_000, 140_000], seed=12, names=["e1", "e2"])
    strs = [(n, synth.codes_to_str(c)) for n, c in contigs]
    args = BossConfig()
    args.general.name = "idxerr"
    args.optional.bucket_threshold = 0
    runs = BossRuns(args)
    runs.init(contigs=strs)
    o = OracleRuns(strs, bucket_threshold=0)

    def good(b):
        batch = synth.make_batch(contigs, 1500, seed=60 + b, mean_len=4000.0)
        o.process_batch(batch["paf"], batch["seqs"], read_lengths=batch["read_lengths"])
        runs.rl_dist.update(batch["read_lengths"])
        runs.process_batch_paf(batch["paf"], batch["seqs"])
        assert runs.threshold == o.threshold
        for n, oc in o.contigs.items():
            _compare_contig(runs.contigs[n], oc, True, (b, n))

    good(0)
    good(1)
    cov_before = {n: runs.contigs[n].coverage for n in runs.contigs}
    strat_before = {n: runs.contigs[n].strat.copy() for n in runs.contigs}
    bad = synth.make_batch(contigs, 1500, seed=99, mean_len=4000.0, extras=False)
    line = bad["paf"].split("\n")[700].split("\t")
    rid, qs, qe = line[0], int(line[2]), int(line[3])
    seqs = dict(bad["seqs"])
    mid = (qs + qe) // 2
    seqs[rid] = seqs[rid][:mid - 3] + "NNNNNN" + seqs[rid][mid + 3:]
    with pytest.raises(IndexError, match="A/C/G/T"):
        runs.process_batch_paf(bad["paf"], seqs)
    # a mapping that runs past the end of its contig (tstart/tend/CIGAR consistent, contig too short)
    f = bad["paf"].split("\n")[10].split("\t")
    L = int(f[6])
    span = int(f[8]) - int(f[7])
    f[7], f[8] = str(L - span + 50), str(L + 50)
    with pytest.raises(IndexError, match="past the end"):
        runs.process_batch_paf("\t".join(f), bad["seqs"])
    for n in runs.contigs:
        assert np.array_equal(runs.contigs[n].coverage, cov_before[n]), n
        assert np.array_equal(runs.contigs[n].strat, strat_before[n]), n
    good(2)
    good(3)


def test_grch38_geometry_one_gpu(in_tmp, monkeypatch):
    """BASELINE configs[3] geometry on ONE GPU (the whole 3.1 Gb reference fits one MI355X): the 24
    chromosome lengths of GRCh38 + MT (dropped by the 100-kb filter, reference.py:330-331) + three
    scaffolds >= 100 kb = 27 contigs, 1.5 M sweep tiles, 31 M bins, chr1's 2.49 M-bin move_sum chain,
    26 rows of drift in _distribute_strategy (core.py:125-155), packed masks (`mask_format="bits"`).
    ~8x preloaded depth, then two 4000-read updates.
      * oracle, every per-site / per-bin array, on the three scaffolds (contig-local stages);
      * the oracle's strategy stage (find_strat_thread + _distribute_strategy) over ALL 31 M bins:
        threshold, chosen exponent, normaliser, and every contig's mask;
      * properties at full size: every 20-kb bucket of every contig gains exactly the aligned bases
        the batch's chosen mappings put there (conservation), bucket sums = coverage sums (chr21),
        masks stay 1 outside switched-on buckets, an update without reads changes nothing."""
    from boss_runs_amd import synth
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    from oracle.contig import OContig, adjust_length, benefit_from_ds, smu_from_ds
    from oracle.dists import OReadlengthDist, OReadStartDist
    from oracle.model import SiteModel, PatternCache
    from oracle.pafcigar import parse_paf, convert_records, best_mapper
    from oracle.strategy import find_strategy, distribute
    names = ["chr%d" % (i + 1) for i in range(22)] + ["chrX", "chrY"]
    allc = list(zip(names, synth.GRCH38_LENS)) + [("MT", synth.MT_LEN), ("scaf_150k", 150_000),
                                                   ("scaf_250k", 250_000), ("scaf_400k", 400_000)]
    kept = [(n, L) for n, L in allc if L >= 100_000]
    assert len(kept) == 27
    codes = {n: np.random.default_rng(9000 + i).integers(0, 4, size=L, dtype=np.uint8) for i, (n, L) in enumerate(allc)}
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    args = BossConfig()
    args.general.name = "grch38"
    args.optional.ploidy = 2
    args.optional.reject_refs = "MT"
    args.optional.bucket_threshold = 8          # Poisson(8) buckets: about half of them switch on
    args.gpu.mask_format = "bits"
    args.gpu.track_entropy = True
    runs = BossRuns(args)
    runs.init(contigs=((n, acgt[codes[n]].tobytes()) for n, _ in allc))
    runs.write_masks = False
    runs.log_fractions = False
    eng = runs.engine
    assert list(runs.contigs) == [n for n, _ in kept]
    assert runs.ref.n_sites == sum(L for _, L in kept) == 3_089_069_832
    assert eng.merged_bins == sum(L // 100 + 1 for _, L in kept)
    eng.preload_coverage(8.0, seed=17)
    scaffolds = ["scaf_150k", "scaf_250k", "scaf_400k"]
    cache = PatternCache(SiteModel(2))
    ocs = {}
    for n in scaffolds:
        oc = OContig(n, acgt[codes[n]].tobytes().decode(), nbarcodes=1)
        pc = runs.contigs[n]
        oc.coverage[:] = pc.coverage
        oc.change_mask[:] = eng.export(pc.index, "touched")[:, None].astype(bool)
        ocs[n] = oc
    o_rl = OReadlengthDist()
    o_rs = OReadStartDist({n: runs.contigs[n] for n, _ in kept})
    expect = {n: np.ones((L // 100, 2, 1), dtype=bool) for n, L in kept}
    thr_b = args.optional.bucket_threshold

    def oracle_update(step, inc):
        for n, oc in ocs.items():
            if inc is not None:
                oc.increment_coverage(inc.get(n, []))
            oc.update_scores(cache)
            oc.modify_scores()
            oc.check_buckets(threshold=thr_b)
            oc.calc_smu()
            oc.calc_u(o_rl.approx_ccl)
            pc = runs.contigs[n]
            assert np.array_equal(pc.coverage, oc.coverage), (step, n)
            assert np.array_equal(pc.scores, oc.scores), (step, n)
            assert np.array_equal(pc.entropy, oc.entropy), (step, n)
            assert np.array_equal(pc.bucket_switches, oc.bucket_switches), (step, n)
            assert np.array_equal(pc.scores_ds, oc.scores_ds), (step, n)
            assert np.array_equal(pc.additional_benefit, oc.additional_benefit), (step, n)
        # chromosome-length chains: chr1's 2.49 M bin sums (and chr2's, chrX's) as the device holds them
        # through the oracle's move_sum arithmetic (oracle/movesum.c: Bottleneck's recurrence) -> additional_benefit, bit for
        # bit: strided rows, composed groups and — from the second batch on — rows left standing by tile stamps, at GRCh38 scale
        for n in ("chr1", "chr2", "chrX"):
            pc = runs.contigs[n]
            ds = pc.scores_ds
            _, add = benefit_from_ds(ds, smu_from_ds(ds), o_rl.approx_ccl)
            assert np.array_equal(pc.additional_benefit, add), (step, n)
            del ds, add
        # strategy stage over all 27 contigs: the other blocks come from the device, whose
        # per-contig stages are the ones verified above (and at 111 Mb in test_chr20_21_full_size_vs_oracle)
        benefit = np.concatenate([runs.contigs[n].additional_benefit for n, _ in kept])
        fhat = o_rs.update_f_pointmass()[:, :, np.newaxis]
        target = runs.ref.n_sites // 100
        detail = {}
        strat, thr = find_strategy(adjust_length(target, benefit), adjust_length(target, benefit),
                                   adjust_length(target, fhat), o_rl.time_cost, detail=detail)
        del benefit, fhat
        assert runs.threshold == thr, step
        assert runs.last_stats["strat_size"] == detail["strat_size"], step
        assert runs.last_stats["normaliser"] == detail["normaliser"], step
        sw = {n: runs.contigs[n].bucket_switches for n, _ in kept}
        stand = {n: _StandIn(L, 1, expect[n], sw[n]) for n, L in kept}
        distribute(stand, strat)
        some_on = some_off = False
        for n, L in kept:
            got = runs.contigs[n].strat
            assert np.array_equal(got, expect[n]), (step, n)
            on_rows = np.repeat(sw[n][:, 0], 200)[: L // 100]
            assert got[~on_rows].all(), (step, n)            # rows of buckets still off keep the initial 1
            some_on, some_off = some_on or on_rows.any(), some_off or (~on_rows).any()
        assert some_on and some_off

    prime_rl = {"p%d" % i: 3000 + 41 * i for i in range(300)}
    runs.rl_dist.update(prime_rl)
    o_rl.update(prime_rl)
    runs.update_wrapper()
    oracle_update("prime", None)
    contig_list = [(n, codes[n]) for n, _ in kept]
    # the scaffolds are 0.03 % of the genome: weight them up so that the oracle sees their increments
    w = np.ones(len(kept))
    w[-3:] = 400.0
    for b in range(2):
        bs0 = {n: eng.bucket_sums(runs.contigs[n].index)[0].copy() for n, _ in kept}
        batch = synth.make_batch(contig_list, 4000, seed=38000 + b, extras=True, start_weights=w)
        paf = parse_paf(batch["paf"], min_len=200)
        o_rl.update(batch["read_lengths"])
        o_rs.count_read_starts(paf)
        inc = convert_records({k: v for k, v in paf.items() if any(r.tname in scaffolds for r in v)}, batch["seqs"])
        runs.rl_dist.update(batch["read_lengths"])
        runs.process_batch_paf(batch["paf"], batch["seqs"])
        assert np.array_equal(runs.rl_dist.approx_ccl, o_rl.approx_ccl)
        assert np.array_equal(runs.read_starts.merge(), o_rs.merge())
        oracle_update(b, inc)
        # conservation, bucket by bucket, genome-wide: every emitted reference base of every chosen mapping
        gain = {n: np.zeros(L // 20000, dtype=np.uint64) for n, L in kept}
        for recs in paf.values():
            r = best_mapper(recs) if len(recs) > 1 else recs[0]
            if r.tname not in gain:
                continue
            g = gain[r.tname]
            for q in range(r.tstart // 20000, min((r.tend - 1) // 20000, g.shape[0] - 1) + 1):
                g[q] += np.uint64(max(0, min(r.tend, (q + 1) * 20000) - max(r.tstart, q * 20000)))
        for n, _ in kept:
            assert np.array_equal(eng.bucket_sums(runs.contigs[n].index)[0] - bs0[n], gain[n]), (b, n)
    c21 = runs.contigs["chr21"]
    depth = c21.coverage.sum(axis=1, dtype=np.uint64)[:, 0]
    nfull = c21.length // 20000
    assert np.array_equal(eng.bucket_sums(c21.index)[0], depth[: nfull * 20000].reshape(-1, 20000).sum(axis=1))
    del depth
    # idempotence: an update without new reads changes neither the threshold nor a single mask bit
    thr1, bits1 = runs.threshold, eng.strat_bits.copy()
    runs.update_wrapper()
    assert runs.threshold == thr1 and np.array_equal(eng.strat_bits, bits1)
    frac = float(np.mean([runs.contigs[n].strat.mean() for n in ("chr1", "chr21", "scaf_400k")]))
    assert 0.0 < frac < 1.0
    # ... and the OTHER form of the chain at chromosome length: the same bin sums through the serial kernel (the gated fallback of
    # the chunk-parallel form) equal the oracle's move_sum arithmetic as well
    from boss_runs_amd.runs import MULT
    monkeypatch.setenv("BOSSX_CHAIN_SERIAL_NOW", "1")
    cs0 = eng.chain_stats()
    eng.benefit(np.concatenate(([400 // 100], runs.rl_dist.approx_ccl // 100)).astype(np.int32), MULT)
    assert eng.chain_stats()["chunk_parallel_launches"] == cs0["chunk_parallel_launches"]      # (it WAS the serial kernel)
    for n in ("chr1", "chrX"):
        ds = runs.contigs[n].scores_ds
        _, add = benefit_from_ds(ds, smu_from_ds(ds), o_rl.approx_ccl)
        assert np.array_equal(runs.contigs[n].additional_benefit, add), n
    assert eng.chain_stats()["failed_checks"] == 0
