#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

Imports the reference (goldman-gp-ebi/BOSS-RUNS v0.4.0) through the import shims in
`tests/golden/_shims/` (tomllib->tomli, stub mappy / minknow_api, Bottleneck `move_sum`
restated), drives `BossRuns` exactly as `process_batch_runs` does minus the mapper
(boss/runs/core.py:202-224) on seeded synthetic inputs, and stores inputs' digests and the
reference's outputs as small `.npz` fixtures next to this script.

    cd /tmp && python3 /root/repo/tests/golden/make_golden.py
    cd /tmp && GOLDEN_ONLY=errors_fuzz python3 /root/repo/tests/golden/make_golden.py     # one fixture only: errors | errors_fuzz | sat

Fixtures whose values pass through `bottleneck.move_sum` carry `movesum_via_shim=1`: the system Python
has no Bottleneck, so the reference ran on `_shims/bottleneck.py`.  `make_movesum_golden.py` (run with
/opt/conda/bin/python3.9, which ships the real Bottleneck 1.3.2) re-derives every `additional_benefit`
column stored here with the real library and asserts it equal: the shim's results ARE Bottleneck's.
Nothing from /root/reference is copied: fixtures hold inputs (or their seeds) and outputs.
"""
import os
import sys
import tempfile
from io import StringIO

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "_shims"), "/root/reference", REPO]

sys.path.insert(0, os.path.join(REPO, "tests"))
from scenarios import (SCENARIOS, E2E_REJECT, E2E_BATCHES, digest, e2e_reference, e2e_batch,  # noqa: E402
                       batch_digest, saturated_coverage)
from boss_runs_amd import synth  # noqa: E402


def gen_tables(out):
    from boss.runs.sequences import Scoring
    rng = np.random.default_rng(42)
    pats = []
    for _ in range(2000):
        s = int(rng.integers(0, 30))
        c = rng.multinomial(s, rng.dirichlet([3, .3, .3, .3, .3]))
        pats.append(rng.permutation(c))
    for s in range(0, 30):                      # edge patterns: everything on one state
        for k in range(5):
            p = [0] * 5
            p[k] = s
            pats.append(p)
    pats = np.array(pats, dtype=np.uint16)
    d = dict(patterns=pats)
    for pl in (1, 2):
        sc = Scoring(ploidy=pl)
        ent, sco = sc.calc_posterior_and_scores(cov_patterns=pats.copy())
        d["phi_p%d" % pl] = sc.priors.phi
        d["priors_p%d" % pl] = sc.priors.priors
        d["score0_p%d" % pl] = sc.score0
        d["ent0_p%d" % pl] = sc.ent0
        d["entropy_p%d" % pl] = ent
        d["score_p%d" % pl] = sco
    np.savez_compressed(os.path.join(out, "g_tables.npz"), **d)


def gen_errors(out):
    """What the reference does with malformed / unusual batches (scenarios.error_cases): the exception class
    of Paf.parse_PAF -> CoverageConverter.convert_records -> Contig.increment_coverage, or a digest of
    the coverage it ends with."""
    import json
    from boss.paf import Paf
    from boss.runs.reference import Contig
    from boss.runs.sequences import CoverageConverter
    from scenarios import error_cases
    contigs, cases = error_cases()
    res = {}
    for name, paf_text, seqs in cases:
        conts = {n: Contig(n, synth.codes_to_str(c)) for n, c in contigs}
        stage = "parse_PAF"
        try:
            paf = Paf.parse_PAF(StringIO(paf_text), min_len=200)
            stage = "convert_records"
            inc = CoverageConverter().convert_records(paf_dict=paf, seqs=seqs,
                                                      quals={k: "I" * len(v) for k, v in seqs.items()})
            stage = "increment_coverage"
            for n, c in conts.items():          # core.py:77-86: every contig of contigs_filt
                c.increment_coverage(inc[n])
            res[name] = {"ok": digest(*[conts[n].coverage for n in sorted(conts)]), "n_reads": len(paf)}
        except Exception as e:                  # noqa: BLE001 - the class is the datum
            res[name] = {"error": type(e).__name__, "stage": stage}
    with open(os.path.join(out, "g_errors.json"), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)


def gen_errors_fuzz(out):
    """The same record of the reference's behaviour for the 2,000 seeded random mutations of scenarios.fuzz_error_cases
    (the mutation set of scripts/fuzz/fuzz_paf.cpp): differential fuzzing of the front end against the reference itself."""
    import json
    from boss.paf import Paf
    from boss.runs.reference import Contig
    from boss.runs.sequences import CoverageConverter
    from scenarios import fuzz_error_cases
    contigs, cases = fuzz_error_cases()
    res = {}
    for name, paf_text, seqs in cases:
        conts = {n: Contig(n, synth.codes_to_str(c)) for n, c in contigs}
        try:
            paf = Paf.parse_PAF(StringIO(paf_text), min_len=200)
            inc = CoverageConverter().convert_records(paf_dict=paf, seqs=seqs,
                                                      quals={k: "I" * len(v) for k, v in seqs.items()})
            for n, c in conts.items():          # core.py:77-86: every contig of contigs_filt
                c.increment_coverage(inc[n])
            res[name] = {"ok": digest(*[conts[n].coverage for n in sorted(conts)])[:16]}
        except Exception as e:                  # noqa: BLE001 - the class is the datum
            res[name] = {"error": type(e).__name__}
    with open(os.path.join(out, "g_errors_fuzz.json"), "w") as fh:
        json.dump(res, fh, sort_keys=True, separators=(",", ":"))


def gen_cigar(out):
    from boss.paf import Paf
    from boss.runs.sequences import CoverageConverter
    contigs = synth.make_reference([5000, 4000], seed=3, names=["c1", "c2"])
    b = synth.make_batch(contigs, 60, seed=77, mean_len=900.0, min_len=250, max_len=2500)
    paf = Paf.parse_PAF(StringIO(b["paf"]), min_len=200)
    inc = CoverageConverter().convert_records(paf_dict=paf, seqs=b["seqs"],
                                              quals={k: "I" * len(v) for k, v in b["seqs"].items()})
    d = dict(paf=np.frombuffer(b["paf"].encode(), dtype=np.uint8),
             read_ids=np.array(list(b["seqs"].keys())),
             read_seqs=np.array(list(b["seqs"].values())))
    k = 0
    for tname, lst in inc.items():
        for (start, end, q, addition, bc) in lst:
            assert np.all(addition == 1)
            d["inc%03d_tname" % k] = np.array(tname)
            d["inc%03d_range" % k] = np.array([start, end])
            d["inc%03d_codes" % k] = q.astype(np.uint8)
            k += 1
    d["n_inc"] = np.array(k)
    np.savez_compressed(os.path.join(out, "g_cigar.npz"), **d)


def gen_dists(out):
    from boss.readlengthdist import ReadlengthDist
    r = ReadlengthDist()
    d = dict(default_approx_ccl=r.approx_ccl.copy())
    rng = np.random.default_rng(5)
    for k in range(3):
        lens = np.clip(rng.gamma(2.0, 3000.0, size=500), 100, 2_000_000).astype(np.int64)
        r.update({"x%d" % i: int(v) for i, v in enumerate(lens)})
        d["lens%d" % k] = lens
        d["approx_ccl%d" % k] = r.approx_ccl.copy()
        d["lam%d" % k] = np.array(r.lam)
        d["time_cost%d" % k] = np.array(r.time_cost)
    np.savez_compressed(os.path.join(out, "g_dists.npz"), **d)


def run_scenario(out, tag, ploidy, nb):
    import boss.config
    import boss.runs.core
    from boss.paf import Paf
    contigs = e2e_reference()
    tmp = tempfile.mkdtemp(prefix="golden_")
    os.chdir(tmp)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, contigs)
    open(os.path.join(tmp, "ref.mmi"), "w").close()
    args = boss.config.BossConfig()
    args.general.ref = fa
    args.general.mmi = os.path.join(tmp, "ref.mmi")
    args.general.name = "golden"
    args.optional.ploidy = ploidy
    args.optional.reject_refs = E2E_REJECT
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = boss.runs.core.BossRuns(args=args)
    runs.init()
    d = dict(movesum_via_shim=np.array(1), ploidy=np.array(ploidy), nb=np.array(nb),
             ref_digest=np.array(digest(*[c[1] for c in contigs])))
    for b in range(E2E_BATCHES):
        batch = e2e_batch(contigs, b, nb)
        d["b%d_input_digest" % b] = np.array(batch_digest(batch))
        runs.rl_dist.update(read_lengths=batch["read_lengths"])
        paf = Paf.parse_PAF(StringIO(batch["paf"]), min_len=200)
        for recs in paf.values():
            for r in recs:
                r.barcode = batch["barcodes"][r.qname] if nb > 1 else None
        inc = runs.cc.convert_records(paf_dict=paf, seqs=batch["seqs"],
                                      quals={k: "I" * len(v) for k, v in batch["seqs"].items()})
        runs._effect_increments(increments=inc)
        runs.tracker.update(n=len(batch["seqs"]), paf_dict=paf)
        runs.read_starts.count_read_starts(paf_dict=paf)
        # capture the threshold chosen inside update_wrapper
        captured = {}
        orig = runs.scoring.find_strat_thread

        def spy(benefit, smu, fhat, time_cost, _orig=orig, _cap=captured):
            strat, thr = _orig(benefit=benefit, smu=smu, fhat=fhat, time_cost=time_cost)
            _cap.update(threshold=thr, merged_strat=strat.copy(), benefit_adj=benefit.copy(),
                        fhat_adj=fhat.copy())
            return strat, thr
        runs.scoring.find_strat_thread = spy
        runs.update_wrapper()
        runs.scoring.find_strat_thread = orig
        d["b%d_updated" % b] = np.array(int(bool(captured)))
        d["b%d_approx_ccl" % b] = runs.rl_dist.approx_ccl.copy()
        d["b%d_time_cost" % b] = np.array(runs.rl_dist.time_cost)
        if captured:
            d["b%d_threshold" % b] = np.array(captured["threshold"])
            d["b%d_merged_strat" % b] = np.packbits(captured["merged_strat"].reshape(-1))
            d["b%d_merged_shape" % b] = np.array(captured["merged_strat"].shape)
            d["b%d_benefit_adj" % b] = captured["benefit_adj"]
            d["b%d_fhat_adj" % b] = captured["fhat_adj"]
        for cname, c in runs.contigs.items():
            key = "b%d_%s_" % (b, cname)
            d[key + "strat"] = np.packbits(c.strat.reshape(-1))
            d[key + "strat_shape"] = np.array(c.strat.shape)
            if c.rej:
                continue
            d[key + "cov_digest"] = np.array(digest(c.coverage))
            d[key + "cov_total"] = np.array(int(c.coverage.sum(dtype=np.uint64)))
            d[key + "change_count"] = np.array(int(c.change_mask.sum()))
            d[key + "scores_digest"] = np.array(digest(c.scores))
            d[key + "entropy_digest"] = np.array(digest(c.entropy))
            d[key + "n_zero_scores"] = np.array(int((c.scores == 0).sum()))
            d[key + "n_tiny_scores"] = np.array(int((c.scores == np.finfo(float).tiny).sum()))
            d[key + "bucket_switches"] = c.bucket_switches.copy()
            d[key + "switched_on"] = c.switched_on.copy()
            if captured:
                d[key + "scores_ds"] = c.scores_ds.copy()
                d[key + "smu"] = c.smu.copy()
                d[key + "additional_benefit"] = c.additional_benefit.copy()
        if b == E2E_BATCHES - 1:
            c = runs.contigs["ctgA"]
            d["final_ctgA_scores"] = c.scores.copy()
            d["final_ctgA_entropy"] = c.entropy.copy()
            d["final_ctgA_coverage"] = c.coverage.copy()
            d["final_read_starts"] = runs.read_starts.merge()
    np.savez_compressed(os.path.join(out, "g_e2e_%s.npz" % tag), **d)
    return d



def run_saturated(out, ploidy=1, nb=1):
    """The reference's update_wrapper on the state a long run converges to (scenarios.saturated_coverage:
    every score `tiny` or 0.0, benefits ~1e-300), then one more ordinary batch on top of it.  The first
    batch only provides read-start counts and the read-length distribution."""
    import boss.config
    import boss.runs.core
    from boss.paf import Paf
    contigs = e2e_reference()
    tmp = tempfile.mkdtemp(prefix="golden_")
    os.chdir(tmp)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, contigs)
    open(os.path.join(tmp, "ref.mmi"), "w").close()
    args = boss.config.BossConfig()
    args.general.ref = fa
    args.general.mmi = os.path.join(tmp, "ref.mmi")
    args.general.name = "golden"
    args.optional.ploidy = ploidy
    args.optional.reject_refs = E2E_REJECT
    runs = boss.runs.core.BossRuns(args=args)
    runs.init()
    d = dict(movesum_via_shim=np.array(1), ploidy=np.array(ploidy), nb=np.array(nb),
             ref_digest=np.array(digest(*[c[1] for c in contigs])))

    def ingest(b):
        batch = e2e_batch(contigs, b, nb)
        d["b%d_input_digest" % b] = np.array(batch_digest(batch))
        runs.rl_dist.update(read_lengths=batch["read_lengths"])
        paf = Paf.parse_PAF(StringIO(batch["paf"]), min_len=200)
        inc = runs.cc.convert_records(paf_dict=paf, seqs=batch["seqs"],
                                      quals={k: "I" * len(v) for k, v in batch["seqs"].items()})
        runs._effect_increments(increments=inc)
        runs.tracker.update(n=len(batch["seqs"]), paf_dict=paf)
        runs.read_starts.count_read_starts(paf_dict=paf)

    def update(tag):
        captured = {}
        orig = runs.scoring.find_strat_thread

        def spy(benefit, smu, fhat, time_cost, _orig=orig, _cap=captured):
            strat, thr = _orig(benefit=benefit, smu=smu, fhat=fhat, time_cost=time_cost)
            _cap.update(threshold=thr, merged_strat=strat.copy(), benefit_adj=benefit.copy(), fhat_adj=fhat.copy())
            return strat, thr
        runs.scoring.find_strat_thread = spy
        runs.update_wrapper()
        runs.scoring.find_strat_thread = orig
        assert captured, "strategies not switched on"
        d[tag + "_threshold"] = np.array(captured["threshold"])
        d[tag + "_approx_ccl"] = np.array(runs.rl_dist.approx_ccl)
        d[tag + "_merged_strat"] = np.packbits(captured["merged_strat"].reshape(-1))
        d[tag + "_merged_shape"] = np.array(captured["merged_strat"].shape)
        d[tag + "_benefit_adj"] = captured["benefit_adj"]
        d[tag + "_benefit_max"] = np.array(captured["benefit_adj"].max())
        for cname, c in runs.contigs.items():
            key = "%s_%s_" % (tag, cname)
            d[key + "strat"] = np.packbits(c.strat.reshape(-1))
            d[key + "strat_shape"] = np.array(c.strat.shape)
            if c.rej:
                continue
            d[key + "n_zero_scores"] = np.array(int((c.scores == 0).sum()))
            d[key + "n_tiny_scores"] = np.array(int((c.scores == np.finfo(float).tiny).sum()))
            d[key + "n_other_scores"] = np.array(int(c.scores.size - (c.scores == 0).sum() - (c.scores == np.finfo(float).tiny).sum()))
            d[key + "scores_ds"] = c.scores_ds.copy()
            d[key + "additional_benefit"] = c.additional_benefit.copy()

    ingest(0)
    runs.update_wrapper()
    for c in runs.contigs.values():
        if c.rej:
            continue
        c.coverage[:] = saturated_coverage(c.seq_int, nb=nb)
        c.change_mask[:] = True
    update("sat")
    ingest(1)
    update("sat1")
    np.savez_compressed(os.path.join(out, "g_sat_p%d_nb%d.npz" % (ploidy, nb)), **d)
    return d


SIM_SCENARIOS = [("nb1", 1, False), ("nb2_unmapped", 2, True)]      # tag, nbarcodes, accept_unmapped
SIM_BATCHES = 5


def sim_batch(contigs, b, nb):
    """Sampled batch of the simulation scenario: full-length and mu-truncated mappings."""
    return synth.make_batch(contigs, 420, seed=40 + b, mean_len=3000.0, nbarcodes=nb,
                            start_weights=[0.6, 1.6, 0.5, 0.5], trunc_mu=400)


def run_sim_scenario(out, tag, nb, accept_unmapped):
    """Drives the reference's `BossRunsSim.process_batch_runs_sim` (boss/runs/simulation.py:139-190;
    make_decisions :37-120) with a stub sampler / read cache (sampling from files and the pseudo-time
    read cache are out of scope) and records decisions, counters and masks per batch."""
    import boss.config
    import boss.runs.simulation
    contigs = e2e_reference()
    tmp = tempfile.mkdtemp(prefix="golden_sim_")
    os.chdir(tmp)
    fa = os.path.join(tmp, "ref.fa")
    synth.write_fasta(fa, contigs)
    open(os.path.join(tmp, "ref.mmi"), "w").close()
    args = boss.config.BossConfig()
    args.general.ref = fa
    args.general.mmi = os.path.join(tmp, "ref.mmi")
    args.general.name = "goldensim"
    args.optional.reject_refs = E2E_REJECT
    args.optional.bucket_threshold = 2
    args.simulation.accept_unmapped = accept_unmapped
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = boss.runs.simulation.BossRunsSim(args=args)
    runs.init()                               # init_sim minus the file-backed Sampler / ReadCache

    class FqStream:
        read_ids, total_bases, read_sequences = set(), 0, {}

    class StubSampler:
        fq_stream = FqStream()

        def sample(self):
            return self.batch

    class StubCache:
        mu = 400

        def update_times_runs(self, **kw):
            pass

        def fill_cache(self, **kw):
            pass
    runs.sampler, runs.read_cache, runs.mu, runs.accept_unmapped = StubSampler(), StubCache(), 400, accept_unmapped
    d = dict(movesum_via_shim=np.array(1), nb=np.array(nb), accept_unmapped=np.array(int(accept_unmapped)))
    for b in range(SIM_BATCHES):
        batch = sim_batch(contigs, b, nb)
        d["b%d_input_digest" % b] = np.array(digest(batch["paf"].encode(), batch["paf_trunc"].encode(),
                                                    "".join(batch["seqs"].values()).encode()))
        names = {rid: (bc + 1 if nb > 1 else 0) for rid, bc in batch["barcodes"].items()}   # sampler.py:214-221
        if nb > 1:
            for k, rid in enumerate(names):
                if k % 37 == 11:
                    names[rid] = 99                                        # 'unclassified' -> index 0 (simulation.py:148)
        runs.sampler.batch = (batch["seqs"], {k: "I" * len(v) for k, v in batch["seqs"].items()}, names,
                              batch["paf"], batch["paf_trunc"])
        runs.sampler.fq_stream.read_ids = set(batch["seqs"].keys())
        captured = {}
        orig = runs.make_decisions

        def spy(**kw):
            res = orig(**kw)
            paf_dict, reads_decision = res[0], res[1]
            captured.update(counts=np.array(res[2:], dtype=np.int64),
                            full_kept=np.array([int(len(reads_decision[r]) == len(kw["seqs"][r])) for r in kw["seqs"]],
                                               dtype=np.uint8),
                            chosen_qlen=np.array([paf_dict[r][0].qlen if r in paf_dict else -1 for r in kw["seqs"]],
                                                 dtype=np.int64))
            return res
        runs.make_decisions = spy
        runs.process_batch_runs_sim()
        runs.make_decisions = orig
        d["b%d_counts" % b] = captured["counts"]             # n_mapped, n_unmapped, n_accepted, n_rejected
        d["b%d_full_kept" % b] = captured["full_kept"]
        d["b%d_chosen_qlen" % b] = captured["chosen_qlen"]
        d["b%d_approx_ccl" % b] = runs.rl_dist.approx_ccl.copy()
        d["b%d_read_starts" % b] = runs.read_starts.merge()
        d["b%d_total_reads" % b] = np.array(runs.tracker.total_reads)
        d["b%d_read_counts" % b] = np.array([runs.tracker.read_counts[n] for n in runs.contigs], dtype=np.int64)
        for cname, c in runs.contigs.items():
            key = "b%d_%s_" % (b, cname)
            d[key + "strat"] = np.packbits(c.strat.reshape(-1))
            d[key + "strat_shape"] = np.array(c.strat.shape)
            if not c.rej:
                d[key + "cov_digest"] = np.array(digest(c.coverage))
                d[key + "cov_total"] = np.array(int(c.coverage.sum(dtype=np.uint64)))
    np.savez_compressed(os.path.join(out, "g_sim_%s.npz" % tag), **d)
    return d


def main():
    out = HERE
    if os.environ.get("GOLDEN_ONLY") == "errors_fuzz":  # (the other fixtures are not touched)
        gen_errors_fuzz(out)
        return
    if os.environ.get("GOLDEN_ONLY") == "errors":       # (the other fixtures are not touched)
        gen_errors(out)
        return
    if os.environ.get("GOLDEN_ONLY") == "sat":          # (the other fixtures are not touched)
        d = run_saturated(out)
        print("sat thresholds", float(d["sat_threshold"]), float(d["sat1_threshold"]), "max benefit", float(d["sat_benefit_max"]))
        return
    gen_tables(out)
    gen_cigar(out)
    gen_errors(out)
    gen_errors_fuzz(out)
    gen_dists(out)
    for tag, pl, nb in SCENARIOS:
        d = run_scenario(out, tag, pl, nb)
        print(tag, "updated:", [int(d["b%d_updated" % b]) for b in range(E2E_BATCHES)],
              "thr:", [float(d.get("b%d_threshold" % b, np.nan)) for b in range(E2E_BATCHES)])
    d = run_saturated(out)
    print("sat thresholds", float(d["sat_threshold"]), float(d["sat1_threshold"]), "max benefit", float(d["sat_benefit_max"]))
    for tag, nb, au in SIM_SCENARIOS:
        d = run_sim_scenario(out, tag, nb, au)
        print("sim", tag, [d["b%d_counts" % b].tolist() for b in range(SIM_BATCHES)])
    for f in sorted(os.listdir(out)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
