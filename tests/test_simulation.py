"""Simulation mode (BASELINE configs[0]) against golden vectors recorded from the
reference's own `BossRunsSim.process_batch_runs_sim` / `make_decisions`
(boss/runs/simulation.py:37-190; generator: tests/golden/make_golden.py run_sim_scenario).

CPU tier: the host logic of `boss_runs_amd.simulation.BossRunsSim` — native best-mapper pass,
vectorised decision lookup, native PAF line selection, accepted-only bookkeeping — over the
oracle-backed engine double.  GPU tier: the same through the HIP engine (fused update)."""
import os

import numpy as np
import pytest

from scenarios import GOLDEN, E2E_REJECT, digest, e2e_contig_strings, e2e_reference

SIM_SCENARIOS = [("nb1", 1, False), ("nb2_unmapped", 2, True)]
SIM_BATCHES = 5


def sim_batch(contigs, b, nb):
    from boss_runs_amd import synth
    return synth.make_batch(contigs, 420, seed=40 + b, mean_len=3000.0, nbarcodes=nb,
                            start_weights=[0.6, 1.6, 0.5, 0.5], trunc_mu=400)


class _FqStream:
    def __init__(self):
        self.read_ids, self.total_bases, self.read_sequences = set(), 0, {}


class _Sampler:
    """boss/sampler.py:20-55 surface: sample() -> (seqs, quals, barcode names, paf_full, paf_trunc)."""

    def __init__(self):
        self.fq_stream = _FqStream()
        self.batch = None

    def sample(self):
        return self.batch


def _drive(runs, tag, nb, accept_unmapped):
    g = np.load(os.path.join(GOLDEN, "g_sim_%s.npz" % tag))
    contigs = e2e_reference()
    sampler = runs.sampler
    n_acc = n_rej = 0
    for b in range(SIM_BATCHES):
        batch = sim_batch(contigs, b, nb)
        assert str(g["b%d_input_digest" % b]) == digest(batch["paf"].encode(), batch["paf_trunc"].encode(),
                                                        "".join(batch["seqs"].values()).encode())
        names = {rid: (bc + 1 if nb > 1 else 0) for rid, bc in batch["barcodes"].items()}
        if nb > 1:
            for k, rid in enumerate(names):
                if k % 37 == 11:
                    names[rid] = 99
        sampler.batch = (batch["seqs"], None, names, batch["paf"], batch["paf_trunc"])
        sampler.fq_stream.read_ids = set(batch["seqs"].keys())
        # make_decisions on its own (simulation.py:37-120): the returned pieces
        read_barcodes = {rid: runs.barcodes_index.get(bc, 0) for rid, bc in names.items()}
        paf_text, reads_decision, *counts = runs.make_decisions(seqs=batch["seqs"], paf_full=batch["paf"],
                                                                paf_trunc=batch["paf_trunc"], barcodes=read_barcodes)
        assert counts == g["b%d_counts" % b].tolist(), b
        full_kept = np.array([int(len(reads_decision[r]) == len(s)) for r, s in batch["seqs"].items()], dtype=np.uint8)
        assert np.array_equal(full_kept, g["b%d_full_kept" % b]), b
        assert all(reads_decision[r] in (s, s[:400]) for r, s in batch["seqs"].items())
        summ = runs.engine.paf_summary(paf_text, list(batch["seqs"]), min_len=1)
        chosen = np.full(len(batch["seqs"]), -1, dtype=np.int64)
        chosen[summ["read_idx"]] = summ["qlen"]
        assert np.array_equal(chosen, g["b%d_chosen_qlen" % b]), b      # full-length vs truncated record per read
        # the whole batch (simulation.py:139-190)
        runs.process_batch_sim(runs.process_batch_runs_sim)
        assert runs.batch == b + 1
        assert list(runs.last_counts.values()) == g["b%d_counts" % b].tolist()
        n_acc += counts[2]
        n_rej += counts[3]
        assert np.array_equal(runs.rl_dist.approx_ccl, g["b%d_approx_ccl" % b]), b
        assert np.array_equal(runs.read_starts.merge(), g["b%d_read_starts" % b]), b
        assert runs.total_reads == int(g["b%d_total_reads" % b]) == n_acc
        assert [runs.read_counts[n] for n in runs.contigs] == g["b%d_read_counts" % b].tolist()
        for cname, c in runs.contigs.items():
            key = "b%d_%s_" % (b, cname)
            shape = tuple(g[key + "strat_shape"])
            want = np.unpackbits(g[key + "strat"])[:int(np.prod(shape))].reshape(shape).astype(bool)
            assert np.array_equal(c.strat, want), (b, cname)             # masks = the reference's, bit for bit
            if not c.rej:
                cov = c.coverage
                assert int(cov.sum(dtype=np.uint64)) == int(g[key + "cov_total"]), (b, cname)
                assert digest(cov) == str(g[key + "cov_digest"]), (b, cname)
    assert n_acc > 0 and n_rej > 0 and runs.threshold is not None


def _args(nb, accept_unmapped, name):
    from boss_runs_amd.config import BossConfig
    args = BossConfig()
    args.general.name = name
    args.optional.reject_refs = E2E_REJECT
    args.optional.bucket_threshold = 2
    args.simulation.accept_unmapped = accept_unmapped
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    return args


@pytest.mark.parametrize("tag,nb,accept_unmapped", SIM_SCENARIOS)
def test_simulation_host_logic_vs_reference_golden(tag, nb, accept_unmapped, in_tmp):
    from fake_engine import FakeEngine
    from boss_runs_amd.simulation import BossRunsSim
    runs = BossRunsSim(_args(nb, accept_unmapped, "simcpu"))
    runs.init_sim(contigs=e2e_contig_strings(), engine=FakeEngine(nbarcodes=nb, ploidy=1), sampler=_Sampler())
    runs.update_wrapper = runs.update_wrapper_staged
    runs._fused = False
    _drive(runs, tag, nb, accept_unmapped)


def test_accepted_read_without_full_mapping_is_an_index_error(in_tmp):
    """simulation.py:90-91: choose_best_mapper on the empty list of a read that has a truncated
    but no full-length mapping."""
    from fake_engine import FakeEngine
    from boss_runs_amd.simulation import BossRunsSim
    runs = BossRunsSim(_args(1, False, "simerr"))
    runs.init_sim(contigs=e2e_contig_strings(), engine=FakeEngine(nbarcodes=1, ploidy=1))
    runs.update_wrapper = runs.update_wrapper_staged
    runs._fused = False
    batch = sim_batch(e2e_reference(), 0, 1)
    rid = [l for l in batch["paf_trunc"].split("\n") if "tp:A:P" in l and "\tctgB\t" in l][3].split("\t")[0]
    paf_full = "\n".join(l for l in batch["paf"].split("\n") if not l.startswith(rid + "\t"))
    with pytest.raises(IndexError, match="no full-length mapping"):
        runs.process_batch_runs_sim((batch["seqs"], None, {r: 0 for r in batch["seqs"]}, paf_full, batch["paf_trunc"]))
    with pytest.raises(RuntimeError, match="no sampler"):
        runs.process_batch_runs_sim()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,nb,accept_unmapped", SIM_SCENARIOS)
def test_simulation_on_gpu_vs_reference_golden(tag, nb, accept_unmapped, in_tmp):
    from boss_runs_amd.simulation import BossRunsSim
    runs = BossRunsSim(_args(nb, accept_unmapped, "simgpu"))
    runs.init_sim(contigs=e2e_contig_strings(), sampler=_Sampler())
    _drive(runs, tag, nb, accept_unmapped)
