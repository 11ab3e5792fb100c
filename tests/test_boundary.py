"""The Python drop-in boundary: the reference's loop `Boss.process_batch(
BossRuns.process_batch_runs)` (boss/core.py:137-157, boss/runs/core.py:202-224) driven with an
object that has exactly the reference `Mapper`'s surface (boss/mapper.py:27-108).  CPU tier: the
engine is the oracle-backed test double, so what is tested here is the host logic of the boundary
— mapper call, min_len = int(mu / 2), read-length update placement, bookkeeping, masks file."""
import os

import numpy as np
import pytest

from scenarios import E2E_REJECT, e2e_batch, e2e_contig_strings, e2e_reference


class StubMapper:
    """Same attributes and method signatures as boss.mapper.Mapper; the 'alignments' are the PAF
    text of a prepared synthetic batch."""

    def __init__(self, mu=400, workers=4):
        self.mu, self.workers = mu, workers
        self.paf = ""
        self.calls = []

    def map_sequences(self, sequences, trunc=False):
        from oracle.pafcigar import parse_paf
        self.calls.append("map_sequences")
        return parse_paf(self._mappy_batch(sequences=sequences), min_len=int(self.mu / 2))

    def _mappy_batch(self, sequences, out=None, log=True):
        self.calls.append("_mappy_batch")
        assert isinstance(sequences, dict)
        return self.paf


class DictOnlyMapper:
    """A mapper that only offers map_sequences (the one call the reference's loop makes)."""

    def __init__(self, mu=400):
        self.mu = mu
        self.paf = ""

    def map_sequences(self, sequences, trunc=False):
        from oracle.pafcigar import parse_paf
        return parse_paf(self.paf, min_len=int(self.mu / 2))


def _runs(in_tmp, mapper, nb=1, staged=True):
    from fake_engine import FakeEngine
    from boss_runs_amd.config import BossConfig
    from boss_runs_amd.runs import BossRuns
    args = BossConfig()
    args.general.name = "boundary"
    args.optional.reject_refs = E2E_REJECT
    if nb > 1:
        args.general.barcodes = ["barcode%02d" % (i + 1) for i in range(nb)]
    runs = BossRuns(args)
    runs.init(contigs=e2e_contig_strings(), engine=FakeEngine(nbarcodes=nb, ploidy=1), mapper=mapper)
    runs.update_wrapper = runs.update_wrapper_staged
    runs._fused = False
    return runs


@pytest.mark.parametrize("mapper_cls,mu", [(StubMapper, 400), (StubMapper, 700), (DictOnlyMapper, 400)])
def test_reference_loop_with_reference_shaped_mapper(in_tmp, mapper_cls, mu):
    from oracle.pipeline import OracleRuns
    contigs = e2e_reference()
    mapper = mapper_cls(mu=mu)
    runs = _runs(in_tmp, mapper)
    assert runs.mapper is mapper
    o = OracleRuns(e2e_contig_strings(contigs), reject_refs={E2E_REJECT})
    state = {}
    runs.data_source = lambda: (state["reads"], {k: "I" * len(v) for k, v in state["reads"].items()})
    for b in range(4):
        batch = e2e_batch(contigs, b, 1)
        mapper.paf = batch["paf"]
        state["reads"] = batch["seqs"]
        wait = runs.process_batch(runs.process_batch_runs)              # the reference's main loop body (BOSS.py:33-38)
        assert isinstance(wait, int) and runs.batch == b + 1
        o.process_batch(batch["paf"], batch["seqs"], min_len=int(mu / 2))   # rl_dist.update(len(seq)) inside
        assert np.array_equal(runs.rl_dist.approx_ccl, o.rl_dist.approx_ccl)
        assert runs.threshold == o.threshold
        for n, oc in o.contigs.items():
            assert np.array_equal(runs.contigs[n].strat, oc.strat), (b, n)
            if not oc.rej:
                assert np.array_equal(runs.contigs[n].coverage, oc.coverage), (b, n)
        assert np.array_equal(runs.read_starts.merge(), o.read_starts.merge())
    assert o.threshold is not None
    if mapper_cls is StubMapper:
        assert set(mapper.calls) == {"_mappy_batch"}      # the raw text is taken, not the parsed dict
    z = np.load(os.path.join(runs.out_dir, "masks", "boss.npz"))
    assert set(z.files) == set(o.contigs)
    # nothing new: the loop defers, as boss/core.py:151-152
    state["reads"] = {}
    assert runs.process_batch(runs.process_batch_runs) == runs.args.general.wait and runs.batch == 4


def test_update_strategy_alias_and_missing_mapper(in_tmp):
    from boss_runs_amd.runs import BossRuns
    assert BossRuns.update_strategy is BossRuns.update_wrapper
    runs = _runs(in_tmp, None)
    with pytest.raises(RuntimeError, match="no mapper"):
        runs.process_batch_runs(new_reads={"r": "ACGT"}, new_quals={"r": "IIII"})
    with pytest.raises(NotImplementedError):
        runs.process_batch(runs.process_batch_runs)


def test_config_sections_match_reference_names():
    """boss/config.py:24-69: the sections and fields reference-shaped callers read."""
    from boss_runs_amd.config import BossConfig
    a = BossConfig()
    assert (a.general.name, a.general.wait, a.general.barcodes) == ("boss", 60, None)
    assert (a.live.device, a.live.host, a.live.port, a.live.data_wait) == (None, "localhost", 9502, 100)
    assert (a.optional.ploidy, a.optional.bucket_threshold, a.optional.reject_refs) == (1, 5, None)
    s = a.simulation
    assert (s.fq, s.batchsize, s.maxb, s.binit, s.dumptime, s.paf_full, s.paf_trunc, s.accept_unmapped) == \
        (None, 4000, 400, 5, 200000000, None, None, False)
