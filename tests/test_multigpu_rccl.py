"""Real multi-process RCCL runs: one process per GPU, the in-stream collectives of
boss_runs_amd/parallel.py between the engine's kernels.  Needs >= 2 GPUs on the node; on a one-GPU
box these tests skip (the same scenario runs on CPUs over gloo in test_parallel_gloo.py, and the
protocol with forced collectives on one rank in test_parity_gpu.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count()          # (counting devices does not initialise the GPU)


@pytest.mark.parametrize("nb,ploidy", [(1, 2), (2, 1)])
def test_two_processes_two_gpus_rccl_vs_oracle(nb, ploidy, tmp_path):
    """core.py:83-121 shards by contig; sequences.py:584-649 needs ONE global threshold: two ranks on
    two devices, every collective over RCCL on the engine's stream — masks, threshold and chosen
    exponent of every rank equal the single-process oracle's, and collectives really ran."""
    if _n_gpus() < 2:
        pytest.skip("needs 2 GPUs")
    import torch.multiprocessing as mp
    import dist_scenario
    expected = dist_scenario.oracle_expected(nb, ploidy)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(dist_scenario.worker, args=(2, port, str(tmp_path), nb, ploidy, ret, "nccl"), nprocs=2, join=True)
    dist_scenario.check(ret, expected, 2, full_stats=False)
    assert all(ret[r][-1]["collectives"] > 0 for r in range(2))


def test_bench_gpus_2_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` (no torchrun around it) must start two ranks itself and print ONE
    JSON line for the whole job."""
    if _n_gpus() < 2:
        pytest.skip("needs 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--reads", "1500"], capture_output=True, text=True, timeout=1500, env=env, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["n_ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["config"]["collectives_per_update"] > 0
    g = d["grch38_strong"]
    assert g["n_ranks_seen"] == 2 and len(g["shards"]) == 2 and sum(s["sites"] for s in g["shards"]) == 3_089_069_832
