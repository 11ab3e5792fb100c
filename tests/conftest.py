import os
import sys

import pytest

# torch brings its own HIP runtime: it has to be loaded BEFORE libbossx.so pulls in the system one,
# or torch.cuda finds no device in tests that use both (the nccl protocol tests).  Collecting the
# whole directory does this by accident (test_parallel_gloo imports torch); make it explicit so
# that any subset of the tests behaves the same.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle's C restatement (move_sum) is compiled on demand; it is test infrastructure."""
    import subprocess
    so = os.path.join(REPO, "oracle", "_build", "liboracle_movesum.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    yield


@pytest.fixture
def in_tmp(tmp_path, monkeypatch):
    """BossRuns writes ./out_<name>/ relative to the cwd (boss/core.py:44-55)."""
    monkeypatch.chdir(tmp_path)
    return tmp_path
