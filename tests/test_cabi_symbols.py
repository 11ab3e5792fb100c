"""The C-ABI library loads and exports every symbol include/bossx.h declares (no compute
calls: there is no GPU in the CPU test tier), and the product refuses to run without a GPU."""
import os
import re

import pytest

from scenarios import REPO

from boss_runs_amd import _lib


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "bossx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bossx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libbossx.so does not export %s" % name
    assert set(declared) == set(_lib.PROTOTYPES), "ctypes prototypes out of sync with bossx.h"
    assert lib.bossx_version().startswith(b"bossx")


def test_no_cpu_fallback():
    """Without a HIP device the engine must fail loudly rather than compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from boss_runs_amd.engine import Engine
    with pytest.raises(_lib.BossxError):
        Engine()


def test_product_does_not_import_oracle():
    pkg = os.path.join(REPO, "boss-runs_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".inc")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle/" not in src or f.endswith(".md"), f


def test_rccl_loopback_double_exports_what_the_driver_binds():
    """tests/rccl_loopback/librccl_loopback.so (BOSSX_RCCL_LIB; test infrastructure) must carry every entry point
    RcclApi::load looks up in librccl (boss-runs_amd/csrc/bossx.hip), and nothing under the product may name it."""
    import ctypes
    import subprocess
    d = os.path.join(REPO, "tests", "rccl_loopback")
    subprocess.run(["make", "-C", d], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(d, "librccl_loopback.so"))
    src = open(os.path.join(REPO, "boss-runs_amd", "csrc", "bossx.hip")).read()
    bound = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert bound >= {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce", "ncclAllGather", "ncclGetErrorString"}
    for name in bound:
        assert hasattr(lib, name), name
    for root, _, files in os.walk(os.path.join(REPO, "boss-runs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".inc")):
                assert "rccl_loopback" not in open(os.path.join(root, f)).read() or f == "bossx.hip", f
