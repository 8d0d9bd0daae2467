"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol
include/ppca_hip.h declares; without a GPU it fails loudly instead of falling back."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ppca_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ppca_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hiplib):
    from ppca_rs_amd import _lib

    names = _declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(hiplib, n), f"{n} declared in include/ppca_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert hiplib.ppca_abi_version() == 6


def test_path_kind_and_stats_len(hiplib):
    assert hiplib.ppca_path_kind(256, 10) == 1
    assert hiplib.ppca_path_kind(32, 4) == 1
    assert hiplib.ppca_path_kind(0, 4) < 0
    assert hiplib.ppca_path_kind(1024, 64) == 0 and hiplib.ppca_path_kind(300, 4) == 0
    assert hiplib.ppca_path_kind(32, 65) == 0 and hiplib.ppca_path_kind(32, 128) == 0  # (round 5: the slow correct path up to k = 128)
    assert hiplib.ppca_path_kind(32, 129) < 0
    d, k = 256, 10
    assert hiplib.ppca_stats_len(d, k) == 2 * d * k + d * 55 + 2 * d + 8


def test_no_silent_cpu_fallback(hiplib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ppca_rs_amd as p

    with pytest.raises(p.PPCAError, match="no CPU fallback"):
        p.Dataset(np.zeros((4, 3)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "ppca_rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "ppca_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_compiled_host_example_builds_and_links():
    """examples/em_train.cpp: plain g++ against include/ppca_hip.h + libppca_hip.so (no Python, no torch in the
    boundary).  Built here; run on the GPU by tests/test_gpu_parity.py::test_compiled_host_example_runs."""
    import subprocess

    from ppca_rs_amd import build

    exe = build.build_example()
    assert os.path.exists(exe)
    needed = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libppca_hip.so" in needed and "libtorch" not in needed and "python" not in needed.lower()


def test_drop_in_package_exports_the_reference_names():
    """`import ppca_rs` must offer the reference's classes (python/ppca_rs/__init__.py, .pyi)."""
    import ppca_rs

    for name in ["Dataset", "DatasetChunks", "Prior", "PPCAModel", "InferredMasked", "PosteriorSampler", "PPCAMix",
                 "InferredMaskedMix", "PosteriorSamplerMix", "PPCATrainer", "PPCAMixTrainer", "TrainMetrics",
                 "DataFrameAdapter", "DataFrameAdapterDescription", "__version__"]:
        assert hasattr(ppca_rs, name), name
    import ppca_rs_amd

    assert ppca_rs.PPCAModel is ppca_rs_amd.PPCAModel
