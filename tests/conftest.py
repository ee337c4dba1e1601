"""pytest configuration: paths, the `gpu` marker, shared fixtures.

-m "not gpu": oracle pins, host logic, C-ABI load/export checks (CPU only, minutes).
-m gpu      : parity of the HIP path (through the C ABI) against the oracle on a real MI355X.
Nothing here reads /root/reference at run time (goldens are committed under tests/golden/).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rad-nerf_amd"), os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _reproducible_training_jitter(monkeypatch):
    """Tests that compare two runs of the training step draw the marcher's jitter with torch.rand (seeded by torch.manual_seed), as
    the reference does; the product default is the marcher's own hash (one launch less), which draws anew on every launch.
    test_gpu_train.py::test_step_marcher_hash_jitter_* and the launch-count test cover that form."""
    monkeypatch.setenv("RN_TRAIN_NOISE", "torch")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def po():
    import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hiplib():
    """Builds (if stale) and imports the ctypes binding of libradnerf_hip.so."""
    sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("rn_build", os.path.join(ROOT, "rad-nerf_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.LIB) or os.environ.get("RN_REBUILD") == "1":
        mod.build_all()
    import radnerf_hip
    return radnerf_hip


@pytest.fixture
def rng():
    return np.random.default_rng(1234)
