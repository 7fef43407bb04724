import os
import sys

import numpy as np
import pytest

try:
    # Load order matters in a process that uses both PyTorch-ROCm and the product library: torch bundles its own HIP runtime, and if
    # libmultivolumes_amd.so (linked against /opt/rocm) comes first, torch's later CUDA initialisation finds "No HIP GPUs".  With torch
    # first, the product library binds to the runtime that is already loaded.
    import torch  # noqa: F401
except Exception:  # CPU-only tests do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


# (-m gpu on a box without a GPU fails loudly by itself: mv_create returns MV_ERR_NO_DEVICE, the product has no CPU fallback)


@pytest.fixture(scope="session")
def oracle():
    from oracle import mv_oracle
    mv_oracle.load()
    return mv_oracle


@pytest.fixture(scope="session")
def mvlib():
    import multivolumes_amd
    return multivolumes_amd.load()


def half_to_float(a):
    return np.asarray(a, np.uint16).view(np.float16).astype(np.float32)
