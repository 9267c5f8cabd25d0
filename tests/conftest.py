import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # the package warns (RuntimeWarning) when a shape leaves its own kernels for a library / ATen path: in the test-suite that is
    # an error, so a golden comparison can never silently validate ATen instead of the HIP kernels
    config.addinivalue_line("filterwarnings", "error::RuntimeWarning:spike2former_amd")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session", autouse=True)
def _build_oracle_c():
    """The C restatement is test infrastructure; build it on demand (gcc only, < 1 s)."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "liblif_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture
def spike_mode():
    """spike_mode(True / False): spike maps between kernels as bf16 (the product default) or as fp32; restored afterwards."""
    from spike2former_amd import ops
    before = ops.SPIKES_BF16

    def set_mode(bf16):
        ops.SPIKES_BF16 = bool(bf16)
    yield set_mode
    ops.SPIKES_BF16 = before
