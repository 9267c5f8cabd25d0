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
    config.addinivalue_line("markers", "allow_fallbacks(*sites): the named ops.fallback sites may be taken by this test "
                                       "(anything else that leaves the package's kernels fails it)")


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


_FALLBACK_LOG = {}


@pytest.fixture(autouse=True)
def _fallback_census(request):
    """Every GPU test proves it ran this package's HIP kernels: an op that leaves them for an ATen / library routine goes through
    ops.fallback, and a test that takes a site it has not DECLARED (`@pytest.mark.allow_fallbacks("site", ...)`) fails -- twice over:
    the package's RuntimeWarning is an error here (pytest_configure), and the per-site counters are compared below.  The committed
    allow-list is the set of markers in tests/: two tests, both about the inference post-processing's arbitrary-size resize.  What was
    taken is written to gpurun_out/fallbacks_by_test.json for audit."""
    if "gpu" not in request.keywords:
        yield
        return
    from spike2former_amd import ops
    mark = request.node.get_closest_marker("allow_fallbacks")
    allowed = set(mark.args) if mark else set()
    before, strict = dict(ops.FALLBACKS), ops.STRICT
    ops.STRICT = not allowed          # S2F_STRICT for EVERY GPU test without a declaration: ops.fallback raises at the site
    try:
        with ops.allowed_fallbacks(*allowed):
            yield
    finally:
        ops.STRICT = strict
    delta = {k: v - before.get(k, 0) for k, v in ops.FALLBACKS.items() if v != before.get(k, 0)}
    if delta:
        _FALLBACK_LOG[request.node.nodeid] = delta
    assert set(delta) <= allowed, f"undeclared fall-backs to ATen / a library: {delta} (declared: {sorted(allowed)})"


def pytest_sessionfinish(session, exitstatus):
    if _FALLBACK_LOG or os.environ.get("S2F_FALLBACK_CENSUS"):
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "fallbacks_by_test.json"), "w") as f:
            json.dump(_FALLBACK_LOG, f, indent=1, sort_keys=True)
