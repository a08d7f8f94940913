import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# every rel_l2() a test evaluates is kept, keyed by the running test, and written
# out at the end of the session (gpurun_out/parity_errors.json on the GPU box):
# the observed errors behind the gates, not only pass / fail
_OBSERVED = {}


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    if not _OBSERVED:
        return
    out = os.environ.get("NSOL_PARITY_LOG")
    if out is None:
        d = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
        except OSError:
            return
        out = os.path.join(d, "parity_errors.json")
    try:
        with open(out, "w") as f:
            json.dump(_OBSERVED, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]
    return load


@pytest.fixture(autouse=True)
def _restore_library_knobs():
    """The nsol_hip_set_param* knobs are process-global: whatever a test sets
    (also one that dies between set and reset) is put back to the built-in
    default before the next test runs."""
    yield
    lib = sys.modules.get("nsol_amd._lib")
    if lib is not None and lib._touched:
        lib.reset_params()


@pytest.fixture(autouse=True)
def _one_launch_per_iteration_unless_asked(request):
    """Small volumes run through the persistent kernel by default
    (ops.PD_PERSIST).  Most GPU tests are about the other kernels -- the
    one-iteration kernel, the two- and three-iterations-per-pass kernels on
    small shapes -- so the persistent path is switched off unless the test asks
    for it by name ("persist" / "configs" / "cli")."""
    ops = sys.modules.get("nsol_amd.ops")
    name = request.node.name.lower()
    want = any(t in name for t in ("persist", "configs", "cli"))
    if ops is None:
        if request.node.get_closest_marker("gpu") is None:
            yield
            return
        import nsol_amd.ops as ops
    old = ops.PD_PERSIST
    ops.PD_PERSIST = want
    yield
    ops.PD_PERSIST = old


def rel_l2(a, b, label=None):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    nb = np.linalg.norm(b)
    err = float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    _OBSERVED.setdefault(test, []).append(
        err if label is None else [label, err])
    return err
