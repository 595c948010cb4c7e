import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Every test session works against a freshly built in-tree library (hipcc cross-compiles on CPU)."""
    import __graft_entry__ as entry

    entry.build()


@pytest.fixture(autouse=True)
def _poisoned_allocator(request):
    """GPU tests start with NaN-filled blocks in torch's caching allocator, so a kernel that reads a scratch buffer it
    (or the host) never initialised shows up as a NaN instead of passing on whatever the previous test left behind
    (fresh hipMalloc pages read as zero, recycled blocks do not).  NVO_TEST_POISON=0 switches it off."""
    if "gpu" in request.keywords and os.environ.get("NVO_TEST_POISON", "1") != "0":
        import torch

        if torch.cuda.is_available():
            blocks = [torch.full((n,), float("nan"), device="cuda:0")
                      for n in (1 << 8, 1 << 12, 1 << 16, 1 << 19, 1 << 22, 1 << 24, 1 << 26) for _ in range(3)]
            del blocks
    yield


def pytest_sessionfinish(session, exitstatus):
    """Parity margins of the session: for every float comparison against the oracle (test_tcnn_gpu._assert_close), the
    largest error in units of the stated bound, the relative L1 error and the share of elements outside the bound --
    written next to the run's other outputs (NVO_PARITY_MARGINS, default gpurun_out/parity_margins.json)."""
    mod = sys.modules.get("test_tcnn_gpu")
    rows = getattr(mod, "MARGINS", None) if mod is not None else None
    if not rows:
        return
    import json

    path = os.environ.get("NVO_PARITY_MARGINS", os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(rows, fh, indent=0)
    except OSError:
        pass
