import json
import os
import sys

import pytest

# there are no checkpoints in this tree: the tests run the drop-in wrappers on seeded synthetic weights (opt-in)
os.environ.setdefault("CYCLEDIFF_SYNTHETIC_WEIGHTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:  # the oracle runs on the host: stay inside the container's CPU quota
        import torch
        torch.set_num_threads(_host_cores())
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="module", autouse=True)
def _reference_stubs_do_not_leak():
    """oracle.ref_import.setup() puts stub `torchvision` / `omegaconf` packages and the reference's own modules into
    sys.modules; they are removed after every test module so that test order does not matter (a stub torchvision
    breaks any later `import transformers`)."""
    yield
    ri = sys.modules.get("oracle.ref_import")
    if ri is not None and getattr(ri, "_done", False):
        ri.teardown()


@pytest.fixture(scope="session")
def engine():
    import cycle_diffusion_amd as cda
    eng = cda.Engine("cuda:0")
    yield eng
    eng.close()


class _Report:
    """Collects numeric parity figures; dumped to gpurun_out/parity_report.json at session end."""

    def __init__(self):
        self.rows = []

    def add(self, name, **kw):
        self.rows.append(dict(name=name, **kw))


_REPORT = _Report()


@pytest.fixture(scope="session")
def report():
    return _REPORT


def pytest_sessionfinish(session, exitstatus):
    if not _REPORT.rows:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        # CYCLEDIFF_PARITY_REPORT: a child session (the bf16-library test runs pytest in a subprocess) keeps its own file
        with open(os.path.join(out, os.environ.get("CYCLEDIFF_PARITY_REPORT") or "parity_report.json"), "w") as f:
            json.dump(_REPORT.rows, f, indent=1)
    except OSError:
        pass
