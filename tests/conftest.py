import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a device: on a host without one they are skipped, not failed (a plain `pytest tests/` there
    is then the CPU suite)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


_REPORT = []


@pytest.fixture
def report():
    """Collects measured parity errors; written to gpurun_out/parity_report.json at session end."""
    def add(name, **vals):
        _REPORT.append(dict(name=name, **vals))
    return add


def pytest_sessionfinish(session, exitstatus):
    if _REPORT:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as f:
            json.dump(_REPORT, f, indent=1)
