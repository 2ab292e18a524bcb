import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"),
          os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are deselected by -m "not gpu"; when selected on a machine
    # without a GPU they must fail loudly, not skip silently.
    pass


@pytest.fixture(scope="session")
def golden():
    from helpers import load_golden
    return load_golden()


@pytest.fixture(autouse=True)
def _knobs_reread_after_each_test():
    """The library caches the YOLO_AMD_* environment knobs.  This fixture is set up
    before monkeypatch and therefore torn down after it has restored the
    environment: the library then reads it again, so a knob set in one test never
    leaks into the next."""
    yield
    mod = sys.modules.get("yolo_amd._lib")
    if mod is not None and getattr(mod, "lib", None) is not None:
        mod.lib.yolo_amd_reload_knobs()
