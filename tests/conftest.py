import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_collection_modifyitems(session, config, items):
    """tests/test_emu.py: the tests that only WAIT for an emulated run of the device tests (fixture device_runs: the runs are started when the
    module's first test begins and go on side by side) come last in their module, so that the tests which drive the matcher hook themselves fill
    the time the runs take."""
    emu = [i for i, it in enumerate(items) if it.nodeid.startswith("tests/test_emu.py") or os.path.basename(str(it.fspath)) == "test_emu.py"]
    if emu:
        block = [items[i] for i in emu]
        block.sort(key=lambda it: "device_runs" in getattr(it, "fixturenames", ()))      # stable: the waiters go to the end
        for i, it in zip(emu, block):
            items[i] = it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the in-tree libraries exist (no-op when they are up to date)."""
    import __graft_entry__
    __graft_entry__.build()
    return True
