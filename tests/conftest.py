import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def frx():
    from frx_import import frx as mod
    return mod


@pytest.fixture(scope="session")
def sc(frx):
    from fast_racing_amd import scenario
    return scenario


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import binding
    binding.lib()
    return binding


def has_gpu() -> bool:
    try:
        from frx_import import frx as mod
        return mod.lib().frx_device_count() > 0
    except Exception:
        return False
