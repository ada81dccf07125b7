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


ITERATION_CAP = 60000


@pytest.fixture(scope="session", autouse=True)
def _bounded_optimisations():
    """The reference runs L-BFGS with max_iterations = 0 = unbounded (CPU.hpp:1243-1247, lbfgs.hpp:128-140), and on an INFEASIBLE scenario that
    can mean for ever: Monte-Carlo scenario 170 sends the reference algorithm into a loop on NaN objectives for some roundings (the CPU oracle
    with the accumulated abscissa, built without FMA contraction: 200 000 iterations and counting; the device drivers stop after 64 non-finite
    values).  Every optimisation a test starts without a limit of its own gets this one - ten times the longest feasible run of the suite -
    so that a test can fail but not hang; the limit shows up as LBFGSERR_MAXIMUMITERATION (-1004)."""
    import inspect
    from frx_import import frx as mod
    classes = [mod.Problem, mod.MultiProblem]
    try:
        from oracle import binding
        classes.append(binding.Oracle)
    except Exception:
        pass
    saved = []
    for cls in classes:
        orig = cls.optimize
        sig = inspect.signature(orig)

        def capped(self, *a, __orig=orig, __sig=sig, **kw):
            ba = __sig.bind(self, *a, **kw)
            ba.apply_defaults()
            if not ba.arguments.get("max_iterations"):
                ba.arguments["max_iterations"] = ITERATION_CAP
            return __orig(*ba.args, **ba.kwargs)
        cls.optimize = capped
        saved.append((cls, orig))
    yield
    for cls, orig in saved:
        cls.optimize = orig


def has_gpu() -> bool:
    try:
        from frx_import import frx as mod
        return mod.lib().frx_device_count() > 0
    except Exception:
        return False
