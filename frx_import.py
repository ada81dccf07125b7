"""The package directory is named `fast-racing_amd` (hyphen, as the task specifies), which Python
cannot import by name; this helper registers it as module `fast_racing_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))


def load():
    if "fast_racing_amd" in sys.modules:
        return sys.modules["fast_racing_amd"]
    pkg_dir = os.path.join(_ROOT, "fast-racing_amd")
    spec = importlib.util.spec_from_file_location("fast_racing_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fast_racing_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


frx = load()
