"""Imports the package directory `alphazero-connect6_amd/` (not a valid identifier) as module
`alphazero_connect6_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_NAME = "alphazero_connect6_amd"


def load_package():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    path = os.path.join(_ROOT, "alphazero-connect6_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(_NAME, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
