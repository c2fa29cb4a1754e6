"""Import shim: ``import adaface_dev_amd`` -> the package directory ``adaface-dev_amd/``.

The package directory carries the reference repo's name (with its hyphen), which
is not a Python identifier; this module replaces itself in ``sys.modules`` with
the package loaded from that directory.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adaface-dev_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
