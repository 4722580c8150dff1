"""Import shim: the package directory is named ``gan-ode_amd`` (not a valid Python identifier), so this module
loads it under the importable name ``gan_ode_amd`` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gan-ode_amd")
_spec = importlib.util.spec_from_file_location("gan_ode_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gan_ode_amd"] = _mod
_spec.loader.exec_module(_mod)
