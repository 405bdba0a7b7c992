"""Import shim: exposes the package directory `cis-565-final-vr-raytracer_amd/` (not a valid Python identifier)
as the module `restir_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cis-565-final-vr-raytracer_amd")
_spec = importlib.util.spec_from_file_location("restir_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["restir_amd"] = _mod
_spec.loader.exec_module(_mod)
