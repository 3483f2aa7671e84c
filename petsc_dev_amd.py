"""Import shim: the package directory is named `petsc-dev_amd` (not a valid identifier), so
`import petsc_dev_amd` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "petsc-dev_amd")
_spec = importlib.util.spec_from_file_location(
    "petsc_dev_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["petsc_dev_amd"] = _mod
_spec.loader.exec_module(_mod)
