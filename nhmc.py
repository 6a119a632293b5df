"""Import shim: `import nhmc` loads the package that lives in ./noise-space-hmc_amd/
(the directory keeps the project's name, which is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'noise-space-hmc_amd')
_spec = importlib.util.spec_from_file_location('nhmc', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['nhmc'] = _mod
_spec.loader.exec_module(_mod)
