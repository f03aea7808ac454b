"""Import alias: `import tagrec_amd` loads the package that lives in the directory
`tag-aware-recommendation_amd/` (a hyphen cannot appear in a Python import name)."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "tag-aware-recommendation_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
