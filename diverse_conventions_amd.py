"""Import shim: the package directory is `diverse-conventions_amd/` (repo layout contract), which is
not a legal Python identifier, so `import diverse_conventions_amd` lands here and this module
replaces itself in sys.modules with the real package loaded from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diverse-conventions_amd")
_spec = importlib.util.spec_from_file_location(
    "diverse_conventions_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["diverse_conventions_amd"] = _mod
_spec.loader.exec_module(_mod)
