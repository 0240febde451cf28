"""Import alias: the package directory is `ai-toolkit_amd/` (not a valid Python identifier), so
`import ai_toolkit_amd` loads it from there and registers it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ai-toolkit_amd")
_spec = importlib.util.spec_from_file_location(
    "ai_toolkit_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ai_toolkit_amd"] = _mod
_spec.loader.exec_module(_mod)
