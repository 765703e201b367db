"""Import alias: the package directory is named ``fhe-linformer_amd`` (not a valid Python identifier),
so ``import fhe_linformer_amd`` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fhe-linformer_amd")
_spec = importlib.util.spec_from_file_location(
    "fhe_linformer_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fhe_linformer_amd"] = _mod
_spec.loader.exec_module(_mod)
