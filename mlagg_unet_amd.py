"""Import alias: the package directory is ``mlagg-unet_amd/`` (hyphenated, not a Python
identifier); ``import mlagg_unet_amd`` loads that directory as the package ``mlagg_unet_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlagg-unet_amd")
_spec = importlib.util.spec_from_file_location(
    "mlagg_unet_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mlagg_unet_amd"] = _mod
_spec.loader.exec_module(_mod)
