"""Import alias: the package directory is `fcl-taco2_amd/` (not a legal Python identifier).

`import fcl_taco2_amd` loads `fcl-taco2_amd/__init__.py` as the package `fcl_taco2_amd`, so
`fcl_taco2_amd.nets...:Tacotron2_sa` works as a `--model-module` string the way the reference's
`nets...:Tacotron2_sa` does (reference tts_train.py:103-109).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fcl-taco2_amd")
_spec = importlib.util.spec_from_file_location(
    "fcl_taco2_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fcl_taco2_amd"] = _mod
_spec.loader.exec_module(_mod)
