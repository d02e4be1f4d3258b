"""Importable alias of the ``multi-modal-qg_amd`` package (whose mandated directory name
contains hyphens): ``import mmqg_amd`` == ``importlib.import_module("multi-modal-qg_amd")``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("multi-modal-qg_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("mmqg_amd", _pkg)
