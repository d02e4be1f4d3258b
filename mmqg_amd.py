"""Importable alias of the ``multi-modal-qg_amd`` package (whose mandated directory name
contains hyphens).  ``import mmqg_amd`` and ``from mmqg_amd import trainer`` resolve to the very
same module objects as ``importlib.import_module("multi-modal-qg_amd[.x]")`` — a meta-path
finder maps the alias names onto the real ones so no module is ever loaded twice."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_ALIAS, _REAL = "mmqg_amd", "multi-modal-qg_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
