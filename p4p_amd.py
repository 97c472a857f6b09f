"""Importable alias for the package whose directory name is not a Python identifier.

``import p4p_amd`` binds the real package object; every submodule is registered under BOTH names (the ones loaded by the
package's __init__ right away, later ones through the finder below), so ``from p4p_amd.optim import FusedAdamW`` returns the
class the trainers use and ``p4p_amd.functional`` is the module the model imported.  (Without this ``import p4p_amd.x`` loaded
a second copy of x.py under the alias name -- with its own module globals and its own classes -- and rebound the package
attribute to the copy.)"""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = 'adapting-2d-vits-for-3d-point-cloud-understanding_amd'
_ALIAS = __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + '.'):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        mod = importlib.import_module(_REAL + spec.name[len(_ALIAS):])
        return mod

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + '.'):
        sys.modules[_ALIAS + _name[len(_REAL):]] = _mod
