"""Import alias: the package directory is named `grounded-video-description_amd` (not a valid Python
identifier), so `import gvd_amd` loads it and registers it (and its submodules) under this name.

Submodules are aliased as well - `from gvd_amd.hip import GvdHipError`, `import gvd_amd.optim` resolve to the SAME module
objects as `grounded-video-description_amd.hip` / `.optim` (a plain `sys.modules` alias of the package alone would let the
import system load a second copy of a submodule under the `gvd_amd.` name: two error classes, two strict switches, two
library handles)."""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_REAL = 'grounded-video-description_amd'
_ALIAS = __name__


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, mod):
        self.mod = mod
        self.keep = {k: getattr(mod, k, None) for k in ('__spec__', '__loader__', '__package__', '__name__')}

    def create_module(self, spec):
        return self.mod

    def exec_module(self, module):
        for k, v in self.keep.items():           # the module keeps its own identity (the import system may have rewritten it)
            if v is not None:
                setattr(module, k, v)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_ALIAS + '.'):
            return None
        real = _REAL + fullname[len(_ALIAS):]
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError:
            return None
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(mod), origin=getattr(mod, '__file__', None))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + '.'):
        sys.modules[_ALIAS + _name[len(_REAL):]] = _mod
sys.modules[_ALIAS] = _pkg
