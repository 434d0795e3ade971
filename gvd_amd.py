"""Import alias: the package directory is named `grounded-video-description_amd` (not a valid Python
identifier), so `import gvd_amd` loads it and registers it (and its submodules) under this name."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module('grounded-video-description_amd')
sys.modules[__name__] = _pkg
