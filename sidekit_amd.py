"""Import shim: the package directory is named `speaker-recognition_amd` (not a Python identifier),
so it is loaded by path.  Usage:  import sidekit_amd as sidekit"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("speaker-recognition_amd")
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
__version__ = _pkg.__version__
package = _pkg
