"""Import alias: the package directory `scene-text-recognition_amd/` has a hyphen in its
name (it follows the reference's repository name), which the `import` statement cannot
spell.  `import str_er_amd` gives the same module object."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("scene-text-recognition_amd")
sys.modules[__name__] = _pkg
