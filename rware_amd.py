"""Import shim: `import rware_amd` loads the package in `robotic-warehouse_amd/` (whose directory
name, fixed by the repository layout, is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("robotic-warehouse_amd")
for _name, _mod in list(sys.modules.items()):
    if _name == "robotic-warehouse_amd" or _name.startswith("robotic-warehouse_amd."):
        sys.modules["rware_amd" + _name[len("robotic-warehouse_amd"):]] = _mod
sys.modules[__name__] = _pkg
