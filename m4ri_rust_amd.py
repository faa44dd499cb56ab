"""Import shim: the package directory is `m4ri-rust_amd/` (hyphen, not importable by name).
`import m4ri_rust_amd` loads that directory as the package `m4ri_rust_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "m4ri-rust_amd")
_spec = importlib.util.spec_from_file_location("m4ri_rust_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["m4ri_rust_amd"] = _mod
_spec.loader.exec_module(_mod)
