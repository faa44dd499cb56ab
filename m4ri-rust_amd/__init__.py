"""m4ri-rust_amd: MI355X-native dense GF(2) matrix multiply behind the M4RI / m4ri-rust ABI.

The product is libm4ri_hip.so (csrc/, C ABI in include/m4ri_hip.h).  This package is the host-side
mirror of the reference's friendly layer plus a thin device-resident API; import it as
`m4ri_rust_amd` (the shim m4ri_rust_amd.py at the repository root maps the name onto this
directory, whose name carries a hyphen).
"""
from . import _lib  # noqa: F401
from .friendly import BinMatrix, BinVector, PanicError, get_mul_strategy, set_mul_strategy, solve_left  # noqa: F401

__all__ = ["BinMatrix", "BinVector", "PanicError", "set_mul_strategy", "get_mul_strategy", "solve_left"]
