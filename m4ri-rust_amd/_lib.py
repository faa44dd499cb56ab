"""ctypes binding of libm4ri_hip.so (declarations: include/m4ri_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no Python or CPU stand-in for the multiply path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libm4ri_hip.so")

c_word_p = ctypes.POINTER(ctypes.c_uint64)


class MzdBlock(ctypes.Structure):
    """mzd_block_t (m4ri-sys/src/mzd.rs:16-21)."""
    _fields_ = [("size", ctypes.c_size_t), ("begin", c_word_p), ("end", c_word_p)]


class Mzd(ctypes.Structure):
    """mzd_t, 64 bytes (m4ri-sys/src/mzd.rs:24-79)."""
    _fields_ = [
        ("nrows", ctypes.c_int),
        ("ncols", ctypes.c_int),
        ("width", ctypes.c_int),
        ("rowstride", ctypes.c_int),
        ("offset_vector", ctypes.c_int),
        ("row_offset", ctypes.c_int),
        ("flags", ctypes.c_uint8),
        ("blockrows_log", ctypes.c_uint8),
        ("padding", ctypes.c_uint8 * 14),
        ("high_bitmask", ctypes.c_uint64),
        ("blocks", ctypes.POINTER(MzdBlock)),
        ("rows", ctypes.POINTER(c_word_p)),
    ]


assert ctypes.sizeof(Mzd) == 64

MzdP = ctypes.POINTER(Mzd)


class DMatStruct(ctypes.Structure):
    """gf2_dmat (include/m4ri_hip.h, section 2)."""
    _fields_ = [("data", ctypes.c_void_p), ("ld", ctypes.c_int64), ("nrows", ctypes.c_int), ("ncols", ctypes.c_int)]


DMatP = ctypes.POINTER(DMatStruct)

ALGO_AUTO, ALGO_M4RM, ALGO_STRASSEN, ALGO_NAIVE = 0, 1, 2, 3

_I = ctypes.c_int
_PROTOS = {
    # name: (restype, argtypes)
    "mzd_init": (MzdP, [_I, _I]),
    "mzd_free": (None, [MzdP]),
    "mzd_init_window": (MzdP, [MzdP, _I, _I, _I, _I]),
    "mzd_copy": (MzdP, [MzdP, MzdP]),
    "mzd_equal": (_I, [MzdP, MzdP]),
    "mzd_randomize": (None, [MzdP]),
    "mzd_set_ui": (None, [MzdP, ctypes.c_uint]),
    "mzd_transpose": (MzdP, [MzdP, MzdP]),
    "mzd_add": (MzdP, [MzdP, MzdP, MzdP]),
    "mzd_sub": (MzdP, [MzdP, MzdP, MzdP]),
    "mzd_concat": (MzdP, [MzdP, MzdP, MzdP]),
    "mzd_stack": (MzdP, [MzdP, MzdP, MzdP]),
    "mzd_submatrix": (MzdP, [MzdP, MzdP, _I, _I, _I, _I]),
    "mzd_is_zero": (_I, [MzdP]),
    "mzd_row_swap": (None, [MzdP, _I, _I]),
    "mzd_copy_row": (None, [MzdP, _I, MzdP, _I]),
    "mzd_col_swap": (None, [MzdP, _I, _I]),
    "mzd_row_clear_offset": (None, [MzdP, _I, _I]),
    "mzd_invert_naive": (MzdP, [MzdP, MzdP, MzdP]),
    "m4ri_opt_k": (_I, [_I, _I, _I]),
    "mzd_make_table": (None, [MzdP, _I, _I, _I, MzdP, ctypes.POINTER(_I)]),
    "mzd_echelonize": (_I, [MzdP, _I]),
    "mzd_echelonize_m4ri": (_I, [MzdP, _I, _I]),
    "mzd_echelonize_pluq": (_I, [MzdP, _I]),
    "mzd_inv_m4ri": (MzdP, [MzdP, MzdP, _I]),
    "mzd_solve_left": (_I, [MzdP, MzdP, _I, _I]),
    "mzd_mul_m4rm": (MzdP, [MzdP, MzdP, MzdP, _I]),
    "mzd_addmul_m4rm": (MzdP, [MzdP, MzdP, MzdP, _I]),
    "mzd_mul": (MzdP, [MzdP, MzdP, MzdP, _I]),
    "mzd_addmul": (MzdP, [MzdP, MzdP, MzdP, _I]),
    "mzd_mul_naive": (MzdP, [MzdP, MzdP, MzdP]),
    "mzd_addmul_naive": (MzdP, [MzdP, MzdP, MzdP]),
    "_mzd_mul_naive": (MzdP, [MzdP, MzdP, MzdP, _I]),
    "_mzd_mul_va": (MzdP, [MzdP, MzdP, MzdP, _I]),
    "gf2_device_count": (_I, []),
    "gf2_last_error": (ctypes.c_char_p, []),
    "gf2_dmat_alloc": (_I, [DMatP, _I, _I]),
    "gf2_dmat_free": (None, [DMatP]),
    "gf2_dmat_free_async": (_I, [DMatP, ctypes.c_void_p]),
    "gf2_dmat_upload": (_I, [DMatP, MzdP, ctypes.c_void_p]),
    "gf2_dmat_download": (_I, [MzdP, DMatP, ctypes.c_void_p]),
    "gf2_dmat_fill_random": (_I, [DMatP, ctypes.c_uint64, ctypes.c_void_p]),
    "gf2_dmat_fill_random_rows": (_I, [DMatP, ctypes.c_uint64, ctypes.c_int64, ctypes.c_void_p]),
    "gf2_dmat_fill_random_block": (_I, [DMatP, ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, _I, ctypes.c_void_p]),
    "gf2_strassen_levels": (_I, [_I, _I, _I, _I, _I]),
    "gf2_strassen_pass_bytes": (ctypes.c_double, [_I, _I, _I, _I]),
    "gf2_mul_plan": (_I, [_I, _I, _I, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gf2_model_time": (ctypes.c_double, [_I, _I, _I, _I]),
    "gf2_tile_plan": (ctypes.c_double, [_I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_longlong)]),
    "gf2_tile_plan_band": (None, [_I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_longlong)]),
    "gf2_mul_dev": (_I, [DMatP, DMatP, DMatP, _I, _I, _I, ctypes.c_void_p]),
    "gf2_mul_nt_dev": (_I, [DMatP, DMatP, DMatP, _I, ctypes.c_void_p]),
    "gf2_add_dev": (_I, [DMatP, DMatP, DMatP, ctypes.c_void_p]),
    "gf2_transpose_dev": (_I, [DMatP, DMatP, ctypes.c_void_p]),
    "gf2_equal_dev": (_I, [DMatP, DMatP, ctypes.POINTER(_I), ctypes.c_void_p]),
    "gf2_echelonize_dev": (_I, [DMatP, _I, _I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.c_void_p]),
    "gf2_inverse_dev": (_I, [DMatP, DMatP, ctypes.POINTER(_I), ctypes.c_void_p]),
    "gf2_mul_workspace_bytes": (ctypes.c_size_t, [_I, _I, _I, _I, _I]),
    "gf2_mul_multi": (MzdP, [MzdP, MzdP, MzdP, _I, _I, ctypes.POINTER(_I), _I]),
    "gf2_mzd_cache_on_device": (_I, [MzdP]),
    "gf2_mzd_uncache": (None, [MzdP]),
    "gf2_mzd_prewarm": (_I, [_I, _I, _I]),
    "gf2_trim": (_I, []),
    "gf2_mul_host_small": (_I, [MzdP, MzdP, MzdP, _I]),
    "gf2_mul_nt_host_small": (_I, [MzdP, MzdP, MzdP, _I]),
    "gf2_echelonize_host_small": (_I, [MzdP, _I]),
    "gf2_host_small_calls": (ctypes.c_longlong, []),
    "gf2_mzd_save": (_I, [ctypes.c_char_p, MzdP]),
    "gf2_mzd_load": (MzdP, [ctypes.c_char_p]),
    "gf2_prof_enable": (None, [_I]),
    "gf2_prof_read": (_I, [ctypes.POINTER(_I), ctypes.POINTER(ctypes.c_double), _I]),
    "gf2_kernel_census": (ctypes.c_size_t, [ctypes.c_char_p, ctypes.c_size_t]),
    "gf2_host_plan_model": (_I, [_I, _I, _I, _I, _I, ctypes.POINTER(ctypes.c_double)]),
}

# every symbol include/m4ri_hip.h declares; tests check the library exports all of them
DECLARED_SYMBOLS = tuple(_PROTOS.keys())

_lib = None


def lib():
    """Load libm4ri_hip.so (once). Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C m4ri-rust_amd/csrc`). There is no fallback for the HIP multiply path.")
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


class HipError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        msg = lib().gf2_last_error()
        raise HipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
