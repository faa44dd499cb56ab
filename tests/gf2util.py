"""Test-side helpers: ctypes view of oracle/libgf2oracle.so, numpy bit packing and the seeded
input generator shared by tests, golden fixtures and bench.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(it loads the oracle, which is test infrastructure, never the product path).
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgf2oracle.so")

_U64P = ctypes.POINTER(ctypes.c_uint64)
_I = ctypes.c_int


def build_oracle():
    src = [os.path.join(ORACLE_DIR, f) for f in ("gf2_oracle.c", "gf2_oracle.h")]
    if not os.path.exists(ORACLE_SO) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = ctypes.CDLL(build_oracle())
        lib.oracle_splitmix64.restype = ctypes.c_uint64
        lib.oracle_splitmix64.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
        lib.oracle_fill_random.argtypes = [_U64P, _I, _I, _I, ctypes.c_uint64]
        mul9 = [_U64P, _I, _U64P, _I, _U64P, _I, _I, _I, _I]
        lib.oracle_mul_bits.argtypes = mul9
        lib.oracle_mul_naive.argtypes = mul9
        lib.oracle_mul_fast.argtypes = mul9
        lib.oracle_mul_naive_t.argtypes = mul9 + [_I]
        lib.oracle_mul_m4rm.argtypes = mul9 + [_I, _I]
        lib.oracle_mul_strassen.argtypes = mul9 + [_I, _I]
        lib.oracle_mul_va.argtypes = [_U64P, _U64P, _U64P, _I, _I, _I, _I]
        lib.oracle_transpose.argtypes = [_U64P, _I, _U64P, _I, _I, _I]
        lib.oracle_add.argtypes = [_U64P, _I, _U64P, _I, _U64P, _I, _I, _I]
        lib.oracle_opt_k.argtypes = [_I, _I, _I]
        lib.oracle_opt_k.restype = _I
        lib.oracle_echelonize.argtypes = [_U64P, _I, _I, _I, _I, _I, ctypes.POINTER(_I)]
        lib.oracle_echelonize.restype = _I
        lib.oracle_inverse.argtypes = [_U64P, _I, _U64P, _I, _I]
        lib.oracle_inverse.restype = _I
        lib.oracle_solve_left.argtypes = [_U64P, _I, _I, _I, _U64P, _I, _I, _I]
        lib.oracle_solve_left.restype = _I
        for f in ("oracle_fill_random", "oracle_mul_bits", "oracle_mul_naive", "oracle_mul_fast",
                  "oracle_mul_naive_t", "oracle_mul_m4rm", "oracle_mul_strassen", "oracle_mul_va",
                  "oracle_transpose", "oracle_add"):
            getattr(lib, f).restype = None
        _oracle = lib
    return _oracle


def width(ncols):
    return (ncols + 63) // 64


def ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_U64P)


# ---- seeded generator (same stream as oracle_splitmix64 / the device generator) -------------

_M64 = (1 << 64) - 1


def splitmix64(seed, t):
    """Vectorised splitmix64: word t (numpy uint64 array) of the stream `seed`."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed & _M64) + (t.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_words(nrows, ncols, seed):
    """nrows x width(ncols) uint64 words, uniform bits, excess bits of the last word zero."""
    w = width(ncols)
    t = np.arange(nrows * w, dtype=np.uint64)
    a = splitmix64(seed, t).reshape(nrows, w)
    if ncols % 64:
        a[:, -1] &= np.uint64((1 << (ncols % 64)) - 1)
    return np.ascontiguousarray(a)


# ---- packing -------------------------------------------------------------------------------

def words_to_bits(a, ncols):
    """(nrows, width) uint64 -> (nrows, ncols) uint8; bit j of a row = bit j%64 of word j//64."""
    b = np.unpackbits(a.view(np.uint8).reshape(a.shape[0], -1), axis=1, bitorder="little")
    return b[:, :ncols]


def bits_to_words(b):
    nrows, ncols = b.shape
    w = width(ncols)
    pad = np.zeros((nrows, w * 64), dtype=np.uint8)
    pad[:, :ncols] = b
    return np.ascontiguousarray(np.packbits(pad, axis=1, bitorder="little").view(np.uint64).reshape(nrows, w))


def numpy_mul(a, b, m, l, n):
    """Independent reference product: integer matmul of the unpacked bits, mod 2."""
    A = words_to_bits(a, l).astype(np.float32) if l < (1 << 24) else words_to_bits(a, l).astype(np.float64)
    B = words_to_bits(b, n).astype(A.dtype)
    C = (A @ B)
    return bits_to_words((np.rint(C).astype(np.int64) & 1).astype(np.uint8))


# ---- oracle wrappers on numpy word arrays ----------------------------------------------------

def _mul(fn, a, b, m, l, n, *extra):
    c = np.zeros((m, width(n)), dtype=np.uint64)
    fn(ptr(c), c.shape[1], ptr(a), a.shape[1], ptr(b), b.shape[1], m, l, n, *extra)
    return c


def o_mul_bits(a, b, m, l, n):
    return _mul(oracle().oracle_mul_bits, a, b, m, l, n)


def o_mul_naive(a, b, m, l, n):
    return _mul(oracle().oracle_mul_naive, a, b, m, l, n)


def o_mul_m4rm(a, b, m, l, n, k=0):
    return _mul(oracle().oracle_mul_m4rm, a, b, m, l, n, k, 1)


def o_mul_strassen(a, b, m, l, n, cutoff=0):
    return _mul(oracle().oracle_mul_strassen, a, b, m, l, n, cutoff, 1)


def o_mul_fast(a, b, m, l, n):
    return _mul(oracle().oracle_mul_fast, a, b, m, l, n)


def o_transpose(a, nrows, ncols):
    d = np.zeros((ncols, width(nrows)), dtype=np.uint64)
    oracle().oracle_transpose(ptr(d), d.shape[1], ptr(a), a.shape[1], nrows, ncols)
    return d


def o_echelonize(a, nrows, ncols, full=True, limit=0):
    """-> (echelon form, rank, pivot columns) of a copy of `a` (oracle_echelonize)."""
    m = np.ascontiguousarray(a).copy()
    piv = (ctypes.c_int * (min(nrows, ncols) + 1))()
    rank = oracle().oracle_echelonize(ptr(m), m.shape[1], nrows, ncols, limit, 1 if full else 0, piv)
    return m, rank, list(piv[:rank])


def o_inverse(a, n):
    """-> inverse words, or None if singular."""
    a = np.ascontiguousarray(a)
    inv = np.zeros_like(a)
    rc = oracle().oracle_inverse(ptr(inv), inv.shape[1], ptr(a), a.shape[1], n)
    return inv if rc == 0 else None


def o_solve_left(a, m, n, b, brows, k):
    """-> (X as brows x width(k) words, consistent?)."""
    a = np.ascontiguousarray(a)
    x = np.ascontiguousarray(b).copy()
    rc = oracle().oracle_solve_left(ptr(a), a.shape[1], m, n, ptr(x), x.shape[1], brows, k)
    return x, rc == 0
