"""Device-resident matrices over the C ABI (include/m4ri_hip.h, section 2).

`DMat` owns (or wraps) a dense bit matrix in HBM: row-major 64-bit words, LSB-first, `ld` words
per row.  Used by bench.py, the multi-GPU layer and tests that must not pay PCIe per product.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import ALGO_AUTO, ALGO_M4RM, ALGO_NAIVE, ALGO_STRASSEN, DMatStruct  # noqa: F401
from .friendly import BinMatrix

ALGOS = {"auto": ALGO_AUTO, "m4rm": ALGO_M4RM, "strassen": ALGO_STRASSEN, "naive": ALGO_NAIVE}


def device_count():
    return _lib.lib().gf2_device_count()


def require_gpu():
    if device_count() <= 0:
        raise _lib.HipError("no usable HIP device: the multiply path has no CPU fallback")


class DMat:
    def __init__(self, nrows, ncols, _wrap=None):
        self.s = DMatStruct()
        self._owned = _wrap is None
        self._keep = None
        self._streams = set()  # streams this matrix has been used on (the device API is asynchronous)
        if _wrap is None:
            _lib.check(_lib.lib().gf2_dmat_alloc(ctypes.byref(self.s), nrows, ncols), "gf2_dmat_alloc")
        else:
            ptr, ld, keep = _wrap
            self.s.data, self.s.ld, self.s.nrows, self.s.ncols = ptr, ld, nrows, ncols
            self._keep = keep

    def __del__(self):
        if getattr(self, "_owned", False) and self.s.data:
            try:
                if len(self._streams) == 1:  # stream-ordered: recycled once that stream has passed this point
                    _lib.lib().gf2_dmat_free_async(ctypes.byref(self.s), next(iter(self._streams)))
                else:  # never used, or used on several streams: hipFree semantics (waits for the device)
                    _lib.lib().gf2_dmat_free(ctypes.byref(self.s))
            except Exception:  # interpreter shutdown: module globals may already be gone
                pass

    def _on(self, stream):
        self._streams.add(stream)
        return ctypes.byref(self.s)

    nrows = property(lambda self: self.s.nrows)
    ncols = property(lambda self: self.s.ncols)
    ld = property(lambda self: self.s.ld)

    @staticmethod
    def wrap(ptr, nrows, ncols, ld, keep=None):
        """Wrap caller-owned device memory (e.g. a torch int64 tensor's data_ptr())."""
        return DMat(nrows, ncols, _wrap=(ptr, ld, keep))

    @staticmethod
    def from_torch(t, ncols):
        """Wrap a 2-D contiguous torch.int64 CUDA tensor of shape (nrows, ld)."""
        assert t.dim() == 2 and t.is_contiguous() and t.element_size() == 8 and t.is_cuda
        assert t.shape[1] * 64 >= ncols
        return DMat.wrap(t.data_ptr(), t.shape[0], ncols, t.shape[1], keep=t)

    @staticmethod
    def random(nrows, ncols, seed, stream=None):
        m = DMat(nrows, ncols)
        m.fill_random(seed, stream)
        return m

    def fill_random(self, seed, stream=None):
        _lib.check(_lib.lib().gf2_dmat_fill_random(self._on(stream), seed, stream), "gf2_dmat_fill_random")

    @staticmethod
    def from_host(bm, stream=None):
        m = DMat(bm.nrows(), bm.ncols())
        _lib.check(_lib.lib().gf2_dmat_upload(m._on(stream), bm.mzd, stream), "gf2_dmat_upload")
        return m

    @staticmethod
    def from_words(arr, ncols, stream=None):
        return DMat.from_host(BinMatrix.from_words(arr, ncols), stream)

    def to_host(self, stream=None):
        bm = BinMatrix.zero(self.nrows, self.ncols)
        _lib.check(_lib.lib().gf2_dmat_download(bm.mzd, self._on(stream), stream), "gf2_dmat_download")
        return bm

    def to_words(self, stream=None):
        return self.to_host(stream).to_words()

    # -- the friendly layer's operators on device-resident operands: same meaning as on
    #    BinMatrix (binary_matrix.rs:434-526), no PCIe traffic; everything runs on the default stream --
    def __mul__(self, other):
        return mul(self, other)

    def __add__(self, other):
        return add(self, other)

    def __eq__(self, other):
        return isinstance(other, DMat) and equal(self, other)

    __hash__ = None

    def transposed(self):
        return transpose(self)

    def clone(self, stream=None):
        zero = add(self, self, stream=stream)  # self ^ self; freed stream-ordered behind the second add
        return add(self, zero, stream=stream)

    def rank(self):
        return echelonize(self.clone(), full=False)[0]

    def inverted(self):
        inv = inverse(self)
        if inv is None:
            raise _lib.HipError("matrix is singular")
        return inv


def mul(A, B, C=None, accumulate=False, algo="auto", param=0, stream=None):
    """C (+)= A*B on the device; asynchronous on `stream` (int hipStream_t or None)."""
    if C is None:
        C = DMat(A.nrows, B.ncols)
    rc = _lib.lib().gf2_mul_dev(C._on(stream), A._on(stream), B._on(stream), int(accumulate), ALGOS[algo], param, stream)
    _lib.check(rc, "gf2_mul_dev")
    return C


def mul_nt(A, Bt, C=None, accumulate=False, stream=None):
    if C is None:
        C = DMat(A.nrows, Bt.nrows)
    _lib.check(_lib.lib().gf2_mul_nt_dev(C._on(stream), A._on(stream), Bt._on(stream), int(accumulate), stream), "gf2_mul_nt_dev")
    return C


def add(A, B, C=None, stream=None):
    if C is None:
        C = DMat(A.nrows, A.ncols)
    _lib.check(_lib.lib().gf2_add_dev(C._on(stream), A._on(stream), B._on(stream), stream), "gf2_add_dev")
    return C


def transpose(S, D=None, stream=None):
    if D is None:
        D = DMat(S.ncols, S.nrows)
    _lib.check(_lib.lib().gf2_transpose_dev(D._on(stream), S._on(stream), stream), "gf2_transpose_dev")
    return D


def equal(A, B, stream=None):
    out = ctypes.c_int(0)
    _lib.check(_lib.lib().gf2_equal_dev(A._on(stream), B._on(stream), ctypes.byref(out), stream), "gf2_equal_dev")
    return bool(out.value)


def echelonize(A, full=True, ncols_limit=0, stream=None):
    """In-place (reduced) row echelon form of the first ncols_limit columns (0 = all) -> (rank, pivot columns)."""
    rank = ctypes.c_int(0)
    cap = min(A.nrows, ncols_limit if 0 < ncols_limit < A.ncols else A.ncols)
    piv = (ctypes.c_int * max(cap, 1))()
    _lib.check(_lib.lib().gf2_echelonize_dev(A._on(stream), int(bool(full)), int(ncols_limit), ctypes.byref(rank), piv,
                                             stream), "gf2_echelonize_dev")
    return rank.value, list(piv[:rank.value])


def inverse(A, stream=None):
    """A^-1 as a new DMat, or None if A is singular."""
    out = DMat(A.nrows, A.ncols)
    singular = ctypes.c_int(0)
    _lib.check(_lib.lib().gf2_inverse_dev(out._on(stream), A._on(stream), ctypes.byref(singular), stream),
               "gf2_inverse_dev")
    return None if singular.value else out


def prof_enable(on):
    _lib.lib().gf2_prof_enable(int(on))


def prof_read(reset=True):
    n, ms = ctypes.c_int(0), ctypes.c_double(0)
    _lib.check(_lib.lib().gf2_prof_read(ctypes.byref(n), ctypes.byref(ms), int(reset)), "gf2_prof_read")
    return n.value, ms.value
