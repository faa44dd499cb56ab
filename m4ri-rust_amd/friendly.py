"""Host-side mirror of m4ri-rust's friendly layer over the C ABI of libm4ri_hip.so.

Mirrors (names, argument meaning, error behaviour) of
  m4ri-rust/src/friendly/binary_matrix.rs  -> BinMatrix
  m4ri-rust/src/friendly/binary_vector.rs  -> BinVector
restricted to what the multiply path needs.  Rust panics are
Python exceptions (`PanicError`).  Every product goes through mzd_mul / mzd_mul_m4rm /
mzd_mul_naive exactly as `mul_impl!` selects them (binary_matrix.rs:53-95), i.e. through the GPU.

The reference has no Rust toolchain in this environment, so this mirror exists for the parity
tests and bench; the binding a Rust maintainer would use is unchanged m4ri-sys.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import MzdP


class PanicError(RuntimeError):
    """A condition on which the Rust reference panics."""


# Cargo features m4rm_mul / naive_mul / strassen_mul (m4ri-rust/Cargo.toml:29-34); default = Strassen
_MUL_STRATEGY = os.environ.get("M4RI_RUST_MUL", "strassen")


def set_mul_strategy(name):
    """Select what `mul_impl!` expands to: 'strassen' (default), 'm4rm' or 'naive'."""
    global _MUL_STRATEGY
    if name not in ("strassen", "m4rm", "naive"):
        raise ValueError("You need to set only one of the feature flags as mul strategy")
    _MUL_STRATEGY = name


def get_mul_strategy():
    return _MUL_STRATEGY


def _mul_impl(dest, a, b):
    """mul_impl! (binary_matrix.rs:53-95)."""
    L = _lib.lib()
    if _MUL_STRATEGY == "m4rm":
        return L.mzd_mul_m4rm(dest, a, b, 0)
    if _MUL_STRATEGY == "naive":
        return L.mzd_mul_naive(dest, a, b)
    return L.mzd_mul(dest, a, b, 0)


def _width(n):
    return (n + 63) // 64


class BinVector:
    """Bit vector, LSB-first 64-bit blocks like vob::Vob (binary_vector.rs:13-17)."""

    __slots__ = ("_w", "_len")

    def __init__(self, words=None, length=0):
        self._w = np.zeros(0, dtype=np.uint64) if words is None else np.array(words, dtype=np.uint64)
        self._len = int(length)
        self.mask_last_block()

    # -- constructors (binary_vector.rs:44-108) --
    @staticmethod
    def new():
        return BinVector()

    @staticmethod
    def from_bools(bools):
        b = np.asarray(list(bools), dtype=np.uint8)
        n = len(b)
        pad = np.zeros(_width(n) * 64, dtype=np.uint8)
        pad[:n] = b
        return BinVector(np.packbits(pad, bitorder="little").view(np.uint64), n)

    @staticmethod
    def from_elem(length, elem):
        return BinVector.from_bools([bool(elem)] * length)

    @staticmethod
    def from_function(length, f):
        return BinVector.from_bools([bool(f(i)) for i in range(length)])

    @staticmethod
    def random(length, rng=None):
        rng = rng or np.random.default_rng()
        return BinVector(rng.integers(0, 1 << 64, size=_width(length), dtype=np.uint64), length)

    @staticmethod
    def with_capacity(_length):
        return BinVector()

    @staticmethod
    def from_bytes(data):
        """MSB-first inside each byte (binary_vector.rs:230-237)."""
        bits = np.unpackbits(np.frombuffer(bytes(data), dtype=np.uint8), bitorder="big")
        return BinVector.from_bools(bits)

    # -- storage --
    def mask_last_block(self):
        w = _width(self._len)
        if len(self._w) != w:
            nw = np.zeros(w, dtype=np.uint64)
            k = min(w, len(self._w))
            nw[:k] = self._w[:k]
            self._w = nw
        if self._len % 64 and w:
            self._w[-1] &= np.uint64((1 << (self._len % 64)) - 1)

    def get_storage(self):
        return self._w

    def __len__(self):
        return self._len

    def len(self):
        return self._len

    def get(self, i):
        if i < 0 or i >= self._len:
            return None
        return bool((int(self._w[i // 64]) >> (i % 64)) & 1)

    def push(self, bit):
        i = self._len
        self._len += 1
        self.mask_last_block()
        if bit:
            self._w[i // 64] |= np.uint64(1 << (i % 64))

    def to_bools(self):
        return [self.get(i) for i in range(self._len)]

    def count_ones(self):
        return int(sum(bin(int(x)).count("1") for x in self._w))

    def extend_from_binvec(self, other):
        for i in range(len(other)):
            self.push(other.get(i))

    def clone(self):
        return BinVector(self._w.copy(), self._len)

    def as_matrix(self):
        """1 x len row matrix (binary_vector.rs:130-132)."""
        return BinMatrix.new([self.clone()])

    def as_column_matrix(self):
        return self.as_matrix().transposed()

    def as_u32(self):
        if not self._len < 32:
            raise PanicError("Can't convert this to a >32 bit number")
        return int(self._w[0]) & 0xFFFFFFFF

    def as_u64(self):
        if not self._len < 64:
            raise PanicError("Can't convert this to a >32 bit number")
        return int(self._w[0])

    # -- operators (binary_vector.rs:156-215; binary_matrix.rs:552-573) --
    def __eq__(self, other):
        return isinstance(other, BinVector) and self._len == other._len and bool(np.array_equal(self._w, other._w))

    def __hash__(self):
        return hash((self._len, self._w.tobytes()))

    def __add__(self, other):
        if self._len != other._len:
            raise PanicError("unequal length vectors")
        return BinVector(self._w ^ other._w, self._len)

    def __iadd__(self, other):
        if self._len != other._len:
            raise PanicError("unequal length vectors")
        self._w ^= other._w
        return self

    def __mul__(self, other):
        if isinstance(other, BinMatrix):  # v^T * A
            return (self.as_matrix() * other).as_vector()
        k = min(len(self._w), len(other._w))  # Vob::and works blockwise on the shorter operand
        x = self._w[:k] & other._w[:k]
        return sum(bin(int(v)).count("1") for v in x) % 2 == 1

    def to_json(self):
        """serde format of BinVector { vec: Vob } (binary_vector.rs:13-17): {"vec":{"len":N,"vec":[u64 words]}}."""
        return '{"vec":{"len":%d,"vec":[%s]}}' % (len(self), ",".join(str(int(w)) for w in self.get_storage()))

    @staticmethod
    def from_json(text):
        import json
        v = json.loads(text)["vec"]
        return BinVector(np.array(v["vec"], dtype=np.uint64), v["len"])

    def __repr__(self):
        return "BinVector(%s)" % "".join("1" if b else "0" for b in self.to_bools())


class BinMatrix:
    """Owns an mzd_t* (binary_matrix.rs:33-45); freed with mzd_free on drop."""

    __slots__ = ("_mzd",)

    def __init__(self, mzd):
        if not mzd:
            raise PanicError("Can't be NULL")
        self._mzd = mzd

    @property
    def mzd(self):
        """The mzd_t* for C-ABI calls.  Every access hands out a FRESH ctypes pointer that holds a reference to this
        owner, so `lib.mzd_mul(None, BinMatrix.from_words(...).mzd, ...)` keeps the temporary matrix alive until the call
        has returned (a bare pointer would outlive its owner: the temporary is collected -- mzd_free -- as soon as `.mzd`
        has been read, and the callee reads freed memory; round 2 hit exactly that in a test)."""
        if not self._mzd:
            return None
        p = ctypes.cast(self._mzd, _lib.MzdP)
        p._owner = self
        return p

    def __del__(self):
        mzd = getattr(self, "_mzd", None)
        self._mzd = None
        if mzd:
            try:
                _lib.lib().mzd_free(mzd)
            except Exception:
                pass

    # -- constructors --
    @staticmethod
    def zero(rows, cols):
        if rows == 0 or cols == 0:
            raise PanicError("Can't create a 0 matrix")
        return BinMatrix(_lib.lib().mzd_init(rows, cols))

    @staticmethod
    def new(rows):
        rowlen = len(rows[0])  # IndexError on an empty list == the reference's panic (binary_matrix.rs:109)
        return BinMatrix.from_slices([r.get_storage() for r in rows], rowlen)

    @staticmethod
    def from_slices(rows, rowlen):
        """binary_matrix.rs:124-167: rows of u64 words, tail masked."""
        if len(rows) == 0 or rowlen == 0:
            raise PanicError("Can't create a 0 matrix")
        w = _width(rowlen)
        arr = np.zeros((len(rows), w), dtype=np.uint64)
        for i, r in enumerate(rows):
            r = np.asarray(r, dtype=np.uint64)
            if len(r) * 64 < rowlen:
                raise PanicError("expected len %d bits but got only %d blocks" % (rowlen, len(r)))
            arr[i, :] = r[:w]
        return BinMatrix.from_words(arr, rowlen)

    @staticmethod
    def from_words(arr, ncols):
        """Bulk form of from_slices: (nrows, width) uint64 array."""
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        m = BinMatrix.zero(arr.shape[0], ncols)
        view = m._words_view()
        view[:, : arr.shape[1]] = arr
        if ncols % 64:
            view[:, arr.shape[1] - 1] &= np.uint64((1 << (ncols % 64)) - 1)
        return m

    @staticmethod
    def random(rows, columns):
        mzd = _lib.lib().mzd_init(rows, columns)
        _lib.lib().mzd_randomize(mzd)
        return BinMatrix(mzd)

    @staticmethod
    def from_mzd(mzd):
        return BinMatrix(mzd)

    @staticmethod
    def identity(rows):
        mzd = _lib.lib().mzd_init(rows, rows)
        _lib.lib().mzd_set_ui(mzd, 1)
        return BinMatrix(mzd)

    # -- raw access --
    def _words_view(self):
        """numpy view (nrows, rowstride) over the single host block."""
        z = self.mzd.contents
        n = z.nrows * z.rowstride
        if n == 0:
            return np.zeros((z.nrows, 0), dtype=np.uint64)
        base = ctypes.cast(z.rows[0], ctypes.POINTER(ctypes.c_uint64 * n)).contents
        return np.frombuffer(base, dtype=np.uint64).reshape(z.nrows, z.rowstride)

    def to_words(self):
        """(nrows, width) uint64 copy of the rows."""
        z = self.mzd.contents
        return self._words_view()[:, : z.width].copy()

    def nrows(self):
        return int(self.mzd.contents.nrows)

    def ncols(self):
        return int(self.mzd.contents.ncols)

    def bit(self, row, col):
        z = self.mzd.contents
        return bool((z.rows[row][col // 64] >> (col % 64)) & 1)

    def get_word(self, row, column):
        if not (row < self.nrows() and column < self.ncols()):
            raise PanicError("assertion failed")
        return int(self.mzd.contents.rows[row][column])

    def count_ones(self):
        if not (self.nrows() == 1 or self.ncols() == 1):
            raise PanicError("only works on single row or single column matrices")
        w = self.to_words()
        return int(np.unpackbits(w.view(np.uint8)).sum())

    # -- structure --
    def augmented(self, other):
        return BinMatrix(_lib.lib().mzd_concat(None, self.mzd, other.mzd))

    def stacked(self, other):
        return BinMatrix(_lib.lib().mzd_stack(None, self.mzd, other.mzd))

    def transposed(self):
        return BinMatrix(_lib.lib().mzd_transpose(None, self.mzd))

    def clone(self):
        return BinMatrix(_lib.lib().mzd_copy(None, self.mzd))

    def get_window(self, start_row, start_col, high_row, high_col):
        return BinMatrix(_lib.lib().mzd_submatrix(None, self.mzd, start_row, start_col, high_row, high_col))

    def set_window(self, start_row, start_col, other):
        _lib.lib().gf2_mzd_uncache(self.mzd)  # plain stores below: a device copy kept for this matrix would go stale
        z = self.mzd.contents
        for r in range(other.nrows()):
            for c in range(other.ncols()):
                rr, cc = start_row + r, start_col + c
                wv = z.rows[rr][cc // 64] & ~(1 << (cc % 64))
                z.rows[rr][cc // 64] = wv | (int(other.bit(r, c)) << (cc % 64))

    def as_vector(self):
        """binary_matrix.rs:332-361."""
        if self.nrows() != 1:
            if self.ncols() != 1:
                raise PanicError("needs to have only one column or row")
            return self.transposed().as_vector()
        return BinVector(self.to_words()[0], self.ncols())

    # -- elimination --
    def rank(self):
        """binary_matrix.rs:246-252: echelonizes a clone and throws it away."""
        return self.clone().echelonize()

    def echelonize(self, full=False):
        """In place; returns the rank (binary_matrix.rs:254-261, which passes full = false)."""
        return int(_lib.lib().mzd_echelonize(self.mzd, 1 if full else 0))

    def inverted(self):
        """binary_matrix.rs:263-268: a NULL result (singular matrix) panics with "Can't be NULL"."""
        ptr = _lib.lib().mzd_inv_m4ri(None, self.mzd, 0)
        if not ptr:
            raise PanicError("Can't be NULL")
        return BinMatrix(ptr)

    # -- products --
    def mul_slice(self, other):
        """A * v^T with v given as u64 words (binary_matrix.rs:416-431): always mzd_mul_naive."""
        other = np.asarray(other, dtype=np.uint64)
        if not self.ncols() <= len(other) * 64:
            raise PanicError("Mismatched sizes: (%dx%d) * (%dx1) (too big)" % (self.nrows(), self.ncols(), len(other) * 64))
        vt = BinMatrix.from_slices([other], self.ncols()).transposed()
        res = _lib.lib().mzd_mul_naive(None, self.mzd, vt.mzd)
        return BinMatrix.from_mzd(res)

    def __mul__(self, other):
        if isinstance(other, BinVector):  # A * v^T (binary_matrix.rs:528-542)
            return self.mul_slice(other.get_storage()).as_vector()
        ptr = _mul_impl(None, self.mzd, other.mzd)
        if not ptr:
            raise PanicError("Multiplication failed")
        return BinMatrix(ptr)

    def __eq__(self, other):
        return isinstance(other, BinMatrix) and _lib.lib().mzd_equal(self.mzd, other.mzd) == 1

    __hash__ = None

    def __add__(self, other):
        return BinMatrix(_lib.lib().mzd_add(None, self.mzd, other.mzd))

    def __iadd__(self, other):
        _lib.lib().mzd_add(self.mzd, self.mzd, other.mzd)
        return self

    def __repr__(self):
        return "BinMatrix(%dx%d)" % (self.nrows(), self.ncols())

    # -- operand cache of the drop-in entry points (include/m4ri_hip.h): products whose operand is this matrix skip its
    #    upload until uncache() / drop; the caller promises not to change the bits through host pointers meanwhile --
    def cache_on_device(self):
        _lib.check(_lib.lib().gf2_mzd_cache_on_device(self.mzd), "gf2_mzd_cache_on_device")
        return self

    def uncache(self):
        _lib.lib().gf2_mzd_uncache(self.mzd)

    # -- serde wire format (feature "serde", binary_matrix.rs:10-35): {"matrix":{"rows":[<Vob>, ...]}}, a Vob being
    #    {"len": bits, "vec": [u64 words, LSB-first]}; byte-for-byte what serde_json::to_string prints (test_serialize,
    #    binary_matrix.rs:693-699).  Upstream only serialises matrices; from_json is the obvious inverse. --
    def to_json(self):
        words = self.to_words()
        n = self.ncols()
        rows = ",".join('{"len":%d,"vec":[%s]}' % (n, ",".join(str(int(w)) for w in r)) for r in words)
        return '{"matrix":{"rows":[%s]}}' % rows

    @staticmethod
    def from_json(text):
        import json
        rows = json.loads(text)["matrix"]["rows"]
        if not rows:
            raise PanicError("Can't create a 0 matrix")
        n = rows[0]["len"]
        return BinMatrix.from_slices([np.array(r["vec"], dtype=np.uint64) for r in rows], n)


def solve_left(a, b):
    """Solve A X = B; b is modified in place and holds X afterwards; True if it succeeded (binary_matrix.rs:575-586).

    The reference consumes `a`; here it is left holding its reduced echelon form."""
    return _lib.lib().mzd_solve_left(a.mzd, b.mzd, 0, 1) == 0
