"""CPU tests of the drop-in boundary: libm4ri_hip.so loads, exports everything include/m4ri_hip.h
declares, keeps M4RI's mzd_t layout, and its host-side container functions behave as the
reference's tests expect (m4ri-sys/src/mzd.rs:374-461, binary_matrix.rs:588-774, binary_vector.rs:217-288).
No compute call is made (no GPU here): the multiply entry points must fail loudly instead."""
import ctypes
import glob
import os
import re
import subprocess

import numpy as np
import pytest

import gf2util as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg(built):
    import m4ri_rust_amd as p
    return p


def test_header_symbols_are_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "m4ri_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(_?mzd_\w+|m4ri_opt_k|gf2_\w+)\s*\(", hdr))
    declared -= {"gf2_dmat"}
    assert len(declared) > 40
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg._lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(pkg._lib.DECLARED_SYMBOLS), sorted(declared ^ set(pkg._lib.DECLARED_SYMBOLS))
    # development hooks (in-kernel clock stamps, tools/clock_probe.py) exist in tools/libm4ri_hip_dev.so only
    assert not any("dev_set" in e or "clock_stamps" in e for e in exported)


def test_header_compiles_as_c_and_struct_is_64_bytes(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "m4ri_hip.h"\n#include <stddef.h>\n'
                   "_Static_assert(sizeof(mzd_t)==64, \"size\");\n"
                   "_Static_assert(offsetof(mzd_t,nrows)==0 && offsetof(mzd_t,ncols)==4 && offsetof(mzd_t,width)==8 && "
                   "offsetof(mzd_t,rowstride)==12 && offsetof(mzd_t,offset_vector)==16 && offsetof(mzd_t,row_offset)==20 && "
                   "offsetof(mzd_t,flags)==24 && offsetof(mzd_t,blockrows_log)==25 && offsetof(mzd_t,high_bitmask)==40 && "
                   "offsetof(mzd_t,blocks)==48 && offsetof(mzd_t,rows)==56, \"offsets\");\n"
                   "_Static_assert(sizeof(mzd_block_t)==24, \"block\");\nint main(void){return 0;}\n")
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_mzd_init_like_reference(pkg):
    """mzd.rs:381-399 `init` and :401-422 (mzd_first_row / mzd_row address rule)."""
    L = pkg._lib.lib()
    for _ in range(20):
        m = L.mzd_init(10, 10)
        z = m.contents
        assert bool(z.blocks) and bool(z.rows)
        L.mzd_randomize(m)
        assert L.mzd_equal(m, m) == 1
        m2 = L.mzd_copy(None, m)
        assert L.mzd_equal(m2, m) == 1
        L.mzd_randomize(m2)
        assert L.mzd_equal(m2, m) == 0
        # mzd_first_row: blocks[0].begin + offset_vector == rows[0]; mzd_row: + row*rowstride
        base = ctypes.addressof(z.blocks[0].begin.contents)
        for r in range(10):
            assert ctypes.addressof(z.rows[r].contents) == base + 8 * (z.offset_vector + r * z.rowstride)
        assert (z.row_offset + 9) >> z.blockrows_log == 0 and not (z.flags & 0x20)
        L.mzd_free(m)
        L.mzd_free(m2)


@pytest.mark.parametrize("cols,width,stride", [(1, 1, 1), (64, 1, 1), (65, 2, 2), (129, 3, 4), (256, 4, 4), (300, 5, 6)])
def test_rowstride_rule(pkg, cols, width, stride):
    """mzd.rs:34-38: rowstride = width, +1 if odd and >= padding width."""
    L = pkg._lib.lib()
    m = L.mzd_init(3, cols)
    z = m.contents
    assert (z.width, z.rowstride) == (width, stride)
    assert z.high_bitmask == ((1 << (cols % 64)) - 1 if cols % 64 else 2 ** 64 - 1)
    L.mzd_free(m)


def test_read_write_bit_identity(pkg):
    """mzd.rs:425-460: set_ui(1) is the unit matrix under the LSB-first bit rule."""
    I = pkg.BinMatrix.identity(1000)
    bits = g.words_to_bits(I.to_words(), 1000)
    assert np.array_equal(bits, np.eye(1000, dtype=np.uint8))
    assert I.bit(999, 999) and not I.bit(999, 998)


def test_binmatrix_construction(pkg):
    """binary_matrix.rs:595-659 (new, identity), :722-755 (zero, set_window), :758 (random unequal)."""
    BM, BV = pkg.BinMatrix, pkg.BinVector
    BM.new([BV.from_bools([True, False, True]), BV.from_bools([True, True, True])])
    ident = BM.new([BV.from_bools([i == j for j in range(10)]) for i in range(10)])
    assert ident == BM.identity(10)
    z = BM.zero(10, 3)
    assert not any(z.bit(i, j) for i in range(10) for j in range(3))
    m1 = BM.zero(10, 10)
    m1.set_window(5, 5, BM.identity(5))
    for i in range(10):
        for j in range(10):
            assert m1.bit(i, j) == (i == j and i >= 5)
    assert BM.random(100, 100) != BM.random(100, 100)
    with pytest.raises(pkg.PanicError):
        BM.zero(0, 5)
    with pytest.raises(pkg.PanicError):
        BM.from_slices([], 5)
    # tail of from_slices is masked (binary_matrix.rs:151-155)
    m = BM.from_slices([[2 ** 64 - 1]], 10)
    assert m.to_words()[0, 0] == 1023


def test_as_vector_roundtrips(pkg):
    """binary_matrix.rs:702-719 and binary_vector.rs:262-277."""
    BM, BV = pkg.BinMatrix, pkg.BinVector
    for i in range(1, 25):
        m1 = BM.random(i, 1)
        vec = m1.as_vector()
        assert len(vec) == i and m1 == vec.as_column_matrix()
        m2 = BM.random(1, i)
        vec = m2.as_vector()
        assert len(vec) == i and m2 == vec.as_matrix()
    a = BV.random(10)
    assert a.as_matrix().ncols() == 10 and a.as_matrix().nrows() == 1 and a.as_matrix().as_vector() == a
    assert a.as_column_matrix().nrows() == 10 and a.as_column_matrix().as_vector() == a


def test_count_ones(pkg):
    """binary_matrix.rs:765-773."""
    rng = np.random.default_rng(1)
    for _ in range(100):
        size = int(rng.integers(1, 1000))
        v = pkg.BinVector.random(size, rng)
        assert v.count_ones() == v.as_matrix().count_ones() == v.as_column_matrix().count_ones()


def test_binvector_semantics(pkg):
    """binary_vector.rs:222-287."""
    BV = pkg.BinVector
    assert len(BV.from_elem(10, False)) == 10
    assert len(BV.from_bytes(bytes([0xFF]))) == 8
    b = BV.from_bytes(bytes([0b1000_0000]))
    assert b.get(0) is True and b.get(1) is False
    a, b = BV.from_elem(10, False), BV.from_elem(10, False)
    c = a + b
    assert len(c) == 10 and c == BV.from_elem(10, False)
    assert (BV.from_elem(10, True) * BV.from_elem(10, False)) is False
    assert (BV.from_elem(11, True) * BV.from_elem(11, True)) is True
    assert BV.from_elem(10, True).count_ones() == 10 and BV.from_bytes(bytes([0b1010_1000])).count_ones() == 3
    v = BV.from_function(4, lambda i: i % 2 == 0)
    assert v.get(0) is True and v.get(1) is False
    with pytest.raises(pkg.PanicError):
        BV.from_elem(3, True) + BV.from_elem(4, True)


def test_host_transpose_add_concat_stack_vs_oracle(pkg):
    BM = pkg.BinMatrix
    for (r, c) in [(1, 1), (64, 64), (65, 63), (100, 257), (1000, 1), (1, 1000), (300, 500)]:
        w = g.random_words(r, c, 3)
        m = BM.from_words(w, c)
        assert np.array_equal(m.transposed().to_words(), g.o_transpose(w, r, c))
        w2 = g.random_words(r, c, 4)
        assert np.array_equal((m + BM.from_words(w2, c)).to_words(), w ^ w2)
        mm = m.clone()
        mm += BM.from_words(w2, c)
        assert np.array_equal(mm.to_words(), w ^ w2)
    a, b = BM.from_words(g.random_words(5, 70, 1), 70), BM.from_words(g.random_words(5, 3, 2), 3)
    cat = g.words_to_bits(a.augmented(b).to_words(), 73)
    assert np.array_equal(cat[:, :70], g.words_to_bits(a.to_words(), 70)) and np.array_equal(cat[:, 70:], g.words_to_bits(b.to_words(), 3))
    s = a.stacked(BM.from_words(g.random_words(2, 70, 9), 70))
    assert s.nrows() == 7 and np.array_equal(s.to_words()[:5], a.to_words())
    win = a.get_window(1, 3, 4, 69)
    assert np.array_equal(g.words_to_bits(win.to_words(), 66), g.words_to_bits(a.to_words(), 70)[1:4, 3:69])


def test_window_shares_memory(pkg):
    L = pkg._lib.lib()
    big = pkg.BinMatrix.from_words(g.random_words(20, 200, 5), 200)
    W = L.mzd_init_window(big.mzd, 2, 64, 10, 150)
    z = W.contents
    assert (z.nrows, z.ncols, z.width) == (8, 86, 2) and z.flags & 0x4
    assert ctypes.addressof(z.rows[0].contents) == ctypes.addressof(big.mzd.contents.rows[2].contents) + 8
    L.mzd_free(W)
    assert big.bit(0, 0) in (True, False)  # parent still alive and readable


def test_mul_fails_loudly_without_gpu(pkg):
    from m4ri_rust_amd import device
    if device.device_count() > 0:
        pytest.skip("a GPU is present: the loud-failure path is for GPU-less hosts")
    with pytest.raises(pkg.PanicError, match="Multiplication failed"):
        pkg.BinMatrix.identity(8) * pkg.BinMatrix.identity(8)
    with pytest.raises(pkg._lib.HipError):
        device.DMat(4, 4)
    with pytest.raises(pkg._lib.HipError):
        device.require_gpu()


def test_elimination_fails_loudly_without_gpu(pkg):
    """mzd_echelonize / mzd_solve_left return a rank / a status, so a missing device cannot be reported through them: the
    process is stopped with a diagnostic (no CPU fallback); mzd_inv_m4ri returns NULL like the products."""
    from m4ri_rust_amd import device
    if device.device_count() > 0:
        pytest.skip("a GPU is present: the loud-failure path is for GPU-less hosts")
    import sys
    code = ("import sys; sys.path.insert(0, %r); import m4ri_rust_amd as p; "
            "print(p.BinMatrix.identity(8).rank())" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "mzd_echelonize failed" in r.stderr and "no CPU fallback" in r.stderr
    with pytest.raises(pkg.PanicError, match="Can't be NULL"):
        pkg.BinMatrix.identity(8).inverted()


def test_product_code_never_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "m4ri-rust_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gf2_oracle" not in txt and "libgf2oracle" not in txt and "gf2util" not in txt, f
    out = subprocess.check_output(["ldd", os.path.join(ROOT, "m4ri-rust_amd", "lib", "libm4ri_hip.so")], text=True)
    assert "oracle" not in out


def test_binary_file_roundtrip(pkg, tmp_path):
    L = pkg._lib.lib()
    for (r, c) in [(1, 1), (7, 64), (33, 130), (100, 1000)]:
        m = pkg.BinMatrix.from_words(g.random_words(r, c, 3), c)
        path = str(tmp_path / ("m_%d_%d.gf2" % (r, c))).encode()
        assert L.gf2_mzd_save(path, m.mzd) == 0
        back = L.gf2_mzd_load(path)
        assert back and pkg.BinMatrix(back) == m
        assert os.path.getsize(path) == 16 + r * g.width(c) * 8
    assert not L.gf2_mzd_load(str(tmp_path / "missing.gf2").encode())


def test_serde_wire_format(pkg):
    """The reference's own known answer (binary_matrix.rs:693-699, feature "serde")."""
    m = pkg.BinMatrix.identity(3)
    assert m.to_json() == '{"matrix":{"rows":[{"len":3,"vec":[1]},{"len":3,"vec":[2]},{"len":3,"vec":[4]}]}}'
    big = pkg.BinMatrix.from_words(g.random_words(5, 130, 9), 130)
    back = pkg.BinMatrix.from_json(big.to_json())
    assert back == big and back.ncols() == 130
    v = pkg.BinVector.from_bools([True, False, True])
    assert v.to_json() == '{"vec":{"len":3,"vec":[5]}}' and pkg.BinVector.from_json(v.to_json()) == v


def test_workspace_and_level_queries_without_gpu(pkg):
    """gf2_mul_workspace_bytes / gf2_strassen_levels are pure host arithmetic (no device needed): the Strassen arena of the
    metric's configuration, and the packed copy of A that a tall plain product stages (include/m4ri_hip.h section 2)."""
    lib = pkg._lib.lib()
    AUTO, M4RM = pkg._lib.ALGO_AUTO, pkg._lib.ALGO_M4RM
    assert lib.gf2_strassen_levels(65536, 65536, 65536, AUTO, 0) == 4
    assert lib.gf2_strassen_levels(65536, 65536, 65536, M4RM, 0) == 0
    assert lib.gf2_strassen_levels(1000, 1000, 1000, AUTO, 0) == 0
    arena = lib.gf2_mul_workspace_bytes(65536, 65536, 65536, AUTO, 0)
    # 4 levels = three fused + one virtual: 3 x 2401 leaves of 2 MiB + the seven 128 MiB level-1 products = 14.9 GiB,
    # plus the partial tiles of the leaf launch's last round (2401 x 8 = 19208 tiles = 75 rounds of 256 and 8 tiles, which
    # are cut into 256 stream-K segments with two 256 KiB slots each)
    plan = (ctypes.c_longlong * 9)()
    assert lib.gf2_tile_plan(4096, 4096, 4096, 2401, 1, plan) > 0
    # 2401 x 8 = 19208 tiles = 75 rounds of 256 and 8 tiles: the last product (its 8 tiles) goes into a launch of its own with
    # short tiles, every one of them cut into segments
    assert list(plan)[:4] == [9, 1, 0, 0] and plan[5] == 1 and plan[6] in (10, 11, 12) and plan[8] == 256 and plan[4] > 0
    assert arena == 8 * (3 * 2401 * 4096 * 64 + 7 * 32768 * 512) + plan[4]
    # plain M4RM at the same size: packed copy of A (same footprint as A); 16 x 128 tiles are 8 full rounds, nothing is cut
    assert lib.gf2_mul_workspace_bytes(65536, 65536, 65536, M4RM, 0) == 65536 * 1024 * 8
    # config 2 (4096^3, M4RM only): fewer than 256 tiles whatever the tile height, so every tile is cut into segments
    assert lib.gf2_tile_plan(4096, 4096, 4096, 1, 0, plan) > 0
    assert plan[0] in (10, 11, 12) and plan[2] == (4096 // {10: 2048, 11: 1024, 12: 512}[plan[0]]) * 8 and plan[3] == 256
    assert lib.gf2_mul_workspace_bytes(1000, 1000, 1000, M4RM, 0) <= (16 << 20)
    # a packed copy of A is staged only when the model says the contiguous loads repay the pass: rows padded to 64, even word count
    ws = lib.gf2_mul_workspace_bytes(2049, 70, 1024, M4RM, 0)
    t_unpacked, t_packed = lib.gf2_tile_plan(2049, 70, 1024, 1, 0, plan), lib.gf2_tile_plan(2049, 70, 1024, 1, 1, plan)
    assert ws in (0, 2112 * 2 * 8) or ws >= plan[4]
    assert (ws == 0) == (t_unpacked <= t_packed + 2.5e-6) or ws > 0


def test_product_opt_k_follows_graycode_rs(built):
    """The library's exported m4ri_opt_k (graycode.rs:44-56: 0.75 * log2(n), n = b for a multiplication (c != 0), min(a, b)
    for an inversion), checked against the documented rule written out independently here -- not against the oracle's copy."""
    import math
    from m4ri_rust_amd import _lib
    L = _lib.lib()

    def rule(a, b, c):
        n = b if c != 0 else min(a, b)
        k = int(0.75 * (1 + math.floor(math.log2(n)))) if n >= 1 else 1
        return max(1, min(16, k))

    table = [(1, 1, 1), (1, 1, 0), (2, 2, 2), (10, 10, 10), (100, 10, 100), (1000, 64, 1000), (1000, 1024, 1000), (1000, 1023, 1),
             (65536, 65536, 65536), (65536, 65535, 65536), (1 << 20, 256, 1), (256, 1 << 20, 0), (1 << 20, 256, 0), (3, 1 << 30, 7),
             (4096, 4096, 0), (5, 1 << 22, 1), (1 << 30, 1 << 30, 1)]
    for a, b, c in table:
        assert L.m4ri_opt_k(a, b, c) == rule(a, b, c), (a, b, c)
    assert L.m4ri_opt_k(1000, 1024, 1000) == 8 and L.m4ri_opt_k(65536, 65536, 1) == 12


# ---- size dispatch: the host routines for tiny products (gf2_small_host.cpp), no device needed to check them ----

def _bm(pkg_words, ncols):
    import m4ri_rust_amd as pkg
    return pkg.BinMatrix.from_words(pkg_words, ncols)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))),
                         ids=os.path.basename)
def test_host_small_product_on_golden_vectors(built, path):
    """The host routine of the size dispatch against every committed golden vector (the reference's bench shapes,
    m4ri-rust/benches/binary_matrix.rs:30-76, ragged sizes, LPN samples) and against the oracle; accumulate form too."""
    from m4ri_rust_amd import _lib
    L = _lib.lib()
    z = np.load(path)
    m, l, n = (int(x) for x in z["dims"])
    A, B = _bm(z["a"], l), _bm(z["b"], n)
    C = _bm(g.random_words(m, n, 5), n)
    assert L.gf2_mul_host_small(C.mzd, A.mzd, B.mzd, 0) == 0
    assert np.array_equal(C.to_words(), z["c"])
    assert np.array_equal(C.to_words(), g.o_mul_m4rm(z["a"], z["b"], m, l, n))
    c0 = g.random_words(m, n, 6)
    C = _bm(c0, n)
    assert L.gf2_mul_host_small(C.mzd, A.mzd, B.mzd, 1) == 0
    assert np.array_equal(C.to_words(), c0 ^ z["c"])


def test_host_small_product_shapes_windows_and_nt(built):
    import m4ri_rust_amd as pkg
    from m4ri_rust_amd import _lib
    L = _lib.lib()
    for (m, l, n) in [(1, 1, 1), (97, 8, 64), (96, 9, 65), (200, 130, 129), (300, 64, 1), (5, 700, 300), (128, 1, 200), (3, 0, 5)]:
        a, b = g.random_words(m, l, 7 + m), g.random_words(l, n, 8 + n)
        A, B = pkg.BinMatrix.from_words(a, l) if l else pkg.BinMatrix(L.mzd_init(m, 0)), \
            pkg.BinMatrix.from_words(b, n) if l else pkg.BinMatrix(L.mzd_init(0, n))
        C = pkg.BinMatrix.from_words(g.random_words(m, n, 9), n)
        assert L.gf2_mul_host_small(C.mzd, A.mzd, B.mzd, 0) == 0
        ref = g.o_mul_naive(a, b, m, l, n) if l else np.zeros((m, g.width(n)), dtype=np.uint64)
        assert np.array_equal(C.to_words(), ref), (m, l, n)
        if l:
            bt = g.o_transpose(b, l, n)
            Bt = pkg.BinMatrix.from_words(bt, l)
            C2 = pkg.BinMatrix.from_words(g.random_words(m, n, 10), n)
            assert L.gf2_mul_nt_host_small(C2.mzd, A.mzd, Bt.mzd, 0) == 0
            assert np.array_equal(C2.to_words(), ref), ("nt", m, l, n)
    # a window as destination keeps the parent's bits around it; windows as sources carry foreign bits past their width
    big = g.random_words(40, 300, 11)
    P = pkg.BinMatrix.from_words(big, 300)
    W = L.mzd_init_window(P.mzd, 5, 64, 25, 164)     # 20 x 100 window
    a, b = g.random_words(20, 70, 12), g.random_words(70, 100, 13)
    assert L.gf2_mul_host_small(W, pkg.BinMatrix.from_words(a, 70).mzd, pkg.BinMatrix.from_words(b, 100).mzd, 0) == 0
    got = g.words_to_bits(P.to_words(), 300)
    exp = g.words_to_bits(big, 300)
    exp[5:25, 64:164] = g.words_to_bits(g.o_mul_naive(a, b, 20, 70, 100), 100)
    assert np.array_equal(got, exp)
    srcA = L.mzd_init_window(P.mzd, 0, 0, 30, 36)      # 30 x 36: its only word also holds columns 36..63 of the parent
    srcB = L.mzd_init_window(P.mzd, 0, 128, 36, 228)   # 36 x 100: its last word holds columns 228..255 of the parent
    wa, wb = g.bits_to_words(exp[0:30, 0:36]), g.bits_to_words(exp[0:36, 128:228])
    C = pkg.BinMatrix.zero(30, 100)
    assert L.gf2_mul_host_small(C.mzd, srcA, srcB, 0) == 0
    assert np.array_equal(C.to_words(), g.o_mul_naive(wa, wb, 30, 36, 100))
    for wdw in (W, srcA, srcB):
        L.mzd_free(wdw)


@pytest.mark.parametrize("m,n,r", [(10, 10, 10), (64, 64, 64), (30, 200, 30), (200, 30, 12), (65, 129, 40), (1, 100, 1), (50, 50, 0)])
def test_host_small_echelon_form(built, m, n, r):
    import m4ri_rust_amd as pkg
    from m4ri_rust_amd import _lib
    L = _lib.lib()
    if r == 0:
        a = np.zeros((m, g.width(n)), dtype=np.uint64)
    elif r >= min(m, n):
        a = g.random_words(m, n, 20 + m)
    else:
        a = g.o_mul_naive(g.random_words(m, r, 21), g.random_words(r, n, 22), m, r, n)
    ref, orank, _ = g.o_echelonize(a, m, n, full=True)
    M = pkg.BinMatrix.from_words(a, n)
    assert L.gf2_echelonize_host_small(M.mzd, 1) == orank
    assert np.array_equal(M.to_words(), ref)
    M = pkg.BinMatrix.from_words(a, n)
    assert L.gf2_echelonize_host_small(M.mzd, 0) == orank
    again, rank2, _ = g.o_echelonize(M.to_words(), m, n, full=True)   # same row space: its reduced form is the unique one
    assert rank2 == orank and np.array_equal(again, ref)


def test_shape_plans_without_gpu(built):
    """The cost model's plan for shapes that do and do not divide: dividing shapes run as given, shapes a
    little short of a multiple are padded up, shapes a little above one are peeled down to a dividing core."""
    import ctypes
    from m4ri_rust_amd import _lib
    L = _lib.lib()

    def plan(m, l, n, algo=0, param=0):
        kind, dims = ctypes.c_int(-1), (ctypes.c_int * 3)()
        lv = L.gf2_mul_plan(m, l, n, algo, param, ctypes.byref(kind), dims)
        return lv, kind.value, tuple(dims)

    assert plan(65536, 65536, 65536) == (4, 0, (65536, 65536, 65536))
    assert plan(32768, 32768, 32768) == (3, 0, (32768, 32768, 32768))
    assert plan(8192, 65536, 65536)[0] == 1                       # two row tiles: ONE level (packed A leaves since round 4: 4.70 against 4.91 ms measured), not two
    lv, kind, dims = plan(60000, 60000, 60000)                        # does not divide: padded up or peeled down, by modelled time
    assert kind in (1, 2) and lv in (3, 4) and all(57344 <= d <= 61440 for d in dims)
    assert dims[0] % (64 << lv) == 0 and dims[1] % (128 << lv) == 0 and dims[2] % (128 << lv) == 0
    lv, kind, dims = plan(65600, 65600, 65600)
    assert kind == 2 and lv == 4 and dims == (65536, 65536, 65536)   # peeled: 64 rows / columns of border
    lv, kind, dims = plan(70000, 70000, 70000)
    assert kind == 2 and lv == 4 and dims[0] == 65536            # a 65536-row core (whole 4096-row leaf tiles), not 69632
    assert plan(12288, 12288, 12288)[0] in (0, 1)                     # one level at most (L0 0.312-0.362 ms, L1 0.310-0.338 measured: a tie)
    assert plan(4096, 4096, 4096) == (0, 0, (4096, 4096, 4096))       # BASELINE config 2: no level pays
    assert plan(1000, 1000, 1000) == (0, 0, (1000, 1000, 1000))
    assert plan(65536, 65536, 65536, algo=1) == (0, 0, (65536, 65536, 65536))   # mzd_mul_m4rm: no levels
    lv, kind, dims = plan(5000, 4000, 4100, algo=2, param=2)         # explicit level count on a shape that does not divide
    assert lv == 2 and kind in (1, 2) and all(d % 256 == 0 for d in dims)


def test_host_small_product_fuzz(built):
    """Seeded random shapes through the three strategies of the host routine (row XOR, byte tables, AND/parity) against the
    oracle's bit-level product; every shape of the size dispatch's range of work."""
    import m4ri_rust_amd as pkg
    from m4ri_rust_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(2024)
    for it in range(60):
        m = int(rng.integers(1, 400))
        l = int(rng.integers(1, 300))
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 100, 128, 200, 257]))
        a, b = g.random_words(m, l, 1000 + it), g.random_words(l, n, 2000 + it)
        if it % 5 == 0:
            a[:] = 0  # an all-zero operand
        A, B = pkg.BinMatrix.from_words(a, l), pkg.BinMatrix.from_words(b, n)
        c0 = g.random_words(m, n, 3000 + it)
        C = pkg.BinMatrix.from_words(c0, n)
        acc = it % 2
        assert L.gf2_mul_host_small(C.mzd, A.mzd, B.mzd, acc) == 0
        ref = g.o_mul_bits(a, b, m, l, n) if m * l * n < 200000 else g.o_mul_naive(a, b, m, l, n)
        assert np.array_equal(C.to_words(), ref ^ c0 if acc else ref), (m, l, n, acc)


def test_mzd_make_table(built):
    """brilliantrussian.rs:8-17: every XOR combination of k rows, Gray-code order (each row of T one row addition away from its
    predecessor), L maps the k-bit selector to the row of T; checked against sums formed bit by bit with numpy."""
    import m4ri_rust_amd as pkg
    from m4ri_rust_amd import _lib
    L = _lib.lib()
    for (rows, cols, r, c, k) in [(20, 200, 3, 0, 5), (9, 64, 1, 0, 8), (40, 300, 30, 128, 6), (4, 70, 0, 64, 1), (3, 10, 2, 0, 0)]:
        a = g.random_words(rows, cols, 50 + k)
        M = pkg.BinMatrix.from_words(a, cols)
        junk = g.random_words(1 << k, cols, 60 + k)
        T = pkg.BinMatrix.from_words(junk, cols)
        Larr = (ctypes.c_int * (1 << k))(*([-1] * (1 << k)))
        L.mzd_make_table(M.mzd, r, c, k, T.mzd, Larr)
        t = T.to_words()
        w0 = c // 64
        bits = g.words_to_bits(a, cols)
        seen = set()
        for v in range(1 << k):
            expect = np.zeros(cols, dtype=np.uint8)
            for j in range(k):
                if (v >> j) & 1:
                    expect ^= bits[r + j]
            row = Larr[v]
            assert 0 <= row < (1 << k) and row not in seen
            seen.add(row)
            got = g.words_to_bits(t[row:row + 1], cols)[0]
            assert np.array_equal(got[64 * w0:], expect[64 * w0:]), (rows, cols, r, c, k, v)
            assert np.array_equal(t[row, :w0], junk[row, :w0])       # words left of column c are not touched
        assert Larr[0] == 0
        for i in range(1, 1 << k):  # Gray order: consecutive rows differ by exactly one row of M
            d = t[i, w0:] ^ t[i - 1, w0:]
            assert any(np.array_equal(d, a[r + j, w0:]) for j in range(k))


def test_col_swap_and_row_clear_offset(pkg):
    """mzd_col_swap (mzd.rs:144) and mzd_row_clear_offset (mzd.rs:235-240) against plain bit arithmetic, including columns
    in the last (partial) word and a window whose parent keeps its bits."""
    L = pkg._lib.lib()
    rng = np.random.default_rng(5)
    for (m, n) in [(1, 1), (5, 64), (9, 65), (40, 130), (33, 200)]:
        a = g.random_words(m, n, 20 + n)
        bits = g.words_to_bits(a, n)
        M = pkg.BinMatrix.from_words(a, n)
        for _ in range(12):
            ca, cb = (int(x) for x in rng.integers(0, n, 2))
            L.mzd_col_swap(M.mzd, ca, cb)
            bits[:, [ca, cb]] = bits[:, [cb, ca]]
        assert np.array_equal(M.to_words(), g.bits_to_words(bits)), (m, n)
        for off in sorted({0, 1, n // 2, max(n - 1, 0), n, 63 % n, 64 % n}):
            r = int(rng.integers(0, m))
            L.mzd_row_clear_offset(M.mzd, r, off)
            bits[r, off:] = 0
            assert np.array_equal(M.to_words(), g.bits_to_words(bits)), (m, n, off)
    # window: clearing a row of the window must not touch the parent's bits right of it
    big = g.random_words(10, 200, 3)
    P = pkg.BinMatrix.from_words(big, 200)
    W = L.mzd_init_window(P.mzd, 2, 64, 8, 150)
    L.mzd_row_clear_offset(W, 1, 10)
    L.mzd_col_swap(W, 0, 85)
    pb = g.words_to_bits(big, 200)
    pb[3, 74:150] = 0
    pb[2:8, [64, 149]] = pb[2:8, [149, 64]]
    assert np.array_equal(P.to_words(), g.bits_to_words(pb))
    L.mzd_free(W)


def test_mzd_pointer_keeps_its_owner_alive(pkg):
    """`BinMatrix(...).mzd` as a call argument: the temporary owner must survive until the call has returned.  (Round 2's
    one unexplained abort, gpurun_out/t8.log, was this: `L.mzd_mul(None, BinMatrix.from_words(..).mzd, ..)` read a matrix
    whose owner had already been collected and freed.)  Every `.mzd` access now returns a pointer that references its owner."""
    import gc
    L = pkg._lib.lib()
    a = g.random_words(300, 300, 4)
    p = pkg.BinMatrix.from_words(a, 300).mzd  # the owner is a temporary
    junk = [pkg.BinMatrix.from_words(g.random_words(300, 300, 5 + i), 300) for i in range(8)]  # would reuse a freed block
    gc.collect()
    assert p.contents.nrows == 300 and L.mzd_equal(p, pkg.BinMatrix.from_words(a, 300).mzd) == 1
    assert L.mzd_equal(pkg.BinMatrix.from_words(a, 300).mzd, pkg.BinMatrix.from_words(a, 300).mzd) == 1
    del junk


def test_integration_lists_every_knob():
    """INTEGRATION.md section 6a is the generated table of tools/knob_table.py: every M4RI_HIP_* variable the sources read is listed,
    with the development-only ones (read only under GF2K_DEV_VARIANTS) in their own table."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("knob_table", os.path.join(ROOT, "tools", "knob_table.py"))
    kt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kt)
    k = kt.knobs()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("<!-- knob table: begin"):doc.index("<!-- knob table: end")]
    shipped, dev = sec.split("**Development builds only")
    for name, e in k.items():
        assert ("`%s`" % name) in (dev if e["dev"] else shipped), name
    assert sum(1 for e in k.values() if not e["dev"]) <= 22  # the shipped library's surface stays small (r5: + result side copy, the two elimination switches ADVICE r4 asked for, the census file, two test hooks)


def test_no_cross_lane_swaps_in_inline_asm():
    """v_permlane16_swap / v_permlane32_swap need two wait states after a VALU write of an operand (gfx950); the compiler pads
    only instructions it can see, so in the product sources they are builtins, never inline asm (round 4: the asm form in the
    transposition read a stale register -- wrong rows from 16 on in every 64 x 64 block)."""
    import re
    src = os.path.join(ROOT, "m4ri-rust_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".inc", ".cpp", ".h")):
            text = open(os.path.join(src, fn)).read()
            for m in re.finditer(r'asm\s*(?:volatile)?\s*\(\s*"([^"]*)"', text):
                assert "permlane" not in m.group(1), (fn, m.group(1))


def _hazard_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("asm_hazards", os.path.join(ROOT, "tools", "asm_hazards.py"))
    ah = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ah)
    return ah


def test_hazard_checker_sees_known_bad_and_known_good_sequences():
    """tools/asm_hazards.py on hand-written assembly: every rule fires on the unpadded pair and is quiet on the padded one -- among
    them the exact shape of round 4's transposition bug (two v_perm_b32 writing the operands of an inline-asm v_permlane16_swap)."""
    ah = _hazard_tool()

    def findings(body):
        text = "kern:\n" + body + "\n.Lfunc_end0:\n"
        (name, ins), = ah.parse_functions(text).items()
        return [f[1] for f in ah.check_function(name, ins)]

    A, E = "\t;;#ASMSTART\n", "\t;;#ASMEND\n"
    bad_swap = "\tv_perm_b32 v1, v3, v4, v5\n\tv_perm_b32 v2, v3, v4, v6\n" + A + "\tv_permlane16_swap_b32 v1, v2\n" + E
    assert findings(bad_swap) == ["VALU write -> v_permlane*_swap operand"] * 2
    assert not findings("\tv_perm_b32 v1, v3, v4, v5\n\tv_perm_b32 v2, v3, v4, v6\n" + A + "\ts_nop 1\n\tv_permlane16_swap_b32 v1, v2\n" + E)
    assert findings("\tv_perm_b32 v1, v3, v4, v5\n" + A + "\ts_nop 0\n\tv_permlane32_swap_b32 v1, v2\n" + E)  # one state is not two
    # both sides visible to the compiler: its own business (it pads what it sees)
    assert not findings("\tv_perm_b32 v1, v3, v4, v5\n\tv_permlane16_swap_b32 v1, v2\n")
    # M0 -> add-TID LDS write / LDS-DMA
    assert findings(A + "\ts_mov_b32 m0, s4\n\tds_write_addtid_b32 v1 offset:256\n" + E) == ["SALU writes M0 -> add-TID LDS / LDS-DMA / s_sendmsg"]
    assert not findings(A + "\ts_mov_b32 m0, s4\n\ts_nop 0\n\tds_write_addtid_b32 v1 offset:256\n" + E)
    assert findings(A + "\ts_mov_b32 m0, s4\n\tglobal_load_lds_dwordx4 v[2:3], off\n" + E)
    assert findings("\ts_mov_b32 m0, s4\n" + A + "\tbuffer_load_dword v1, s[8:11], 0 offen lds\n" + E)
    assert findings(A + "\ts_mov_b32 m0, s4\n\tv_writelane_b32 v1, s5, m0\n" + E) == ["SALU writes M0 -> v_readlane / v_writelane lane select in M0"]
    assert not findings(A + "\ts_mov_b32 m0, s4\n\ts_nop 0\n\tv_writelane_b32 v1, s5, m0\n\tv_writelane_b32 v2, s6, m0\n" + E)
    # DPP
    assert findings(A + "\tv_xor_b32 v1, v2, v3\n\tv_mov_b32_dpp v4, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n" + E) == ["VALU write -> DPP src0"]
    assert not findings(A + "\tv_xor_b32 v1, v2, v3\n\ts_nop 1\n\tv_mov_b32_dpp v4, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n" + E)
    assert findings(A + "\tv_cmpx_eq_u32_e32 v1, v2\n\tv_nop\n\tv_mov_b32_dpp v4, v7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" + E) == ["VALU writes EXEC -> DPP"]
    # readlane / readfirstlane
    assert findings("\tv_add_u32_e32 v1, v2, v3\n" + A + "\tv_readfirstlane_b32 s4, v1\n" + E) == ["VALU write -> v_readlane / v_readfirstlane source"]
    assert not findings("\tv_add_u32_e32 v1, v2, v3\n" + A + "\ts_nop 0\n\tv_readfirstlane_b32 s4, v1\n" + E)
    assert findings(A + "\tv_readfirstlane_b32 s5, v9\n\ts_nop 2\n\tv_readlane_b32 s4, v1, s5\n" + E) == ["VALU writes SGPR -> lane select of v_readlane / v_writelane"]
    # a VALU-written SGPR as a VMEM address operand
    assert findings(A + "\tv_readfirstlane_b32 s8, v1\n\ts_nop 3\n\tbuffer_load_dword v2, v3, s[8:11], 0 offen\n" + E) == ["VALU writes SGPR -> VMEM reads it"]
    assert not findings(A + "\tv_readfirstlane_b32 s8, v1\n\ts_nop 4\n\tbuffer_load_dword v2, v3, s[8:11], 0 offen\n" + E)
    assert findings(A + "\tv_cmp_eq_u32_e64 s[2:3], v1, v2\n\tglobal_load_dword v4, v5, s[2:3]\n" + E)
    # a wide store's data registers rewritten right behind it
    assert findings(A + "\tglobal_store_dwordx4 v[10:11], v[4:7], off\n" + E + "\tv_mov_b32_e32 v5, 0\n") == ["VMEM store of > 64 bits -> VALU rewrites its data"]
    assert not findings(A + "\tglobal_store_dwordx4 v[10:11], v[4:7], off\n\ts_nop 1\n" + E + "\tv_mov_b32_e32 v5, 0\n")
    assert not findings(A + "\tglobal_store_dwordx2 v[10:11], v[4:5], off\n" + E + "\tv_mov_b32_e32 v5, 0\n")  # 64 bits: no hazard


def test_no_unpadded_hazard_around_inline_asm_in_the_device_code():
    """The audit itself: csrc/*.hip compiled to gfx950 assembly with the Makefile's flags, every hazard pair with an
    instruction between ;;#ASMSTART and ;;#ASMEND on either side checked for its wait states.  ~45 s (hipcc -S of the two kernel files)."""
    ah = _hazard_tool()
    findings, stats = ah.audit()
    assert stats["asm_instructions"] > 10000 and stats["functions"] > 100, stats  # the audit saw the library, not an empty file
    assert stats["examined"].get("SALU writes M0 -> add-TID LDS / LDS-DMA / s_sendmsg", 0) > 1000, stats["examined"]
    assert not findings, "\n".join("%s %s line %d: %s (%d of %d states)\n   %s\n   %s" % f for f in findings[:20])
