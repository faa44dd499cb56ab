"""Every tile-kernel variant that ships in libm4ri_hip.so (the defaults and the read-window variants kept for A/B runs) must
produce the oracle's bits: they are driven here through the internal launcher gf2k_m4rm (m4ri-rust_amd/csrc/gf2_kernels.h) on
device buffers and compared with oracle_mul_m4rm.  The development-only generations and the timing-only ablations (wrong
results by design) live in tools/libm4ri_hip_dev.so and must be refused by the shipped library."""
import ctypes

import numpy as np
import pytest

import gf2util as g

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(built):
    from m4ri_rust_amd import device
    device.require_gpu()
    return device


class MulArgs(ctypes.Structure):  # gf2k_mul_args
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
                ("lda", ctypes.c_longlong), ("ldb", ctypes.c_longlong), ("ldc", ctypes.c_longlong),
                ("sA", ctypes.c_longlong), ("sB", ctypes.c_longlong), ("sC", ctypes.c_longlong),
                ("m", ctypes.c_int), ("l", ctypes.c_int), ("n", ctypes.c_int),
                ("tiles_m", ctypes.c_int), ("tiles_n", ctypes.c_int), ("ksplit", ctypes.c_int), ("kwords", ctypes.c_int),
                ("batch", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("Bp", ctypes.c_void_p), ("sBp", ctypes.c_longlong), ("bp_nc", ctypes.c_int),
                ("P", ctypes.c_void_p), ("ldp", ctypes.c_longlong), ("sP", ctypes.c_longlong),
                ("a_packed", ctypes.c_int),
                ("n_rem", ctypes.c_int), ("nseg", ctypes.c_int), ("p_words", ctypes.c_longlong),
                ("n_full", ctypes.c_int), ("seg_slabs", ctypes.c_int), ("tile_slabs", ctypes.c_int)]


@pytest.mark.parametrize("m,l,n,batch", [(2048, 2048, 2048, 2), (4160, 1000, 1100, 1), (300, 520, 2300, 3)])
def test_tile_kernel_generations_agree(dev, m, l, n, batch):
    import torch
    from m4ri_rust_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.gf2k_m4rm.restype = ctypes.c_int
    lib.gf2k_m4rm.argtypes = [MulArgs, ctypes.c_int, ctypes.c_void_p]
    lib.gf2k_packA.restype = ctypes.c_int
    lib.gf2k_packA.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                               ctypes.c_void_p]
    wa, wb = (l + 63) // 64, (n + 63) // 64
    lda, ldb = (wa + 1) & ~1, (wb + 1) & ~1
    As = [dev.DMat.random(m, l, 10 + i) for i in range(batch)]
    Bs = [dev.DMat.random(l, n, 20 + i) for i in range(batch)]
    A = torch.zeros((batch, m, lda), dtype=torch.int64, device="cuda")
    B = torch.zeros((batch, l, ldb), dtype=torch.int64, device="cuda")
    for i in range(batch):
        A[i, :, :wa] = torch.from_numpy(As[i].to_words().view(np.int64)).cuda()
        B[i, :, :wb] = torch.from_numpy(Bs[i].to_words().view(np.int64)).cuda()
    expect = np.stack([g.o_mul_m4rm(As[i].to_words(), Bs[i].to_words(), m, l, n) for i in range(batch)])  # the CPU oracle
    mp = (m + 63) & ~63
    Apk = torch.zeros((batch, mp * lda), dtype=torch.int64, device="cuda")
    for i in range(batch):
        assert lib.gf2k_packA(Apk[i].data_ptr(), lda, A[i].data_ptr(), lda, m, wa, None) == 0

    lib.gf2k_m4rm_streamk_words.restype = ctypes.c_longlong
    lib.gf2k_m4rm_streamk_words.argtypes = [ctypes.c_int, ctypes.c_int]

    V89 = {9: (4096, 512), 10: (2048, 512), 11: (1024, 512), 12: (512, 512)}  # the v8 family: tile rows x columns

    def run(cfg, packed=False, ksplit=1, n_rem=0, nseg=0, accumulate=False):
        C = torch.full((batch, m, ldb), -1, dtype=torch.int64, device="cuda")
        if accumulate:
            C.copy_(C0)
        a = MulArgs()
        if n_rem or (cfg in V89 and ksplit > 1):  # stream-K split of the v8 / v9 family: scratch for the partial tiles
            rows, cols = V89[cfg]
            tiles = -(-m // rows) * -(-n // cols) * batch
            want = nseg or (tiles * ksplit if not n_rem else 256)
            words = lib.gf2k_m4rm_streamk_words(cfg, want + tiles + 8)
            scratch = torch.full((words,), -1, dtype=torch.int64, device="cuda")
            a.P, a.p_words, a.n_rem, a.nseg = scratch.data_ptr(), words, n_rem, nseg
        a.A = (Apk if packed else A).data_ptr()
        a.B, a.C = B.data_ptr(), C.data_ptr()
        a.lda, a.ldb, a.ldc = lda, ldb, ldb
        a.sA, a.sB, a.sC = (mp * lda if packed else m * lda), l * ldb, m * ldb
        a.m, a.l, a.n, a.batch, a.ksplit, a.accumulate, a.a_packed = m, l, n, batch, ksplit, int(accumulate), int(packed)
        assert lib.gf2k_m4rm(a, cfg, None) == 0, cfg
        torch.cuda.synchronize()
        return C.cpu().numpy().view(np.uint64)[:, :, :wb]

    for cfg, packed, ks in [(7, False, 1), (20, False, 1), (8, False, 1), (9, False, 1), (81, False, 1), (82, False, 1),
                            (8, True, 1), (9, True, 1), (7, False, 3), (20, False, 2), (8, False, 3), (9, True, 3),
                            (10, False, 1), (10, True, 1), (11, False, 1), (11, True, 1), (12, False, 1), (12, True, 1),
                            (10, True, 2), (11, False, 5), (12, True, 3)]:
        got = run(cfg, packed, ks)
        assert np.array_equal(got, expect), (cfg, packed, ks)
    # stream-K: the last n_rem tiles of the launch cut into segments (slices of one tile, or the tail of one tile plus the head
    # of the next), whole tiles before them; segment counts that do not divide the slabs; accumulate form
    C0 = torch.from_numpy(g.random_words(batch * m, ldb * 64, 77).view(np.int64).reshape(batch, m, ldb)).cuda()
    c0 = C0.cpu().numpy().view(np.uint64)[:, :, :wb]
    for cfg in sorted(V89):
        rows, cols = V89[cfg]
        tiles = -(-m // rows) * -(-n // cols) * batch
        for packed in (False, True):
            for n_rem, nseg in [(tiles, 0), (tiles, tiles + 1), (max(1, tiles // 2), 7), (1, 3), (tiles, 3 * tiles + 2)]:
                got = run(cfg, packed, 1, n_rem, nseg)
                assert np.array_equal(got, expect), (cfg, packed, n_rem, nseg)
        got = run(cfg, True, 1, tiles, 2 * tiles + 1, accumulate=True)
        assert np.array_equal(got, c0 ^ expect), (cfg, "accumulate")  # (the product's bits past n are zero)
    # first-generation kernels, v5, packed B and every timing-only ablation: not in the shipped library
    a = MulArgs()
    a.A, a.B, a.lda, a.ldb, a.ldc, a.m, a.l, a.n, a.batch, a.ksplit = A.data_ptr(), B.data_ptr(), lda, ldb, ldb, m, l, n, 1, 1
    a.C = torch.zeros((m, ldb), dtype=torch.int64, device="cuda").data_ptr()
    for cfg in (0, 1, 13, 17, 21, 22, 23, 80, 50, 40, 41, 42, 43, 44, 45, 49, 83, 84, 85, 86, 87, 88, 89, 90, 92, 93, 94, 95, 96):
        assert lib.gf2k_m4rm(a, cfg, None) != 0, cfg
    torch.cuda.synchronize()
