// gf2_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for dense GF(2) products.
//
// Replaces, on the device, the M4RI routines reached by m4ri-rust's `*`:
//   mzd_mul_m4rm / mzd_addmul_m4rm   (/root/reference/m4ri-sys/src/brilliantrussian.rs:210-224)
//   mzd_mul / mzd_addmul             (/root/reference/m4ri-sys/src/strassen.rs:8-31)
//   mzd_mul_naive / _mzd_mul_naive   (/root/reference/m4ri-sys/src/mzd.rs:150-168)
//   _mzd_mul_va                      (/root/reference/m4ri-sys/src/mzd.rs:175-181)
// plus mzd_add / mzd_transpose / mzd_equal / mzd_randomize equivalents on device buffers.
//
// Data layout in HBM: row-major 64-bit words, bit j of a row = bit (j%64) of word j/64 (LSB first,
// mzd.rs:246-269), row stride `ld` words (even), excess bits of the last word zero.
//
// Design notes:
//  * M4RM tile kernels: one workgroup owns an R x 2048-column tile of C held entirely in VGPRs
//    (R = WAVES*RPW rows).  The inner dimension is consumed 8 bits at a time: the workgroup builds
//    the 256-entry Four-Russians table of 8 rows of B (256 entries x 256 B = 64 KiB, double buffered
//    in LDS, Gray-code order per half-wave), then every 16-lane group looks up the table row selected
//    by a byte of A with one conflict-free ds_read_b128 and XORs it into its accumulators.  The table
//    row is exactly one LDS bank row (64 banks x 4 B), so any mix of entries is conflict free.
//    The LDS byte address is formed by ONE v_perm_b32: {0, table-select, A byte, lane offset}.
//    Shipped generations: _v3 (instruction-count minimal; short operands), _v6 (two chunks per lookup step folded with
//    v_bitop3_b32, 2048 x 1024 tile, one row of A per lane), _v8 (four chunks per table generation, 512 columns, tile height 4096 /
//    2048 / 1024 / 512 chosen at launch, stream-K segments for the last round of a launch; the default nearly everywhere).  The first
//    generation, v5, v7 (v8's fixed-height predecessor with its ablation branches) and the tall-narrow experiment v9 live under
//    tools/ (development builds only).
//  * few-tile products cut the inner dimension: stream-K segments + gf2_streamk_reduce_kernel (v8), uniform slices + gf2_splitk_reduce_kernel (v3, v6).
//  * products with n <= 256, many rows and a short inner dimension (batches of matrix x vector products) have their own
//    Four-Russians kernels with tables over B in LDS: gf2_tallskinny6_kernel (n <= 64: 4-bit tables, small streaming workgroups),
//    gf2_tallskinny5_kernel (64 < n <= 256: 8-bit tables, every row read once), gf2_tallskinny4 / 3_kernel for 256 < l <= 1024;
//    n <= 8 uses an AND/popcount kernel (gf2_narrow_kernel) that streams A at HBM speed; up to 64 vectors against LONG rows
//    (l > 512: `&A * &v` on a big square A) take gf2_widevec_kernel, a wave per row with lanes along the row, or gf2_tallskinny7_kernel,
//    4-bit tables rebuilt per 512-bit slab with the inner dimension divided among workgroups (9-64 vectors).
//  * the Strassen passes fuse three levels (and a virtual fourth) per kernel in registers (gf2_strassen_split3 / merge3_kernel);
//    transpose, XOR, compare, fill, padding are HBM-streaming kernels with 16- or 8-byte accesses.  The elimination kernels live in gf2_elim.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>
#include "gf2_kernels.h"

typedef uint64_t u64;
typedef uint32_t u32;

// A/B switches of the launchers: read from the environment in development builds only (tools/libm4ri_hip_dev.so); the shipped
// library uses the default
#ifdef GF2K_DEV_VARIANTS
#define GF2K_DEV_ENV(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
// Which clock does the tile kernel hold?  (tools/clock_probe.py, development builds only: in the shipped kernel no stamp executes.)
// When set, every workgroup of gf2_m4rm_kernel_v8 stores {s_memrealtime, s_memtime} at its start and {s_memtime, s_memrealtime} at its
// end: 4 u64 per workgroup, the tall packed instantiation <8, *, 1> from entry 0 on, every other instantiation from entry 2^20 on.
// clock = d(s_memtime) / d(s_memrealtime) x 100 MHz inside one workgroup (s_memtime counts per XCD; MI355X_MICROARCH.md, DVFS item 6).
__device__ unsigned long long *gf2k_clock_stamps;
extern "C" hipError_t gf2k_dev_set_clock_stamps(unsigned long long *p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gf2k_clock_stamps), &p, sizeof(p));
}
#else
#define GF2K_DEV_ENV(name, dflt) (dflt)
#endif

// ---------------------------------------------------------------------------------------------
// M4RM tile kernel
// ---------------------------------------------------------------------------------------------

static constexpr int kTileWords = 32;          // 2048 columns per tile = one 256-byte LDS bank row
static constexpr int kTableBytes = 256 * 256;  // 2^8 entries x 256 B
static constexpr int kStageBytes = 64 * 256;   // 64 rows of B x 256 B
static constexpr int kLdsBytes = 2 * kTableBytes + kStageBytes;  // 144 KiB: also covers v4's rings (2 x 2 KiB + 2 x 4 KiB)

__device__ __forceinline__ uint4 xor4(uint4 a, uint4 b) {
  return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w);
}

template <int N>
struct Log2 {
  static constexpr int value = 1 + Log2<N / 2>::value;
};
template <>
struct Log2<1> {
  static constexpr int value = 0;
};

// ---------------------------------------------------------------------------------------------
// helpers for the v3 kernel
// ---------------------------------------------------------------------------------------------

template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  static_for_impl<N>(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4 lds_cu32x4;
typedef u32 u32x2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const u32x2v lds_cu32x2;
typedef __attribute__((address_space(3))) const u64 lds_cu64;
typedef __attribute__((address_space(3))) u64 lds_u64;
typedef __attribute__((address_space(3))) u32 lds_u32;

#ifndef GF2_LPNVEC_STRIDED
#define GF2_LPNVEC_STRIDED 0  // the table-free vector kernel keeps the blocked order: 2^20 x 256 x 1 cold 8.5 us blocked, 8.9 grid-stride (A/B builds: -DGF2_LPNVEC_STRIDED=1)
#endif
#include "gf2_lpn.inc"  // gf2_lpn8_kernel / gf2_lpn256_kernel: the l <= 256 tall-skinny products (BASELINE config 5)

// ---------------------------------------------------------------------------------------------
// M4RM tile kernel v3: instruction-count-minimal form ("issue model").
// Measured on gfx950 (tools/ubench): a SIMD issues at most one instruction per ~2.2 cycles whatever
// its type (s_waitcnt / s_mov / s_nop included), v_perm_b32 is half rate, ds_read_b128 costs 4 LDS
// cycles per CU and ds_write_addtid_b32 2.2.  One chunk (8 bits of the inner dimension) of a
// 1024 x 2048 tile is 256 table-row reads + 256 table-entry writes: ~1600 LDS cycles, and about as
// many issue cycles if every step is {wait, 4 xor, perm, read, xor, write}.  So:
//  * B rows come straight from global memory (buffer loads, two chunks ahead) into registers:
//    no LDS staging, no staging barriers;
//  * A words are re-fetched in place with buffer loads (row bound check by the buffer descriptor);
//  * the table of chunk i+1 is written (ds_write_addtid_b32, M0 set once per chunk) between the
//    lookups of chunk i; G lookups are in flight per wave (rolling window, hand-placed s_waitcnt).
// ---------------------------------------------------------------------------------------------

// outstanding-LDS-op count to wait for at step st: the reads issued after read(st) plus the table
// writes issued since (all in order behind it); one write per step when `wps` is 1.
constexpr int v3_wait_count(int st, int G, int STEPS, int wps) {
  int reads = 0, writes = 0;
  if (st >= G) {
    for (int j = st - G + 1; j <= st - 1; ++j) reads += (j + G < STEPS) ? 1 : 0;
    writes = G * wps;
  } else {
    for (int j = st + 1; j < G; ++j) reads += 1;
    for (int j = 0; j < st; ++j) reads += (j + G < STEPS) ? 1 : 0;
    writes = st * wps;
  }
  const int n = reads + writes;
  return n > 15 ? 15 : n;
}

__device__ unsigned long long gf2_dbg_sec[16];  // diagnostic builds only (DBG != 0): per-section cycle sums

// BPACK: B is given in the chunk-packed layout written by gf2_packB_kernel / the packed Strassen split: for column
// tile tn and chunk c one 2 KiB block at ((tn * bp_nc + c) * 2048) bytes, lane-major: lane d holds its dword d of the
// 8 rows in 32 consecutive bytes.  One wave then fetches a whole chunk with TWO 16-byte buffer loads per lane
// (instead of eight 4-byte ones), rows and columns outside the matrix are stored as zeros (no guards, one loop).
template <int WAVES, int RPW, int G, int DBG = 0, int PRIO = 0, int BPACK = 0>
__global__ __launch_bounds__(WAVES * 64) void gf2_m4rm_kernel_v3(const gf2k_mul_args p) {
  unsigned long long sec[4] = {0, 0, 0, 0};
  auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long {
    unsigned long long tt;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return tt;
  };
  constexpr int R = WAVES * RPW;
  constexpr int STEPS = RPW / 4;
  constexpr int EPW = 256 / WAVES;  // table entries built per wave
  constexpr int LOWB = Log2<EPW>::value;
  constexpr int WPS = (EPW + STEPS - 1) / STEPS;  // entry writes per lookup step
  static_assert(EPW * WAVES == 256 && G <= STEPS && EPW <= WPS * STEPS, "geometry");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (PRIO == 1) {  // the later-dispatched half loses VALU arbitration to the older half: lift it
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);
  } else if constexpr (PRIO == 2) {
    if (wave < WAVES / 2) __builtin_amdgcn_s_setprio(1);
  } else if constexpr (PRIO == 3) {
    if (wave & 1) __builtin_amdgcn_s_setprio(1);
  }
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = t % p.tiles_m;
  t /= p.tiles_m;
  const int ks = t % p.ksplit;  // slice of the inner dimension (split-K): fills the chip when tiles are few
  t /= p.ksplit;
  const int tn = t % p.tiles_n;
  const int bt = t / p.tiles_n;
  const u64 *__restrict__ A = p.A + (long long)bt * p.sA;
  const u64 *__restrict__ B = p.B + (long long)bt * p.sB;
  // split-K: either every slice stores its partial product (combined by gf2_splitk_reduce_kernel) or all slices meet
  // in C with atomic XOR
  const bool part = p.P != nullptr && p.ksplit > 1;
  u64 *__restrict__ C = part ? p.P + ((long long)bt * p.ksplit + ks) * p.sP : p.C + (long long)bt * p.sC;
  const long long ldc = part ? p.ldp : p.ldc;
  const bool accum = !part && p.accumulate;

  const int row0 = tm * R, w0 = tn * kTileWords;
  const int widthA = (p.l + 63) >> 6, widthB = (p.n + 63) >> 6;
  const u64 maskC = (p.n & 63) ? ((1ull << (p.n & 63)) - 1) : ~0ull;
  const int nw32 = (p.l + 31) >> 5;
  const int jbeg = ks * p.kwords;                   // first 32-bit word of this slice
  const int jend = min(nw32, jbeg + p.kwords);      // one past its last
  const int ibeg = jbeg * 4, nchunks = jend * 4;    // chunk range [ibeg, nchunks)
  (void)widthA;

  const int g = lane >> 4, qd = lane & 15;
  const u32 laneoff0 = (u32)qd * 16u, laneoff1 = laneoff0 | 0x10000u;
  const int myrow0 = row0 + wave * RPW + g;

  // accumulators as scalars (not 128-bit tuples): the allocator can place them anywhere
  u32 acc[STEPS][4];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0;

  // ---- A: 32-bit word j of the row of each step; rows past m read as zero (descriptor bound) ----
  const u32 ldaB = (u32)p.lda * 8u;
  const int rows_here = min(p.m - row0, R);
  const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc((void *)(A + (long long)row0 * p.lda), (short)0, (int)((u32)rows_here * ldaB), 0x00020000);
  const u32 voffA0 = (u32)(wave * RPW + g) * ldaB;
  const u32 tailA = (p.l & 31) ? ((1u << (p.l & 31)) - 1u) : 0xffffffffu;  // valid bits of the last 32-bit word
  u32 aw[STEPS];
  auto loadA_all = [&](int j) __attribute__((always_inline)) {
    u32 vo = voffA0;
    asm volatile("" : "+v"(vo));
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      aw[s] = __builtin_amdgcn_raw_buffer_load_b32(rsrcA, vo, j * 4, 0);
      vo += 4u * ldaB;
    }
  };

  // ---- B: dword `lane` of the 8 rows of a chunk, straight into registers ----
  const u32 ldbB = (u32)p.ldb * 8u;
  const int validB = min(256, (widthB - w0) * 8);            // bytes of this tile's columns that exist
  const u32 voffB = ((int)(lane * 4) < validB) ? (u32)lane * 4u : 0x80000000u;  // out of range -> reads 0
  auto rsrcB_for = [&](int c) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)(B + (long long)c * 8 * p.ldb + w0), (short)0, (int)(8u * ldbB), 0x00020000);
  };
  // packed operand: descriptor over this tile column's run of 2 KiB chunk blocks
  const __amdgpu_buffer_rsrc_t rsrcBp = __builtin_amdgcn_make_buffer_rsrc(
      (void *)(reinterpret_cast<const char *>(p.Bp) + ((long long)bt * p.sBp + (long long)tn * p.bp_nc) * 2048), (short)0,
      (int)((u32)p.bp_nc * 2048u), 0x00020000);
  const u32 voffBp = (u32)lane * 32u;
  auto loadBpacked = [&](int c, int half, u32 *dst4) __attribute__((always_inline)) {
    const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsrcBp, voffBp + (u32)half * 16u, c * 2048, 0);
    dst4[0] = x.x;
    dst4[1] = x.y;
    dst4[2] = x.z;
    dst4[3] = x.w;
  };
  // FAST: all 8 rows exist.  Otherwise rows past the inner dimension read as zero (wave-uniform test).
  auto loadBrow = [&](auto fast, const __amdgpu_buffer_rsrc_t &rs, int c, int b) __attribute__((always_inline)) -> u32 {
    if constexpr (decltype(fast)::value) {
      return __builtin_amdgcn_raw_buffer_load_b32(rs, voffB, b * (int)ldbB, 0);
    } else {
      u32 x = 0;
      if (c * 8 + b < p.l) x = __builtin_amdgcn_raw_buffer_load_b32(rs, voffB, b * (int)ldbB, 0);
      return x;
    }
  };

  // ---- table build: wave w owns entries [EPW*w, EPW*(w+1)), one ds_write_addtid_b32 per entry ----
  u32 cur32 = 0;
  auto build_begin = [&](const u32 (&rr)[8], u32 tbase) __attribute__((always_inline)) {
    cur32 = 0;
#pragma unroll
    for (int b = LOWB; b < 8; ++b)
      if ((wave >> (b - LOWB)) & 1) cur32 ^= rr[b];
    // LDS address of an addtid write = M0[15:0] + imm16 + 4*lane; the upper table lies above 64 KiB, so the
    // constant is split (kOff): M0 = (w+1)*EPW*256 - 4 <= 0xFFFC and imm <= 0xFF04 for every wave (verified on
    // gfx950: the sum is not wrapped at 16 bits)
    const u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;
    const u32 m0v = tbase + (u32)wave * (u32)(EPW * 256) - kOff;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");  // (one wait state before an LDS add-TID instruction reads M0: the compiler cannot see that the asm below does)
  };
  auto build_write = [&cur32](auto itag, auto ttag) __attribute__((always_inline)) {  // entry i (Gray order) <- cur32
    constexpr int i = decltype(itag)::value;
    constexpr u32 tbase = decltype(ttag)::value;
    constexpr unsigned e = (unsigned)i ^ ((unsigned)i >> 1);
    constexpr u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;
    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(cur32), "n"(kOff + e * 256u) : "memory");
  };
  auto build_entry = [&](auto itag, auto ttag, const u32 (&rr)[8]) __attribute__((always_inline)) {
    constexpr int i = decltype(itag)::value;
    if constexpr (i > 0) cur32 ^= rr[__builtin_ctz(i | 256)];
    build_write(itag, ttag);
  };

  // ---- prologue: rows of chunk 0 -> table 0, rows of chunk 1 -> rrB, A column 0 ----
  u32 rrA[8], rrB[8];
  {
    if constexpr (BPACK) {
      loadBpacked(ibeg, 0, rrA);
      loadBpacked(ibeg, 1, rrA + 4);
      loadBpacked(ibeg + 1, 0, rrB);
      loadBpacked(ibeg + 1, 1, rrB + 4);
    } else {
      const __amdgpu_buffer_rsrc_t rs0 = rsrcB_for(ibeg), rs1 = rsrcB_for(ibeg + 1);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        rrA[b] = loadBrow(std::false_type{}, rs0, ibeg, b);
        rrB[b] = loadBrow(std::false_type{}, rs1, ibeg + 1, b);
      }
    }
  }
  loadA_all(jbeg);
  if (jbeg == nw32 - 1) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) aw[s] &= tailA;
  }
  build_begin(rrA, 0u);
  static_for<EPW>([&](auto it) __attribute__((always_inline)) { build_entry(it, std::integral_constant<u32, 0u>{}, rrA); });
  __syncthreads();

  // one chunk: look up chunk i in table (C4&1); build chunk i+1 from `rows` into the other table; fetch the
  // rows of chunk i+2 into `next` and (last chunk of a 32-bit column) the next A column in place.  Every load
  // is issued between lookup steps, so the texture path works under the LDS traffic instead of after it.
  auto chunk_iter = [&](int i, auto c4tag, auto fast, const u32 (&rows)[8], u32 (&next)[8]) __attribute__((always_inline)) {
    constexpr int C4 = decltype(c4tag)::value;
    const u32 sel = 0x0c020000u | ((4u + (u32)C4) << 8);
    const u32 lo = (C4 & 1) ? laneoff1 : laneoff0;
    using tnext = std::integral_constant<u32, (C4 & 1) ? 0u : (u32)kTableBytes>;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if constexpr (DBG == 1) t0 = stamp();
    build_begin(rows, tnext::value);
    const __amdgpu_buffer_rsrc_t rsN = rsrcB_for(i + 2);
    const int jn = (i >> 2) + 1;
    u32 voA = voffA0;
    asm volatile("" : "+v"(voA));
    u32x4 tv[G];
    auto issue = [&](int st, u32x4 &dst) __attribute__((always_inline)) {
      u32 addr;
      asm volatile("v_perm_b32 %1, %2, %3, %4\n\tds_read_b128 %0, %1" : "=v"(dst), "=&v"(addr) : "v"(aw[st]), "v"(lo), "s"(sel) : "memory");
    };
#pragma unroll
    for (int k = 0; k < G; ++k) issue(k, tv[k]);
    static_for<STEPS>([&](auto stag) __attribute__((always_inline)) {
      constexpr int st = decltype(stag)::value;
      // one s_waitcnt per PAIR of steps (it covers the younger read of the pair): every instruction,
      // waits included, costs a SIMD issue slot
      if constexpr (st % 2 == 0 || st + 1 >= STEPS) {
        constexpr int sw = (st % 2 == 0 && st + 1 < STEPS) ? st + 1 : st;
        constexpr int N = v3_wait_count(sw, G, STEPS, WPS) - (sw - st) * (1 + WPS);
        static_assert(N >= 0, "window too small for paired waits");
        if constexpr (sw != st)
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(tv[st % G]), "+v"(tv[sw % G]) : "n"(N) : "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(tv[st % G]) : "n"(N) : "memory");
      }
      // the table-build XOR goes first: it is independent of the reads and fills the wait state hipcc
      // otherwise pads with s_nop between an inline-asm result and its first VALU use
      static_for<WPS>([&](auto ktag) __attribute__((always_inline)) {
        constexpr int e = st * WPS + decltype(ktag)::value;
        if constexpr (e < EPW && e > 0 && decltype(ktag)::value == 0 && DBG != 4 && DBG != 5) cur32 ^= rows[__builtin_ctz(e | 256)];
      });
      // XOR in place through asm: opaque to LLVM (no reassociation of the four chunks' XOR chains into a
      // tree that keeps every table row alive) and pinned to the accumulator's own register
      asm("v_xor_b32 %0, %1, %0" : "+v"(acc[st][0]) : "v"(tv[st % G].x));
      asm("v_xor_b32 %0, %1, %0" : "+v"(acc[st][1]) : "v"(tv[st % G].y));
      asm("v_xor_b32 %0, %1, %0" : "+v"(acc[st][2]) : "v"(tv[st % G].z));
      asm("v_xor_b32 %0, %1, %0" : "+v"(acc[st][3]) : "v"(tv[st % G].w));
      if constexpr (st + G < STEPS) issue(st + G, tv[st % G]);
      static_for<WPS>([&](auto ktag) __attribute__((always_inline)) {
        constexpr int k = decltype(ktag)::value;
        constexpr int e = st * WPS + k;
        if constexpr (e < EPW) {
          if constexpr (k > 0 && e > 0) cur32 ^= rows[__builtin_ctz(e | 256)];
          build_write(std::integral_constant<int, e>{}, tnext{});
        }
      });
      if constexpr (BPACK) {
        if constexpr (st < 2) loadBpacked(i + 2, st, next + 4 * st);
      } else if constexpr (st < 8 && DBG != 3 && DBG != 5 && DBG != 6) {
        next[st] = loadBrow(fast, rsN, i + 2, st);
      }
      if constexpr (C4 == 3 && DBG != 3 && DBG != 5 && DBG != 7) {  // aw[st] was consumed G steps ago: fetch the next column's word in place
        aw[st] = __builtin_amdgcn_raw_buffer_load_b32(rsrcA, voA, jn * 4, 0);
        voA += 4u * ldaB;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DBG == 1) t1 = stamp();
    if constexpr (C4 == 3) {
      if (jn == nw32 - 1 && tailA != 0xffffffffu) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) aw[s] &= tailA;
      }
    }
    if constexpr (DBG == 1) t2 = stamp();
    if constexpr (DBG != 2 && DBG != 5) __syncthreads();  // DBG >= 2: timing-only ablations (wrong results)
    if constexpr (DBG == 1) {
      t3 = stamp();
      sec[0] += t1 - t0;  // steps (lookups + build + interleaved loads)
      sec[1] += t2 - t1;  // tail mask
      sec[2] += t3 - t2;  // barrier wait
      sec[3] += 1;
    }
  };

  // two separate loops (not one loop with a branch): the register allocator otherwise spills the
  // accumulators around the merge point
  int i = ibeg;
#pragma unroll 1
  for (; i < nchunks && (BPACK || (i + 6) * 8 <= p.l); i += 4) {  // chunks i+2 .. i+5 lie wholly inside the inner dimension
    chunk_iter(i + 0, std::integral_constant<int, 0>{}, std::true_type{}, rrB, rrA);
    chunk_iter(i + 1, std::integral_constant<int, 1>{}, std::true_type{}, rrA, rrB);
    chunk_iter(i + 2, std::integral_constant<int, 2>{}, std::true_type{}, rrB, rrA);
    chunk_iter(i + 3, std::integral_constant<int, 3>{}, std::true_type{}, rrA, rrB);
  }
#pragma unroll 1
  for (; i < nchunks; i += 4) {  // ragged end
    chunk_iter(i + 0, std::integral_constant<int, 0>{}, std::false_type{}, rrB, rrA);
    chunk_iter(i + 1, std::integral_constant<int, 1>{}, std::false_type{}, rrA, rrB);
    chunk_iter(i + 2, std::integral_constant<int, 2>{}, std::false_type{}, rrB, rrA);
    chunk_iter(i + 3, std::integral_constant<int, 3>{}, std::false_type{}, rrA, rrB);
  }

  if constexpr (DBG == 1) {
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == WAVES - 1)) {
      const int o = wave ? 4 : 0;
      gf2_dbg_sec[o + 0] = sec[0];
      gf2_dbg_sec[o + 1] = sec[1];
      gf2_dbg_sec[o + 2] = sec[2];
      gf2_dbg_sec[o + 3] = sec[3];
    }
  }
  const int wc = w0 + 2 * qd;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const int row = myrow0 + 4 * s;
    if (row < p.m && wc < widthB) {
      u64 *dst = C + (long long)row * ldc + wc;
      u64 v0 = (u64)acc[s][0] | ((u64)acc[s][1] << 32);
      u64 v1 = (u64)acc[s][2] | ((u64)acc[s][3] << 32);
      if (wc == widthB - 1) v0 &= maskC;
      if (wc + 1 == widthB - 1) v1 &= maskC;
      if (p.ksplit > 1 && !part) {  // partial sums of the slices meet in C (zeroed by the launcher unless accumulating)
        if (v0) atomicXor(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v0);
        if (wc + 1 < widthB && v1) atomicXor(reinterpret_cast<unsigned long long *>(dst + 1), (unsigned long long)v1);
      } else if (wc + 1 < widthB) {
        if (accum) {
          const uint4 old = *reinterpret_cast<const uint4 *>(dst);
          v0 ^= (u64)old.x | ((u64)old.y << 32);
          v1 ^= (u64)old.z | ((u64)old.w << 32);
        }
        *reinterpret_cast<uint4 *>(dst) = make_uint4((u32)v0, (u32)(v0 >> 32), (u32)v1, (u32)(v1 >> 32));
      } else {
        if (accum) v0 ^= dst[0];
        dst[0] = v0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// M4RM tile kernel v5: two chunks per lookup step, combined with one three-input XOR (v_bitop3_b32).
// The v3 kernel spends five VALU instructions per 16-byte lookup (address v_perm_b32 + four v_xor_b32) and the VALU
// port is what it runs out of.  Here a lane looks its row up in the tables of TWO consecutive chunks and folds both
// 16-byte results into the accumulator with four v_bitop3_b32 (acc ^= x ^ y): three VALU instructions per lookup.
// Geometry: tile of 2048 rows x 1024 columns (8 waves x 256 rows; 8 lanes x 16 bytes per row, 8 rows per step).  A
// 256-byte LDS row holds entry e of the even chunk's table in its lower half and of the odd chunk's table in its upper
// half ("pair table", 64 KiB; two of them: one looked up, one being built), so a table row of a pair is still ONE
// ds_write_addtid_b32 (lanes 0..31 hold dwords of the even chunk's rows of B, lanes 32..63 of the odd chunk's).  The 16
// lanes the LDS serves together are two rows: the lanes of the odd row read the upper half (odd chunk) first and the
// lower half second, the lanes of the even row the other way round -- 64 different banks in both reads
// (tools/ubench: 16.0 cycles per ds_read_b128 and SIMD, same as a full 256-byte row; 32.0 without the swap).
// Rows of B past the inner dimension read as zero through the buffer descriptor's bound, so there is one loop only.
// ---------------------------------------------------------------------------------------------

// LDS operations that may still be outstanding when step `st_wait` needs the two reads of step `target` (ops complete in
// order).  Issue order: prologue 2 reads for each step < G; then per step s: [wait], two reads of step s+G (if any),
// one table write.
constexpr int v5_wait_count(int st_wait, int target, int G, int STEPS) {
  int after = 0;      // ops issued after the target's second read
  bool seen = false;
  for (int k = 0; k < G && k < STEPS; ++k) {
    if (seen) after += 2;
    if (k == target) seen = true;
  }
  for (int s = 0; s < st_wait; ++s) {
    if (s + G < STEPS) {
      if (seen) after += 2;
      if (s + G == target) seen = true;
    }
    if (seen) after += 1;
  }
  return after > 15 ? 15 : after;
}

static constexpr int kTileWords5 = 16;  // 1024 columns per tile

#ifdef GF2K_DEV_VARIANTS
#include "../../tools/gf2_kernels_legacy.inc"  // v1 and v5, kbench A/B runs only
#endif

// ---------------------------------------------------------------------------------------------
// M4RM tile kernel v6: the paired lookups of v5 with ONE ROW PER LANE.  In v5 the eight lanes that share a row all fetch
// the same word of A (32 buffer loads per wave and 32 bits of the inner dimension, a cache line of A touched 32 times);
// here lane L of a wave owns rows {64 r + L} of the wave's 256 rows and walks the eight 16-byte column pieces of its row
// itself, in the lane-dependent order piece = k ^ (L & 7) (static register indices: acc[r][k] simply holds piece
// k ^ (L & 7), sorted out when C is stored).  The 16 lanes the LDS serves together then read 8 different pieces x 2
// different halves of a pair-table row: conflict-free whatever their table entries are.  A is fetched as one 8-byte
// load per row and 64 bits of the inner dimension (4 loads per wave instead of 64), double buffered in registers.
// Same tile (2048 x 1024), pair tables, table build and wait accounting as v5.
// The inner dimension is walked in blocks of 128 bits; the launcher makes slices start at even 32-bit words.
// ---------------------------------------------------------------------------------------------
// APACK: A is given in the row-group-packed layout written by gf2_strassen_split2_kernel (side 2): m % 64 == 0
template <int WAVES, int G, int DBG = 0, int APACK = 0>
__global__ __launch_bounds__(WAVES * 64) void gf2_m4rm_kernel_v6(const gf2k_mul_args p) {
  constexpr int RPW = 256, RG = 4;  // rows per wave, row groups of 64
  constexpr int R = WAVES * RPW;
  constexpr int STEPS = 32;         // (row group, piece) pairs per wave and chunk pair
  constexpr int EPW = 256 / WAVES;
  constexpr int LOWB = Log2<EPW>::value;
  static_assert(EPW * WAVES == 256 && G <= STEPS && EPW <= STEPS, "geometry");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = t % p.tiles_m;
  t /= p.tiles_m;
  const int ks = t % p.ksplit;
  t /= p.ksplit;
  const int tn = t % p.tiles_n;
  const int bt = t / p.tiles_n;
  const u64 *__restrict__ A = p.A + (long long)bt * p.sA;
  const u64 *__restrict__ B = p.B + (long long)bt * p.sB;
  const bool part = p.P != nullptr && p.ksplit > 1;
  u64 *__restrict__ C = part ? p.P + ((long long)bt * p.ksplit + ks) * p.sP : p.C + (long long)bt * p.sC;
  const long long ldc = part ? p.ldp : p.ldc;
  const bool accum = !part && p.accumulate;

  const int row0 = tm * R, w0 = tn * kTileWords5;
  const int widthB = (p.n + 63) >> 6;
  const u64 maskC = (p.n & 63) ? ((1ull << (p.n & 63)) - 1) : ~0ull;
  const int nw32 = (p.l + 31) >> 5;
  const int jbeg = ks * p.kwords;  // even (launcher)
  const int jend = min(nw32, jbeg + p.kwords);

  const int l7 = lane & 7, h = (lane >> 3) & 1;
  // lo[k]: byte 0 = offset of the first read inside a pair-table row, byte 1 = of the second read (other half),
  // byte 2 = 0, byte 3 = 1 (pair table select)
  u32 lo[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const u32 sa = (u32)((k ^ l7) * 16 + h * 128);
    lo[k] = sa | ((sa ^ 128u) << 8) | 0x01000000u;
  }
  // selectors {0, table byte of lo, byte of the A word, offset byte of lo}; pp = pair inside the 32-bit word
  const u32 sel0a = 0x0c020000u | ((4u + (u32)h) << 8) | 0u, sel0b = 0x0c020000u | ((5u - (u32)h) << 8) | 1u;
  const u32 sel1a = 0x0c030000u | ((6u + (u32)h) << 8) | 0u, sel1b = 0x0c030000u | ((7u - (u32)h) << 8) | 1u;

  u32 acc[STEPS][4];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0;

  // ---- A: 64 bits of row (64 r + lane) per load; rows past m read as zero (descriptor bound) ----
  const u32 ldaB = (u32)p.lda * 8u;
  const int rows_here = min(p.m - row0, R);
  // packed: group g of 64 rows starts at byte g * 64 * ldaB, lane = row inside the group, 512 bytes per 64-bit column
  const __amdgpu_buffer_rsrc_t rsrcA =
      APACK ? __builtin_amdgcn_make_buffer_rsrc((void *)A, (short)0, (int)((u32)((p.m + 63) & ~63) * ldaB), 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc((void *)(A + (long long)row0 * p.lda), (short)0, (int)((u32)rows_here * ldaB), 0x00020000);
  const u32 voffA0 = APACK ? (u32)(row0 + wave * RPW) * ldaB + (u32)lane * 8u : (u32)(wave * RPW + lane) * ldaB;
  constexpr int kAColB = APACK ? 256 : 4;  // bytes per 32-bit column index in the scalar offset (packed: 512 per 64-bit slab)
  const u32 tailA = (p.l & 31) ? ((1u << (p.l & 31)) - 1u) : 0xffffffffu;
  const int jlast = (nw32 - 1) & ~1;  // last 64-bit slab that exists
  u32 awX[RG][2], awY[RG][2];
  auto loadA = [&](u32 (&dst)[RG][2], int j) __attribute__((always_inline)) {  // slab of words j, j+1 (j even)
    const int jl = min(j, jlast);
    u32 vo = voffA0;
    asm volatile("" : "+v"(vo));
#pragma unroll
    for (int r = 0; r < RG; ++r) {
      const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrcA, vo, jl * kAColB, 0);
      dst[r][0] = v.x;
      dst[r][1] = v.y;
      vo += 64u * ldaB;
    }
  };
  // words outside [jbeg, jend) contribute nothing; the last word of the inner dimension loses its padding bits
  auto maskA = [&](u32 (&dst)[RG][2], int j) __attribute__((always_inline)) {
    const u32 m0 = j >= jend ? 0u : (j == nw32 - 1 ? tailA : 0xffffffffu);
    const u32 m1 = j + 1 >= jend ? 0u : (j + 1 == nw32 - 1 ? tailA : 0xffffffffu);
    if ((m0 & m1) != 0xffffffffu) {
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        dst[r][0] &= m0;
        dst[r][1] &= m1;
      }
    }
  };

  // ---- B: lanes 0..31 hold dword `lane` of the 8 rows of the pair's even chunk, lanes 32..63 of its odd chunk ----
  const u32 ldbB = (u32)p.ldb * 8u;
  const int validB = min(128, (widthB - w0) * 8);
  const u32 voffB = ((int)((lane & 31) * 4) < validB) ? (u32)(lane & 31) * 4u + (u32)(lane >> 5) * 8u * ldbB : 0x80000000u;
  auto rsrcB_for = [&](int pr) __attribute__((always_inline)) {  // rows [16 pr, 16 pr + 16) of B, cut at l
    const int rows = min(16, p.l - 16 * pr);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(B + (long long)pr * 16 * p.ldb + w0), (short)0,
                                             rows > 0 ? (int)((u32)rows * ldbB) : 0, 0x00020000);
  };

  u32 cur32 = 0;
  auto build_begin = [&](const u32 (&rr)[8], u32 tbase) __attribute__((always_inline)) {
    cur32 = 0;
#pragma unroll
    for (int b = LOWB; b < 8; ++b)
      if ((wave >> (b - LOWB)) & 1) cur32 ^= rr[b];
    const u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;  // see gf2_m4rm_kernel_v3
    const u32 m0v = tbase + (u32)wave * (u32)(EPW * 256) - kOff;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");  // (one wait state before an LDS add-TID instruction reads M0: the compiler cannot see that the asm below does)
  };
  auto build_write = [&cur32](auto itag, auto ttag) __attribute__((always_inline)) {
    constexpr int i = decltype(itag)::value;
    constexpr u32 tbase = decltype(ttag)::value;
    constexpr unsigned e = (unsigned)i ^ ((unsigned)i >> 1);
    constexpr u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;
    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(cur32), "n"(kOff + e * 256u) : "memory");
  };

  // ---- prologue ----
  u32 rrA[8], rrB[8];
  {
    const __amdgpu_buffer_rsrc_t rs0 = rsrcB_for(2 * jbeg), rs1 = rsrcB_for(2 * jbeg + 1);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      rrA[b] = __builtin_amdgcn_raw_buffer_load_b32(rs0, voffB + (u32)b * ldbB, 0, 0);
      rrB[b] = __builtin_amdgcn_raw_buffer_load_b32(rs1, voffB + (u32)b * ldbB, 0, 0);
    }
  }
  loadA(awX, jbeg);
  maskA(awX, jbeg);
  build_begin(rrA, 0u);
  static_for<EPW>([&](auto it) __attribute__((always_inline)) {
    constexpr int i = decltype(it)::value;
    if constexpr (i > 0) cur32 ^= rrA[__builtin_ctz(i | 256)];
    build_write(it, std::integral_constant<u32, 0u>{});
  });
  __syncthreads();

  // one pair: look pair `pr` (word W of the slab in `aw`, pair PP of the word) up; build pair pr+1 from `rows` into the
  // other table; fetch the rows of pair pr+2 into `next`; the first pair of a slab also fetches the next slab of A
  auto pair_iter = [&](int pr, auto wtag, auto pptag, const u32 (&aw)[RG][2], u32 (&awn)[RG][2], const u32 (&rows)[8],
                       u32 (&next)[8]) __attribute__((always_inline)) {
    constexpr int PP = decltype(pptag)::value, W = decltype(wtag)::value;
    const u32 sela = PP ? sel1a : sel0a, selb = PP ? sel1b : sel0b;
    u32 lk[8];  // (copied: an asm operand of the nested lambda does not capture the enclosing function's array)
#pragma unroll
    for (int k = 0; k < 8; ++k) lk[k] = lo[k];
    using tnext = std::integral_constant<u32, PP ? 0u : (u32)kTableBytes>;
    build_begin(rows, tnext::value);
    const __amdgpu_buffer_rsrc_t rsN = rsrcB_for(pr + 2);
    const int jnext = min((pr >> 2) * 2 + 2, jlast);  // next slab, clamped: never past the end of a row
    u32 voA = voffA0;
    asm volatile("" : "+v"(voA));
    u32x4 ta[G], tb[G];
    auto issue = [&](int st, u32x4 &da, u32x4 &db) __attribute__((always_inline)) {
      u32 a0, a1;
      asm volatile("v_perm_b32 %2, %4, %5, %6\n\tds_read_b128 %0, %2\n\tv_perm_b32 %3, %4, %5, %7\n\tds_read_b128 %1, %3"
                   : "=&v"(da), "=&v"(db), "=&v"(a0), "=&v"(a1)
                   : "v"(aw[st >> 3][W]), "v"(lk[st & 7]), "v"(sela), "v"(selb)
                   : "memory");
    };
    if constexpr (DBG != 5) {
#pragma unroll
      for (int k = 0; k < G; ++k) issue(k, ta[k], tb[k]);
    } else {
#pragma unroll
      for (int k = 0; k < G; ++k) ta[k] = tb[k] = u32x4{aw[0][W], aw[1][W], aw[2][W], aw[3][W]};
    }
    static_for<STEPS>([&](auto stag) __attribute__((always_inline)) {
      constexpr int st = decltype(stag)::value;
      if constexpr (DBG == 5) {
      } else if constexpr (st % 2 == 0 || st + 1 >= STEPS) {
        constexpr int sw = (st % 2 == 0 && st + 1 < STEPS) ? st + 1 : st;
        constexpr int N = v5_wait_count(st, sw, G, STEPS);
        if constexpr (sw != st)
          asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(ta[st % G]), "+v"(tb[st % G]), "+v"(ta[sw % G]), "+v"(tb[sw % G]) : "n"(N) : "memory");
        else
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ta[st % G]), "+v"(tb[st % G]) : "n"(N) : "memory");
      }
      if constexpr (st < EPW && st > 0) cur32 ^= rows[__builtin_ctz(st | 256)];
      if constexpr (DBG != 5 && DBG != 6) {
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][0]) : "v"(ta[st % G].x), "v"(tb[st % G].x));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][1]) : "v"(ta[st % G].y), "v"(tb[st % G].y));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][2]) : "v"(ta[st % G].z), "v"(tb[st % G].z));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][3]) : "v"(ta[st % G].w), "v"(tb[st % G].w));
      } else if constexpr (DBG == 6) {
        asm volatile("" ::"v"(ta[st % G]), "v"(tb[st % G]));
      }
      if constexpr (st + G < STEPS && DBG != 5) issue(st + G, ta[st % G], tb[st % G]);
      if constexpr (st < EPW && DBG != 4) build_write(std::integral_constant<int, st>{}, tnext{});
      if constexpr (DBG == 3) {
      } else if constexpr (st < 8) {
        if constexpr (DBG != 8) next[st] = __builtin_amdgcn_raw_buffer_load_b32(rsN, voffB + (u32)st * ldbB, 0, 0);
      } else if constexpr (W == 0 && PP == 0 && st >= 8 && st < 8 + RG && DBG != 7) {  // next slab of A into the other buffer
        const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrcA, DBG == 9 ? (u32)(wave * RPW + (st - 8) * 64) * ldaB + (u32)lane * 8u : voA, jnext * kAColB, 0);
        awn[st - 8][0] = v.x;
        awn[st - 8][1] = v.y;
        voA += 64u * ldaB;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (DBG != 2) __syncthreads();
  };
  auto slab_iter = [&](int j, const u32 (&aw)[RG][2], u32 (&awn)[RG][2]) __attribute__((always_inline)) {
    pair_iter(2 * j + 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, aw, awn, rrB, rrA);
    pair_iter(2 * j + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, aw, awn, rrA, rrB);
    pair_iter(2 * j + 2, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, aw, awn, rrB, rrA);
    pair_iter(2 * j + 3, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, aw, awn, rrA, rrB);
    maskA(awn, j + 2);
  };

#pragma unroll 1
  for (int j = jbeg; j < jend; j += 4) {  // 128 bits of the inner dimension: slab X, then slab Y (all zero if past the slice)
    slab_iter(j, awX, awY);
    slab_iter(j + 2, awY, awX);
  }

  // ---- epilogue: a lane holds whole rows, so direct stores would be 64 scattered 16-byte pieces per instruction.  The
  // tables are dead now (the last pair ended with a barrier): every wave transposes its rows through its own 16 KiB of
  // LDS, 128 rows at a time, and stores them with 8 lanes per row (128 contiguous bytes). ----
  {
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
    const u32 wbase = (u32)wave * 16384u;
    const int prow = lane >> 3, pq = lane & 7;
    const int wc = w0 + 2 * pq;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int s = (2 * half + rr) * 8 + k;
          const u32 off = wbase + (u32)(rr * 64 + lane) * 128u + (u32)((k ^ l7) * 16);
          *reinterpret_cast<lds_u32x4 *>(off) = u32x4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const u32x4 v = *reinterpret_cast<lds_cu32x4 *>(wbase + (u32)i * 1024u + (u32)lane * 16u);
        const int row = row0 + wave * RPW + half * 128 + i * 8 + prow;
        if (row < p.m && wc < widthB) {
          u64 *dst = C + (long long)row * ldc + wc;
          u64 v0 = (u64)v.x | ((u64)v.y << 32);
          u64 v1 = (u64)v.z | ((u64)v.w << 32);
          if (wc == widthB - 1) v0 &= maskC;
          if (wc + 1 == widthB - 1) v1 &= maskC;
          if (p.ksplit > 1 && !part) {
            if (v0) atomicXor(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v0);
            if (wc + 1 < widthB && v1) atomicXor(reinterpret_cast<unsigned long long *>(dst + 1), (unsigned long long)v1);
          } else if (wc + 1 < widthB) {
            if (accum) {
              const uint4 old = *reinterpret_cast<const uint4 *>(dst);
              v0 ^= (u64)old.x | ((u64)old.y << 32);
              v1 ^= (u64)old.z | ((u64)old.w << 32);
            }
            *reinterpret_cast<uint4 *>(dst) = make_uint4((u32)v0, (u32)(v0 >> 32), (u32)v1, (u32)(v1 >> 32));
          } else {
            if (accum) v0 ^= dst[0];
            dst[0] = v0;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// M4RM tile kernel v8: FOUR chunks (one 32-bit word of the inner dimension) per table generation, tile of 512 RG rows x 512
// columns (RG = 8, 4, 2, 1 chosen at launch), stream-K scheduling of the launch's last round.
// Lookup scheme (unchanged from the round-1/2 kernel v7, tools/gf2_kernels_legacy_v7.inc): one v_perm_b32, one ds_read_b128
// and two v_bitop3_b32 per 16 bytes; a table row (one ds_write_addtid_b32: 4 chunks x 64 bytes) and a fetched row of B serve
// all rows of the tile.  A 256-byte LDS row holds entry e of the four chunks' tables side by side (16 slots of 16 bytes:
// slot = 4 chunk + piece).  Lane L owns rows {64 r + L}, r < RG, of its wave's 64 RG rows; in step (r, k) it reads, for
// c = 0..3, slot 4 (c ^ cl) + (k ^ l3) with cl = (L >> 2) & 3, l3 = L & 3: the 16 lanes served together hit 16 different
// slots, acc[r][k] holds piece k ^ l3 of the row's 64 bytes, and the four reads of a step belong to four different chunks
// (byte c ^ cl of the A word selects the entry).
//
// Tile height.  The table of a generation costs 256 ds_write_addtid_b32 per workgroup whatever the tile's height, so a
// tall tile amortises it best (4096 rows: 12 % of the LDS cycles) -- but a product with few rows, or few tiles, fills the
// chip only with shorter tiles: 4096^3 has 8 tiles of 4096 x 512 and 32 of 1024 x 512.
//
// Stream-K.  Every tile of a launch takes the same time, so T tiles on 256 CUs cost ceil(T / 256) rounds.  The launcher
// may therefore declare the last n_rem tiles "remainder": workgroups [0, n_full) compute whole tiles as before, workgroups
// n_full + s (s < nseg) compute SEGMENTS -- seg_slabs consecutive 64-bit slabs of the remainder tiles' inner dimensions laid
// end to end, so a segment is a slice of one tile or the tail of one tile plus the head of the next.  A segment stores each
// of its (at most two) partial tiles into its own slot of a scratch buffer, straight from the accumulator registers (slot
// layout: 16 bytes of step s of thread t at (s * 512 + t) * 16 -- no transposition, 1 KiB per wave instruction), and
// gf2_streamk_reduce_kernel XORs the slots of every remainder tile into C.  No workgroup ever waits for another one.
// With T < 256 everything is remainder: that is split-K with slices that need not divide the inner dimension evenly.
// ---------------------------------------------------------------------------------------------
constexpr int v8_wait_count(int st_wait, int target, int G, int STEPS, int RPS, int WPS) {
  int after = 0;  // LDS operations issued after the last read of step `target` when step `st_wait` waits for it
  bool seen = false;
  for (int k = 0; k < G && k < STEPS; ++k) {
    if (seen) after += RPS;
    if (k == target) seen = true;
  }
  for (int s = 0; s < st_wait; ++s) {
    if (s + G < STEPS) {
      if (seen) after += RPS;
      if (s + G == target) seen = true;
    }
    if (seen) after += WPS;
  }
  return after > 15 ? 15 : after;
}

static constexpr int kTileWords7 = 8;  // 512 columns per tile

// tile index -> (row tile, column tile, batch item); row tiles fastest: the row tiles that share a column panel of B are neighbours
struct v8_tile {
  int tm, tn, bt;
};
__device__ __forceinline__ v8_tile v8_decode(int t, const gf2k_mul_args &p) {
  v8_tile r;
  r.tm = t % p.tiles_m;
  t /= p.tiles_m;
  r.tn = t % p.tiles_n;
  r.bt = t / p.tiles_n;
  return r;
}

// NB: register buffers for rows of B (a quad's rows are requested NB - 1 quads before its table is built; the loop body covers NB
// quads); NA: 1 = a row group's slab of A is re-fetched in place after its last use, 2 = the next slab goes into a second
// buffer at the start of the current one.  Tall tiles (RG = 8) have no registers to spare and quads long enough to hide a
// memory latency (NB = 2, NA = 1); a quad of a short tile lasts well under a microsecond, so short tiles look further ahead.
template <int RG, int G, int APACK, int NB = (RG >= 8 ? 2 : 4), int NA = (RG >= 8 ? 1 : 2)>
__global__ __launch_bounds__(512) void gf2_m4rm_kernel_v8(const gf2k_mul_args p) {
  constexpr int WAVES = 8, RPW = 64 * RG, R = WAVES * RPW;
  static_assert(NB >= 2 && NB % 2 == 0 && (NA == 1 || (NA == 2 && NB % 4 == 0)), "buffers");
  constexpr int STEPS = 4 * RG;  // (row group, piece)
  constexpr int EPW = 256 / WAVES, LOWB = Log2<EPW>::value;
  constexpr int WPS = EPW / STEPS > 0 ? EPW / STEPS : 1;  // table entries written per step
  constexpr int BPS = STEPS >= 8 ? 1 : 8 / STEPS;         // rows of B fetched per step
  static_assert(WPS * STEPS == EPW || STEPS > EPW, "geometry");
  static_assert(G <= STEPS && G <= 3, "read window");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- what this workgroup computes: one whole tile, or a segment = (tile0, slabs [sl0, slN)) [+ (tile0 + 1, slabs [0, rest))] ----
  const int Q = p.tile_slabs;
  int tile0, sl0 = 0, slN = Q, rest = 0;
  long long slot0 = -1;  // first partial slot of a segment; -1: whole tile, stored to C
  {
    const int bid = blockIdx.x;
    if (bid < p.n_full) {  // XCD-aware order: blocks b, b + 8 share an XCD; each XCD gets a contiguous range of tiles
      const int nwg = p.n_full, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      tile0 = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    } else {
      const int s = bid - p.n_full;
      const long long g0 = (long long)s * p.seg_slabs, gtot = (long long)p.n_rem * Q;
      const long long g1 = min(g0 + (long long)p.seg_slabs, gtot);
      if (g1 <= g0) return;  // (whole workgroup)
      tile0 = p.n_full + (int)(g0 / Q);
      sl0 = (int)(g0 % Q);
      const int len = (int)(g1 - g0);
      slN = min(Q, sl0 + len);
      rest = len - (slN - sl0);
      slot0 = 2ll * s;
    }
  }
  const int nparts = rest > 0 ? 2 : 1;
#ifdef GF2K_DEV_VARIANTS
  unsigned long long *const clk_stamps = gf2k_clock_stamps;
  unsigned long long clk_rt0 = 0, clk_c0 = 0;
  if (clk_stamps) clk_rt0 = __builtin_amdgcn_s_memrealtime(), clk_c0 = __builtin_amdgcn_s_memtime();
  if ((p.kwords & 1) && wave >= 4) __builtin_amdgcn_s_setprio(1);  // static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if ((p.kwords & 2) && wave < 4) __builtin_amdgcn_s_setprio(1);
#endif

  const int l3 = lane & 3, cl = (lane >> 2) & 3;
  // lo[c >> 1][k]: byte 0 = slot offset of read c even, byte 1 = of read c odd, byte 2 = 0, byte 3 = 1 (table select)
  u32 lo[2][4];
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 s0 = (u32)((((2 * c2) ^ cl) * 4 + (k ^ l3)) * 16), s1 = (u32)((((2 * c2 + 1) ^ cl) * 4 + (k ^ l3)) * 16);
      lo[c2][k] = s0 | (s1 << 8) | 0x01000000u;
    }
  // sel[c] = {0, table byte of lo (2: table 0; the second table's selector is sel[c] + 0x10000, formed where it is used so that it
  // does not occupy four more registers), byte c ^ cl of the A word, slot byte c & 1 of lo}
  u32 sel[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) sel[c] = 0x0c000000u | (2u << 16) | ((4u + (u32)(c ^ cl)) << 8) | (u32)(c & 1);

  const int widthB = (p.n + 63) >> 6;
  const u64 maskC = (p.n & 63) ? ((1ull << (p.n & 63)) - 1) : ~0ull;
  const int nw32 = (p.l + 31) >> 5;
  const u32 ldaB = (u32)p.lda * 8u, ldbB = (u32)p.ldb * 8u;
  constexpr int kAColB = APACK ? 256 : 4;
  const u32 tailA = (p.l & 31) ? ((1u << (p.l & 31)) - 1u) : 0xffffffffu;
  const int jlast = (nw32 - 1) & ~1;

#pragma unroll 1
  for (int part = 0; part < nparts; ++part) {
    const v8_tile tl = v8_decode(tile0 + part, p);
    const int jbeg = 2 * (part ? 0 : sl0);
    const int jend = min(nw32, 2 * (part ? rest : slN));
    const u64 *__restrict__ A = p.A + (long long)tl.bt * p.sA;
    const u64 *__restrict__ B = p.B + (long long)tl.bt * p.sB;
    const int row0 = tl.tm * R, w0 = tl.tn * kTileWords7;

    u32 acc[STEPS][4];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc[s][0] = acc[s][1] = acc[s][2] = acc[s][3] = 0;

    // ---- A: 64 bits of row (64 r + lane) per load, reloaded in place; rows past m read as zero (descriptor bound) ----
    const int rows_here = min(p.m - row0, R);
    const __amdgpu_buffer_rsrc_t rsrcA =
        APACK ? __builtin_amdgcn_make_buffer_rsrc((void *)A, (short)0, (int)((u32)((p.m + 63) & ~63) * ldaB), 0x00020000)
              : __builtin_amdgcn_make_buffer_rsrc((void *)(A + (long long)row0 * p.lda), (short)0, (int)((u32)rows_here * ldaB), 0x00020000);
    const u32 voffA0 = APACK ? (u32)(row0 + wave * RPW) * ldaB + (u32)lane * 8u : (u32)(wave * RPW + lane) * ldaB;
    u32 aw[NA][RG][2];
    auto maskA = [&](u32 (&dst)[RG][2], int j) __attribute__((always_inline)) {  // slab of words j, j+1 (loaded into dst)
      const u32 m0 = j >= jend ? 0u : (j == nw32 - 1 ? tailA : 0xffffffffu);
      const u32 m1 = j + 1 >= jend ? 0u : (j + 1 == nw32 - 1 ? tailA : 0xffffffffu);
      if ((m0 & m1) != 0xffffffffu) {
#pragma unroll
        for (int r = 0; r < RG; ++r) {
          dst[r][0] &= m0;
          dst[r][1] &= m1;
        }
      }
    };

    // ---- B: lane L holds dword L & 15 of the 8 rows of chunk L >> 4 of the quad ----
    const int validB = min(64, (widthB - w0) * 8);
    const u32 voffB = ((int)((lane & 15) * 4) < validB) ? (u32)(lane & 15) * 4u + (u32)(lane >> 4) * 8u * ldbB : 0x80000000u;
    auto rsrcB_for = [&](int q) __attribute__((always_inline)) {  // rows [32 q, 32 q + 32) of B, cut at l
      const int rows = min(32, p.l - 32 * q);
      return __builtin_amdgcn_make_buffer_rsrc((void *)(B + (long long)q * 32 * p.ldb + w0), (short)0,
                                               rows > 0 ? (int)((u32)rows * ldbB) : 0, 0x00020000);
    };

    u32 cur32 = 0;
    auto build_begin = [&](const u32 (&rr)[8], u32 tbase) __attribute__((always_inline)) {
      cur32 = 0;
#pragma unroll
      for (int b = LOWB; b < 8; ++b)
        if ((wave >> (b - LOWB)) & 1) cur32 ^= rr[b];
      const u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;  // see gf2_m4rm_kernel_v3
      const u32 m0v = tbase + (u32)wave * (u32)(EPW * 256) - kOff;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");  // (one wait state before an LDS add-TID instruction reads M0: the compiler cannot see that the asm below does)
    };
    auto build_write = [&cur32](auto itag, auto ttag) __attribute__((always_inline)) {
      constexpr int i = decltype(itag)::value;
      constexpr u32 tbase = decltype(ttag)::value;
      constexpr unsigned e = (unsigned)i ^ ((unsigned)i >> 1);
      constexpr u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;
      asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(cur32), "n"(kOff + e * 256u) : "memory");
    };

    // ---- prologue: rows of B for the first NB quads, first slab of A, first table ----
    u32 rr[NB][8];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const __amdgpu_buffer_rsrc_t rs = rsrcB_for(jbeg + u);
#pragma unroll
      for (int b = 0; b < 8; ++b) rr[u][b] = __builtin_amdgcn_raw_buffer_load_b32(rs, voffB + (u32)b * ldbB, 0, 0);
    }
    {
      const int jl = min(jbeg, jlast);
      u32 vo = voffA0;
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int r = 0; r < RG; ++r) {
        const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrcA, vo, jl * kAColB, 0);
        aw[0][r][0] = v.x;
        aw[0][r][1] = v.y;
        vo += 64u * ldaB;
      }
    }
    maskA(aw[0], jbeg);
    build_begin(rr[0], 0u);
    static_for<EPW>([&](auto it) __attribute__((always_inline)) {
      constexpr int i = decltype(it)::value;
      if constexpr (i > 0) cur32 ^= rr[0][__builtin_ctz(i | 256)];
      build_write(it, std::integral_constant<u32, 0u>{});
    });
    __syncthreads();

    // one quad: look word W of the slab (quad q) up in table W; build quad q+1 from `rows` into the other table; fetch the
    // rows of quad q+NB into `next`; A: (NA == 1) the second quad of a slab reloads each row group's slab in place after its
    // last use, (NA == 2) the first quad of a slab fetches the next slab into `awn`
    auto quad_iter = [&](int q, auto wtag, u32 (&awp)[RG][2], u32 (&awn)[RG][2], const u32 (&rows)[8], u32 (&next)[8]) __attribute__((always_inline)) {
      constexpr int W = decltype(wtag)::value;
      u32 lk[2][4], sk[4];  // (copied: an asm operand of the nested lambda does not capture the enclosing function's arrays)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int k = 0; k < 4; ++k) lk[c2][k] = lo[c2][k];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (W == 0) sk[c] = sel[c];
        else asm volatile("v_add_u32 %0, %1, %2" : "=v"(sk[c]) : "v"(sel[c]), "s"(0x10000u));
      }
      using tnext = std::integral_constant<u32, W ? 0u : (u32)kTableBytes>;
      build_begin(rows, tnext::value);
      const __amdgpu_buffer_rsrc_t rsN = rsrcB_for(q + NB);
      const int jnext = min((q & ~1) + 2, jlast);  // next slab, clamped: never past the end of a row
      u32 voA = voffA0;
      asm volatile("" : "+v"(voA));
      u32x4 t0[G], t1[G], t2[G], t3[G];
      auto issue = [&](int st, u32x4 &d0, u32x4 &d1, u32x4 &d2, u32x4 &d3) __attribute__((always_inline)) {
        u32 a0, a1;
        asm volatile("v_perm_b32 %4, %6, %7, %9\n\tds_read_b128 %0, %4\n\tv_perm_b32 %5, %6, %7, %10\n\tds_read_b128 %1, %5\n\t"
                     "v_perm_b32 %4, %6, %8, %11\n\tds_read_b128 %2, %4\n\tv_perm_b32 %5, %6, %8, %12\n\tds_read_b128 %3, %5"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(a0), "=&v"(a1)
                     : "v"(awp[st >> 2][W]), "v"(lk[0][st & 3]), "v"(lk[1][st & 3]), "v"(sk[0]), "v"(sk[1]), "v"(sk[2]), "v"(sk[3])
                     : "memory");
      };
#pragma unroll
      for (int k = 0; k < G; ++k) issue(k, t0[k], t1[k], t2[k], t3[k]);
      static_for<STEPS>([&](auto stag) __attribute__((always_inline)) {
        constexpr int st = decltype(stag)::value;
        constexpr int N = v8_wait_count(st, st, G, STEPS, 4, WPS);
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(t0[st % G]), "+v"(t1[st % G]), "+v"(t2[st % G]), "+v"(t3[st % G]) : "n"(N) : "memory");
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][0]) : "v"(t0[st % G].x), "v"(t1[st % G].x));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][1]) : "v"(t0[st % G].y), "v"(t1[st % G].y));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][2]) : "v"(t0[st % G].z), "v"(t1[st % G].z));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][3]) : "v"(t0[st % G].w), "v"(t1[st % G].w));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][0]) : "v"(t2[st % G].x), "v"(t3[st % G].x));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][1]) : "v"(t2[st % G].y), "v"(t3[st % G].y));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][2]) : "v"(t2[st % G].z), "v"(t3[st % G].z));
        asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[st][3]) : "v"(t2[st % G].w), "v"(t3[st % G].w));
        if constexpr (st + G < STEPS) issue(st + G, t0[st % G], t1[st % G], t2[st % G], t3[st % G]);
        static_for<WPS>([&](auto wt) __attribute__((always_inline)) {  // entries st * WPS .. of the next table (Gray order)
          constexpr int i = st * WPS + decltype(wt)::value;
          if constexpr (i < EPW) {
            if constexpr (i > 0) cur32 ^= rows[__builtin_ctz(i | 256)];
            build_write(std::integral_constant<int, i>{}, tnext{});
          }
        });
#pragma unroll
        for (int b = st * BPS; b < (st + 1) * BPS && b < 8; ++b)
          next[b] = __builtin_amdgcn_raw_buffer_load_b32(rsN, voffB + (u32)b * ldbB, 0, 0);
        if constexpr (NA == 1) {
          // the reads of step st + G (issued above) are the last users of row group (st + G) >> 2 when (st + G) & 3 == 3
          if constexpr (W == 1 && ((st + G) & 3) == 3 && st + G < STEPS) {
            constexpr int r = (st + G) >> 2;
            const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrcA, voA + (u32)r * 64u * ldaB, jnext * kAColB, 0);
            awp[r][0] = v.x;
            awp[r][1] = v.y;
          }
        } else if constexpr (W == 0 && (st & 3) == 0) {  // one row group every fourth step (awn's last readers were issued in the previous quad)
          constexpr int r = st >> 2;
          const u32x2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrcA, voA + (u32)r * 64u * ldaB, jnext * kAColB, 0);
          awn[r][0] = v.x;
          awn[r][1] = v.y;
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (W == 1) maskA(NA == 1 ? awp : awn, (q & ~1) + 2);
      __syncthreads();
    };

#pragma unroll 1
    for (int j = jbeg; j < jend; j += NB) {
      bool done = false;
      static_for<NB / 2>([&](auto sl) __attribute__((always_inline)) {  // NB / 2 slabs; a short slice leaves after any of them
        constexpr int u = 2 * decltype(sl)::value, ab = (NA == 2) ? (decltype(sl)::value & 1) : 0, an = (NA == 2) ? (ab ^ 1) : 0;
        if (!done) {
          quad_iter(j + u, std::integral_constant<int, 0>{}, aw[ab], aw[an], rr[(u + 1) % NB], rr[u]);
          quad_iter(j + u + 1, std::integral_constant<int, 1>{}, aw[ab], aw[an], rr[(u + 2) % NB], rr[u + 1]);
          if (j + u + 2 >= jend) done = true;
        }
      });
    }

    if (slot0 >= 0) {
      // ---- segment: the partial tile goes to its slot as it lies in the registers (1 KiB per wave instruction) ----
      u64 *__restrict__ S = p.P + (slot0 + part) * p.sP + (long long)tid * 2;
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
        *reinterpret_cast<uint4 *>(S + (long long)s * 1024) = make_uint4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
      // (the last quad ended with a barrier: the tables are free for the second part's prologue)
    } else {
      // ---- whole tile: transpose through LDS (tables are dead), up to 128 rows x 64 bytes per wave at a time, 4 lanes per row ----
      u64 *__restrict__ C = p.C + (long long)tl.bt * p.sC;
      const long long ldc = p.ldc;
      const bool accum = p.accumulate != 0;
      typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
      constexpr int GPP = RG >= 2 ? 2 : 1;  // row groups per pass
      const u32 wbase = (u32)wave * 8192u;
      const int prow = lane >> 2, pq = lane & 3;
      const int wc = w0 + 2 * pq;
#pragma unroll
      for (int pass = 0; pass < RG / GPP; ++pass) {
#pragma unroll
        for (int rr = 0; rr < GPP; ++rr)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int s = (GPP * pass + rr) * 4 + k;
            const u32 off = wbase + (u32)(rr * 64 + lane) * 64u + (u32)((k ^ l3) * 16);
            *reinterpret_cast<lds_u32x4 *>(off) = u32x4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
          }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 4 * GPP; ++i) {
          const u32x4 v = *reinterpret_cast<lds_cu32x4 *>(wbase + (u32)i * 1024u + (u32)lane * 16u);
          const int row = row0 + wave * RPW + pass * (64 * GPP) + i * 16 + prow;
          if (row < p.m && wc < widthB) {
            u64 *dst = C + (long long)row * ldc + wc;
            u64 v0 = (u64)v.x | ((u64)v.y << 32);
            u64 v1 = (u64)v.z | ((u64)v.w << 32);
            if (wc == widthB - 1) v0 &= maskC;
            if (wc + 1 == widthB - 1) v1 &= maskC;
            if (wc + 1 < widthB) {
              if (accum) {
                const uint4 old = *reinterpret_cast<const uint4 *>(dst);
                v0 ^= (u64)old.x | ((u64)old.y << 32);
                v1 ^= (u64)old.z | ((u64)old.w << 32);
              }
              *reinterpret_cast<uint4 *>(dst) = make_uint4((u32)v0, (u32)(v0 >> 32), (u32)v1, (u32)(v1 >> 32));
            } else {
              if (accum) v0 ^= dst[0];
              dst[0] = v0;
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
#ifdef GF2K_DEV_VARIANTS
  if (clk_stamps && tid == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long *o = clk_stamps + ((RG == 8 && APACK == 1) ? 0 : (1 << 20)) + 4ll * (blockIdx.x & 0x3ffff);
    o[0] = clk_rt0, o[1] = clk_c0, o[2] = c1, o[3] = rt1;
  }
#endif
}

// C (+)= the partial tiles of a stream-K launch (gf2_m4rm_kernel_v8): one 256-thread block per (remainder tile, 64 rows), thread =
// (row, 16-byte column piece).  Remainder tile u covers slabs [u Q, (u + 1) Q) of the remainder space; segment s covers
// [s seg, (s + 1) seg) and keeps the part that lies in its first tile in slot 2 s, the part in the following tile in slot
// 2 s + 1.  Slot layout: see the kernel (16 bytes of step (rg, k') of thread (wave, lane) at ((rg * 4 + k') * 512 + 64 wave + lane)
// * 16, holding piece k' ^ (lane & 3) of row 64 RG wave + 64 rg + lane).
__global__ __launch_bounds__(256) void gf2_streamk_reduce_kernel(const gf2k_mul_args p, int RG) {
  const int R = 512 * RG, RPW = 64 * RG, bpt = R / 64;
  const int u = blockIdx.x / bpt, rb = blockIdx.x % bpt;
  const int row = rb * 64 + ((int)threadIdx.x >> 2), k = threadIdx.x & 3;
  const v8_tile tl = v8_decode(p.n_full + u, p);
  const long long Q = p.tile_slabs, seg = p.seg_slabs, glo = (long long)u * Q, ghi = glo + Q;
  const int s_first = (int)(glo / seg);
  int s_last = (int)((ghi - 1) / seg);
  if (s_last > p.nseg - 1) s_last = p.nseg - 1;
  const int wv = row / RPW, rg = (row % RPW) >> 6, ln = row & 63, kp = k ^ (ln & 3);
  const long long e = ((long long)(rg * 4 + kp) * 512 + wv * 64 + ln) * 2;  // u64 index inside a slot
  uint4 a = make_uint4(0, 0, 0, 0);
  for (int s = s_first; s <= s_last; s += 4) {  // four slots in flight (a dependent load per slot would serialise the latencies)
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int sk = min(s + k, s_last);
      const long long sl = 2ll * sk + ((long long)sk * seg < glo ? 1 : 0);
      v[k] = *reinterpret_cast<const uint4 *>(p.P + sl * p.sP + e);
      if (s + k > s_last) v[k] = make_uint4(0, 0, 0, 0);
    }
    a = xor4(xor4(a, v[0]), xor4(xor4(v[1], v[2]), v[3]));
  }
  const int widthB = (p.n + 63) >> 6;
  const u64 maskC = (p.n & 63) ? ((1ull << (p.n & 63)) - 1) : ~0ull;
  const int grow = tl.tm * R + row, wc = tl.tn * kTileWords7 + 2 * k;
  if (grow >= p.m || wc >= widthB) return;
  u64 *dst = p.C + (long long)tl.bt * p.sC + (long long)grow * p.ldc + wc;
  u64 v0 = (u64)a.x | ((u64)a.y << 32), v1 = (u64)a.z | ((u64)a.w << 32);
  if (wc == widthB - 1) v0 &= maskC;
  if (wc + 1 == widthB - 1) v1 &= maskC;
  if (wc + 1 < widthB) {
    if (p.accumulate) {
      const uint4 old = *reinterpret_cast<const uint4 *>(dst);
      v0 ^= (u64)old.x | ((u64)old.y << 32);
      v1 ^= (u64)old.z | ((u64)old.w << 32);
    }
    *reinterpret_cast<uint4 *>(dst) = make_uint4((u32)v0, (u32)(v0 >> 32), (u32)v1, (u32)(v1 >> 32));
  } else {
    if (p.accumulate) v0 ^= dst[0];
    dst[0] = v0;
  }
}

#ifdef GF2K_DEV_VARIANTS
#include "../../tools/gf2_kernels_v9_experiment.inc"  // tall, narrow tiles (4096 x 128): measured, not adopted
#endif
#ifdef GF2K_DEV_VARIANTS
#include "../../tools/gf2_kernels_legacy_v7.inc"  // the fixed-tile predecessor with its ablation branches, kbench only
#endif

// A (m x w words, row stride lds_) -> row-group-packed copy for the APACK tile kernels: word c of row r at u64 index
// ((r / 64) * wp + c) * 64 + r % 64 (wp even, >= w); rows past m and words past w are written as zeros.  A wave takes the 64
// rows of a group: 8-byte reads of 64 rows, contiguous 512-byte writes.
__global__ __launch_bounds__(256) void gf2_packA_kernel(u64 *__restrict__ dst, long long wp, const u64 *__restrict__ src,
                                                        long long lds_, int m, int w) {
  const long long pairs = wp >> 1, total = (long long)((m + 63) >> 6) * pairs * 64;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long rest = idx >> 6;
    const int g = (int)(rest / pairs), c = (int)(rest % pairs) * 2, rl = (int)(idx & 63);
    const long long r = (long long)g * 64 + rl;
    const u64 v0 = (r < m && c < w) ? src[r * lds_ + c] : 0, v1 = (r < m && c + 1 < w) ? src[r * lds_ + c + 1] : 0;
    const long long pk = ((long long)g * wp + c) * 64 + rl;
    dst[pk] = v0;
    dst[pk + 64] = v1;
  }
}

// C (+)= XOR of the ksplit partial products of a split-K launch (dense, row stride ldp, 16-byte accesses; ldp even)
__global__ __launch_bounds__(256) void gf2_splitk_reduce_kernel(u64 *__restrict__ C, long long ldc, long long sC,
                                                                const u64 *__restrict__ P, long long ldp, long long sP,
                                                                int ksplit, int m, int words, int batch, int accumulate) {
  const int pairs = (words + 1) >> 1;
  const long long per = (long long)m * pairs, total = per * batch;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / per);
    const long long rem = idx - (long long)b * per;
    const int r = (int)(rem / pairs), w = (int)(rem % pairs) * 2;
    const u64 *src = P + (long long)b * ksplit * sP + (long long)r * ldp + w;
    uint4 a = make_uint4(0, 0, 0, 0);
    for (int s = 0; s < ksplit; ++s) a = xor4(a, *reinterpret_cast<const uint4 *>(src + (long long)s * sP));
    u64 *dst = C + (long long)b * sC + (long long)r * ldc + w;
    if (w + 1 < words) {
      if (accumulate) a = xor4(a, *reinterpret_cast<const uint4 *>(dst));
      *reinterpret_cast<uint4 *>(dst) = a;
    } else {
      u64 v = (u64)a.x | ((u64)a.y << 32);
      if (accumulate) v ^= dst[0];
      dst[0] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// row-parity kernel: C (m x n) (+)= A (m x l) * Bt^T, Bt is n x l   (_mzd_mul_naive, mzd.rs:154-168)
// one lane per (row of A, 64-bit output word).  HBM-streaming over A.
// ---------------------------------------------------------------------------------------------

template <int WL>  // words of the inner dimension held in registers
__global__ __launch_bounds__(256) void gf2_rowparity_kernel(const u64 *__restrict__ A, long long lda,
                                                            const u64 *__restrict__ Bt, long long ldbt,
                                                            u64 *__restrict__ C, long long ldc, int m, int l, int n,
                                                            int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int jw = blockIdx.y;
  if (i >= m) return;
  const int wl = (l + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const int nb = min(64, n - 64 * jw);
  u64 out = 0;
  if constexpr (WL > 0) {
    u64 a[WL];
    const u64 *ar = A + i * lda;
    if constexpr (WL % 2 == 0) {
#pragma unroll
      for (int t = 0; t < WL; t += 2) {
        if (t + 1 < wl) {
          const uint4 v = *reinterpret_cast<const uint4 *>(ar + t);
          a[t] = (u64)v.x | ((u64)v.y << 32);
          a[t + 1] = (u64)v.z | ((u64)v.w << 32);
        } else {
          a[t] = (t < wl) ? ar[t] : 0;
          a[t + 1] = 0;
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < WL; ++t) a[t] = (t < wl) ? ar[t] : 0;
    }
#pragma unroll
    for (int t = 0; t < WL; ++t)
      if (t == wl - 1) a[t] &= maskL;
    for (int jj = 0; jj < nb; ++jj) {
      const u64 *b = Bt + (long long)(64 * jw + jj) * ldbt;  // wave-uniform -> scalar loads
      u64 x = 0;
#pragma unroll
      for (int t = 0; t < WL; ++t)
        if (t < wl) x ^= a[t] & b[t];
      out |= (u64)(__popcll(x) & 1) << jj;
    }
  } else {
    const u64 *ar = A + i * lda;
    for (int jj = 0; jj < nb; ++jj) {
      const u64 *b = Bt + (long long)(64 * jw + jj) * ldbt;
      u64 x = 0;
      for (int t = 0; t < wl; ++t) {
        u64 v = ar[t] & b[t];
        if (t == wl - 1) v &= maskL;
        x ^= v;
      }
      out |= (u64)(__popcll(x) & 1) << jj;
    }
  }
  u64 *dst = C + i * ldc + jw;
  if (accumulate) out ^= *dst;
  *dst = out;
}

// ---------------------------------------------------------------------------------------------
// narrow product kernel: C (m x n) (+)= A (m x l) * B (l x n) with n <= 64 -- mzd_mul_naive as mul_slice calls
// it (binary_matrix.rs:416-431: A * v^T with v an l x 1 column).  One launch: every block first transposes B
// into LDS with wave ballots (word tw of B^T row j = ballot over the 64 lanes holding rows 64tw..64tw+63 of
// bit j), then streams A exactly like the row-parity kernel with B^T read from LDS (broadcast reads).
// ---------------------------------------------------------------------------------------------

template <int WL>
__global__ __launch_bounds__(256) void gf2_narrow_kernel(const u64 *__restrict__ A, long long lda,
                                                         const u64 *__restrict__ B, long long ldb, u64 *__restrict__ C,
                                                         long long ldc, int m, int l, int n, int accumulate) {
  extern __shared__ __align__(16) unsigned char lds[];
  u64 *bt = reinterpret_cast<u64 *>(lds);  // n rows x wl words
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wl = (l + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // rows are fetched one iteration ahead, and the first one BEFORE the transpose of B: its latency then overlaps B's
  constexpr int NA = WL > 0 ? WL : 1;
  u64 cur[NA];
  auto load_row = [&](long long row, u64 (&a)[NA]) __attribute__((always_inline)) {
    if constexpr (WL > 0) {
      const u64 *ar = A + row * lda;
      if constexpr (WL % 2 == 0) {
#pragma unroll
        for (int t = 0; t < WL; t += 2) {
          if (t + 1 < wl) {
            const uint4 v = *reinterpret_cast<const uint4 *>(ar + t);
            a[t] = (u64)v.x | ((u64)v.y << 32);
            a[t + 1] = (u64)v.z | ((u64)v.w << 32);
          } else {
            a[t] = (t < wl) ? ar[t] : 0;
            a[t + 1] = 0;
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < WL; ++t) a[t] = (t < wl) ? ar[t] : 0;
      }
    }
  };
  if (WL > 0 && i < m) load_row(i, cur);
  for (int tw = wave; tw < wl; tw += 4) {
    const int r = 64 * tw + lane;
    const u64 v = (r < l) ? B[(long long)r * ldb] : 0;
    for (int j = 0; j < n; ++j) {
      const u64 mk = __ballot((v >> j) & 1);
      if (lane == 0) bt[j * wl + tw] = mk;
    }
  }
  __syncthreads();
  for (; i < m; i += stride) {
    u64 out = 0;
    if constexpr (WL > 0) {
      u64 a[WL];
#pragma unroll
      for (int t = 0; t < WL; ++t) a[t] = (t == wl - 1) ? (cur[t] & maskL) : cur[t];
      if (i + stride < m) load_row(i + stride, cur);
      for (int j = 0; j < n; ++j) {
        u64 x = 0;
#pragma unroll
        for (int t = 0; t < WL; ++t)
          if (t < wl) x ^= a[t] & bt[j * wl + t];
        out |= (u64)(__popcll(x) & 1) << j;
      }
    } else {
      const u64 *ar = A + i * lda;
      for (int j = 0; j < n; ++j) {
        u64 x = 0;
        for (int t = 0; t < wl; ++t) {
          u64 v = ar[t] & bt[j * wl + t];
          if (t == wl - 1) v &= maskL;
          x ^= v;
        }
        out |= (u64)(__popcll(x) & 1) << j;
      }
    }
    u64 *dst = C + i * ldc;
    if (accumulate) out ^= *dst;
    *dst = out;
  }
}

// ---------------------------------------------------------------------------------------------
// wide matrix x vector(s): C (m x n) (+)= A (m x l) * Bt^T with n <= 32 and a LONG inner dimension (mul_slice / `&A * &v` on a
// large square A, binary_matrix.rs:416-431,528-542; a block of Wiedemann / Lanczos vectors).  The row-parity and narrow kernels
// keep a row in the registers of ONE lane, which is right for l <= 512; a lane walking an 8-KiB row on its own makes every load
// of its wave touch 64 cache lines (65536^2 times a vector: 0.30 ms = 1.8 TB/s).  Here a WAVE owns a row: lane t takes the
// words t, t + 64, ... -- 512 contiguous bytes per load, eight loads in flight per lane --, ANDs them with the same words of
// the vectors (Bt: n rows of l bits, staged in LDS slab by slab: lane-consecutive 8-byte reads, conflict-free) and keeps one
// 64-bit XOR accumulator per vector; the parity of a vector's accumulators over the wave is one popcount, one ballot and one
// scalar popcount.  A is read once; n = 1 ... 8 stream at the rate of a copy, 32 vectors cost about three times that (VALU).
// The inner dimension is walked in slabs of `slab` words (n * slab * 8 bytes of LDS); rows are revisited per slab and C is
// accumulated, by the one wave that owns the row.
// ---------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(1024) void gf2_widevec_kernel(const u64 *__restrict__ A, long long lda, const u64 *__restrict__ Bt,
                                                            long long ldbt, u64 *__restrict__ C, long long ldc, int m, int l,
                                                            int n, int accumulate, int slab, int jshift) {
  extern __shared__ __align__(16) unsigned char lds[];
  u64 *bt = reinterpret_cast<u64 *>(lds);  // NJ rows x slab words (rows n .. NJ-1 are zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int wl = (l + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const long long gw = (long long)blockIdx.x * nwaves + wave, nw = (long long)gridDim.x * nwaves;
  constexpr int U = 8;  // loads in flight per lane
  for (int s0 = 0; s0 < wl; s0 += slab) {
    const int sw = min(slab, wl - s0);
    if (s0) __syncthreads();  // the previous slab's readers are done
    for (int idx = tid; idx < NJ * sw; idx += blockDim.x) {
      const int j = idx / sw, t = idx - j * sw;
      u64 v = j < n ? Bt[(long long)j * ldbt + s0 + t] : 0;
      if (s0 + t == wl - 1) v &= maskL;  // bits past the inner dimension never count, whatever A holds there
      bt[j * slab + t] = v;
    }
    __syncthreads();
    for (long long i = gw; i < m; i += nw) {
      u64 acc[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[j] = 0;
      const u64 *ar = A + i * lda + s0;
      for (int t0 = 0; t0 < sw; t0 += 64 * U) {
        u64 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = t0 + 64 * u + lane;
          a[u] = t < sw ? ar[t] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (t0 + 64 * u >= sw) break;  // (uniform: a short row does not pay for eight words per lane)
          const int t = min(t0 + 64 * u + lane, sw - 1);  // (a[u] is zero past the slab)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j] ^= a[u] & bt[j * slab + t];
        }
      }
      u64 out = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const u64 bal = __ballot(__popcll(acc[j]) & 1);
        out |= (u64)(__popcll(bal) & 1) << j;
      }
      if (lane == 0) {
        u64 *dst = C + i * ldc;
        out <<= jshift;  // (vectors jshift .. jshift + n - 1 of a wider C: the caller accumulates)
        if (accumulate || s0) out ^= *dst;
        *dst = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// tall-skinny kernels: C (m x n) (+)= A (m x l) * B (l x n) with n <= 256 and m large -- a BATCH of LPN-style matrix x vector
// products (BASELINE config 5: A is 2^20 x 256, B holds the vectors as columns).  Four-Russians with the roles the shape
// dictates: B is tiny, so the chunks of the inner dimension get their tables of n-bit rows (NW words each) in LDS; each lane
// owns rows of A and XORs the table entries selected by the bytes (or nibbles) of its rows into NW 64-bit accumulators per row.
// (The first form of this kernel -- 8-byte-aligned entries stored table after table, bank-conflicted lookups -- was removed in
// round 2; its successors follow.)
// ---------------------------------------------------------------------------------------------
// tall-skinny kernel with skewed lookups (used for n > 64): one lane per row like the first kernel, but with
// conflict-free LDS lookups.  A 256-B LDS row holds entry e of TPR = 32/NW tables side by side (two such row sets =
// 128 KiB).  Lane L visits the TPR chunks of a row set in the order c ^ s, s = L mod TPR, so that the lanes the LDS serves
// together read TPR different tables, i.e. different banks.  To keep every register index static, the row's bytes are
// XOR-permuted by s once after loading (dword butterfly with v_cndmask, bytes inside a dword with one v_perm_b32):
// byte c of the permuted row is byte c ^ s of the row.  NW = 4 (two 16-byte reads per entry): lanes 8..15 of every 16
// read the halves in the opposite order and swap their accumulators at the end.
// ---------------------------------------------------------------------------------------------
template <int NW, int RPT, int NT>
__global__ __launch_bounds__(NT) void gf2_tallskinny3_kernel(const u64 *__restrict__ A, long long lda, const u64 *__restrict__ B,
                                                              long long ldb, u64 *__restrict__ C, long long ldc, int m, int l,
                                                              int n, int accumulate) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int TPR = 32 / NW;            // tables per row set
  constexpr int ND = TPR / 4;             // dwords of a row that select inside one row set
  constexpr int WRS = TPR / 8;            // 64-bit words of the inner dimension per row set (NW=4: 1)
  constexpr int kWordsPerGroup = 2 * WRS; // two row sets in LDS
  constexpr int kStage = 128 * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wl = (l + 63) >> 6, wn = (n + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const long long row_base = (long long)blockIdx.x * (NT * RPT);
  const int s = lane & (TPR - 1);
  const u32 bsel = (s & 3) == 0 ? 0x03020100u : (s & 3) == 1 ? 0x02030001u : (s & 3) == 2 ? 0x01000302u : 0x00010203u;
  const int hsw = NW == 4 ? (lane >> 3) & 1 : 0;  // NW=4: which 16-byte half this lane reads first
  const u32 xl0 = (u32)s * (8u * NW) + (u32)hsw * 16u, xl1 = xl0 + 65536u;

  u32 acc[RPT][2 * NW];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int w = 0; w < 2 * NW; ++w) acc[r][w] = 0;
  const u64 maskC = (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  // a row is stored as soon as its last lookup is done, so that the writes of C overlap the lookups of the other rows
  auto store_row = [&](int r) {
    const long long row = row_base + (long long)r * NT + tid;
    if (row < m) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w < wn) {
          // NW=4: lanes that read the upper half first hold words 2,3 in acc[0..3] and words 0,1 in acc[4..7]
          u32 lo = acc[r][2 * w], hi = acc[r][2 * w + 1];
          if constexpr (NW == 4) {
            lo = hsw ? acc[r][(2 * w) ^ 4] : lo;
            hi = hsw ? acc[r][(2 * w + 1) ^ 4] : hi;
          }
          u64 v = (u64)lo | ((u64)hi << 32);
          if (w == wn - 1) v &= maskC;
          u64 *dd = C + row * ldc + w;
          if (accumulate) v ^= *dd;
          *dd = v;
        }
    }
  };

  for (int w0 = 0; w0 < wl; w0 += kWordsPerGroup) {  // group of 64-bit words of the inner dimension
    __syncthreads();  // previous group's lookups are done
    // stage the group's rows of B behind the tables, coalesced (rows past l are zero: their tables select nothing)
    u64 *bst = reinterpret_cast<u64 *>(lds + kStage);
    for (int idx = tid; idx < kWordsPerGroup * 64 * NW; idx += NT) {
      const int rr_ = idx / NW, w = idx % NW;
      const long long brow = (long long)w0 * 64 + rr_;
      bst[idx] = (brow < l && w < wn) ? B[brow * ldb + w] : 0;
    }
    __syncthreads();
    // build: item = (row set, table, word, high nibble); lanes differ in (word, table) first -> conflict-free writes;
    // 8 rows of B, 4 select-XORs, 15 Gray-code steps for the 16 entries of the nibble
    for (int item = tid; item < 2 * TPR * NW * 16; item += NT) {
      const int w = item % NW, t = (item / NW) % TPR;
      int rest = item / (NW * TPR);
      const int rs = rest & 1, h = rest >> 1;
      const u64 *rows = bst + ((rs * TPR + t) * 8) * NW + w;
      u64 v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) v ^= rows[(4 + b) * NW] & (0ull - (u64)((h >> b) & 1));
      const u64 r0 = rows[0], r1 = rows[NW], r2 = rows[2 * NW], r3 = rows[3 * NW];
      unsigned char *tb = lds + rs * 65536 + t * (8 * NW) + w * 8;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int gc = i ^ (i >> 1);
        if (i) {
          const int flip = __builtin_ctz(i);
          v ^= flip == 0 ? r0 : flip == 1 ? r1 : flip == 2 ? r2 : r3;
        }
        *reinterpret_cast<u64 *>(tb + (h * 16 + gc) * 256) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      if ((w0 + rs * WRS) * 64 < l) {  // uniform: the row set holds bits of the inner dimension
        // the words of ALL of this lane's rows first (one memory latency per row set instead of one per row)
        u64 aw[RPT][WRS];
        const int wi0 = w0 + rs * WRS;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const long long row = min(row_base + (long long)r * NT + tid, (long long)m - 1);  // clamped: stores are guarded below
          const u64 *ap = A + row * lda;
#pragma unroll
          for (int q = 0; q < WRS; ++q) aw[r][q] = ap[min(wi0 + q, wl - 1)];  // clamped as well, masked below
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          __builtin_amdgcn_sched_barrier(0);  // one row at a time: interleaving the rows' lookups spills
          u32 d[ND];
#pragma unroll
          for (int q = 0; q < WRS; ++q) {
            u64 x = wi0 + q < wl ? aw[r][q] : 0;
            if (wi0 + q == wl - 1) x &= maskL;
            d[2 * q] = (u32)x;
            d[2 * q + 1] = (u32)(x >> 32);
          }
          // byte c of the permuted dwords = byte c ^ s of the row set's bytes
#pragma unroll
          for (int k = 0; (1 << k) < ND; ++k) {
            const bool sw = (s >> (2 + k)) & 1;
            u32 t2[ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) t2[i] = sw ? d[i ^ (1 << k)] : d[i];
#pragma unroll
            for (int i = 0; i < ND; ++i) d[i] = t2[i];
          }
#pragma unroll
          for (int i = 0; i < ND; ++i) d[i] = __builtin_amdgcn_perm(d[i], d[i], bsel);
          const u32 xl = rs ? xl1 : xl0;
#pragma unroll
          for (int c = 0; c < TPR; c += 2) {
            // LDS byte address {0, row set, selecting byte, skewed table offset} in one v_perm_b32 (bytes 0, 2, 3 from the
            // lane's table offset, byte 1 = byte c&3 of the permuted dword); integer -> LDS pointer, no base to add.
            // Two chunks per step: their entries are folded into the accumulators with one three-input XOR per dword.
            const u32 off = __builtin_amdgcn_perm(d[c >> 2], xl ^ (u32)(c * 8 * NW), 0x03020000u | ((4u + (c & 3)) << 8));
            const u32 off1 = __builtin_amdgcn_perm(d[(c + 1) >> 2], xl ^ (u32)((c + 1) * 8 * NW), 0x03020000u | ((4u + ((c + 1) & 3)) << 8));
            if constexpr (NW == 1) {
              const u32x2v v = *reinterpret_cast<lds_cu32x2 *>(off), w = *reinterpret_cast<lds_cu32x2 *>(off1);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(v.x), "v"(w.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(v.y), "v"(w.y));
            } else if constexpr (NW == 2) {
              const u32x4 v = *reinterpret_cast<lds_cu32x4 *>(off), w = *reinterpret_cast<lds_cu32x4 *>(off1);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(v.x), "v"(w.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(v.y), "v"(w.y));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][2]) : "v"(v.z), "v"(w.z));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][3]) : "v"(v.w), "v"(w.w));
            } else {
              const u32x4 v = *reinterpret_cast<lds_cu32x4 *>(off), w = *reinterpret_cast<lds_cu32x4 *>(off1);
              const u32x4 v2 = *reinterpret_cast<lds_cu32x4 *>(off ^ 16u), w2 = *reinterpret_cast<lds_cu32x4 *>(off1 ^ 16u);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(v.x), "v"(w.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(v.y), "v"(w.y));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][2]) : "v"(v.z), "v"(w.z));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][3]) : "v"(v.w), "v"(w.w));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][4]) : "v"(v2.x), "v"(w2.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][5]) : "v"(v2.y), "v"(w2.y));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][6]) : "v"(v2.z), "v"(w2.z));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][7]) : "v"(v2.w), "v"(w2.w));
            }
          }
          // last row set of the last group that holds bits of the inner dimension (uniform)
          if ((w0 + (rs + 1) * WRS) * 64 >= l) store_row(r);
        }
      }
    }
  }
}

// The round-2/3 kernels for l <= 256 (gf2_tallskinny5_kernel: 8-bit tables, two phases at n > 128; gf2_tallskinny6_kernel: 4-bit
// tables, small streaming workgroups) were replaced by gf2_lpn.inc in round 4 and no shape selects them any more: they are
// compiled into development builds only (tools/libm4ri_hip_dev.so, M4RI_HIP_LPN=0 for A/B runs).
#ifdef GF2K_DEV_VARIANTS
// ---------------------------------------------------------------------------------------------
// tall-skinny kernel for an inner dimension of at most 256 bits (BASELINE config 5: l = 256) and 64 < n <= 256: the skewed
// lookups of gf2_tallskinny3_kernel with every row of A read exactly ONCE.
// gf2_tallskinny3_kernel fetches the words of a row again for every row set (16 bytes per lane and row: half of every
// 32-byte sector, once per row set -- 1.19x the algorithmic traffic at n = 256 and a load latency in the middle of the
// kernel each time).  Here a lane loads its RPT whole rows (32 bytes each, two 16-byte loads) before anything else, the 256
// rows of B are staged once, and the tables of 2 row sets (128 KiB: 256 bits of the inner dimension at NW = 2, 128 bits at
// NW = 4) are built per phase.  NW = 2 has ONE phase: a row goes through both row sets, is stored and forgotten (4 accumulator
// registers in all).  NW = 4 has two phases with 8 accumulator registers per row; it runs with 512 threads and 8 rows per
// lane so that 8 whole rows and their accumulators fit into registers.  The build of a phase needs only LDS, so the loads
// of A issued at the top stay in flight across it (the barriers wait for LDS operations only).
// ---------------------------------------------------------------------------------------------
template <int NW, int RPT, int NT, bool FULL, bool DBG = false, int EARLY = (RPT >= 8 ? 2 : 1), bool AHEAD = false>  // FULL: l in (192, 256], rows 16-byte aligned: two 16-byte loads per row, no branches
__global__ __launch_bounds__(NT) void gf2_tallskinny5_kernel(const u64 *__restrict__ A, long long lda, const u64 *__restrict__ B,
                                                              long long ldb, u64 *__restrict__ C, long long ldc, int m, int l,
                                                              int n, int accumulate, u64 *__restrict__ stamps = nullptr) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int TPR = 32 / NW;       // tables per row set (a 256-byte LDS row holds entry e of TPR tables)
  constexpr int ND = TPR / 4;        // dwords of a row that select inside one row set
  constexpr int WRS = TPR / 8;       // 64-bit words of the inner dimension per row set
  constexpr int SETS = NW == 1 ? 1 : 2;  // row sets in LDS at a time (64 KiB each); NW = 1: one row set holds all 256 bits
  constexpr int PHASES = 4 / (SETS * WRS);  // 4 words = SETS * WRS * PHASES
  const int tid = threadIdx.x, lane = tid & 63;
  const int wl = (l + 63) >> 6, wn = (n + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const u64 maskC = (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  const long long row_base = (long long)blockIdx.x * (NT * RPT);
  const int s = lane & (TPR - 1);
  const u32 bsel = (s & 3) == 0 ? 0x03020100u : (s & 3) == 1 ? 0x02030001u : (s & 3) == 2 ? 0x01000302u : 0x00010203u;
  const int hsw = NW == 4 ? (lane >> 3) & 1 : 0;  // NW = 4: which 16-byte half this lane reads first
  const u32 xl0 = (u32)s * (8u * NW) + (u32)hsw * 16u, xl1 = xl0 + 65536u;
  auto lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  // DBG (development builds): wall-clock stamps (100 MHz) per wave: start, tables built, first row done, last row done
  u64 *st = nullptr;
  if constexpr (DBG) {
    st = stamps + ((long long)blockIdx.x * (NT / 64) + (tid >> 6)) * 8;
    if (lane == 0) st[0] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- the 256 rows of B first (256 x NW words, one or two coalesced 8-byte loads per thread; rows past l are zero: their
  // tables select nothing): in-order return lets the build wait for them without waiting for the rows of A requested behind
  // them.  (Loading each build item's eight words of B straight from global memory instead -- scattered 8-byte loads, 128
  // instructions per CU ahead of the loads of A -- measured 0.9 us slower to the first table.)
  constexpr int ITEMS = SETS * TPR * NW * 16, IPT = (ITEMS + NT - 1) / NT;  // build items per phase, per thread
  constexpr int NB = (256 * NW + NT - 1) / NT;
  u64 bv[NB];
#pragma unroll
  for (int kk = 0; kk < NB; ++kk) {
    const int idx = tid + kk * NT, brow_ = idx / NW, w = idx % NW;
    bv[kk] = (idx < 256 * NW && brow_ < l && w < wn) ? B[(long long)brow_ * ldb + w] : 0;
  }
  // ---- every row of this lane, whole (words past the inner dimension read as zero).  Only the first row is requested
  // before the build: a CU holds a limited number of outstanding misses, and waves that cannot issue their loads would
  // reach the build barrier microseconds late (stamps: tables ready at 7.5 us with all rows requested up front) ----
  u64 aw[RPT][4];
  auto load_row = [&](int r) __attribute__((always_inline)) {
    const long long row = min(row_base + (long long)r * NT + tid, (long long)m - 1);  // clamped: stores are guarded
    const u64 *ap = A + row * lda;
    if constexpr (FULL) {  // straight-line code: the compiler can then wait for the rows one by one
      const uint4 *ap4 = reinterpret_cast<const uint4 *>(__builtin_assume_aligned(ap, 16));
      const uint4 lo = ap4[0], hi = ap4[1];
      aw[r][0] = (u64)lo.x | ((u64)lo.y << 32);
      aw[r][1] = (u64)lo.z | ((u64)lo.w << 32);
      aw[r][2] = (u64)hi.x | ((u64)hi.y << 32);
      aw[r][3] = (u64)hi.z | ((u64)hi.w << 32);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) aw[r][q] = q < wl ? ap[q] : 0;
    }
  };
  // EARLY rows are requested before the first build; AHEAD: the others one row ahead of the lookups instead of all at once
#pragma unroll
  for (int r = 0; r < EARLY; ++r) load_row(r);
  u64 *bst = reinterpret_cast<u64 *>(lds + SETS * 65536);
#pragma unroll
  for (int kk = 0; kk < NB; ++kk)
    if (tid + kk * NT < 256 * NW) bst[tid + kk * NT] = bv[kk];
  lds_barrier();  // LDS operations only: the loads of A stay in flight across the build

  u32 acc[RPT][2 * NW];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int w = 0; w < 2 * NW; ++w) acc[r][w] = 0;

  auto store_row = [&](int r) {
    const long long row = row_base + (long long)r * NT + tid;
    if (row < m) {
      // NW = 4: lanes that read the upper half first hold words 2, 3 in acc[0..3] and words 0, 1 in acc[4..7]
      u32 o[2 * NW];
#pragma unroll
      for (int i = 0; i < 2 * NW; ++i) o[i] = (NW == 4 && hsw) ? acc[r][i ^ 4] : acc[r][i];
      if (FULL && NW >= 2 && wn == NW) {  // whole rows of C, 16 bytes per store (FULL also says: ldc even, C 16-byte aligned)
        const u32 mlo = (u32)maskC, mhi = (u32)(maskC >> 32);
        o[2 * NW - 2] &= mlo;
        o[2 * NW - 1] &= mhi;
        uint4 *dd = reinterpret_cast<uint4 *>(__builtin_assume_aligned(C + row * ldc, 16));
#pragma unroll
        for (int q = 0; q < NW / 2; ++q) {
          uint4 v = make_uint4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
          if (accumulate) v = xor4(v, dd[q]);
          dd[q] = v;
        }
      } else {
#pragma unroll
        for (int w = 0; w < NW; ++w)
          if (w < wn) {
            u64 v = (u64)o[2 * w] | ((u64)o[2 * w + 1] << 32);
            if (w == wn - 1) v &= maskC;
            u64 *dd = C + row * ldc + w;
            if (accumulate) v ^= *dd;
            *dd = v;
          }
      }
    }
  };

#pragma unroll
  for (int ph = 0; ph < PHASES; ++ph) {
    if (ph * SETS * WRS * 64 >= l) break;  // uniform: nothing of the inner dimension left
    if (ph) lds_barrier();              // the previous phase's lookups are done
    // build: 4 select-XORs for the high nibble, 15 Gray-code steps for the 16 entries of the low nibble
#pragma unroll
    for (int it = 0; it < IPT; ++it) {
      const int item = tid + it * NT;
      if (ITEMS % NT != 0 && item >= ITEMS) break;
      const int w = item % NW, t = (item / NW) % TPR;
      const int rest = item / (NW * TPR);
      const int rs = rest & (SETS - 1), h = rest / SETS;
      const u64 *rows = bst + (((ph * SETS + rs) * TPR + t) * 8) * NW + w;
      u64 v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) v ^= rows[(4 + b) * NW] & (0ull - (u64)((h >> b) & 1));
      const u64 r0 = rows[0], r1 = rows[NW], r2 = rows[2 * NW], r3 = rows[3 * NW];
      unsigned char *tb = lds + rs * 65536 + t * (8 * NW) + w * 8;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int gc = i ^ (i >> 1);
        if (i) {
          const int flip = __builtin_ctz(i);
          v ^= flip == 0 ? r0 : flip == 1 ? r1 : flip == 2 ? r2 : r3;
        }
        *reinterpret_cast<u64 *>(tb + (h * 16 + gc) * 256) = v;
      }
    }
    lds_barrier();
    if constexpr (DBG) {
      if (lane == 0) st[1 + 3 * ph] = __builtin_amdgcn_s_memrealtime();
    }
    if (ph == 0 && !AHEAD) {  // the remaining rows: nothing waits behind these requests but this wave's own lookups
#pragma unroll
      for (int r = EARLY; r < RPT; ++r) load_row(r);
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      __builtin_amdgcn_sched_barrier(0);  // one row at a time
      if (AHEAD && ph == 0 && r + EARLY < RPT) load_row(r + EARLY);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rs = 0; rs < SETS; ++rs) {
        const int wi0 = (ph * SETS + rs) * WRS;
        if (wi0 * 64 < l) {  // uniform
          u32 d[ND];
#pragma unroll
          for (int q = 0; q < WRS; ++q) {
            u64 x = aw[r][wi0 + q];
            if (wi0 + q == wl - 1) x &= maskL;
            d[2 * q] = (u32)x;
            d[2 * q + 1] = (u32)(x >> 32);
          }
          // byte c of the permuted dwords = byte c ^ s of the row set's bytes
#pragma unroll
          for (int k = 0; (1 << k) < ND; ++k) {
            const bool sw = (s >> (2 + k)) & 1;
            u32 t2[ND];
#pragma unroll
            for (int i = 0; i < ND; ++i) t2[i] = sw ? d[i ^ (1 << k)] : d[i];
#pragma unroll
            for (int i = 0; i < ND; ++i) d[i] = t2[i];
          }
#pragma unroll
          for (int i = 0; i < ND; ++i) d[i] = __builtin_amdgcn_perm(d[i], d[i], bsel);
          const u32 xl = rs ? xl1 : xl0;
#pragma unroll
          for (int c = 0; c < TPR; c += 2) {
            const u32 off = __builtin_amdgcn_perm(d[c >> 2], xl ^ (u32)(c * 8 * NW), 0x03020000u | ((4u + (c & 3)) << 8));
            const u32 off1 = __builtin_amdgcn_perm(d[(c + 1) >> 2], xl ^ (u32)((c + 1) * 8 * NW), 0x03020000u | ((4u + ((c + 1) & 3)) << 8));
            if constexpr (NW == 1) {
              const u32x2v v = *reinterpret_cast<lds_cu32x2 *>(off), w = *reinterpret_cast<lds_cu32x2 *>(off1);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(v.x), "v"(w.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(v.y), "v"(w.y));
            } else {
              const u32x4 v = *reinterpret_cast<lds_cu32x4 *>(off), w = *reinterpret_cast<lds_cu32x4 *>(off1);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(v.x), "v"(w.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(v.y), "v"(w.y));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][2]) : "v"(v.z), "v"(w.z));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][3]) : "v"(v.w), "v"(w.w));
            }
            if constexpr (NW == 4) {
              const u32x4 v2 = *reinterpret_cast<lds_cu32x4 *>(off ^ 16u), w2 = *reinterpret_cast<lds_cu32x4 *>(off1 ^ 16u);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][4]) : "v"(v2.x), "v"(w2.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][5]) : "v"(v2.y), "v"(w2.y));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][6]) : "v"(v2.z), "v"(w2.z));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][7]) : "v"(v2.w), "v"(w2.w));
            }
          }
        }
      }
      // the last phase that holds bits of the inner dimension (uniform): the row is complete
      if ((ph + 1) * SETS * WRS * 64 >= l) store_row(r);
      if constexpr (DBG) {
        if ((r == 0 || r == RPT - 1) && lane == 0) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          st[2 + 3 * ph + (r ? 1 : 0)] = __builtin_amdgcn_s_memrealtime();
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// tall-skinny kernel with 4-BIT tables (n <= 256, l <= 256): small workgroups that stream.
// The kernels above build 8-bit tables (64 KiB per 256 bits of l at n = 64, up to 256 KiB at n = 256): one 512- or 1024-thread
// workgroup per CU, whose phases -- wait for B, build, wait for rows, look up, store -- every wave of the chip goes through at
// the same time, so that the HBM pipe idles during the build and the stores come in one burst.  With FOUR bits per table the
// tables of all 256 bits are 8 / 16 / 32 KiB for NW = 1 / 2 / 4 words per entry: a 256-thread workgroup builds them in a
// microsecond, eight (four) such workgroups fit a CU and run out of phase, and the rows stream through a grid-stride loop
// with the next row in flight -- the shape of gf2_narrow_kernel, which reaches the rate of a plain copy.  Twice the lookups
// (64 per row), but a 16-entry table spans each LDS bank at most once, so ANY mix of entries within a wave is conflict-free
// without skewing or byte permutations: lookup = shift, and, ds_read with the table's offset as an immediate.
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void gf2_tallskinny6_kernel(const u64 *__restrict__ A, long long lda, const u64 *__restrict__ B,
                                                              long long ldb, u64 *__restrict__ C, long long ldc, int m, int l,
                                                              int n, int accumulate, int vec_ok) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int EB = 8 * NW;           // bytes per entry
  constexpr int TB = 16 * EB;          // bytes per table
  const int tid = threadIdx.x;
  const int wl = (l + 63) >> 6, wn = (n + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const u64 maskC = (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + tid;
  u64 *bst = reinterpret_cast<u64 *>(lds + 64 * TB);  // the 256 rows of B, NW words each

  // B first (in-order return: the build then does not wait for the row requested behind it), then this lane's first row
  u64 bv[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) bv[w] = (tid < l && w < wn) ? B[(long long)tid * ldb + w] : 0;
  u64 cur[4];
  auto load_row = [&](long long row, u64 (&a)[4]) __attribute__((always_inline)) {
    const u64 *ar = A + row * lda;
    if (vec_ok) {  // l > 192, 16-byte aligned rows: two 16-byte loads
      const uint4 lo = *reinterpret_cast<const uint4 *>(ar), hi = *reinterpret_cast<const uint4 *>(ar + 2);
      a[0] = (u64)lo.x | ((u64)lo.y << 32);
      a[1] = (u64)lo.z | ((u64)lo.w << 32);
      a[2] = (u64)hi.x | ((u64)hi.y << 32);
      a[3] = (u64)hi.z | ((u64)hi.w << 32);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[q] = q < wl ? ar[q] : 0;
    }
  };
  if (i < m) load_row(i, cur);
#pragma unroll
  for (int w = 0; w < NW; ++w) bst[tid * NW + w] = bv[w];
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) only
  __builtin_amdgcn_s_barrier();
  // build: 64 tables x 16 entries; entry e of table t = XOR of rows 4t + b of B for the set bits b of e
  for (int idx = tid; idx < 64 * 16; idx += 256) {
    const int t = idx >> 4, e = idx & 15;
    u64 v[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) v[w] = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const u64 sel = 0ull - (u64)((e >> b) & 1);
#pragma unroll
      for (int w = 0; w < NW; ++w) v[w] ^= bst[(4 * t + b) * NW + w] & sel;
    }
    u64 *dst = reinterpret_cast<u64 *>(lds + idx * EB);
#pragma unroll
    for (int w = 0; w < NW; ++w) dst[w] = v[w];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();

  for (; i < m; i += stride) {
    u32 d[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u64 x = (q == wl - 1) ? (cur[q] & maskL) : cur[q];
      d[2 * q] = (u32)x;
      d[2 * q + 1] = (u32)(x >> 32);
    }
    if (i + stride < m) load_row(i + stride, cur);  // the next row of this lane: in flight during the lookups
    u32 acc[2 * NW];
#pragma unroll
    for (int w = 0; w < 2 * NW; ++w) acc[w] = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      // the eight entry offsets of a dword, pre-scaled, as the bytes of two registers (even nibbles in xe, odd ones in xo): one
      // byte extraction per lookup instead of a shift and a mask
      u32 xe = (d[q] & 0x0f0f0f0fu) * (u32)EB, xo = ((d[q] >> 4) & 0x0f0f0f0fu) * (u32)EB;
      asm("" : "+v"(xe), "+v"(xo));  // (opaque: the optimiser would fold the bytes back into eight shift-and-mask pairs)
#pragma unroll
      for (int k = 0; k < 8; k += 2) {  // nibbles k and k + 1 of dword q: tables 8 q + k, 8 q + k + 1 (offsets fold to immediates)
        const u32 t0 = (u32)(8 * q + k) * TB, t1 = t0 + TB;
        const u32 o0 = EB <= 16 ? __builtin_amdgcn_ubfe(xe, 4 * k, 8) : ((d[q] >> (4 * k)) & 15u) * EB,
                  o1 = EB <= 16 ? __builtin_amdgcn_ubfe(xo, 4 * k, 8) : ((d[q] >> (4 * k + 4)) & 15u) * EB;
        if constexpr (NW == 1) {
          const u32x2v x = *reinterpret_cast<lds_cu32x2 *>(o0 + t0), y = *reinterpret_cast<lds_cu32x2 *>(o1 + t1);
          asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[0]) : "v"(x.x), "v"(y.x));
          asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[1]) : "v"(x.y), "v"(y.y));
        } else {
          const u32x4 x = *reinterpret_cast<lds_cu32x4 *>(o0 + t0), y = *reinterpret_cast<lds_cu32x4 *>(o1 + t1);
          asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[0]) : "v"(x.x), "v"(y.x));
          asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[1]) : "v"(x.y), "v"(y.y));
          asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[2]) : "v"(x.z), "v"(y.z));
          asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[3]) : "v"(x.w), "v"(y.w));
          if constexpr (NW == 4) {
            const u32x4 x2 = *reinterpret_cast<lds_cu32x4 *>(o0 + t0 + 16u), y2 = *reinterpret_cast<lds_cu32x4 *>(o1 + t1 + 16u);
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[4]) : "v"(x2.x), "v"(y2.x));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[5]) : "v"(x2.y), "v"(y2.y));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[6]) : "v"(x2.z), "v"(y2.z));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[7]) : "v"(x2.w), "v"(y2.w));
          }
        }
      }
    }
    u64 *dd = C + i * ldc;
#pragma unroll
    for (int w = 0; w < NW; ++w)
      if (w < wn) {
        u64 v = (u64)acc[2 * w] | ((u64)acc[2 * w + 1] << 32);
        if (w == wn - 1) v &= maskC;
        if (accumulate) v ^= dd[w];
        dd[w] = v;
      }
  }
}

#endif  // GF2K_DEV_VARIANTS (gf2_tallskinny5 / 6)

// ---------------------------------------------------------------------------------------------
// up to 64 vectors against an inner dimension of ANY length: the 4-bit tables of gf2_tallskinny6_kernel, rebuilt per 512-bit
// slab of the inner dimension, with the inner dimension also divided among workgroups (blockIdx.y) whose partial words
// meet in C by 64-bit atomic XOR (C is one word per row: the atomics are m x splits, not a pass over a matrix).
// A block of vectors times a big matrix (block Wiedemann / Lanczos; `mul_slice` with many rows of B) used to take the tile
// kernel, which computes 512 columns to deliver 64 and finds a handful of tiles (4096 x 65536 x 64: ONE tile, 92 us for 32 MiB
// of A), or the wave-per-row kernel, whose AND / XOR work grows with the vector count.  Here a lane owns RPT rows, reads 64
// bytes of each per slab (one whole line: rows of any length stream), looks its 128 nibbles up in 128 sixteen-entry tables of
// 8-byte entries (16 KiB; a table spans every LDS bank pair once: any mix of entries is conflict-free) and folds two lookups
// per v_bitop3_b32 pair.  The build of a slab's tables is 8 entries per thread, ~6 % of the lookups at four rows per lane.
// ---------------------------------------------------------------------------------------------
template <int RPT, int NW>  // NW = words per row of C (n <= 64 NW): entries of 8 NW bytes, 16 KiB x NW of tables
__global__ __launch_bounds__(256) void gf2_tallskinny7_kernel(const u64 *__restrict__ A, long long lda, const u64 *__restrict__ B,
                                                              long long ldb, u64 *__restrict__ C, long long ldc, int m, int l,
                                                              int n, int accumulate, int slabs_per_split, int atomic, int vec_ok,
                                                              int remap_splits, int remap_rblocks) {
  static_assert(NW == 1 || NW == 2, "a 16-entry table must not span more than the 64 banks");
  // Which (row block, division of the inner dimension) a workgroup takes: a workgroup reads 64-byte pieces a row pitch apart, the
  // access pattern of the 512-tile transposition, whose rate depends on which pieces are in flight together (gf2_transpose512_kernel,
  // tools/tilecopy.hip).  remap_splits > 0: a 1-D grid whose workgroups b, b + 8, ... (one XCD, dispatched together) take blocks of
  // 4 divisions x 4 row blocks out of super-tiles of 16 x 8, super-tiles walked diagonally, the XCD's place rotating -- 256
  // contiguous bytes per row and XCD, and no two super-tiles in flight with the same low address bits.
  int bx = blockIdx.x, by = blockIdx.y;
  if (remap_splits > 0) {
    const int b = blockIdx.x, sx_n = (remap_splits + 15) >> 4, sy_n = (remap_rblocks + 7) >> 3;
    const int st = b >> 7, in = b & 127, x = in & 7, slot = in >> 3;
    const int stx = st % sx_n, sty = (st / sx_n + stx) % sy_n;
    const int px = ((x & 3) + stx + sty) & 3, py = ((x >> 2) + stx + (sty >> 2)) & 1;
    by = stx * 16 + 4 * px + (slot & 3), bx = sty * 8 + 4 * py + (slot >> 2);
    if (by >= remap_splits || bx >= remap_rblocks) return;  // (uniform: the grid is padded to whole super-tiles)
  }
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int EB = 8 * NW, TB = 16 * EB;       // bytes per entry / per table
  u64 *bst = reinterpret_cast<u64 *>(lds + 128 * TB);  // the 512 rows of B of the slab (NW words each)
  const int tid = threadIdx.x;
  const int wl = (l + 63) >> 6, wn = (n + 63) >> 6;
  const u64 maskC = (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  const int nslabs = (l + 511) >> 9;
  const int slab0 = by * slabs_per_split, slab1 = min(nslabs, slab0 + slabs_per_split);
  const long long row0 = (long long)bx * (256 * RPT) + tid;
  u32 acc[RPT][2 * NW];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int w = 0; w < 2 * NW; ++w) acc[r][w] = 0;
  for (int slab = slab0; slab < slab1; ++slab) {
    const int w0 = slab * 8;  // first word of the slab
    // this lane's rows first: their loads are in flight across the staging and the build (words past the row read as zero;
    // bits past the inner dimension meet tables built from zero rows of B)
    u32 d[RPT][16];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const long long row = min(row0 + (long long)r * 256, (long long)m - 1);  // clamped: stores are guarded
      const u64 *ar = A + row * lda + w0;
      if (vec_ok && w0 + 8 <= wl) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 v = reinterpret_cast<const uint4 *>(ar)[q];
          d[r][4 * q] = v.x, d[r][4 * q + 1] = v.y, d[r][4 * q + 2] = v.z, d[r][4 * q + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const u64 v = w0 + q < wl ? ar[q] : 0;
          d[r][2 * q] = (u32)v, d[r][2 * q + 1] = (u32)(v >> 32);
        }
      }
    }
    if (slab != slab0) __syncthreads();  // the previous slab's lookups are done
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = slab * 512 + h * 256 + tid;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        u64 v = (k < l && w < wn) ? B[(long long)k * ldb + w] : 0;
        if (w == wn - 1) v &= maskC;
        bst[(h * 256 + tid) * NW + w] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // 128 tables x 16 entries: entry e of table t = XOR of rows 4t + b of the slab for the set bits b of e
      const int idx = tid + 256 * j, t = idx >> 4, e = idx & 15;
      u64 v[NW];
#pragma unroll
      for (int w = 0; w < NW; ++w) v[w] = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const u64 sel = 0ull - (u64)((e >> b) & 1);
#pragma unroll
        for (int w = 0; w < NW; ++w) v[w] ^= bst[(4 * t + b) * NW + w] & sel;
      }
#pragma unroll
      for (int w = 0; w < NW; ++w) *reinterpret_cast<u64 *>(lds + idx * EB + 8 * w) = v[w];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        u32 xe = (d[r][q] & 0x0f0f0f0fu) * (u32)EB, xo = ((d[r][q] >> 4) & 0x0f0f0f0fu) * (u32)EB;
        asm("" : "+v"(xe), "+v"(xo));  // (pre-scaled entry offsets as bytes, see gf2_tallskinny6_kernel)
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const u32 t0 = (u32)(8 * q + k) * TB, t1 = t0 + TB;
          const u32 o0 = __builtin_amdgcn_ubfe(xe, 4 * k, 8), o1 = __builtin_amdgcn_ubfe(xo, 4 * k, 8);
          if constexpr (NW == 1) {
            const u32x2v x = *reinterpret_cast<lds_cu32x2 *>(o0 + t0), y = *reinterpret_cast<lds_cu32x2 *>(o1 + t1);
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(x.x), "v"(y.x));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(x.y), "v"(y.y));
          } else {
            const u32x4 x = *reinterpret_cast<lds_cu32x4 *>(o0 + t0), y = *reinterpret_cast<lds_cu32x4 *>(o1 + t1);
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][0]) : "v"(x.x), "v"(y.x));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][1]) : "v"(x.y), "v"(y.y));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][2]) : "v"(x.z), "v"(y.z));
            asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[r][3]) : "v"(x.w), "v"(y.w));
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const long long row = row0 + (long long)r * 256;
    if (row < m) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w < wn) {
          u64 v = (u64)acc[r][2 * w] | ((u64)acc[r][2 * w + 1] << 32);
          u64 *dst = C + row * ldc + w;
          if (atomic) {
            if (v) atomicXor(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v);
          } else {
            if (accumulate) v ^= *dst;
            *dst = v;
          }
        }
    }
  }
}

// tall-skinny kernel with conflict-free lookups and no byte permutation ("generation" kernel; n <= 256).
// The first kernel's lookups collide in the LDS banks (8-byte entries: 32 lanes on 32 random bank pairs, ~3.5 lanes on the
// busiest); the skewed kernel avoids that by spreading the lanes over 32 / NW tables, which costs a byte permutation of every
// row and reloads A after every table build.  Here the skew stays inside one 64-bit word of the row -- lane L visits its 8
// chunks in the order c ^ (L & 7), and the selecting byte comes out of the two dwords of the word with ONE v_perm_b32 whose
// selector is per lane -- and whatever is missing to fill a 256-byte LDS row comes from COPIES of every table: the row holds
// entry e of 8 chunks x (4 / NW) copies x 8 NW bytes (slot = copies * chunk + copy), so the lanes the LDS serves together
// read different slots whatever their entries are (NW = 4: one copy, the two 16-byte halves of an entry are read in opposite
// order by lanes 8..15 of every 16, as in the skewed kernel).
// A "generation" = the tables of one 64-bit word of the inner dimension = 64 KiB; two generations are resident (one being
// looked up while the next is built, ONE barrier per generation, and that barrier waits for LDS operations only, so the loads
// of A issued at the top stay in flight across the builds); rows of the table are written with one ds_write_addtid_b32 each
// (lane = dword of the row), Gray-code order, like the tile kernels; the rows of B of four generations are staged in LDS.
// One lane per row of A, RPT rows per lane, two chunks per three-input XOR.
// ---------------------------------------------------------------------------------------------
template <int NW, int RPT, int NT>
__global__ __launch_bounds__(NT) void gf2_tallskinny4_kernel(const u64 *__restrict__ A, long long lda, const u64 *__restrict__ B,
                                                              long long ldb, u64 *__restrict__ C, long long ldc, int m, int l,
                                                              int n, int accumulate) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int WAVES = NT / 64, EPW = 256 / WAVES, LOWB = Log2<EPW>::value;
  constexpr int COPIES = 4 / NW, EB = 8 * NW;  // copies of a table in an LDS row, bytes per entry
  static_assert(EPW * WAVES == 256 && EPW >= 1 && (NW == 1 || NW == 2 || NW == 4), "geometry");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wl = (l + 63) >> 6, wn = (n + 63) >> 6;
  const u64 maskL = (l & 63) ? ((1ull << (l & 63)) - 1) : ~0ull;
  const u64 maskC = (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
  const long long row_base = (long long)blockIdx.x * (NT * RPT);
  // lookups: chunk order c ^ s, copy cp; NW = 4: lanes with hsw read the upper 16 bytes of an entry first
  const int s = lane & 7, cp = (lane >> 3) & (COPIES - 1);
  const int hsw = NW == 4 ? (lane >> 3) & 1 : 0;
  u32 selc[8], loc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int cs = c ^ s;
    selc[c] = 0x0c0c000cu | ((u32)cs << 8);  // {0, 0, byte cs of the 64-bit word, 0}: v_perm_b32(high dword, low dword, sel)
    loc[c] = (u32)((cs * COPIES + cp) * EB + hsw * 16);
  }
  // build: lane = dword of the 256-byte table row: slot = lane / (2 NW) = COPIES * chunk + copy, dword lane % (2 NW) of the entry
  const int bch = (lane / (2 * NW)) / COPIES, bdw = lane % (2 * NW);

  u32 acc[RPT][2 * NW];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int w = 0; w < 2 * NW; ++w) acc[r][w] = 0;

  // barrier that waits for this wave's LDS operations only: the loads of A stay in flight across it
  auto lds_barrier = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  u64 *bstage = reinterpret_cast<u64 *>(lds + 128 * 1024);  // the 256 rows of B of a block of four generations (2 NW KiB)

  for (int w0 = 0; w0 < wl; w0 += 4) {  // four 64-bit words of the inner dimension = four generations
    // the block's rows of B first (one or two coalesced loads per thread): in-order return then lets the staging wait for
    // them without waiting for the rows of A requested behind them
    constexpr int NBV = (256 * NW + NT - 1) / NT;
    u64 bvals[NBV];
#pragma unroll
    for (int kk = 0; kk < NBV; ++kk) {
      const int idx = tid + kk * NT;
      const long long brow = (long long)w0 * 64 + idx / NW;
      const int w = idx % NW;
      bvals[kk] = (idx < 256 * NW && brow < l && w < wn) ? B[brow * ldb + w] : 0;
    }
    // HALF of this lane's rows are requested before the barriers, the other half behind them: a CU holds a limited number
    // of outstanding misses, and with all RPT rows requested up front the waves that could not issue their loads reached
    // the staging barrier microseconds late (per-wave stamps: first table at 7.5 us instead of 3)
    u64 aw[RPT][4];
    auto load_row = [&](int r) __attribute__((always_inline)) {
      const long long row = min(row_base + (long long)r * NT + tid, (long long)m - 1);  // clamped: stores are guarded
      const u64 *ap = A + row * lda;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        u64 x = ap[min(w0 + q, wl - 1)];
        if (w0 + q >= wl) x = 0;
        if (w0 + q == wl - 1) x &= maskL;
        aw[r][q] = x;
      }
    };
    constexpr int EARLY4 = RPT > 1 ? RPT / 2 : 1;
#pragma unroll
    for (int r = 0; r < EARLY4; ++r) load_row(r);
    lds_barrier();  // every wave is done with the previous block's rows of B
#pragma unroll
    for (int kk = 0; kk < NBV; ++kk)
      if (tid + kk * NT < 256 * NW) bstage[tid + kk * NT] = bvals[kk];
    lds_barrier();
#pragma unroll
    for (int r = EARLY4; r < RPT; ++r) load_row(r);
    static_for<4>([&](auto gtag) __attribute__((always_inline)) {
      constexpr int g = decltype(gtag)::value;
      constexpr u32 tbase = (g & 1) ? 65536u : 0u;  // w0 is a multiple of 4: generation w0 + g lives in buffer g & 1
      if (w0 + g < wl) {                            // uniform
        // ---- build the generation: wave w owns entries [EPW w, EPW (w + 1)) ----
        u32 rr[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) rr[b] = reinterpret_cast<const u32 *>(bstage + (g * 64 + bch * 8 + b) * NW)[bdw];
        u32 cur = 0;
#pragma unroll
        for (int b = LOWB; b < 8; ++b)
          if ((wave >> (b - LOWB)) & 1) cur ^= rr[b];
        {
          constexpr u32 kOff = tbase ? (0x10004u - (u32)(EPW * 256)) : 0u;  // see gf2_m4rm_kernel_v3
          const u32 m0v = tbase + (u32)wave * (u32)(EPW * 256) - kOff;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(m0v) : "memory");  // (one wait state before an LDS add-TID instruction reads M0: the compiler cannot see that the asm below does)
          static_for<EPW>([&](auto itag) __attribute__((always_inline)) {
            constexpr int i = decltype(itag)::value;
            constexpr unsigned e = (unsigned)i ^ ((unsigned)i >> 1);
            if constexpr (i > 0) cur ^= rr[__builtin_ctz(i | 256)];
            asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(cur), "n"(kOff + e * 256u) : "memory");
          });
        }
        lds_barrier();  // the generation is complete; every wave has also finished the lookups of generation - 2 in this buffer
        // ---- lookups ----
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const u32 a0 = (u32)aw[r][g], a1 = (u32)(aw[r][g] >> 32);
          u32 ac[2 * NW];  // (locals: an asm operand inside this lambda cannot name the enclosing function's array)
#pragma unroll
          for (int w = 0; w < 2 * NW; ++w) ac[w] = acc[r][w];
#pragma unroll
          for (int c = 0; c < 8; c += 2) {
            const u32 o0 = __builtin_amdgcn_perm(a1, a0, selc[c]) | loc[c];
            const u32 o1 = __builtin_amdgcn_perm(a1, a0, selc[c + 1]) | loc[c + 1];
            if constexpr (NW == 1) {
              const u32x2v x = *reinterpret_cast<lds_cu32x2 *>(o0), y = *reinterpret_cast<lds_cu32x2 *>(o1);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[0]) : "v"(x.x), "v"(y.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[1]) : "v"(x.y), "v"(y.y));
            } else {
              const u32x4 x = *reinterpret_cast<lds_cu32x4 *>(o0), y = *reinterpret_cast<lds_cu32x4 *>(o1);
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[0]) : "v"(x.x), "v"(y.x));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[1]) : "v"(x.y), "v"(y.y));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[2]) : "v"(x.z), "v"(y.z));
              asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[3]) : "v"(x.w), "v"(y.w));
              if constexpr (NW == 4) {
                const u32x4 x2 = *reinterpret_cast<lds_cu32x4 *>(o0 ^ 16u), y2 = *reinterpret_cast<lds_cu32x4 *>(o1 ^ 16u);
                asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[4]) : "v"(x2.x), "v"(y2.x));
                asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[5]) : "v"(x2.y), "v"(y2.y));
                asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[6]) : "v"(x2.z), "v"(y2.z));
                asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(ac[7]) : "v"(x2.w), "v"(y2.w));
              }
            }
          }
#pragma unroll
          for (int w = 0; w < 2 * NW; ++w) acc[r][w] = ac[w];
        }
      }
      // the next generation lives in the other buffer
#pragma unroll
      for (int c = 0; c < 8; ++c) loc[c] ^= 65536u;
    });
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const long long row = row_base + (long long)r * NT + tid;
    if (row < m) {
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (w < wn) {
          // NW = 4: lanes that read the upper half first hold words 2, 3 in acc[0..3] and words 0, 1 in acc[4..7]
          u32 lo = acc[r][2 * w], hi = acc[r][2 * w + 1];
          if constexpr (NW == 4) {
            lo = hsw ? acc[r][(2 * w) ^ 4] : lo;
            hi = hsw ? acc[r][(2 * w + 1) ^ 4] : hi;
          }
          u64 v = (u64)lo | ((u64)hi << 32);
          if (w == wn - 1) v &= maskC;
          u64 *d = C + row * ldc + w;
          if (accumulate) v ^= *d;
          *d = v;
        }
    }
  }
}

// v*A kernel: C (m x n) (+)= A (m x l) * B (l x n) for a handful of rows m <= 8 (_mzd_mul_va,
// mzd.rs:175-181 and `&v * &A`, binary_matrix.rs:552-563).  Streams B once; the inner dimension
// is split over blockIdx.y and partial sums are combined with 64-bit atomic XOR.
// ---------------------------------------------------------------------------------------------

template <int M>
__global__ __launch_bounds__(256) void gf2_va_kernel(const u64 *__restrict__ A, long long lda,
                                                     const u64 *__restrict__ B, long long ldb, u64 *__restrict__ C,
                                                     long long ldc, int m, int l, int n, int rows_per_split, int wshift) {
  // rows_per_split is a multiple of 64: a split starts on a word of A.  The rows of B are fetched eight at a time (one row
  // per iteration left every load exposed: 8 x 64 x 64 took 26 us, most of it 64 dependent latencies).  A narrow B does not
  // fill a workgroup with column words: its 256 threads are 2^wshift words x (256 >> wshift) interleaved 64-row blocks of
  // the split (8 x 65536 x 256: four of 256 threads had work, 58 us)
  const int lanes_w = 1 << wshift, sub = threadIdx.x >> wshift, nsub = 256 >> wshift;
  const int w = blockIdx.x * lanes_w + (threadIdx.x & (lanes_w - 1));
  const int wn = (n + 63) >> 6;
  const int t0 = blockIdx.y * rows_per_split;
  const int t1 = min(l, t0 + rows_per_split);
  u64 acc[M];
#pragma unroll
  for (int i = 0; i < M; ++i) acc[i] = 0;
  if (w < wn) {
    for (int tb = t0 + 64 * sub; tb < t1; tb += 64 * nsub) {
      u64 aw[M];
#pragma unroll
      for (int i = 0; i < M; ++i) aw[i] = i < m ? A[(long long)i * lda + (tb >> 6)] : 0;  // wave-uniform
      for (int tt = 0; tt < 64 && tb + tt < t1; tt += 8) {
        u64 bw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bw[u] = tb + tt + u < t1 ? B[(long long)(tb + tt + u) * ldb + w] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int i = 0; i < M; ++i) acc[i] ^= bw[u] & (0ull - ((aw[i] >> (tt + u)) & 1ull));
      }
    }
  }
  // the interleaved blocks of a workgroup meet in LDS: one atomic per (workgroup, word, row), not one per thread
  __shared__ u64 red[M][256];
  const int nact = min(nsub, (max(t1 - t0, 0) + 63) >> 6);  // interleaved blocks that had rows at all
  if (nact > 1) {
    if (sub < nact) {
#pragma unroll
      for (int i = 0; i < M; ++i) red[i][threadIdx.x] = acc[i];
    }
    __syncthreads();
    if (sub == 0) {
      for (int s2 = 1; s2 < nact; ++s2)
#pragma unroll
        for (int i = 0; i < M; ++i) acc[i] ^= red[i][threadIdx.x + s2 * lanes_w];
    }
  }
  if (w < wn && sub == 0) {
    const u64 maskC = (n & 63) ? ((1ull << (n & 63)) - 1) : ~0ull;
#pragma unroll
    for (int i = 0; i < M; ++i)
      if (i < m) {
        u64 v = acc[i];
        if (w == wn - 1) v &= maskC;
        if (v) atomicXor(reinterpret_cast<unsigned long long *>(C + (long long)i * ldc + w), (unsigned long long)v);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// streaming helpers
// ---------------------------------------------------------------------------------------------

// C = A ^ B over a rows x words region (mzd_add, mzd.rs:220-223); also copy (B == nullptr) and zero.
__global__ __launch_bounds__(256) void gf2_xor2d_kernel(u64 *__restrict__ C, long long ldc, const u64 *A, long long lda,
                                                        const u64 *B, long long ldb, int rows, int words) {
  const int pairs = (words + 1) >> 1;
  const long long total = (long long)rows * pairs;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / pairs), w = (int)(idx % pairs) * 2;
    if (w + 1 < words) {
      uint4 a = A ? *reinterpret_cast<const uint4 *>(A + (long long)r * lda + w) : make_uint4(0, 0, 0, 0);
      if (B) a = xor4(a, *reinterpret_cast<const uint4 *>(B + (long long)r * ldb + w));
      *reinterpret_cast<uint4 *>(C + (long long)r * ldc + w) = a;
    } else {
      u64 a = A ? A[(long long)r * lda + w] : 0;
      if (B) a ^= B[(long long)r * ldb + w];
      C[(long long)r * ldc + w] = a;
    }
  }
}

// dst (drows x dwords words, dense) = src (srows x swords) in its top left corner, zeros elsewhere: operands padded up to
// dimensions that divide by the Strassen level plan (m4ri_hip_api.cpp, mul_strassen_padded)
// dst (drows x dwords, zero padded) <- src (srows x swords); 16-byte accesses where both rows allow them (ldd, lds_ even and
// 16-byte aligned bases: `vec`), one block row per blockIdx.y step (no division in the loop)
__global__ __launch_bounds__(256) void gf2_padcopy_kernel(u64 *__restrict__ dst, long long ldd, int drows, int dwords,
                                                          const u64 *__restrict__ src, long long lds_, int srows, int swords, int vec) {
  const int pairs = (dwords + 1) >> 1;
  for (int r = blockIdx.y; r < drows; r += gridDim.y) {
    u64 *d = dst + (long long)r * ldd;
    const u64 *sr = src + (long long)r * lds_;
    const bool live = r < srows;
    for (int pr = blockIdx.x * blockDim.x + threadIdx.x; pr < pairs; pr += gridDim.x * blockDim.x) {
      const int w = 2 * pr;
      u64 v0 = 0, v1 = 0;
      if (live && w + 1 < swords && vec) {
        const uint4 t = *reinterpret_cast<const uint4 *>(sr + w);
        v0 = (u64)t.x | ((u64)t.y << 32);
        v1 = (u64)t.z | ((u64)t.w << 32);
      } else if (live) {
        if (w < swords) v0 = sr[w];
        if (w + 1 < swords) v1 = sr[w + 1];
      }
      if (w + 1 < dwords && vec) {
        *reinterpret_cast<uint4 *>(d + w) = make_uint4((u32)v0, (u32)(v0 >> 32), (u32)v1, (u32)(v1 >> 32));
      } else {
        d[w] = v0;
        if (w + 1 < dwords) d[w + 1] = v1;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gf2_fill_random_kernel(u64 *M, long long ld, int rows, int cols, u64 seed,
                                                              long long row0, long long fullw, long long colw0) {
  const int w = (cols + 63) >> 6;
  const u64 tm = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
  const long long total = (long long)rows * w;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / w), j = (int)(idx % w);
    u64 z = seed + ((u64)((row0 + r) * fullw + colw0 + j) + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    if (j == w - 1) z &= tm;
    M[(long long)r * ld + j] = z;
  }
}

// mzd_equal (mzd.rs:187): *diff != 0 iff some valid word differs
__global__ __launch_bounds__(256) void gf2_diff_kernel(const u64 *A, long long lda, const u64 *B, long long ldb, int rows,
                                                       int cols, int *diff) {
  const int w = (cols + 63) >> 6;
  const u64 tm = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
  const long long total = (long long)rows * w;
  int bad = 0;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / w), j = (int)(idx % w);
    u64 x = A[(long long)r * lda + j] ^ B[(long long)r * ldb + j];
    if (j == w - 1) x &= tm;
    bad |= (x != 0);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(diff, 1);
}

// mzd_transpose (mzd.rs:146-148): 64x64 bit blocks.  One wave per block: lane r loads word
// (row 64*bi + r, word bj), the block is transposed with 6 butterfly exchange rounds over lanes,
// lane c stores word (row 64*bj + c, word bi).
__global__ __launch_bounds__(256) void gf2_transpose_kernel(u64 *__restrict__ D, long long ldd, const u64 *__restrict__ S,
                                                            long long lds_, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int wavesPerBlock = blockDim.x >> 6;
  const int bj = blockIdx.x * wavesPerBlock + (threadIdx.x >> 6);  // source word column
  const int bi = blockIdx.y;                                       // source row block
  const int sw = (cols + 63) >> 6, dw = (rows + 63) >> 6;
  if (bj >= sw) return;
  const int r = 64 * bi + lane;
  u64 x = (r < rows) ? S[(long long)r * lds_ + bj] : 0;
  if (bj == sw - 1 && (cols & 63)) x &= (1ull << (cols & 63)) - 1;
  // butterfly transpose: after round with distance d, bit positions and lane indices swap bit log2(d)
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const u64 mask = (d == 32)  ? 0x00000000FFFFFFFFull
                     : (d == 16) ? 0x0000FFFF0000FFFFull
                     : (d == 8)  ? 0x00FF00FF00FF00FFull
                     : (d == 4)  ? 0x0F0F0F0F0F0F0F0Full
                     : (d == 2)  ? 0x3333333333333333ull
                                 : 0x5555555555555555ull;
    const u64 y = __shfl_xor(x, d, 64);
    if (lane & d)
      x = (x & ~mask) | ((y >> d) & mask);
    else
      x = (x & mask) | ((y << d) & ~mask);
  }
  const int orow = 64 * bj + lane;
  if (orow < cols) D[(long long)orow * ldd + bi] = x;
  (void)dw;
}

// ---------------------------------------------------------------------------------------------
// Strassen level kernels (mzd_mul, strassen.rs:8-18).  One pass per level and side:
//   split:  X (2h x 2w words-blocks) -> 7 operands of h rows x w words each
//   merge:  7 products -> the 4 quadrants of C
// Classic Strassen over GF(2) (all signs +):
//   M1=(A11+A22)(B11+B22) M2=(A21+A22)B11 M3=A11(B12+B22) M4=A22(B21+B11) M5=(A11+A12)B22
//   M6=(A21+A11)(B11+B12) M7=(A12+A22)(B21+B22)
//   C11=M1+M4+M5+M7 C12=M3+M5 C21=M2+M4 C22=M1+M2+M3+M6
// Each source word is read once and each destination word written once (11 block-units of traffic).
// blockIdx.z = batch (operands produced by the previous level).
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void gf2_strassen_split_kernel(u64 *__restrict__ dst, long long ldd, long long dstStride,
                                                                 const u64 *__restrict__ src, long long lds_,
                                                                 long long srcStride, int h, int w, int side_) {
  // dst holds 7 consecutive operands per batch element: operand q at dst + (7*b + q)*dstStride
  // side_ 2 (round 4): the A side with its operands written ROW-GROUP PACKED (word c of row r at ((r / 64) w + c) 64 + r % 64), the
  // layout the paired tile kernels read with contiguous loads -- a single level used to hand them unpacked leaves, whose rows of
  // 8192 bits and more run the tile kernel at half its rate (14336^3 with one level: 1.52 ms against 0.55 without any)
  const bool pack = side_ == 2;
  const int side = pack ? 0 : side_;
  const int b = blockIdx.z;
  const u64 *X = src + (long long)b * srcStride;
  u64 *Y = dst + (long long)b * 7 * dstStride;
  const int pairs = w >> 1;
  const long long total = (long long)h * pairs;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int r, c;
    if (pack) {  // a wave takes the 64 rows of a group at one pair of words: its 8-byte stores are 512 consecutive bytes
      const long long rest = idx >> 6;
      r = (int)(rest / pairs) * 64 + (int)(idx & 63);
      c = (int)(rest % pairs) * 2;
    } else {
      r = (int)(idx / pairs), c = (int)(idx % pairs) * 2;
    }
    const uint4 x11 = *reinterpret_cast<const uint4 *>(X + (long long)r * lds_ + c);
    const uint4 x12 = *reinterpret_cast<const uint4 *>(X + (long long)r * lds_ + w + c);
    const uint4 x21 = *reinterpret_cast<const uint4 *>(X + (long long)(r + h) * lds_ + c);
    const uint4 x22 = *reinterpret_cast<const uint4 *>(X + (long long)(r + h) * lds_ + w + c);
    uint4 o[7];
    if (side == 0) {  // A side
      o[0] = xor4(x11, x22);
      o[1] = xor4(x21, x22);
      o[2] = x11;
      o[3] = x22;
      o[4] = xor4(x11, x12);
      o[5] = xor4(x21, x11);
      o[6] = xor4(x12, x22);
    } else {  // B side
      o[0] = xor4(x11, x22);
      o[1] = x11;
      o[2] = xor4(x12, x22);
      o[3] = xor4(x21, x11);
      o[4] = x22;
      o[5] = xor4(x11, x12);
      o[6] = xor4(x21, x22);
    }
    if (pack) {
      const long long pk = ((long long)(r >> 6) * w + c) * 64 + (r & 63);
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        u64 *Yq = Y + q * dstStride;
        Yq[pk] = (u64)o[q].x | ((u64)o[q].y << 32);
        Yq[pk + 64] = (u64)o[q].z | ((u64)o[q].w << 32);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 7; ++q) *reinterpret_cast<uint4 *>(Y + q * dstStride + (long long)r * ldd + c) = o[q];
    }
  }
}
__global__ __launch_bounds__(256) void gf2_strassen_merge_kernel(u64 *__restrict__ dst, long long ldd, long long dstStride,
                                                                 const u64 *__restrict__ src, long long lds_,
                                                                 long long srcStride, int h, int w, int accumulate) {
  const int b = blockIdx.z;
  const u64 *M = src + (long long)b * 7 * srcStride;
  u64 *Cq = dst + (long long)b * dstStride;
  const int pairs = w >> 1;
  const long long total = (long long)h * pairs;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / pairs), c = (int)(idx % pairs) * 2;
    uint4 m[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) m[q] = *reinterpret_cast<const uint4 *>(M + q * srcStride + (long long)r * lds_ + c);
    uint4 c11 = xor4(xor4(m[0], m[3]), xor4(m[4], m[6]));
    uint4 c12 = xor4(m[2], m[4]);
    uint4 c21 = xor4(m[1], m[3]);
    uint4 c22 = xor4(xor4(m[0], m[1]), xor4(m[2], m[5]));
    uint4 *p11 = reinterpret_cast<uint4 *>(Cq + (long long)r * ldd + c);
    uint4 *p12 = reinterpret_cast<uint4 *>(Cq + (long long)r * ldd + w + c);
    uint4 *p21 = reinterpret_cast<uint4 *>(Cq + (long long)(r + h) * ldd + c);
    uint4 *p22 = reinterpret_cast<uint4 *>(Cq + (long long)(r + h) * ldd + w + c);
    if (accumulate) {
      c11 = xor4(c11, *p11);
      c12 = xor4(c12, *p12);
      c21 = xor4(c21, *p21);
      c22 = xor4(c22, *p22);
    }
    *p11 = c11;
    *p12 = c12;
    *p21 = c21;
    *p22 = c22;
  }
}

// ---------------------------------------------------------------------------------------------
// Two Strassen levels in one pass.  A source operand is cut into 4 x 4 sub-blocks S[r][c]; the 49 operands of
// the level below the next are out[7*q1 + q2] = combo_q2(combo_q1(S)).  Each source word is read once and each
// destination word written once: 16 + 49 block units of traffic instead of (4 + 7) + 7*(4 + 7)/4 * ... two passes
// (121 units in the same unit).  The merge is the mirror image: 49 products -> 16 sub-blocks of the parent.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void strassen_combo(const uint4 (&x)[4], uint4 (&o)[7], int side) {
  // x = {X11, X12, X21, X22}; same combinations as gf2_strassen_split_kernel
  if (side == 0) {
    o[0] = xor4(x[0], x[3]);
    o[1] = xor4(x[2], x[3]);
    o[2] = x[0];
    o[3] = x[3];
    o[4] = xor4(x[0], x[1]);
    o[5] = xor4(x[2], x[0]);
    o[6] = xor4(x[1], x[3]);
  } else {
    o[0] = xor4(x[0], x[3]);
    o[1] = x[0];
    o[2] = xor4(x[1], x[3]);
    o[3] = xor4(x[2], x[0]);
    o[4] = x[3];
    o[5] = xor4(x[0], x[1]);
    o[6] = xor4(x[2], x[3]);
  }
}

// side: 0 = A operand, 1 = B operand, 2 = A operand with ROW-GROUP-PACKED output for gf2_m4rm_kernel_v6<.., APACK>: word c of
// row r of an output operand goes to u64 index ((r / 64) * w + c) * 64 + r % 64, i.e. the 64 rows of a group lie side by
// side for every 64-bit column (one 8-byte load per lane of the tile kernel = 512 contiguous bytes).  The threads of a
// wave then take the 64 rows of a group (h % 64 == 0), so that these 8-byte stores are contiguous.
__global__ __launch_bounds__(256) void gf2_strassen_split2_kernel(u64 *__restrict__ dst, long long ldd, long long dstStride,
                                                                  const u64 *__restrict__ src, long long lds_,
                                                                  long long srcStride, int h, int w, int side_) {
  // h, w: rows / words of one OUTPUT operand (a quarter of the source in each dimension)
  const bool pack = side_ == 2;
  const int side = pack ? 0 : side_;
  const int b = blockIdx.z;
  const u64 *X = src + (long long)b * srcStride;
  u64 *Y = dst + (long long)b * 49 * dstStride;
  const int pairs = w >> 1;
  const long long total = (long long)h * pairs;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    int r, c;
    if (pack) {
      const long long rest = idx >> 6;
      r = (int)(rest / pairs) * 64 + (int)(idx & 63);
      c = (int)(rest % pairs) * 2;
    } else {
      r = (int)(idx / pairs);
      c = (int)(idx % pairs) * 2;
    }
    // level-1 combination applied to each of the four inner positions (ir, ic)
    uint4 l1[4][7];
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      const int ir = ip >> 1, ic = ip & 1;
      uint4 x[4];
#pragma unroll
      for (int op = 0; op < 4; ++op) {
        const int orow = op >> 1, ocol = op & 1;
        x[op] = *reinterpret_cast<const uint4 *>(X + (long long)(r + (2 * orow + ir) * h) * lds_ + (2 * ocol + ic) * w + c);
      }
      strassen_combo(x, l1[ip], side);
    }
    const long long pk = ((long long)(r >> 6) * w + c) * 64 + (r & 63);
#pragma unroll
    for (int q1 = 0; q1 < 7; ++q1) {
      const uint4 x[4] = {l1[0][q1], l1[1][q1], l1[2][q1], l1[3][q1]};
      uint4 o[7];
      strassen_combo(x, o, side);
#pragma unroll
      for (int q2 = 0; q2 < 7; ++q2) {
        u64 *Yq = Y + (long long)(7 * q1 + q2) * dstStride;
        if (pack) {
          Yq[pk] = (u64)o[q2].x | ((u64)o[q2].y << 32);
          Yq[pk + 64] = (u64)o[q2].z | ((u64)o[q2].w << 32);
        } else {
          *reinterpret_cast<uint4 *>(Yq + (long long)r * ldd + c) = o[q2];
        }
      }
    }
  }
}

__device__ __forceinline__ void strassen_fold(const uint4 (&m)[7], uint4 (&c)[4]) {
  // c = {C11, C12, C21, C22}
  c[0] = xor4(xor4(m[0], m[3]), xor4(m[4], m[6]));
  c[1] = xor4(m[2], m[4]);
  c[2] = xor4(m[1], m[3]);
  c[3] = xor4(xor4(m[0], m[1]), xor4(m[2], m[5]));
}

__global__ __launch_bounds__(256) void gf2_strassen_merge2_kernel(u64 *__restrict__ dst, long long ldd, long long dstStride,
                                                                  const u64 *__restrict__ src, long long lds_,
                                                                  long long srcStride, int h, int w, int accumulate) {
  // h, w: rows / words of one INPUT product (a quarter of the destination in each dimension)
  const int b = blockIdx.z;
  const u64 *M = src + (long long)b * 49 * srcStride;
  u64 *Cq = dst + (long long)b * dstStride;
  const int pairs = w >> 1;
  const long long total = (long long)h * pairs;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / pairs), c = (int)(idx % pairs) * 2;
    uint4 inner[7][4];  // inner[q1] = the four quadrants of level-1 product q1
#pragma unroll
    for (int q1 = 0; q1 < 7; ++q1) {
      uint4 m[7];
#pragma unroll
      for (int q2 = 0; q2 < 7; ++q2)
        m[q2] = *reinterpret_cast<const uint4 *>(M + (long long)(7 * q1 + q2) * srcStride + (long long)r * lds_ + c);
      strassen_fold(m, inner[q1]);
    }
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {
      const int ir = ip >> 1, ic = ip & 1;
      const uint4 m[7] = {inner[0][ip], inner[1][ip], inner[2][ip], inner[3][ip], inner[4][ip], inner[5][ip], inner[6][ip]};
      uint4 o[4];
      strassen_fold(m, o);
#pragma unroll
      for (int op = 0; op < 4; ++op) {
        const int orow = op >> 1, ocol = op & 1;
        uint4 *pd = reinterpret_cast<uint4 *>(Cq + (long long)(r + (2 * orow + ir) * h) * ldd + (2 * ocol + ic) * w + c);
        uint4 v = o[op];
        if (accumulate) v = xor4(v, *pd);
        *pd = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Three Strassen levels in one pass, with an optional "virtual" fourth level on top.
// A source operand is cut into 8 x 8 sub-blocks x[i1][i2][i3] (i = 2 * row half + column half at each level, i1 the
// coarsest); operand 49 q1 + 7 q2 + q3 of the third level below is T(q1) T(q2) T(q3) applied along the three axes, T being
// the 7 x 4 combination table of the side (two non-zeros per row at most).  A thread owns ONE 64-bit word position of the
// output operands: it loads the 64 source words of that position (128 registers), and walks q1 -> 16 values, q2 -> 4
// values, q3 -> one word to store.  Each source word is read once and each destination word written once: 64 + 343 block
// units, against (16 + 49) + 49 (16 + 49) / 16 = 264 for two fused-pair passes in the same unit -- the intermediate
// level is never written or re-read.
// Virtual level: blockIdx.y = q0 picks ONE or TWO quadrants of the grandparent matrix (src0[q0], src1[q0] or null) whose
// XOR is the source operand, so that four levels cost one read of (12 / 4 of) the matrix and one write of the 2401 leaves.
// Accesses are 8 bytes per lane (512 contiguous bytes per wave and stream): the register budget is what sets the width.
// PACK: row-group-packed output for the paired tile kernels (see gf2_strassen_split2_kernel).
// ---------------------------------------------------------------------------------------------

struct gf2k_split3_srcs {
  const u64 *a[7];
  const u64 *b[7];  // second quadrant of the virtual level's combination, or nullptr
};

// quadrants (0 = X11, 1 = X12, 2 = X21, 3 = X22) that combination q of a side adds up; second entry -1: a plain copy
__device__ constexpr int kStrassenSupp[2][7][2] = {
    {{0, 3}, {2, 3}, {0, -1}, {3, -1}, {0, 1}, {2, 0}, {1, 3}},   // A side: A11+A22, A21+A22, A11, A22, A11+A12, A21+A11, A12+A22
    {{0, 3}, {0, -1}, {1, 3}, {2, 0}, {3, -1}, {0, 1}, {2, 3}}};  // B side: B11+B22, B11, B12+B22, B21+B11, B22, B11+B12, B21+B22

template <int SIDE, bool PACK, bool NT = false>
__global__ __launch_bounds__(256) void gf2_strassen_split3_kernel(u64 *__restrict__ dst, long long ldd, long long dstStride,
                                                                  const gf2k_split3_srcs srcs, long long lds_,
                                                                  long long srcStride, int h, int w) {
  // h, w: rows / words of one OUTPUT operand (an eighth of the source operand in each dimension)
  const int b = blockIdx.z, g = blockIdx.y;
  const u64 *X0 = srcs.a[g] + (long long)b * srcStride;
  const u64 *X1 = srcs.b[g] ? srcs.b[g] + (long long)b * srcStride : nullptr;
  u64 *Y = dst + ((long long)b * gridDim.y + g) * 343 * dstStride;
  const long long total = (long long)h * w;
  // PACK with w % 16 == 0: a workgroup takes a tile of 16 rows x 16 words -- wave v the words 4v..4v+3, lane L row L % 16 of
  // word L / 16 -- so that its reads are whole 128-byte lines (a lane-per-row mapping touches 64 lines per load and ran at
  // half the bandwidth) and its packed stores 128-byte runs (16 rows of one 64-bit column); tiles in packed-address order
  const bool tiled = PACK && (w & 15) == 0;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    int r, c;
    if (tiled) {
      const long long tile = idx >> 8;           // 256 positions per tile
      const int t = (int)(idx & 255);
      const int rb = (int)(tile & 3);            // 16-row block inside the 64-row group
      const long long rest = tile >> 2;
      const int cb = (int)(rest % (w >> 4));
      const int grp = (int)(rest / (w >> 4));
      r = grp * 64 + rb * 16 + (t & 15);
      c = cb * 16 + (t >> 4);
    } else if (PACK) {
      const long long rest = idx >> 6;
      r = (int)(rest / w) * 64 + (int)(idx & 63);
      c = (int)(rest % w);
    } else {
      r = (int)(idx / w);
      c = (int)(idx % w);
    }
    u64 x[64];
    static_for<64>([&](auto K) {
      constexpr int k = decltype(K)::value;
      constexpr int R = 4 * ((k >> 5) & 1) + 2 * ((k >> 3) & 1) + ((k >> 1) & 1);
      constexpr int Cc = 4 * ((k >> 4) & 1) + 2 * ((k >> 2) & 1) + (k & 1);
      x[k] = X0[(long long)(r + R * h) * lds_ + (long long)Cc * w + c];
    });
    if (X1) {
      static_for<64>([&](auto K) {
        constexpr int k = decltype(K)::value;
        constexpr int R = 4 * ((k >> 5) & 1) + 2 * ((k >> 3) & 1) + ((k >> 1) & 1);
        constexpr int Cc = 4 * ((k >> 4) & 1) + 2 * ((k >> 2) & 1) + (k & 1);
        x[k] ^= X1[(long long)(r + R * h) * lds_ + (long long)Cc * w + c];
      });
    }
    const long long at = PACK ? ((long long)(r >> 6) * w + c) * 64 + (r & 63) : (long long)r * ldd + c;
    static_for<7>([&](auto Q1) {
      constexpr int q1 = decltype(Q1)::value;
      constexpr int a1 = kStrassenSupp[SIDE][q1][0], b1 = kStrassenSupp[SIDE][q1][1];
      u64 y[16];
      static_for<16>([&](auto J) {
        constexpr int j = decltype(J)::value;
        y[j] = b1 >= 0 ? (x[a1 * 16 + j] ^ x[(b1 < 0 ? 0 : b1) * 16 + j]) : x[a1 * 16 + j];
      });
      static_for<7>([&](auto Q2) {
        constexpr int q2 = decltype(Q2)::value;
        constexpr int a2 = kStrassenSupp[SIDE][q2][0], b2 = kStrassenSupp[SIDE][q2][1];
        u64 z[4];
        static_for<4>([&](auto J) {
          constexpr int j = decltype(J)::value;
          z[j] = b2 >= 0 ? (y[a2 * 4 + j] ^ y[(b2 < 0 ? 0 : b2) * 4 + j]) : y[a2 * 4 + j];
        });
        static_for<7>([&](auto Q3) {
          constexpr int q3 = decltype(Q3)::value;
          constexpr int a3 = kStrassenSupp[SIDE][q3][0], b3 = kStrassenSupp[SIDE][q3][1];
          const u64 o = b3 >= 0 ? (z[a3] ^ z[b3 < 0 ? 0 : b3]) : z[a3];
          if (NT) __builtin_nontemporal_store(o, &Y[(long long)(49 * q1 + 7 * q2 + q3) * dstStride + at]);
          else Y[(long long)(49 * q1 + 7 * q2 + q3) * dstStride + at] = o;
        });
      });
    });
  }
}

// quadrants of C (0 = C11, 1 = C12, 2 = C21, 3 = C22) that product q is added to; -1: none further
//   C11 = M1+M4+M5+M7, C12 = M3+M5, C21 = M2+M4, C22 = M1+M2+M3+M6
__device__ constexpr int kStrassenFoldTo[7][2] = {{0, 3}, {2, 3}, {1, 3}, {0, 2}, {0, 1}, {3, -1}, {0, -1}};

// The mirror image: the 343 products three levels down -> the 8 x 8 sub-blocks of their great-grandparent product.  A
// thread owns one 64-bit word position: it loads the 343 product words as it goes (seven at a time) and keeps the 64
// results in registers.  blockIdx.y = g: several independent parents per batch element (the seven level-1 products of a
// four-level plan), parent (b * gridDim.y + g) at dst + that * dstStride.
template <bool NTL>
__global__ __launch_bounds__(256) void gf2_strassen_merge3_kernel(u64 *__restrict__ dst, long long ldd, long long dstStride,
                                                                  const u64 *__restrict__ src, long long lds_,
                                                                  long long srcStride, int h, int w, int accumulate) {
  // h, w: rows / words of one INPUT product (an eighth of the destination in each dimension)
  const long long parent = (long long)blockIdx.z * gridDim.y + blockIdx.y;
  const u64 *M = src + parent * 343 * srcStride;
  u64 *Cq = dst + parent * dstStride;
  const long long total = (long long)h * w;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(idx / w), c = (int)(idx % w);
    const u64 *Mp = M + (long long)r * lds_ + c;
    u64 z[64];
    static_for<64>([&](auto K) { z[decltype(K)::value] = 0; });
    // q1 is a run-time loop on purpose (fully unrolled, the scheduler hoists all 343 loads and spills): its contribution
    // to the four coarse quadrants is applied under all-ones / zero masks, so register indices stay static
#pragma unroll 1
    for (int q1 = 0; q1 < 7; ++q1) {
      u64 y[16];
      static_for<16>([&](auto J) { y[decltype(J)::value] = 0; });
      const u64 *Mq = Mp + (long long)(49 * q1) * srcStride;
      static_for<7>([&](auto Q2) {
        constexpr int q2 = decltype(Q2)::value;
        u64 m[7];
        static_for<7>([&](auto Q3) {
          constexpr int q3 = decltype(Q3)::value;
          m[q3] = NTL ? __builtin_nontemporal_load(&Mq[(long long)(7 * q2 + q3) * srcStride]) : Mq[(long long)(7 * q2 + q3) * srcStride];
        });
        const u64 f[4] = {m[0] ^ m[3] ^ m[4] ^ m[6], m[2] ^ m[4], m[1] ^ m[3], m[0] ^ m[1] ^ m[2] ^ m[5]};
        constexpr int t0 = kStrassenFoldTo[q2][0], t1 = kStrassenFoldTo[q2][1];
        static_for<4>([&](auto J) {
          constexpr int j = decltype(J)::value;
          y[t0 * 4 + j] ^= f[j];
          if (t1 >= 0) y[(t1 < 0 ? 0 : t1) * 4 + j] ^= f[j];
        });
      });
      const int u0 = kStrassenFoldTo[q1][0], u1 = kStrassenFoldTo[q1][1];
      static_for<4>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const u64 mask = (u == u0 || u == u1) ? ~0ull : 0ull;
        static_for<16>([&](auto J) {
          constexpr int j = decltype(J)::value;
          z[u * 16 + j] ^= y[j] & mask;
        });
      });
    }
    static_for<64>([&](auto K) {
      constexpr int k = decltype(K)::value;
      constexpr int R = 4 * ((k >> 5) & 1) + 2 * ((k >> 3) & 1) + ((k >> 1) & 1);
      constexpr int Cc = 4 * ((k >> 4) & 1) + 2 * ((k >> 2) & 1) + (k & 1);
      u64 *pd = Cq + (long long)(r + R * h) * ldd + (long long)Cc * w + c;
      u64 v = z[k];
      if (accumulate) v ^= *pd;
      *pd = v;
    });
  }
}

#ifdef GF2K_DEV_VARIANTS
// ---------------------------------------------------------------------------------------------
// B (l x n, row-major) -> chunk-packed layout of the tile kernel (see gf2_m4rm_kernel_v3, BPACK).
// thread <-> (chunk c, column tile tn, lane d): reads dword d of rows 8c..8c+7 (a wave reads 256 contiguous bytes of
// each row), writes its 32 bytes (a wave writes one whole 2 KiB block).  Rows >= l, columns >= n and the nc - ceil(l/8)
// padding chunks are written as zeros.  blockIdx.z = batch.
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void gf2_packB_kernel(u32 *__restrict__ Bp, long long bpStride, const u64 *__restrict__ B,
                                                        long long ldb, long long bStride, int l, int n, int nc, int tiles_n) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tn = blockIdx.y;
  if (c >= nc) return;
  const u32 *B32 = reinterpret_cast<const u32 *>(B + (long long)blockIdx.z * bStride);
  const int ndw = ((n + 63) >> 6) * 2;  // dwords per row that exist
  const int dcol = tn * 64 + lane;
  const u32 tail = (n & 31) ? ((1u << (n & 31)) - 1u) : 0xffffffffu;
  u32 v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = c * 8 + r;
    u32 x = 0;
    if (row < l && dcol < ndw) {
      x = B32[(long long)row * ldb * 2 + dcol];
      if (dcol == ((n + 31) >> 5) - 1) x &= tail;
      if (dcol > ((n + 31) >> 5) - 1) x = 0;
    }
    v[r] = x;
  }
  uint4 *dst = reinterpret_cast<uint4 *>(Bp + ((long long)blockIdx.z * bpStride + ((long long)tn * nc + c) * 512) + lane * 8);
  dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
  dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
}

#endif  // GF2K_DEV_VARIANTS

// ---------------------------------------------------------------------------------------------
// launchers (internal C ABI used by m4ri_hip_api.cpp)
// ---------------------------------------------------------------------------------------------

static inline int grid_for(long long total, int block = 256, int cap = 256 * 8) {
  long long g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// variants: 9 / 10 / 11 / 12 = v8 with 4096 / 2048 / 1024 / 512-row tiles (512 columns); 21 / 22 / 23 = v9 with 4096 / 2048 / 1024-row
// tiles of 128 columns; 8x = v6; 90-99 = the legacy v7 (development builds)
static int cfg_v8_rg(int cfg) {
#ifdef GF2K_DEV_VARIANTS
  if (cfg >= 13 && cfg <= 16) return 8 >> (cfg - 13);  // read window of three steps (kbench A/B)
  if (cfg >= 17 && cfg <= 19) return 8 >> (cfg - 17);  // read window of one step
#endif
  return cfg == 9 ? 8 : cfg == 10 ? 4 : cfg == 11 ? 2 : cfg == 12 ? 1 : 0;
}
static int cfg_v9_rg(int cfg) {  // the tall-narrow experiment (tools/gf2_kernels_v9_experiment.inc): development builds only
#ifdef GF2K_DEV_VARIANTS
  return cfg == 21 ? 8 : cfg == 22 ? 4 : cfg == 23 ? 2 : 0;
#else
  (void)cfg;
  return 0;
#endif
}
static bool cfg_is_v7(int cfg) { return cfg_v8_rg(cfg) > 0 || (cfg >= 90 && cfg < 100); }
static bool cfg_is_v56(int cfg) { return cfg == 8 || (cfg >= 80 && cfg < 90); }
extern "C" int gf2k_m4rm_rows_per_tile(int cfg) {
  if (cfg_v8_rg(cfg)) return 512 * cfg_v8_rg(cfg);
  if (cfg_v9_rg(cfg)) return 512 * cfg_v9_rg(cfg);
  return (cfg == 1 || cfg == 20) ? 256 : cfg_is_v7(cfg) ? 4096 : cfg_is_v56(cfg) ? 2048 : 1024;
}
extern "C" int gf2k_m4rm_cols_per_tile(int cfg) { return cfg_v9_rg(cfg) ? 128 : cfg_is_v7(cfg) ? 512 : cfg_is_v56(cfg) ? 1024 : 2048; }
extern "C" long long gf2k_m4rm_streamk_words(int cfg, int nseg) {
  if (cfg_v9_rg(cfg)) return 2ll * nseg * 512 * cfg_v9_rg(cfg) * 2;
  return cfg_v8_rg(cfg) ? 2ll * nseg * 512 * cfg_v8_rg(cfg) * 8 : 0;
}

// The dynamic-LDS limit of a kernel is per device; hipFuncSetAttribute costs host time that short kernels launched back
// to back notice, so it is issued once per (kernel, device).
static hipError_t lds_limit_once(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void *, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({kernel, dev})) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert({kernel, dev});
  return e;
}

template <typename K>
static hipError_t launch_tile_kernel(K kernel, int threads, const gf2k_mul_args &a, long long nwg, hipStream_t stream) {
  hipError_t e = lds_limit_once(reinterpret_cast<const void *>(kernel), kLdsBytes);
  if (e != hipSuccess) {
    fprintf(stderr, "gf2k: hipFuncSetAttribute failed: %s\n", hipGetErrorString(e));
    return e;
  }
  hipLaunchKernelGGL(kernel, dim3((unsigned)nwg), dim3(threads), kLdsBytes, stream, a);
  e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "gf2k: tile kernel launch failed: %s\n", hipGetErrorString(e));
  return e;
}

// v8 (cfg 9-12) and v9 (cfg 21-23): tiles, the stream-K split the caller asked for, the tile kernel, the reduction of the partial tiles
static hipError_t launch_v8(gf2k_mul_args a, int cfg, int RG, hipStream_t stream) {
  const bool v9 = cfg_v9_rg(cfg) != 0;
  const int R = 512 * RG, TC = v9 ? 128 : 512;
  a.tiles_m = (a.m + R - 1) / R;
  a.tiles_n = (a.n + TC - 1) / TC;
  const long long T = (long long)a.tiles_m * a.tiles_n * a.batch;
  if (T > 0x7fffff00LL) return hipErrorInvalidValue;
  const int nw32 = (a.l + 31) / 32, Q = v9 ? (nw32 + 3) / 4 : (nw32 + 1) / 2;  // units of the inner dimension: 128 / 64 bits
  a.tile_slabs = Q;
  // the uniform split-K of the older kernels, expressed as a stream-K split of all tiles
  if (a.ksplit > 1 && a.n_rem <= 0) a.n_rem = (int)T, a.nseg = (int)std::min<long long>(T * a.ksplit, 1 << 20);
  a.ksplit = 1;
  a.kwords = 0;
#ifdef GF2K_DEV_VARIANTS
  static const int v8_flags = getenv("GF2K_V8_FLAGS") ? atoi(getenv("GF2K_V8_FLAGS")) : 0;  // kbench A/B: 1 = s_setprio 1 for waves 4-7
  a.kwords = v8_flags;
#endif
  long long n_rem = a.P && Q > 0 ? std::min<long long>(std::max(a.n_rem, 0), T) : 0;
  int nseg = 0, seg = 0;
  if (n_rem > 0) {
    const long long gtot = n_rem * Q, want = a.nseg > 0 ? a.nseg : 256;
    long long sg = (gtot + want - 1) / want;
    if (sg > Q) sg = Q;  // a segment spans at most two tiles
    if (sg < 1) sg = 1;
    const long long ns = (gtot + sg - 1) / sg;
    if (ns > 0x7fffff00LL - T || gf2k_m4rm_streamk_words(cfg, (int)ns) > a.p_words || (ns <= n_rem && sg == Q)) {
      n_rem = 0;  // no room for the partial tiles (or nothing would be split): whole tiles
    } else {
      nseg = (int)ns;
      seg = (int)sg;
    }
  }
  a.n_rem = (int)n_rem;
  a.nseg = nseg;
  a.seg_slabs = seg;
  a.n_full = (int)(T - n_rem);
  a.sP = (long long)R * (v9 ? 2 : 8);
  const long long nwg = (long long)a.n_full + nseg;
  if (nwg <= 0) return hipSuccess;
  hipError_t e = hipErrorInvalidValue;
#define GF2K_V8G(RGv, Gv)                                                                              \
  e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v8<RGv, Gv, 1>, 512, a, nwg, stream)            \
                 : launch_tile_kernel(&gf2_m4rm_kernel_v8<RGv, Gv, 0>, 512, a, nwg, stream)
#define GF2K_V8(RGv) GF2K_V8G(RGv, 2)
#define GF2K_V9(RGv)                                                                                   \
  e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v9<RGv, 2, 1>, 512, a, nwg, stream)             \
                 : launch_tile_kernel(&gf2_m4rm_kernel_v9<RGv, 2, 0>, 512, a, nwg, stream)
#ifdef GF2K_DEV_VARIANTS
  if (v9) {
    switch (RG) {
      case 8: GF2K_V9(8); break;
      case 4: GF2K_V9(4); break;
      default: GF2K_V9(2); break;
    }
  } else if (cfg >= 13 && cfg <= 16) {
    switch (RG) {
      case 8: GF2K_V8G(8, 3); break;
      case 4: GF2K_V8G(4, 3); break;
      case 2: GF2K_V8G(2, 3); break;
      default: GF2K_V8G(1, 3); break;
    }
  } else if (cfg >= 17 && cfg <= 19) {
    switch (RG) {
      case 8: GF2K_V8G(8, 1); break;
      case 4: GF2K_V8G(4, 1); break;
      default: GF2K_V8G(2, 1); break;
    }
  } else
#endif
  switch (RG) {
    case 8: GF2K_V8(8); break;
    case 4: GF2K_V8(4); break;
    case 2: GF2K_V8(2); break;
    default: GF2K_V8(1); break;
  }
#undef GF2K_V9
#undef GF2K_V8
#undef GF2K_V8G
  if (e != hipSuccess || n_rem == 0) return e;
#ifdef GF2K_DEV_VARIANTS
  if (v9) {
    hipLaunchKernelGGL(gf2_streamk_reduce9_kernel, dim3((unsigned)(n_rem * (R / 256))), dim3(256), 0, stream, a, RG);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(gf2_streamk_reduce_kernel, dim3((unsigned)(n_rem * (R / 64))), dim3(256), 0, stream, a, RG);
  return hipGetLastError();
}

// cfg (shipped): 7 = v3 1024 x 2048 tile, 20 = v3 256 x 2048 (4 waves), 8 = v6 2048 x 1024, 81 / 82 = v6 with a deeper /
// shallower read window, 9 / 10 / 11 / 12 = v8 with 4096 / 2048 / 1024 / 512-row tiles (launch_v8).  Everything else -- the first-generation kernels 0 / 1, v5 (80), packed B (50)
// and the timing-only ablations whose results are wrong by design (40-45, 49, 83-89, 92-96) -- exists only in builds
// with -DGF2K_DEV_VARIANTS (tools/libm4ri_hip_dev.so for tools/kbench) and is hipErrorInvalidValue in libm4ri_hip.so.
extern "C" hipError_t gf2k_m4rm(gf2k_mul_args a, int cfg, hipStream_t stream) {
  if (a.m <= 0 || a.n <= 0 || a.batch <= 0) return hipSuccess;
  if (a.a_packed && cfg != 8 && !cfg_is_v7(cfg) && !cfg_v9_rg(cfg)) return hipErrorInvalidValue;  // only v6 / v7 / v8 / v9 read the packed layout
  if (const int RG = cfg_v8_rg(cfg)) return launch_v8(a, cfg, RG, stream);
  if (const int RG = cfg_v9_rg(cfg)) return launch_v8(a, cfg, RG, stream);
  const int R = gf2k_m4rm_rows_per_tile(cfg);
  a.tiles_m = (a.m + R - 1) / R;
  const int TC = gf2k_m4rm_cols_per_tile(cfg);
  a.tiles_n = (a.n + TC - 1) / TC;
  const int nw32 = (a.l + 31) / 32;
  if (cfg == 0 || cfg == 1 || a.ksplit < 1) a.ksplit = 1;  // first-generation kernels have no split-K
  if (a.ksplit > nw32) a.ksplit = nw32 > 0 ? nw32 : 1;
  a.kwords = (nw32 + a.ksplit - 1) / a.ksplit;
  if (cfg == 8 || (cfg > 80 && cfg < 100)) a.kwords = (a.kwords + 1) & ~1;  // v6 reads A in 64-bit slabs: slices start at even words
  a.ksplit = a.kwords > 0 ? (nw32 + a.kwords - 1) / a.kwords : 1;  // no empty slices
  if (a.ksplit <= 1 || cfg == 0 || cfg == 1 || (a.ldp & 1)) a.P = nullptr;
  if (a.ksplit > 1 && !a.accumulate && !a.P) {  // slices are combined with atomic XOR: start from zero
    for (int b = 0; b < a.batch; ++b) {
      hipError_t e = gf2k_xor2d(a.C + (long long)b * a.sC, a.ldc, nullptr, 0, nullptr, 0, a.m, (a.n + 63) / 64, stream);
      if (e != hipSuccess) return e;
    }
  }
  const long long nwg = (long long)a.tiles_m * a.tiles_n * a.batch * a.ksplit;
  if (nwg > 0x7fffffffLL) return hipErrorInvalidValue;
  hipError_t e = hipErrorInvalidValue;
  switch (cfg) {
    case 7: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4>, 512, a, nwg, stream); break;
    case 20: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<4, 64, 4>, 256, a, nwg, stream); break;
    case 8:  // paired chunks, one row per lane
      e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 0, 1>, 512, a, nwg, stream)
                     : launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4>, 512, a, nwg, stream);
      break;
    case 81: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 6>, 512, a, nwg, stream); break;
    case 82: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 3>, 512, a, nwg, stream); break;
#ifdef GF2K_DEV_VARIANTS
    case 0: e = launch_tile_kernel(&gf2_m4rm_kernel<8, 128>, 512, a, nwg, stream); break;
    case 1: e = launch_tile_kernel(&gf2_m4rm_kernel<4, 64>, 256, a, nwg, stream); break;
    case 90:  // the round-1/2 kernel v7 (fixed 4096 x 512 tile, uniform split-K)
      e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2, 0, 1>, 512, a, nwg, stream)
                     : launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2>, 512, a, nwg, stream);
      break;
    // timing-only ablations of v7 on packed A (wrong results by design)
    case 92: e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2, 2, 1>, 512, a, nwg, stream) : hipErrorInvalidValue; break;  // no barriers
    case 93: e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2, 3, 1>, 512, a, nwg, stream) : hipErrorInvalidValue; break;  // no loads in the loop
    case 94: e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2, 4, 1>, 512, a, nwg, stream) : hipErrorInvalidValue; break;  // no table writes
    case 95: e = a.a_packed ? launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2, 6, 1>, 512, a, nwg, stream) : hipErrorInvalidValue; break;  // no XORs
    case 96: e = (a.a_packed && a.Bp) ? launch_tile_kernel(&gf2_m4rm_kernel_v7<8, 2, 1, 1>, 512, a, nwg, stream) : hipErrorInvalidValue; break;  // phase stamps -> Bp
    case 80: e = launch_tile_kernel(&gf2_m4rm_kernel_v5<8, 256, 4>, 512, a, nwg, stream); break;  // paired chunks, 8 lanes per row
    case 86: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 2>, 512, a, nwg, stream); break;  // timing only: no barriers
    case 87: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 3>, 512, a, nwg, stream); break;  // timing only: no loads in the loop
    case 88: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 4>, 512, a, nwg, stream); break;  // timing only: no table writes
    case 89: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 6>, 512, a, nwg, stream); break;  // timing only: no XORs
    case 85: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 9>, 512, a, nwg, stream); break;  // timing only: coalesced (wrong) A loads
    case 83: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 7>, 512, a, nwg, stream); break;  // timing only: no A loads in the loop
    case 84: e = launch_tile_kernel(&gf2_m4rm_kernel_v6<8, 4, 8>, 512, a, nwg, stream); break;  // timing only: no B loads in the loop
    case 50: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 0, 0, 1>, 512, a, nwg, stream); break;  // packed B
    case 40: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 2>, 512, a, nwg, stream); break;  // no barriers (timing only)
    case 41: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 3>, 512, a, nwg, stream); break;  // no loads in the loop
    case 42: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 4>, 512, a, nwg, stream); break;  // no build xor chain
    case 44: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 6>, 512, a, nwg, stream); break;  // no B loads in the loop
    case 45: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 7>, 512, a, nwg, stream); break;  // no A loads in the loop
    case 43: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 4, 5>, 512, a, nwg, stream); break;  // none of the three
    case 49: e = launch_tile_kernel(&gf2_m4rm_kernel_v3<8, 128, 6, 1>, 512, a, nwg, stream); break;  // section stamps
#endif
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || !a.P) return e;
  const int words = (a.n + 63) / 64;
  const long long total = (long long)a.m * ((words + 1) / 2) * a.batch;
  hipLaunchKernelGGL(gf2_splitk_reduce_kernel, dim3(grid_for(total)), dim3(256), 0, stream, a.C, a.ldc, a.sC, a.P, a.ldp, a.sP,
                     a.ksplit, a.m, words, a.batch, a.accumulate);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_packA(u64 *dst, long long wp, const u64 *src, long long lds_, int m, int w, hipStream_t stream) {
  if (m <= 0 || w <= 0) return hipSuccess;
  if (wp < w || (wp & 1)) return hipErrorInvalidValue;
  const long long total = (long long)((m + 63) >> 6) * (wp >> 1) * 64;
  hipLaunchKernelGGL(gf2_packA_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dst, wp, src, lds_, m, w);
  return hipGetLastError();
}

#ifdef GF2K_DEV_VARIANTS
extern "C" hipError_t gf2k_dbg_sec(unsigned long long *out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(gf2_dbg_sec), 8 * sizeof(unsigned long long));
}
#endif

#ifdef GF2K_DEV_VARIANTS
// chunks per tile column of a packed operand with inner dimension l (4 padding chunks: the kernel prefetches two
// chunks past the last 32-bit word)
extern "C" int gf2k_packB_chunks(int l) { return ((l + 31) / 32) * 4 + 4; }

// bpStride: dwords between batch elements of Bp (>= tiles_n * gf2k_packB_chunks(l) * 512)
extern "C" hipError_t gf2k_packB(uint32_t *Bp, long long bpStride, const u64 *B, long long ldb, long long bStride, int l,
                                 int n, int batch, hipStream_t stream) {
  if (l <= 0 || n <= 0 || batch <= 0) return hipSuccess;
  const int nc = gf2k_packB_chunks(l), tiles_n = (n + 2047) / 2048;
  hipLaunchKernelGGL(gf2_packB_kernel, dim3((nc + 3) / 4, tiles_n, batch), dim3(256), 0, stream, Bp, bpStride, B, ldb,
                     bStride, l, n, nc, tiles_n);
  return hipGetLastError();
}
#endif

extern "C" hipError_t gf2k_rowparity(const u64 *A, long long lda, const u64 *Bt, long long ldbt, u64 *C, long long ldc,
                                     int m, int l, int n, int accumulate, hipStream_t stream) {
  if (m <= 0 || n <= 0) return hipSuccess;
  const int wl = (l + 63) >> 6;
  dim3 grid((m + 255) / 256, (n + 63) / 64), block(256);
  const bool vec_ok = (lda % 2 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  if (wl <= 1)
    hipLaunchKernelGGL((gf2_rowparity_kernel<1>), grid, block, 0, stream, A, lda, Bt, ldbt, C, ldc, m, l, n, accumulate);
  else if (wl <= 2 && vec_ok)
    hipLaunchKernelGGL((gf2_rowparity_kernel<2>), grid, block, 0, stream, A, lda, Bt, ldbt, C, ldc, m, l, n, accumulate);
  else if (wl <= 4 && vec_ok)
    hipLaunchKernelGGL((gf2_rowparity_kernel<4>), grid, block, 0, stream, A, lda, Bt, ldbt, C, ldc, m, l, n, accumulate);
  else if (wl <= 8 && vec_ok)
    hipLaunchKernelGGL((gf2_rowparity_kernel<8>), grid, block, 0, stream, A, lda, Bt, ldbt, C, ldc, m, l, n, accumulate);
  else
    hipLaunchKernelGGL((gf2_rowparity_kernel<0>), grid, block, 0, stream, A, lda, Bt, ldbt, C, ldc, m, l, n, accumulate);
  return hipGetLastError();
}

// n <= 64 and n * ceil(l/64) * 8 <= 64 KiB of LDS; returns hipErrorInvalidValue otherwise (caller takes the two-kernel path)
extern "C" hipError_t gf2k_narrow(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m,
                                  int l, int n, int accumulate, hipStream_t stream) {
  if (m <= 0 || n <= 0) return hipSuccess;
  const int wl = (l + 63) >> 6;
  const size_t lds = (size_t)n * wl * 8;
  if (n > 64 || lds > 65536 || l <= 0) return hipErrorInvalidValue;
  long long blocks = ((long long)m + 255) / 256;
  static const int cap_env = GF2K_DEV_ENV("M4RI_HIP_NARROW_BLOCKS", 0);  // (A/B measurements)
  if (blocks > (cap_env > 0 ? cap_env : 2048)) blocks = cap_env > 0 ? cap_env : 2048;  // grid-stride: every block pays the B transpose once
  dim3 grid((unsigned)blocks), block(256);
  const bool vec_ok = (lda % 2 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  if (wl <= 1)
    hipLaunchKernelGGL((gf2_narrow_kernel<1>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate);
  else if (wl <= 2 && vec_ok)
    hipLaunchKernelGGL((gf2_narrow_kernel<2>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate);
  else if (wl <= 4 && vec_ok)
    hipLaunchKernelGGL((gf2_narrow_kernel<4>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate);
  else if (wl <= 8 && vec_ok)
    hipLaunchKernelGGL((gf2_narrow_kernel<8>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate);
  else
    hipLaunchKernelGGL((gf2_narrow_kernel<0>), grid, block, lds, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate);
  return hipGetLastError();
}

// n <= 32 (Bt: n rows of l bits, row stride ldbt words); any l, meant for l > 512.  The n result bits of a row go to bits
// jshift .. jshift + n - 1 of its word of C (jshift > 0 only with accumulate: the second half of a 64-vector block).
extern "C" hipError_t gf2k_widevec(const u64 *A, long long lda, const u64 *Bt, long long ldbt, u64 *C, long long ldc, int m,
                                   int l, int n, int accumulate, int jshift, hipStream_t stream) {
  if (m <= 0 || n <= 0) return hipSuccess;
  if (n > 32 || l <= 0 || jshift < 0 || jshift + n > 64 || (jshift && !accumulate)) return hipErrorInvalidValue;
  const int wl = (l + 63) >> 6;
  const int NJ = n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : n <= 16 ? 16 : 32;
  int slab = (128 * 1024) / (NJ * 8);  // words of the inner dimension per slab: 128 KiB of LDS
  if (slab > wl) slab = (wl + 63) & ~63;
  const size_t lds = (size_t)NJ * slab * 8;
  const int threads = 1024, nwaves = threads / 64;
  long long blocks = ((long long)m + nwaves - 1) / nwaves;
  if (blocks > 512) blocks = 512;
  hipError_t e = hipSuccess;
#define GF2K_WV(NJv)                                                                                          \
  e = lds_limit_once(reinterpret_cast<const void *>(&gf2_widevec_kernel<NJv>), 128 * 1024);                   \
  if (e == hipSuccess)                                                                                        \
    hipLaunchKernelGGL((gf2_widevec_kernel<NJv>), dim3((unsigned)blocks), dim3(threads), lds, stream, A, lda, Bt, ldbt, C, ldc, m, l, \
                       n, accumulate, slab, jshift)
  switch (NJ) {
    case 1: GF2K_WV(1); break;
    case 2: GF2K_WV(2); break;
    case 4: GF2K_WV(4); break;
    case 8: GF2K_WV(8); break;
    case 16: GF2K_WV(16); break;
    default: GF2K_WV(32); break;
  }
#undef GF2K_WV
  return e != hipSuccess ? e : hipGetLastError();
}

// n <= 128 (one or two words per row of C), any l and m: 4-bit tables per 512-bit slab, the inner dimension divided among workgroups.  With more than one
// division the partial words are XORed into C atomically: C must then hold the value to accumulate into (the launcher
// zeroes it for a plain product).
extern "C" hipError_t gf2k_tallskinny_long(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m,
                                           int l, int n, int accumulate, hipStream_t stream) {
  if (m <= 0 || n <= 0) return hipSuccess;
  if (n > 128 || l <= 0) return hipErrorInvalidValue;
  constexpr int RPT = 4;
  const int nwC = (n + 63) >> 6;
  const int nslabs = (l + 511) >> 9;
  const long long rblocks = ((long long)m + 256 * RPT - 1) / (256 * RPT);
  static const int want = GF2K_DEV_ENV("M4RI_HIP_TS7_BLOCKS", 8192);  // (65536^2 x 64: 0.21 ms with 1024 workgroups, 0.17 with 8192)
  // (two words per row of C: fewer divisions -- the atomics on a C with 32-byte rows cost more: 65536^2 x 256 0.81 ms with 8192 workgroups, 0.53 with 2048)
  const int want_eff = nwC == 1 ? want : (want + 3) / 4;
  long long splits = rblocks >= want_eff ? 1 : (want_eff + rblocks - 1) / rblocks;
  if (splits > nslabs) splits = nslabs;
  const int sps = (int)((nslabs + splits - 1) / splits);
  splits = (nslabs + sps - 1) / sps;
  const int atomic = splits > 1;
  if (atomic && !accumulate) {
    hipError_t e = gf2k_xor2d(C, ldc, nullptr, 0, nullptr, 0, m, nwC, stream);  // zero the word(s) of every row
    if (e != hipSuccess) return e;
  }
  const int vec_ok = (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const size_t lds7 = (size_t)(128 * 16 * 8 + 512 * 8) * nwC;
  // many divisions and row blocks: the workgroups take them in the XCD-blocked, diagonal order (see the kernel)
  static const int remap_on = GF2K_DEV_ENV("M4RI_HIP_TS7_REMAP", 1);
  const bool remap = remap_on && splits >= 16 && rblocks >= 8 && ((splits + 15) / 16) * ((rblocks + 7) / 8) * 128 < (1ll << 30);
  const dim3 grid = remap ? dim3((unsigned)(((splits + 15) / 16) * ((rblocks + 7) / 8) * 128)) : dim3((unsigned)rblocks, (unsigned)splits);
  const int rs = remap ? (int)splits : 0, rb = remap ? (int)rblocks : 0;
  if (nwC == 1)
    hipLaunchKernelGGL((gf2_tallskinny7_kernel<RPT, 1>), grid, dim3(256), lds7, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate, sps, atomic,
                       vec_ok, rs, rb);
  else
    hipLaunchKernelGGL((gf2_tallskinny7_kernel<RPT, 2>), grid, dim3(256), lds7, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate, sps, atomic,
                       vec_ok, rs, rb);
  return hipGetLastError();
}

// n <= 256; returns hipErrorInvalidValue otherwise
// Which kernel takes a tall-skinny product (n <= 256; measured cold at 2^20 x 256, us):
//   l <= 256, n <= 64        gf2_tallskinny6_kernel<1>   4-bit tables, small streaming workgroups       11.8  (generation kernel 15.0)
//   l <= 256, 64 < n <= 128  gf2_tallskinny5_kernel<2>   8-bit tables, whole rows read once, one phase  14.7  (4-bit form 16.0)
//   l <= 256, 128 < n        gf2_tallskinny5_kernel<4>   the same in two phases, 512 threads x 8 rows   20.9  (4-bit form 35.7)
//   256 < l, n <= 64         gf2_tallskinny4_kernel<1>   generations of 64 bits of l, one built while the previous is looked up
//   256 < l, 64 < n          gf2_tallskinny3_kernel<NW>  row sets of 32 / NW tables, rows re-fetched per row set
// (The caller keeps l <= 1024: beyond that the tile kernel with split-K is faster.)  M4RI_HIP_TS6 = mask of the entry widths
// (1, 2, 4 words) the 4-bit kernel takes instead (default 1); M4RI_HIP_TALLSKINNY_OLD=1 sends l <= 256 to the round-1 kernels (A/B runs).
// `side` (optional): the table-free kernel for one to four vectors also packs bit j of every 64 rows' results into word
// side[j * side_ld + row / 64] -- the transposed form of the product, for free (a ballot per vector and 64 rows); other kernels cannot
// (hipErrorNotSupported: the caller transposes C itself).  A, C and side may be PINNED HOST blocks (gf2k_tallskinny_side's caller).
static hipError_t tallskinny_impl(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m, int l, int n,
                                  int accumulate, u64 *side, long long side_ld, hipStream_t stream);
extern "C" hipError_t gf2k_tallskinny(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m,
                                      int l, int n, int accumulate, hipStream_t stream) {
  return tallskinny_impl(A, lda, B, ldb, C, ldc, m, l, n, accumulate, nullptr, 0, stream);
}
extern "C" hipError_t gf2k_tallskinny_side(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m,
                                           int l, int n, u64 *side, long long side_ld, hipStream_t stream) {
  if (!side || n > 4 || l > 256 || l <= 64) return hipErrorNotSupported;
  return tallskinny_impl(A, lda, B, ldb, C, ldc, m, l, n, 0, side, side_ld, stream);
}
static hipError_t tallskinny_impl(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m, int l, int n,
                                  int accumulate, u64 *side, long long side_ld, hipStream_t stream) {
  if (m <= 0 || n <= 0) return hipSuccess;
  if (n > 256 || l <= 0) return hipErrorInvalidValue;
  const int nw = (n + 63) / 64;
  static const int old_only = GF2K_DEV_ENV("M4RI_HIP_TALLSKINNY_OLD", 0);
#ifdef GF2K_DEV_VARIANTS
  static const int ts6 = GF2K_DEV_ENV("M4RI_HIP_TS6", 1);
#endif
  // ---- l <= 256 (round 4): the single-phase streaming kernels of gf2_lpn.inc.  Measured cold at 2^20 x 256 (tools/lpn_lab, us):
  //   n <= 64         gf2_lpn8_kernel<1>    8-bit tables, 8-byte entries, 2 workgroups per CU     8.7-8.9   (4-bit kernel 11.2-13.5)
  //   64 < n <= 128   gf2_lpn8_kernel<2>    8-bit tables, 16-byte entries                         11.3      (14.5-15.1)
  //   128 < n <= 256  gf2_lpn256_kernel     6/6/6/7/7-bit fields, 32-byte entries, ONE phase      17.3-17.7 (two phases: 20.0-21.4)
  // The RPT row steps of a workgroup are taken in GRID-STRIDE order (step r of workgroup b = rows (r * grid + b) * 512 ...): the launch
  // then reads one contiguous window of A and writes one of C at any time, like a plain stream (cold, one box: V = 64 8.88 -> 8.60 us,
  // V = 128 11.26 -> 10.64, V = 256 17.61 -> 17.27; the load / store skeleton of V = 256 alone 13.75 -> 13.06).
  // Rows per workgroup = 512 x RPT with RPT chosen so that the launch has about one workgroup per CU (two for n <= 64): with
  // few rows a batch of 4096 per workgroup leaves most of the chip idle (65536 x 256 x 256: 14.5 us with RPT = 8, 5.4 with 1).
  // M4RI_HIP_LPN=0 restores the round-2/3 kernels (A/B runs).
  static const int lpn = GF2K_DEV_ENV("M4RI_HIP_LPN", 1);
  if (l <= 256 && !old_only && lpn) {
    const bool a16 = (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    const bool c16 = nw == 1 || ((ldc & 1) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0);
    const int mode = (l > 192 && a16 && c16) ? (lda == 4 ? 2 : 1) : 0;
    hipError_t e = hipSuccess;
    // one to four vectors: no tables (gf2_lpnvec_kernel); 256 rows x RPT per workgroup, four workgroups per CU wanted
    static const int vec_on = GF2K_DEV_ENV("M4RI_HIP_LPNVEC", 1);
    if (n <= 4 && vec_on) {
      // (2^20 x 256 x 1 cold on one box: rows per lane 4 / 2 / 1 with >= 1024 workgroups 8.5 / 8.5 / 9.1 us, 8 with 512 or 256
      // workgroups 8.7 / 8.6; the table kernel 9.0.  Warm, A in the Infinity Cache: 5.7 against 7.5 us.)
      static const int vrpt = GF2K_DEV_ENV("M4RI_HIP_LPNVEC_RPT", 4);
      int rpt = vrpt;
      static const int vwant = GF2K_DEV_ENV("M4RI_HIP_LPNVEC_GRID", 1024);
      while (rpt > 1 && ((long long)m + 256LL * rpt - 1) / (256LL * rpt) < vwant) rpt >>= 1;
      const unsigned grid = (unsigned)(((long long)m + 256LL * rpt - 1) / (256LL * rpt));
#define GF2_LPNVEC_GO(NVV, RPTV, MODEV)                                                                                \
  hipLaunchKernelGGL((gf2_lpnvec_kernel<NVV, 256, RPTV, MODEV, GF2_LPNVEC_STRIDED>), dim3(grid), dim3(256), 0, stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate, \
                     side, side_ld)
#define GF2_LPNVEC_RPT(NVV, MODEV)                                                                                     \
  do {                                                                                                                 \
    if (rpt >= 4) GF2_LPNVEC_GO(NVV, 4, MODEV);                                                                        \
    else if (rpt == 2) GF2_LPNVEC_GO(NVV, 2, MODEV);                                                                   \
    else GF2_LPNVEC_GO(NVV, 1, MODEV);                                                                                 \
  } while (0)
#define GF2_LPNVEC_MODE(NVV)                                                                                           \
  do {                                                                                                                 \
    if (mode == 2) GF2_LPNVEC_RPT(NVV, 2);                                                                             \
    else if (mode == 1) GF2_LPNVEC_RPT(NVV, 1);                                                                        \
    else GF2_LPNVEC_RPT(NVV, 0);                                                                                       \
  } while (0)
      if (n == 1) GF2_LPNVEC_MODE(1);
      else if (n == 2) GF2_LPNVEC_MODE(2);
      else GF2_LPNVEC_MODE(4);
#undef GF2_LPNVEC_MODE
#undef GF2_LPNVEC_RPT
#undef GF2_LPNVEC_GO
      return hipGetLastError();
    }
    const int target = nw == 1 ? 512 : 256;  // workgroups wanted
    int rpt = nw == 1 ? 4 : 8;
    while (rpt > 1 && ((long long)m + 512LL * rpt - 1) / (512LL * rpt) < target) rpt >>= 1;
    const unsigned grid = (unsigned)(((long long)m + 512LL * rpt - 1) / (512LL * rpt));
#define GF2_LPN_GO(KERNEL, LDSB)                                                                                       \
  do {                                                                                                                 \
    e = lds_limit_once(reinterpret_cast<const void *>(&KERNEL), (int)(LDSB));                                          \
    if (e != hipSuccess) return e;                                                                                     \
    hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(512), (LDSB), stream, A, lda, B, ldb, C, ldc, m, l, n, accumulate);    \
  } while (0)
#define GF2_LPN_RPT(NWV, MODEV)                                                                                        \
  do {                                                                                                                 \
    if (NWV == 2 && rpt == 8) GF2_LPN_GO((gf2_lpn8_kernel<2, 512, 8, MODEV, 3, 0, 1>), kLpn8LdsBytes(2)); /* (n <= 64 starts at 4) */ \
    else if (rpt == 4) GF2_LPN_GO((gf2_lpn8_kernel<NWV, 512, 4, MODEV, 3, 0, 1>), kLpn8LdsBytes(NWV));                        \
    else if (rpt == 2) GF2_LPN_GO((gf2_lpn8_kernel<NWV, 512, 2, MODEV, 2, 0, 1>), kLpn8LdsBytes(NWV));                        \
    else GF2_LPN_GO((gf2_lpn8_kernel<NWV, 512, 1, MODEV, 1, 0, 1>), kLpn8LdsBytes(NWV));                                      \
  } while (0)
#define GF2_LPN256_RPT(MODEV)                                                                                          \
  do {                                                                                                                 \
    if (rpt == 8) GF2_LPN_GO((gf2_lpn256_kernel<512, 8, MODEV, 3, 0, 13>), kLpn256LdsBytes);                             \
    else if (rpt == 4) GF2_LPN_GO((gf2_lpn256_kernel<512, 4, MODEV, 3, 0, 13>), kLpn256LdsBytes);                        \
    else if (rpt == 2) GF2_LPN_GO((gf2_lpn256_kernel<512, 2, MODEV, 2, 0, 13>), kLpn256LdsBytes);                        \
    else GF2_LPN_GO((gf2_lpn256_kernel<512, 1, MODEV, 1, 0, 13>), kLpn256LdsBytes);                                      \
  } while (0)
    if (nw == 1) {
      if (mode == 2) GF2_LPN_RPT(1, 2);
      else if (mode == 1) GF2_LPN_RPT(1, 1);
      else GF2_LPN_RPT(1, 0);
    } else if (nw == 2) {
      if (mode == 2) GF2_LPN_RPT(2, 2);
      else if (mode == 1) GF2_LPN_RPT(2, 1);
      else GF2_LPN_RPT(2, 0);
    } else {
      if (mode == 2) GF2_LPN256_RPT(2);
      else if (mode == 1) GF2_LPN256_RPT(1);
      else GF2_LPN256_RPT(0);
    }
#undef GF2_LPN256_RPT
#undef GF2_LPN_RPT
#undef GF2_LPN_GO
    return hipGetLastError();
  }
#ifdef GF2K_DEV_VARIANTS  // (reached with M4RI_HIP_LPN=0 in development builds only)
  if (l <= 256 && !old_only && (ts6 & (nw == 3 ? 4 : nw))) {
    const int vec_ok = l > 192 && (lda & 1) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    long long blocks = ((long long)m + 255) / 256;
    static const int cap_env = GF2K_DEV_ENV("M4RI_HIP_TS6_BLOCKS", 0);  // (A/B measurements)
    const long long cap = cap_env > 0 ? cap_env : (nw <= 2 ? 2048 : 1024);  // 8 (4) workgroups per CU
    if (blocks > cap) blocks = cap;
#define GF2_TS6_LAUNCH(NWV)                                                                                                       \
  do {                                                                                                                            \
    const size_t lds6 = 64 * 16 * 8 * NWV + 256 * 8 * NWV;                                                                        \
    hipLaunchKernelGGL((gf2_tallskinny6_kernel<NWV>), dim3((unsigned)blocks), dim3(256), lds6, stream, A, lda, B, ldb, C, ldc, m, l, n, \
                       accumulate, vec_ok);                                                                                       \
  } while (0)
    if (nw == 1) GF2_TS6_LAUNCH(1);
    else if (nw == 2) GF2_TS6_LAUNCH(2);
    else GF2_TS6_LAUNCH(4);
#undef GF2_TS6_LAUNCH
    return hipGetLastError();
  }
  if (nw >= 2 && l <= 256 && !old_only) {  // every row of A read once (gf2_tallskinny5_kernel)
    const size_t lds5 = 128 * 1024 + 256 * 8 * (nw <= 2 ? 2 : 4);
    const bool full = l > 192 && ((lda | ldc) & 1) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(C)) & 15) == 0;
    hipError_t e5 = hipSuccess;
#define GF2_TS5_LAUNCH(NWV, RPTV, NTV, FULLV, EV)                                                                             \
  do {                                                                                                                      \
    e5 = lds_limit_once(reinterpret_cast<const void *>(&gf2_tallskinny5_kernel<NWV, RPTV, NTV, FULLV, false, EV, true>), (int)lds5); \
    if (e5 != hipSuccess) return e5;                                                                                        \
    const unsigned grid5 = (unsigned)(((long long)m + NTV * RPTV - 1) / (NTV * RPTV));                                      \
    hipLaunchKernelGGL((gf2_tallskinny5_kernel<NWV, RPTV, NTV, FULLV, false, EV, true>), dim3(grid5), dim3(NTV), lds5, stream, A, lda, B, ldb, C, \
                       ldc, m, l, n, accumulate);                                                                           \
  } while (0)
    // rows requested before the first build (EARLY) / the others one row ahead of the lookups: per-wave stamps, cold A
    if (nw == 2 && full) GF2_TS5_LAUNCH(2, 4, 1024, true, 1);
    else if (nw == 2) GF2_TS5_LAUNCH(2, 4, 1024, false, 1);
    else if (full) GF2_TS5_LAUNCH(4, 8, 512, true, 2);
    else GF2_TS5_LAUNCH(4, 8, 512, false, 2);
#undef GF2_TS5_LAUNCH
    return hipGetLastError();
  }
#endif
  if (nw == 1) {  // generation kernel (replicated tables, skew inside a 64-bit word)
#ifndef GF2K_DEV_VARIANTS
    // Every shape that used to come here (n <= 64, 256 < l <= 1024, at least 2^19 rows) is taken by the slab table kernel first
    // (ts_long_shape, m4ri_hip_api.cpp), so the shipped library does not carry an instantiation no call can reach
    // (tools/kernel_coverage.py, retirement rule of DESIGN 4.1); development builds keep it for A/B runs (M4RI_HIP_TS7=0).
    return hipErrorInvalidValue;
#else
    constexpr int RPT4 = 4, NT4 = 1024;
    const unsigned grid4 = (unsigned)(((long long)m + NT4 * RPT4 - 1) / (NT4 * RPT4));
    const size_t lds4 = 128 * 1024 + 8 * 1024;
    hipError_t e4 = lds_limit_once(reinterpret_cast<const void *>(&gf2_tallskinny4_kernel<1, RPT4, NT4>), (int)lds4);
    if (e4 != hipSuccess) return e4;
    hipLaunchKernelGGL((gf2_tallskinny4_kernel<1, RPT4, NT4>), dim3(grid4), dim3(NT4), lds4, stream, A, lda, B, ldb, C, ldc, m, l, n,
                       accumulate);
    return hipGetLastError();
#endif
  }
  constexpr int RPT3 = 4, NT3 = 1024;
  const unsigned grid3 = (unsigned)(((long long)m + NT3 * RPT3 - 1) / (NT3 * RPT3));
  const size_t lds3 = 128 * 1024 + 4 * 1024;
  hipError_t e3;
#define GF2_TS3_LAUNCH(NWV)                                                                                              \
  e3 = lds_limit_once(reinterpret_cast<const void *>(&gf2_tallskinny3_kernel<NWV, RPT3, NT3>), (int)lds3);              \
  if (e3 != hipSuccess) return e3;                                                                                       \
  hipLaunchKernelGGL((gf2_tallskinny3_kernel<NWV, RPT3, NT3>), dim3(grid3), dim3(NT3), lds3, stream, A, lda, B, ldb, C, ldc, m, \
                     l, n, accumulate)
  if (nw == 2) {
    GF2_TS3_LAUNCH(2);
  } else {
    GF2_TS3_LAUNCH(4);
  }
#undef GF2_TS3_LAUNCH
  return hipGetLastError();
}

#ifdef GF2K_DEV_VARIANTS
// development: gf2_tallskinny5_kernel with wall-clock stamps (8 u64 per wave: start, then per phase: built, first row, last row)
extern "C" hipError_t gf2k_tallskinny5_dbg(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m,
                                           int l, int n, u64 *stamps, hipStream_t stream) {
  const int nw = (n + 63) / 64;
  const size_t lds5 = (nw == 1 ? 64 : 128) * 1024 + 256 * 8 * nw;
  const int variant = getenv("TS5_VARIANT") ? atoi(getenv("TS5_VARIANT")) : 0;  // 10 * EARLY + AHEAD
#define TS5_DBG(NWV, RPTV, NTV, EARLYV, AHEADV)                                                                                   \
  do {                                                                                                                            \
    hipError_t e = lds_limit_once(reinterpret_cast<const void *>(&gf2_tallskinny5_kernel<NWV, RPTV, NTV, true, true, EARLYV, AHEADV>), \
                                  (int)lds5);                                                                                     \
    if (e != hipSuccess) return e;                                                                                                \
    hipLaunchKernelGGL((gf2_tallskinny5_kernel<NWV, RPTV, NTV, true, true, EARLYV, AHEADV>), dim3((m + NTV * RPTV - 1) / (NTV * RPTV)), \
                       dim3(NTV), lds5, stream, A, lda, B, ldb, C, ldc, m, l, n, 0, stamps);                                      \
  } while (0)
  if (nw == 1) {
    switch (variant) {
      case 111: TS5_DBG(1, 8, 512, 1, true); break;
      case 121: TS5_DBG(1, 8, 512, 2, true); break;
      case 211: TS5_DBG(1, 4, 512, 1, true); break;
      case 221: TS5_DBG(1, 4, 512, 2, true); break;
      case 240: TS5_DBG(1, 4, 512, 4, false); break;
      case 311: TS5_DBG(1, 2, 512, 1, true); break;
      case 320: TS5_DBG(1, 2, 512, 2, false); break;
      default: TS5_DBG(1, 4, 1024, 1, true); break;
    }
  } else if (nw == 2) {
    switch (variant) {
      case 10: TS5_DBG(2, 4, 1024, 1, false); break;
      case 11: TS5_DBG(2, 4, 1024, 1, true); break;
      case 20: TS5_DBG(2, 4, 1024, 2, false); break;
      case 21: TS5_DBG(2, 4, 1024, 2, true); break;
      case 31: TS5_DBG(2, 4, 1024, 3, true); break;
      case 111: TS5_DBG(2, 8, 512, 1, true); break;
      case 121: TS5_DBG(2, 8, 512, 2, true); break;
      case 131: TS5_DBG(2, 8, 512, 3, true); break;
      case 141: TS5_DBG(2, 8, 512, 4, true); break;
      default: TS5_DBG(2, 4, 1024, 1, false); break;
    }
  } else {
    switch (variant) {
      case 10: TS5_DBG(4, 8, 512, 1, false); break;
      case 11: TS5_DBG(4, 8, 512, 1, true); break;
      case 21: TS5_DBG(4, 8, 512, 2, true); break;
      case 31: TS5_DBG(4, 8, 512, 3, true); break;
      case 41: TS5_DBG(4, 8, 512, 4, true); break;
      default: TS5_DBG(4, 8, 512, 2, false); break;
    }
  }
#undef TS5_DBG
  return hipGetLastError();
}
#endif

extern "C" hipError_t gf2k_va(const u64 *A, long long lda, const u64 *B, long long ldb, u64 *C, long long ldc, int m, int l,
                              int n, hipStream_t stream) {
  // C must already hold the value to accumulate into (zero for a plain product)
  if (m <= 0 || n <= 0 || l <= 0) return hipSuccess;
  const int wn = (n + 63) >> 6;
  int wshift = 8;  // threads of a workgroup along the words of a row of B: 256, or the power of two that covers a narrow row
  while (wshift > 0 && (1 << (wshift - 1)) >= wn) --wshift;
  const int lanes_w = 1 << wshift, nsub = 256 >> wshift;
  const int gx = (wn + lanes_w - 1) / lanes_w;
  // aim at ~2048 blocks, but at no less than 16 KiB of B per block (every block ends in atomics on the few words of a narrow C)
  long long want = ((long long)l * wn * 8) >> 14;
  want = want < 64 ? 64 : want > 2048 ? 2048 : want;
  int splits = (int)((want + gx - 1) / gx);
  const int unit = 64 * nsub;         // a split gives every interleaved block of 64 rows its share
  int rps = ((l + splits - 1) / splits + unit - 1) / unit * unit;
  if (rps < unit) rps = unit;
  splits = (l + rps - 1) / rps;
  dim3 grid(gx, splits), block(256);
  for (int i0 = 0; i0 < m; i0 += 8) {
    const int mm = (m - i0 < 8) ? (m - i0) : 8;
    hipLaunchKernelGGL((gf2_va_kernel<8>), grid, block, 0, stream, A + (long long)i0 * lda, lda, B, ldb,
                       C + (long long)i0 * ldc, ldc, mm, l, n, rps, wshift);
  }
  return hipGetLastError();
}

extern "C" hipError_t gf2k_xor2d(u64 *C, long long ldc, const u64 *A, long long lda, const u64 *B, long long ldb, int rows,
                                 int words, hipStream_t stream) {
  if (rows <= 0 || words <= 0) return hipSuccess;
  const long long total = (long long)rows * ((words + 1) / 2);
  hipLaunchKernelGGL(gf2_xor2d_kernel, dim3(grid_for(total)), dim3(256), 0, stream, C, ldc, A, lda, B, ldb, rows, words);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_padcopy(u64 *dst, long long ldd, int drows, int dwords, const u64 *src, long long lds_, int srows,
                                   int swords, hipStream_t stream) {
  if (drows <= 0 || dwords <= 0) return hipSuccess;
  const int vec = !((ldd | lds_) & 1) && !(((uintptr_t)dst | (uintptr_t)src) & 15);
  const int pairs = (dwords + 1) >> 1;
  const unsigned gx = (unsigned)std::min(8, (pairs + 255) / 256), gy = (unsigned)std::min(drows, 65535);
  hipLaunchKernelGGL(gf2_padcopy_kernel, dim3(gx, gy), dim3(256), 0, stream, dst, ldd, drows, dwords, src, lds_, srows, swords, vec);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_fill_random(u64 *M, long long ld, int rows, int cols, u64 seed, long long row0, long long fullw,
                                       long long colw0, hipStream_t stream) {
  if (fullw <= 0) fullw = (cols + 63) / 64;
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const long long total = (long long)rows * ((cols + 63) / 64);
  hipLaunchKernelGGL(gf2_fill_random_kernel, dim3(grid_for(total)), dim3(256), 0, stream, M, ld, rows, cols, seed, row0, fullw, colw0);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_diff(const u64 *A, long long lda, const u64 *B, long long ldb, int rows, int cols, int *diff,
                                hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const long long total = (long long)rows * ((cols + 63) / 64);
  hipLaunchKernelGGL(gf2_diff_kernel, dim3(grid_for(total)), dim3(256), 0, stream, A, lda, B, ldb, rows, cols, diff);
  return hipGetLastError();
}

// The same transposition with both sides coalesced: a workgroup of 8 waves takes a 512 x 512-bit tile.  Wave w loads source
// rows [64 w, 64 w + 64) of the tile 64 bytes per row at a time (lanes 0-7 the eight words of a row: eight whole 64-byte
// segments per load, where the kernel above touched 64 lines for 8 bytes each -- 65536^2 took 1.1 ms, 0.9 TB/s), hands them
// to the lane-per-row layout of the butterfly through LDS (rows padded to 72 bytes: conflict-free both ways), transposes its
// eight 64 x 64 blocks in registers and puts word w of the 512 output rows into LDS; the output rows then leave 64 bytes
// at a time as well.
// Round 4: what bounded the first form of this kernel was its instruction stream, not its bytes -- rocprofv3 SQ counters on
// 65536^2 (profiles/r04_transpose_counters.txt): 995 VALU + 1147 SALU + 128 LDS instructions per wave and tile (a 64-bit
// __shfl_xor is two ds_bpermute_b32, and the lane-dependent 64-bit select around it about twenty more instructions, 48 times
// per wave), a wave alive for 43600 cycles per tile of which 62 % parked on s_waitcnt / barriers (nothing of the next tile was
// in flight while this one was shuffled).  Now the six butterfly rounds of a 64 x 64 block are register operations: distance
// 32 and 16 are ONE v_permlane32_swap_b32 / v_permlane16_swap_b32 on the two halves that change places (gfx950), distances
// 8, 4, 2, 1 a DPP move (row_ror:8; row_half_mirror + quad_perm for 4; quad_perm for 2 and 1), a per-lane rotate and a
// v_bfi_b32 per dword -- 32 VALU instructions per block, no LDS traffic --, the staging of a wave's own rows needs no workgroup
// barrier (only the exchange of the transposed blocks between the waves does), and the two exchanges share one 36 KiB buffer so
// that four workgroups fit a CU: 65536^2 0.61 -> 0.36 ms.
__device__ __forceinline__ u32 tr_dpp_xor(u32 y, int d) {
  if (d == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)y, 0x128, 0xf, 0xf, true);   // row_ror:8: lane i <- lane i ^ 8 of its row
  if (d == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)y, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
  if (d == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)y, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
  const u32 m = (u32)__builtin_amdgcn_update_dpp(0, (int)y, 0x141, 0xf, 0xf, true);        // row_half_mirror: i <- 7 - i = i ^ 7
  return (u32)__builtin_amdgcn_update_dpp(0, (int)m, 0x1B, 0xf, 0xf, true);                // quad_perm [3,2,1,0]: ^ 3, together ^ 4
}

// transposes the 64 x 64 bit block whose row L is {lo, hi} of lane L (lo = columns 0-31)
__device__ __forceinline__ void tr_block64(u32 &lo, u32 &hi, const u32 (&keep)[4], const u32 (&rot)[4]) {
  // distance 32: columns 32-63 of lanes 0-31 <-> columns 0-31 of lanes 32-63
  // (builtins, not inline asm: the swaps need two wait states after a VALU write of an operand, which the compiler pads only around
  // instructions it can see -- the asm form of the 16-lane swap right behind the two v_perm_b32 read a stale register: wrong rows
  // from 16 on in every block, caught by test_dev_transpose_large_tiles)
  {
    const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
    lo = r[0], hi = r[1];
  }
  // distance 16: split by column bit 4 (p0 = the low halves of both dwords), swap p0 of rows 16-31 / 48-63 with p1 of rows 0-15 / 32-47.
  // The remaining rounds act inside 16-bit groups, so they run on (p0, p1) as they are and the halves are put back at the end.
  u32 p0 = __builtin_amdgcn_perm(hi, lo, 0x05040100u), p1 = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
  {
    const auto r = __builtin_amdgcn_permlane16_swap(p0, p1, false, false);
    p0 = r[0], p1 = r[1];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = 8 >> k;
    // a lane whose bit d is set keeps the bits at column positions with bit d set and takes the others from its partner's
    // upper positions (partner >> d); the other lane the mirror image.  rot = d or 32 - d: the rotation's wrapped bits fall
    // where `keep` selects the lane's own value
    const u32 y0 = tr_dpp_xor(p0, d), y1 = tr_dpp_xor(p1, d);
    const u32 z0 = __builtin_amdgcn_alignbit(y0, y0, rot[k]), z1 = __builtin_amdgcn_alignbit(y1, y1, rot[k]);
    p0 = (p0 & keep[k]) | (z0 & ~keep[k]);
    p1 = (p1 & keep[k]) | (z1 & ~keep[k]);
  }
  lo = __builtin_amdgcn_perm(p1, p0, 0x05040100u);
  hi = __builtin_amdgcn_perm(p1, p0, 0x07060302u);
}

__global__ __launch_bounds__(512, 8) void gf2_transpose512_kernel(u64 *__restrict__ D, long long ldd, const u64 *__restrict__ S,
                                                                  long long lds_, int rows, int cols, int ntiles_padded, int flags) {
  // ONE 36 KiB buffer for both exchanges (four workgroups per CU): rows 64 w .. 64 w + 63 are wave w's private staging rows on
  // the way in and the output rows wave w stores on the way out; in between every wave writes its word of all 512 output rows,
  // after the barrier that also says every wave has taken its staged rows out.
  __shared__ u64 sbuf[512][9];
  u64 (*sin)[64][9] = reinterpret_cast<u64 (*)[64][9]>(sbuf);
  u64 (*sout)[9] = sbuf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rr = lane >> 3, w = lane & 7;
  const int sw = (cols + 63) >> 6, dwn = (rows + 63) >> 6;
  const u64 maskS = (cols & 63) ? ((1ull << (cols & 63)) - 1) : ~0ull;
  // per-lane constants of the four in-row rounds (distance 8, 4, 2, 1)
  u32 keep[4], rot[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = 8 >> k;
    const u32 mask = d == 8 ? 0x00FF00FFu : d == 4 ? 0x0F0F0F0Fu : d == 2 ? 0x33333333u : 0x55555555u;
    keep[k] = (lane & d) ? ~mask : mask;
    rot[k] = (lane & d) ? (u32)d : (u32)(32 - d);
  }
  // Which tile a workgroup takes decides the rate: the kernel runs at what the memory system gives for 64-byte pieces 8 KiB
  // apart, and a copy with exactly this access pattern (tools/tilecopy.hip, 65536^2) goes 2.3 TB/s with tiles in plain order,
  // 2.75 with the order below up to round 4a (flags & 64), 3.8 with the present one:
  //  * workgroups b, b + 8, b + 16, ... share an XCD, hence an L2, and start together: out of a super-tile of 16 x 8 tiles an
  //    XCD takes a block of 4 x 4 (2 x 4 until round 4a), so the 64-byte pieces of four neighbours make 256 contiguous bytes on
  //    the source AND the destination side (with neighbours on different XCDs every 128-byte line was fetched twice);
  //  * super-tiles are walked diagonally (the ~8 in flight together differ in both coordinates: their destination pieces no
  //    longer share their low address bits) and an XCD's place inside the super-tile rotates from one super-tile to the next.
  const int tx_n = (((cols + 63) >> 6) + 7) >> 3, ty_n = (rows + 511) >> 9;
  const int bxs = (flags & 64) ? 2 : 4, slots = bxs * 4;
  const int sx_n = (tx_n + 4 * bxs - 1) / (4 * bxs), sy_n = (ty_n + 7) >> 3;
  auto tile_of = [&](int b, int &tX, int &tY) -> bool {
    const int st = b / (8 * slots), in = b % (8 * slots);
    int stx = st % sx_n, sty = st / sx_n;
    if (flags & 1) {
      const int x = in & 7, slot = in >> 3;
      int px = x & 3, py = x >> 2;
      if (!(flags & 64)) {
        sty = (sty + stx) % sy_n;
        px = (px + stx + sty) & 3, py = (py + stx + (sty >> 2)) & 1;
      }
      tX = stx * 4 * bxs + bxs * px + slot % bxs, tY = sty * 8 + 4 * py + slot / bxs;
    } else {  // (column-adjacent tiles per XCD, plain order otherwise: A/B)
      tX = stx * 4 * bxs + in / 8, tY = sty * 8 + (in & 7);
    }
    return tX < tx_n && tY < ty_n;  // (uniform: the tile count is padded to whole super-tiles)
  };
  for (int b = blockIdx.x; b < ntiles_padded; b += gridDim.x) {
    int tX, tY;
    if (!tile_of(b, tX, tY)) continue;
    // ---- this wave's 64 rows of the tile (8 rows x 64 bytes per instruction) into its private staging rows (pitch 9 words:
    // conflict-free both ways), back with lane = row.  (Requesting the next tile's rows before this one is shuffled -- persistent
    // workgroups, 16 more registers -- measured 0.39-0.43 ms at 65536^2 against 0.36 for a plain walk: four workgroups per CU
    // overlap their phases better than one workgroup overlaps its own.)
    const int C0w = tX * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long row = (long long)tY * 512 + 64 * wave + 8 * k + rr;
      const int wc = C0w + w;
#ifdef GF2K_DEV_VARIANTS  // timing-only ablations (wrong results): 4 = no loads, 8 = no stores, 16 = no butterfly
      u64 v = (flags & 4) ? (u64)row * 0x9E3779B97F4A7C15ull : (row < rows && wc < sw) ? __builtin_nontemporal_load(S + row * lds_ + wc) : 0;
#else
      u64 v = (row < rows && wc < sw) ? __builtin_nontemporal_load(S + row * lds_ + wc) : 0;
#endif
      if (wc == sw - 1) v &= maskS;
      sin[wave][8 * k + rr][w] = v;
    }
    u32 lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const u64 x = sin[wave][lane][j];  // (same wave wrote it: LDS operations of a wave execute in order)
      lo[j] = (u32)x, hi[j] = (u32)(x >> 32);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#ifdef GF2K_DEV_VARIANTS
      if (flags & 16) continue;
#endif
      tr_block64(lo[j], hi[j], keep, rot);
    }
    __syncthreads();  // every wave has taken its staged rows out (and, from the second tile on, stored its output rows)
#pragma unroll
    for (int j = 0; j < 8; ++j) sout[64 * j + lane][wave] = (u64)lo[j] | ((u64)hi[j] << 32);  // output row 64 j + lane, word = this wave's row block
    __syncthreads();
    const long long R0 = (long long)tY * 512;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ol = 64 * wave + 8 * k + rr;             // output row inside the tile
      const long long orow = 64ll * C0w + ol;            // = source column
      const long long ow = R0 / 64 + w;                  // output word = source row block
#ifdef GF2K_DEV_VARIANTS
      if ((flags & 8) && sout[ol][w] != 0x123456789ull) continue;
#endif
      if (orow < cols && ow < dwn) {
        if (flags & 2) D[orow * ldd + ow] = sout[ol][w];
        else __builtin_nontemporal_store(sout[ol][w], D + orow * ldd + ow);
      }
    }
  }
}

extern "C" hipError_t gf2k_transpose(u64 *D, long long ldd, const u64 *S, long long lds_, int rows, int cols,
                                     hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return hipSuccess;
  const int sw = (cols + 63) / 64, rb = (rows + 63) / 64;
  static const int t512 = GF2K_DEV_ENV("M4RI_HIP_TRANSPOSE512", 1);  // (A/B measurements)
  if (t512 && (long long)rows * cols >= (1ll << 29)) {  // from 64 MiB on: below, both operands live in the Infinity Cache and the small blocks win
    const int tx_n = (sw + 7) / 8, ty_n = (rows + 511) / 512;
    static const int tflags = GF2K_DEV_ENV("M4RI_HIP_TRANSPOSE_FLAGS", 1);  // (A/B on one box at 65536^2: 0 = column-adjacent tiles per XCD 0.405 ms, 65 = 2 x 4 tile blocks per XCD 0.380, 1 = 4 x 4 blocks, diagonal walk; 2 = plain instead of non-temporal stores: no difference)
    const int stw = (tflags & 64) ? 8 : 16;  // super-tile: stw x 8 tiles
    const long long ntp = (long long)((tx_n + stw - 1) / stw) * ((ty_n + 7) / 8) * (stw * 8);
    if (ntp > 0x7fffffffLL) return hipErrorInvalidValue;
    static const int tgrid = GF2K_DEV_ENV("M4RI_HIP_TRANSPOSE_GRID", (1 << 20));  // (A/B: workgroups that walk several tiles)
    dim3 grid((unsigned)std::min<long long>(ntp, tgrid)), block(512);
    hipLaunchKernelGGL(gf2_transpose512_kernel, grid, block, 0, stream, D, ldd, S, lds_, rows, cols, (int)ntp, tflags);
    return hipGetLastError();
  }
  dim3 grid((sw + 3) / 4, rb), block(256);
  hipLaunchKernelGGL(gf2_transpose_kernel, grid, block, 0, stream, D, ldd, S, lds_, rows, cols);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_strassen_split(u64 *dst, long long ldd, long long dstStride, const u64 *src, long long lds_,
                                          long long srcStride, int h, int w, int side, int batch, hipStream_t stream) {
  if (h <= 0 || w <= 0 || batch <= 0) return hipSuccess;
  const long long total = (long long)h * (w / 2);
  int gx = grid_for(total, 256, (4096 + batch - 1) / batch);
  hipLaunchKernelGGL(gf2_strassen_split_kernel, dim3(gx, 1, batch), dim3(256), 0, stream, dst, ldd, dstStride, src, lds_,
                     srcStride, h, w, side);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_strassen_split2(u64 *dst, long long ldd, long long dstStride, const u64 *src, long long lds_,
                                           long long srcStride, int h, int w, int side, int batch, hipStream_t stream) {
  if (h <= 0 || w <= 0 || batch <= 0) return hipSuccess;
  const long long total = (long long)h * (w / 2);
  int gx = grid_for(total, 256, (4096 + batch - 1) / batch);
  hipLaunchKernelGGL(gf2_strassen_split2_kernel, dim3(gx, 1, batch), dim3(256), 0, stream, dst, ldd, dstStride, src, lds_,
                     srcStride, h, w, side);
  return hipGetLastError();
}

// Three fused levels.  src0 / src1: per virtual-level group g < groups the one or two source quadrants whose XOR is the source
// operand of that group (src1[g] may be null; groups == 1 with src1[0] == null: an ordinary three-level split of the batch).
// Outputs: operand 49 q1 + 7 q2 + q3 of (batch element b, group g) at dst + ((b * groups + g) * 343 + that) * dstStride.
// side: 0 = A, 1 = B, 2 = A with row-group-packed outputs (h % 64 == 0).
extern "C" hipError_t gf2k_strassen_split3(u64 *dst, long long ldd, long long dstStride, const u64 *const *src0,
                                           const u64 *const *src1, int groups, long long lds_, long long srcStride, int h, int w,
                                           int side, int batch, hipStream_t stream) {
  if (h <= 0 || w <= 0 || batch <= 0 || groups <= 0) return hipSuccess;
  if (groups > 7 || (side == 2 && (h & 63))) return hipErrorInvalidValue;
  gf2k_split3_srcs srcs{};
  for (int g = 0; g < groups; ++g) {
    srcs.a[g] = src0[g];
    srcs.b[g] = src1 ? src1[g] : nullptr;
  }
  const long long total = (long long)h * w;
  static const int cap = GF2K_DEV_ENV("M4RI_HIP_PASS_GRID", 8192);
  const int gx = grid_for(total, 256, (cap + batch * groups - 1) / (batch * groups));
  const dim3 grid(gx, groups, batch);
  // non-temporal stores: the 343 operand streams are not re-read before the leaf launch (measured: -3 % / -6 % pass time)
  static const int nt = GF2K_DEV_ENV("M4RI_HIP_PASS_NT", 1);
#ifdef GF2K_DEV_VARIANTS  // (the plain-store forms exist for A/B runs only: no instantiation the shipped launcher cannot reach)
  if (!nt && side == 2)
    hipLaunchKernelGGL((gf2_strassen_split3_kernel<0, true>), grid, dim3(256), 0, stream, dst, ldd, dstStride, srcs, lds_, srcStride, h, w);
  else if (!nt && side == 1)
    hipLaunchKernelGGL((gf2_strassen_split3_kernel<1, false>), grid, dim3(256), 0, stream, dst, ldd, dstStride, srcs, lds_, srcStride, h, w);
  else
#endif
  if (side == 2)
    hipLaunchKernelGGL((gf2_strassen_split3_kernel<0, true, true>), grid, dim3(256), 0, stream, dst, ldd, dstStride, srcs, lds_, srcStride, h, w);
  else if (side == 0)
    hipLaunchKernelGGL((gf2_strassen_split3_kernel<0, false>), grid, dim3(256), 0, stream, dst, ldd, dstStride, srcs, lds_, srcStride, h, w);
  else
    hipLaunchKernelGGL((gf2_strassen_split3_kernel<1, false, true>), grid, dim3(256), 0, stream, dst, ldd, dstStride, srcs, lds_, srcStride, h, w);
  (void)nt;
  return hipGetLastError();
}

// 343 products per parent -> the parent (8h rows x 8w words); parent p of batch * groups at dst + p * dstStride, its products
// at src + p * 343 * srcStride
extern "C" hipError_t gf2k_strassen_merge3(u64 *dst, long long ldd, long long dstStride, const u64 *src, long long lds_,
                                           long long srcStride, int h, int w, int accumulate, int groups, int batch,
                                           hipStream_t stream) {
  if (h <= 0 || w <= 0 || batch <= 0 || groups <= 0) return hipSuccess;
  const long long total = (long long)h * w;
  const int gx = grid_for(total, 256, (8192 + batch * groups - 1) / (batch * groups));
  static const int ntl = GF2K_DEV_ENV("M4RI_HIP_PASS_NTL", 1);  // the products are read once: non-temporal loads, -7 % (1.25 -> 1.15 ms)
#ifdef GF2K_DEV_VARIANTS
  if (!ntl)
    hipLaunchKernelGGL(gf2_strassen_merge3_kernel<false>, dim3(gx, groups, batch), dim3(256), 0, stream, dst, ldd, dstStride, src, lds_,
                       srcStride, h, w, accumulate);
  else
#endif
    hipLaunchKernelGGL(gf2_strassen_merge3_kernel<true>, dim3(gx, groups, batch), dim3(256), 0, stream, dst, ldd, dstStride, src, lds_,
                       srcStride, h, w, accumulate);
  (void)ntl;
  return hipGetLastError();
}

extern "C" hipError_t gf2k_strassen_merge2(u64 *dst, long long ldd, long long dstStride, const u64 *src, long long lds_,
                                           long long srcStride, int h, int w, int accumulate, int batch,
                                           hipStream_t stream) {
  if (h <= 0 || w <= 0 || batch <= 0) return hipSuccess;
  const long long total = (long long)h * (w / 2);
  int gx = grid_for(total, 256, (4096 + batch - 1) / batch);
  hipLaunchKernelGGL(gf2_strassen_merge2_kernel, dim3(gx, 1, batch), dim3(256), 0, stream, dst, ldd, dstStride, src, lds_,
                     srcStride, h, w, accumulate);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_strassen_merge(u64 *dst, long long ldd, long long dstStride, const u64 *src, long long lds_,
                                          long long srcStride, int h, int w, int accumulate, int batch,
                                          hipStream_t stream) {
  if (h <= 0 || w <= 0 || batch <= 0) return hipSuccess;
  const long long total = (long long)h * (w / 2);
  int gx = grid_for(total, 256, (4096 + batch - 1) / batch);
  hipLaunchKernelGGL(gf2_strassen_merge_kernel, dim3(gx, 1, batch), dim3(256), 0, stream, dst, ldd, dstStride, src, lds_,
                     srcStride, h, w, accumulate);
  return hipGetLastError();
}
