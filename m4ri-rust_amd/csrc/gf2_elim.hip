// gf2_elim.hip -- row echelon forms over GF(2) on gfx950, device resident.
//
// Replaces the elimination entry points the friendly layer reaches (paths relative to /root/reference):
//   mzd_echelonize / _m4ri / _pluq   m4ri-sys/src/echelonform.rs:16-37   (BinMatrix::echelonize / rank,
//                                                                          binary_matrix.rs:246-261)
//   mzd_inv_m4ri                     m4ri-sys/src/brilliantrussian.rs:201-208  (BinMatrix::inverted, :263-268)
//   mzd_solve_left                   m4ri-sys/src/solve.rs:12-29          (solve_left, binary_matrix.rs:575-586)
//
// Blocked Gauss-Jordan.  Columns are processed in blocks of up to 2048 (32 words).  Inside a block the columns of one
// 64-bit word form a step:
//   pivot   one workgroup scans the word of the rows that are not pivots of this block yet (256 first, then 1024 per
//           pass), keeps a fully reduced GF(2) basis of at most 64 words (candidates reduced against it with v_readlane
//           broadcasts, insertion by ballot inside one wave), orders the chosen rows by pivot column, flags them, and
//           writes the np reduced pivot rows restricted to the block's columns plus the block's tracking matrix U
//           (staged through LDS);
//   update  every other row XORs the pivot rows its word selects: Four Russians with 4-bit groups, 16 x 16 entries
//           of 512 B in 128 KiB of LDS per workgroup, one wave per row, lane = word, 16 conflict-free lookups per row;
//           the step's pivot rows are overwritten with their reduced form.
// Rows stay where they are inside a block (a byte per row says "pivot of this step / of this block"); when the block is
// finished one permutation (plan, gather, scatter of whole rows) brings block pivot j to row rank0 + j.
// U (rows x 2048 bits) records, for each row, which of the block's pivot rows (as they were when the block started)
// have been added to it; when the block is finished, everything right of it is updated with ONE product
//   A[:, right] ^= U' * A[pivot rows of the block, right]
// through the M4RM tile kernel (gf2_kernels.hip), which is where almost all bit operations of a large elimination go.
// All per-step decisions live in a device-side state record, so a block is enqueued without host round trips.
#include "gf2_kernels.h"

#include <atomic>
#include <type_traits>

typedef uint64_t u64;
typedef uint32_t u32;

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 readfirst64(u64 v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((u64)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------------------
// pivot search in one word column + the reduced pivot rows over the block
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 rdlane64(u64 v, int lane) {  // lane must be wave-uniform
  const int l = __builtin_amdgcn_readfirstlane(lane);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ int rdlane32(int v, int lane) {
  return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
}

// Where does a fused step's time go?  (tools/elim_stamps.py, development builds only.)  When set, thread 0 of the look-ahead workgroup adds
// the shader cycles of its stages to stamps[0..11] (launch start -> search starts -> first pass's candidates loaded -> basis complete ->
// the update workgroups' counter reached -> published; 7 / 9 / 10 / 11: inside the search loop) and counts the steps in stamps[8].
#ifdef GF2K_DEV_VARIANTS
__device__ unsigned long long *gf2k_elim_stamps;
extern "C" hipError_t gf2k_dev_set_elim_stamps(unsigned long long *p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(gf2k_elim_stamps), &p, sizeof(p));
}
// (-DELIM_STAMP_TOTAL_ONLY: only the last stamp of either workgroup fires, i.e. stamps[5] and stamps[21] hold the whole chain from the
// workgroup's start without the stamps' own cost in between.  A stamp is a global read-modify-write by one thread, ~500 cycles on its own --
// but one that FOLLOWS another closely reads the running clock the other has just stored and waits for that store: the intervals "barrier
// behind the re-reductions" and "barrier behind another wave's insertions" (5,000 and 2,500 cycles in the stamped runs) are mostly that;
// every wave has arrived at the first of those barriers 1,700-2,800 cycles after the loop's entry (measured with per-wave stamps).  The
// whole look-ahead chain of a 4096-row step: 33,500 cycles unstamped against 38,000 as the sum of the stamped stages.)
#ifdef ELIM_STAMP_TOTAL_ONLY
#define ELIM_STAMP_ON(k) ((k) == 5)
#else
#define ELIM_STAMP_ON(k) true
#endif
#define ELIM_STAMP(k)                                                                  \
  do {                                                                                 \
    if (ELIM_STAMP_ON(k) && COH && gf2k_elim_stamps && threadIdx.x == 0) {             \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
      gf2k_elim_stamps[k] += now_ - gf2k_elim_stamps[15];                              \
      gf2k_elim_stamps[15] = now_;                                                     \
    }                                                                                  \
  } while (0)
// the same for ONE update workgroup (the middle one of the launch): stamps[16 + k], its own running clock in stamps[31]
#define UPD_STAMP(k)                                                                   \
  do {                                                                                 \
    if ((ELIM_STAMP_ON(k) || (k) == 0) && LOOK && gf2k_elim_stamps && threadIdx.x == 0 && blockIdx.x == (unsigned)(nupd / 2)) { \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
      if (k) gf2k_elim_stamps[16 + k] += now_ - gf2k_elim_stamps[31];                  \
      gf2k_elim_stamps[31] = now_;                                                     \
    }                                                                                  \
  } while (0)
#else
#define UPD_STAMP(k) do { } while (0)
#define ELIM_STAMP(k) do { } while (0)
#endif

// Six registers' slot `lane_sel` := six scalar values (v_writelane_b32; this clang has no builtin for it).  VOP3 takes ONE scalar register
// on gfx9, so the lane select goes through M0, saved and restored inside the statement (M0 is the compiler's).  The scalar operands come
// from the scalar unit or from v_readlane many instructions earlier: no wait state is owed (the 4-state rule is about a lane select
// written by the vector unit).
__device__ __forceinline__ void elim_writelane6(u32 &v0, u32 &v1, u32 &v2, u32 &v3, int &v4, int &v5, u32 s0, u32 s1, u32 s2, u32 s3, int s4,
                                                int s5, int lane_sel) {
  unsigned keep;
  asm("s_mov_b32 %6, m0\n\ts_mov_b32 m0, %13\n\ts_nop 0\n\t"  // (one state between the scalar write of M0 and its first reader, as for the other M0 readers)
      "v_writelane_b32 %0, %7, m0\n\tv_writelane_b32 %1, %8, m0\n\tv_writelane_b32 %2, %9, m0\n\t"
      "v_writelane_b32 %3, %10, m0\n\tv_writelane_b32 %4, %11, m0\n\tv_writelane_b32 %5, %12, m0\n\t"
      "s_mov_b32 m0, %6"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "=&s"(keep)
      : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "s"(lane_sel));
}

// shared state of the pivot search (one workgroup): a member of the calling kernel's LDS
struct ElimPivotShared {
  // basis vector k: b_word[k] is clear on the pivot column of every other vector (kept fully reduced), b_trk[k] says
  // which of the chosen rows (by insertion index, as originally read) it is the XOR of
  u64 b_word[64], b_trk[64];
  int b_row[64], b_col[64];
  int s_nb;
  int s_nz[16];
  int f_pos[64];
  unsigned char s_chosen[256];  // rows scan0 .. scan0+255 picked by this step
  int s_lead[4];
  int s_col2k[64];
  // the stash (see elim_pivot_step): the step's selector map and its slice of the tables for the next word column, the new scan start
  u64 s_smap[256], s_nxt[256], s_pcmask;
  int s_ns, s_far;
};
// The stash: u64 words [0, 256) = the next word column of rows scan .. scan + 255 as it will be once the step just published has been
// applied, [256, 512) = those rows' flags; lives behind ptab's 64 x 64 + 256 words.
constexpr int kElimStashOff = 64 * 64 + 256;

// loads that see what other workgroups of the SAME launch stored with agent scope (the look-ahead workgroup reads rows the update
// workgroups have just rewritten; L2 is not coherent across XCDs for plain accesses)
template <bool COH, typename T>
__device__ __forceinline__ T elim_ld(const T *p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

// The pivot step of word column j of the block: SEARCH (reads the column's word of the candidate rows and their flags, keeps the
// basis in LDS, writes nothing to global memory), then `between(need_all)` (the look-ahead form waits there until the rows it is
// going to read have landed: the PRIORITY rows, or -- need_all -- every update workgroup's; see EARLY PUBLICATION in the update
// kernel), then PUBLISH (flags, state, pivot columns, the raw chosen rows and the selector map).
// THE STASH (round 5).  The look-ahead search of the next step needs the next word column of its candidate rows AFTER the step
// published here has been applied -- it used to wait for every update workgroup of its launch to rewrite that column first (5-9 us
// of a step, the largest single stage once the search loops were rewritten).  But the publisher can work those 256 words out
// itself, right here, while nothing else runs: the rows' selector words and the next column are final (every update of the previous
// step has landed: that is what `between` waited for), the selector map and the column's slice of the tables are in its hands.  It
// leaves them -- and the rows' flags -- behind ptab (`use_stash` of the NEXT search reads them instead of waiting); the update
// workgroups rewrite the column for every row with the rest of the step, and a search that needs more than its first 256 rows waits
// for them (`wait_column`).  `make_stash`: a next step exists in this block and its search will be a look-ahead one.
template <bool COH, typename Between, typename WaitColumn>
__device__ __forceinline__ bool elim_pivot_step(ElimPivotShared &sm, const u64 *A, long long lda, int m, long long c0w,
                                                int sw, int j, u64 colmask, const u64 *U, long long ldu, int uw,
                                                gf2k_elim_state *st, int *pivcols, u64 *__restrict__ ptab, unsigned char *rowflag,
                                                int *__restrict__ blkpiv, Between between, WaitColumn wait_column, bool use_stash,
                                                bool make_stash) {
  u64 (&b_word)[64] = sm.b_word, (&b_trk)[64] = sm.b_trk;
  int (&b_row)[64] = sm.b_row, (&b_col)[64] = sm.b_col, &s_nb = sm.s_nb, (&s_nz)[16] = sm.s_nz, (&f_pos)[64] = sm.f_pos;
  unsigned char (&s_chosen)[256] = sm.s_chosen;
  int (&s_lead)[4] = sm.s_lead, (&s_col2k)[64] = sm.s_col2k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r_cur = st->r_cur;
  const int jbase = r_cur - st->r0;
  const int scan0 = st->scan;
  const long long wc = c0w + j;
  if (tid == 0) s_nb = 0;
  if (tid < 256) s_chosen[tid] = 0;
  unsigned char f0 = 1;                                // flag of row scan0 + tid before this step (first pass, tid < 256)
  __syncthreads();
  ELIM_STAMP(1);  // (the search starts: until the stash there was a wait for the next column in front of this)
  // the first pass looks at 256 rows only (one wave per SIMD: nearly always enough for 64 pivots, and the waves that
  // re-reduce their candidates after the first wave's insertions do not compete for issue slots); then 1024 per pass
  int csz = 256;
  bool beyond = false;  // (uniform) the search went past its first 256 rows
  for (int base = scan0; base < m; base += csz, csz = 1024) {
    if (base != scan0) beyond = true;
    const int i = base + tid;
    // w: candidate reduced against the basis; t: which chosen rows were added to it
    // flag and word are requested together (the word of a flagged row is simply discarded): one memory latency, not two
    const bool inr = tid < csz && i < m;
    unsigned char fl = 1;  // pivots of this block (and of the step being applied) are no candidates
    u64 wraw = 0;
    if (use_stash && base == scan0) {  // (uniform) the previous publication left this pass's words and flags behind
      if (inr) {
        fl = (unsigned char)ptab[kElimStashOff + 256 + tid];
        wraw = ptab[kElimStashOff + tid];
      }
    } else {
      if (use_stash && base == scan0 + 256)  // (uniform) beyond the stash: the update workgroups' rewrite of the column
        if (!wait_column()) return false;
      if (inr) {
        fl = elim_ld<COH>(rowflag + i);
        wraw = elim_ld<COH>(A + (long long)i * lda + wc);
      }
    }
    if (base == scan0) f0 = fl;
    u64 w = fl == 0 ? (wraw & colmask) : 0, t = 0;
#ifdef GF2K_DEV_VARIANTS
    if (base == scan0) {
      asm volatile("" : "+v"(w));  // (the stamp stands behind the arrival of the candidates)
      ELIM_STAMP(2);
    }
#endif
    int done = 0;  // basis vectors already applied to w
    for (;;) {
      const int nb = s_nb;
      if (nb == 64) break;  // uniform over the workgroup
      if (done < nb && __ballot(w != 0)) {
        // The basis is FULLY reduced (every vector is clear on the pivot columns of all others), so adding vector k never changes a
        // candidate's bit on another vector's pivot column: whether vector k is added is decided by the candidate's bits as they are
        // NOW.  That takes the candidate out of the loop's dependency chain -- 63 turns of five broadcasts, a mask and four XORs that
        // overlap freely instead of 63 turns that each wait for the one before (round 5: the search loop was 16 of a step's 28 us).
        // The vectors come straight from LDS (one address for the wave: a broadcast read; no hop through scalar registers), four
        // turns in flight.
        const u64 w0 = w;
        u32 wl = (u32)w, wh = (u32)(w >> 32), tl = (u32)t, th = (u32)(t >> 32);
#pragma unroll 4
        for (int k = done; k < nb; ++k) {
          const u64 pw = b_word[k], pt = b_trk[k];
          const u32 sel = (u32)__builtin_amdgcn_sbfe((int)(u32)(w0 >> b_col[k]), 0, 1);  // all ones where the candidate has vector k's column
          wl = __builtin_amdgcn_bitop3_b32((u32)pw, sel, wl, 0x6a);
          wh = __builtin_amdgcn_bitop3_b32((u32)(pw >> 32), sel, wh, 0x6a);
          tl = __builtin_amdgcn_bitop3_b32((u32)pt, sel, tl, 0x6a);
          th = __builtin_amdgcn_bitop3_b32((u32)(pt >> 32), sel, th, 0x6a);
        }
        w = (u64)wl | ((u64)wh << 32), t = (u64)tl | ((u64)th << 32);
      }
      ELIM_STAMP(7);  // (re-reduction of this wave's candidates; wave 0's view)
      done = nb;
      const u64 nzb = __ballot(w != 0);
      if (lane == 0) s_nz[wave] = nzb != 0;
      __syncthreads();
      ELIM_STAMP(9);  // (waiting for the other waves' re-reductions)
      int fw = -1;
      for (int wv = 15; wv >= 0; --wv)
        if (s_nz[wv]) fw = wv;
      if (fw < 0) break;
      if (wave == fw && nb == 0) {
        // THE FIRST INSERTIONS (no older vectors; on full-rank input all 64 pivots of a step).  A chosen candidate STAYS IN ITS LANE as
        // the basis vector: "reduce the later candidates by the new pivot" and "keep the older vectors clear on the new pivot column"
        // are then ONE operation on one register pair -- 13 vector instructions per pivot instead of 24 (one shift + v_bfe_i32 instead
        // of two, four v_bitop3_b32 instead of eight, a compare and two selects instead of six v_writelane_b32 through M0): the lone
        // inserting wave pays four-plus cycles for every instruction it issues (in-kernel stamps: 296 cycles per pivot before).  The
        // pivot's own lane is masked out of the XOR; its own tracking bit (toggled in), insertion index and LDS slot follow after the loop from the mask
        // of chosen lanes.  Same visiting order, same pivots, same insertion indices as the general loop below.
        u32 wl = (u32)w, wh = (u32)(w >> 32), tl = (u32)t, th = (u32)(t >> 32);
        const int row0w = __builtin_amdgcn_readfirstlane(base + fw * 64);
        int mc = 0, nbl = 0;
        u64 chosen = 0;
        // Sixty-four turns, no branch in them (a zero candidate's turn changes nothing: its word and its tracking word enter as zero):
        // with the test-and-skip branch a turn cost 253 cycles for 30 instructions -- the wave waited on the chain v_readlane ->
        // s_or / s_cmp / s_cbranch -> s_ff1 -> shift -> v_bfe -> select -> v_bitop3 -> the next v_readlane, not on issue slots.
#pragma unroll 8
        for (int p = 0; p < 64; ++p) {  // lowest row first
          const u32 pwl = __builtin_amdgcn_readlane(wl, p), pwh = __builtin_amdgcn_readlane(wh, p);
          // 1 / 0: a candidate / no candidate in this lane, or a dependent one.  (s_min_u32 by hand: written as a compare or as min() the
          // flag becomes a lane mask, and the insertion count, the chosen mask and the tracking bit go through the vector unit with it)
          u32 nz1;
          asm("s_min_u32 %0, %1, 1" : "=s"(nz1) : "s"(pwl | pwh) : "scc");
          const u32 nzm = 0u - nz1;
          const int c = __builtin_ctzll(((u64)(pwh | 0x80000000u) << 32) | pwl);  // (63 for a zero word: any column will do)
          const u64 nbit = 1ull << nbl;
          const u32 ptl = (__builtin_amdgcn_readlane(tl, p) | (u32)nbit) & nzm, pth = (__builtin_amdgcn_readlane(th, p) | (u32)(nbit >> 32)) & nzm;
          u32 wm = (u32)__builtin_amdgcn_sbfe((int)(u32)((((u64)wh << 32) | wl) >> c), 0, 1);  // all ones where a lane's word has the column
          const bool self = lane == p;
          wm = self ? 0u : wm;
          mc = self ? c : mc;
          wl = __builtin_amdgcn_bitop3_b32(pwl, wm, wl, 0x6a);
          wh = __builtin_amdgcn_bitop3_b32(pwh, wm, wh, 0x6a);
          tl = __builtin_amdgcn_bitop3_b32(ptl, wm, tl, 0x6a);
          th = __builtin_amdgcn_bitop3_b32(pth, wm, th, 0x6a);
          chosen |= (u64)nz1 << p;
          nbl += (int)nz1;
        }
        const bool ch = (chosen >> lane) & 1;
        const int kidx = __popcll(chosen & ((1ull << lane) - 1));  // insertion index: the chosen lanes in lane order
        w = (u64)wl | ((u64)wh << 32), t = (u64)tl | ((u64)th << 32);
        if (ch) {
          b_word[kidx] = w;
          b_trk[kidx] = t ^ (1ull << kidx);  // (XOR, not OR: a later pivot that contains this one toggles the bit when it is added)
          b_col[kidx] = mc;
          b_row[kidx] = row0w + lane;
          w = 0;  // (consumed: no candidate any more)
        }
        done = nbl;
        if (lane == 0) s_nb = nbl;
        ELIM_STAMP(10);  // (insertions by wave 0)
      } else if (wave == fw) {  // this wave's candidates are reduced against the whole basis: insert the independent ones
        u64 mw = lane < nb ? b_word[lane] : 0, mt = lane < nb ? b_trk[lane] : 0;
        int mc = lane < nb ? b_col[lane] : 0, mrow = lane < nb ? b_row[lane] : 0;
        int nbl = nb;
        // One pivot per turn, up to 64 turns in ONE wave: the search's longest serial piece (in-kernel stamps, round 5: 22,000 of a
        // step's 60,000 cycles -- 350 per turn -- in the form that took the lowest non-zero candidate by ballot / s_ff1 and tested the
        // column through exec-masked branches).  Every hop between the vector and the scalar unit costs 15-25 cycles, so the loop-carried
        // chain is kept to five: the candidates are visited in LANE order (a scalar counter: no ballot, no s_ff1 on it; a zero candidate
        // is skipped by a scalar branch), v_readlane of the candidate's word -> s_ff1_i32_b64 for its lowest column -> one 64-bit shift
        // of every lane's word by that column -> v_bfe_i32 (all ones / zero) -> v_bitop3_b32 ((pivot & mask) ^ word) -> the next
        // v_readlane.  The tracking words and the older basis vectors are updated off the chain with the same masks.  The pivot's own
        // lane needs no special case: its bit on the column is set, so it XORs itself to zero.
        u32 wl = (u32)w, wh = (u32)(w >> 32), tl = (u32)t, th = (u32)(t >> 32);
        u32 bl = (u32)mw, bh = (u32)(mw >> 32), btl = (u32)mt, bth = (u32)(mt >> 32);
        (void)nzb;
        // (ONE wave runs here while the others wait at the barrier, so nothing hides an instruction's four issue cycles: a turn costs
        // its instruction count.  Hence v_writelane_b32 for the new vector's slot -- one instruction per register instead of a compare,
        // a move and a select each --, the tracking bit set without branches, one backward branch.)
        const int row0w = __builtin_amdgcn_readfirstlane(base + fw * 64);
        int p = 0;
        do {  // lowest row first
          const u32 pwl = __builtin_amdgcn_readlane(wl, p), pwh = __builtin_amdgcn_readlane(wh, p);
          if ((pwl | pwh) != 0) {  // (uniform; zero: no candidate in this lane, or a dependent one)
            const int c = __builtin_ctzll(((u64)pwh << 32) | pwl);
            const u64 nbit = 1ull << nbl;
            const u32 ptl = __builtin_amdgcn_readlane(tl, p) | (u32)nbit, pth = __builtin_amdgcn_readlane(th, p) | (u32)(nbit >> 32);
            const u32 wm = (u32)__builtin_amdgcn_sbfe((int)(u32)((((u64)wh << 32) | wl) >> c), 0, 1);  // all ones where the candidate has the column
            const u32 bm = (u32)__builtin_amdgcn_sbfe((int)(u32)((((u64)bh << 32) | bl) >> c), 0, 1);  // ... where an older basis vector has it
            wl = __builtin_amdgcn_bitop3_b32(pwl, wm, wl, 0x6a);
            wh = __builtin_amdgcn_bitop3_b32(pwh, wm, wh, 0x6a);
            tl = __builtin_amdgcn_bitop3_b32(ptl, wm, tl, 0x6a);
            th = __builtin_amdgcn_bitop3_b32(pth, wm, th, 0x6a);
            bl = __builtin_amdgcn_bitop3_b32(pwl, bm, bl, 0x6a);  // keep the older vectors clear on the new pivot column
            bh = __builtin_amdgcn_bitop3_b32(pwh, bm, bh, 0x6a);
            btl = __builtin_amdgcn_bitop3_b32(ptl, bm, btl, 0x6a);
            bth = __builtin_amdgcn_bitop3_b32(pth, bm, bth, 0x6a);
            // the new vector takes slot nbl (that lane's slot was zero: untouched above)
            elim_writelane6(bl, bh, btl, bth, mc, mrow, pwl, pwh, ptl, pth, c, row0w + p, nbl);
            ++nbl;
          }
        } while (++p < 64 && nbl < 64);
        w = (u64)wl | ((u64)wh << 32), t = (u64)tl | ((u64)th << 32);
        mw = (u64)bl | ((u64)bh << 32), mt = (u64)btl | ((u64)bth << 32);
        done = nbl;
        if (lane < nbl) {
          b_word[lane] = mw;
          b_trk[lane] = mt;
          b_col[lane] = mc;
          b_row[lane] = mrow;
        }
        if (lane == 0) s_nb = nbl;
        ELIM_STAMP(10);  // (insertions by wave 0)
      }
      __syncthreads();
      ELIM_STAMP(11);  // (waiting for another wave's insertions)
    }
    __syncthreads();  // s_nz is rewritten by the next pass
    if (s_nb == 64) break;
  }
  __syncthreads();
  const int np = s_nb;
  ELIM_STAMP(3);  // basis complete
  // (the previous step's pivots: stable since that step was published -- fetched before the wait instead of behind it)
  const int prev_row = tid < st->np ? st->cur_row[tid] : -1;

  // ---- PREPARE: everything of the publication that needs only the basis, into LDS and registers (nothing global is written yet:
  // the update workgroups of this launch still read the previous step's state).  While they finish, this runs for free; until round 5
  // all of it stood behind the wait, on the critical path of every step of a large matrix. ----
  // order the pivots by column: vector k is pivot jbase + pos[k] of the block
  u64 o_pcmask = 0;
  int o_c = 0, o_row = -1, o_pos = 0;  // (wave 0, lane < np)
  if (wave == 0) {
    if (lane < np) {
      o_c = b_col[lane];
      o_row = b_row[lane];
      o_pcmask = 1ull << o_c;
    }
    for (int o = 32; o; o >>= 1) o_pcmask |= shfl64(o_pcmask, lane ^ o);
    const bool far = lane < np && o_row - scan0 >= 256;  // chosen by a later pass (rare)
    const u64 anyfar = __ballot(far);
    if (lane < np) {
      o_pos = __popcll(o_pcmask & ((1ull << o_c) - 1));
      f_pos[lane] = o_pos;
      if (!far) s_chosen[o_row - scan0] = 1;
    }
    if (lane == 0) {
      sm.s_pcmask = o_pcmask;
      sm.s_far = anyfar != 0;
    }
  }
  if (tid < 64) s_col2k[tid] = -1;
  __syncthreads();
  if (tid < np) s_col2k[b_col[tid]] = tid;
  // the search of the next step starts behind the leading run of block pivots
  if (tid < 256) {
    const bool taken = f0 != 0 || s_chosen[tid];
    const u64 mk = __ballot(taken);
    if (lane == 0) s_lead[wave] = ~mk ? __builtin_ctzll(~mk) : 64;
  }
  __syncthreads();
  if (tid == 0) {
    int adv = 0;
    for (int w2 = 0; w2 < 4; ++w2) {
      adv += s_lead[w2];
      if (s_lead[w2] < 64) break;
    }
    const int ns = scan0 + adv;
    sm.s_ns = ns < m ? ns : m;
  }
  // The 16 x 16 selector map that turns a row's bits on the pivot columns into the set of the raw chosen rows to add: reduced pivot
  // row k = XOR of the raw rows in b_trk[k], so a row whose word selects the pivot columns s must add the raw rows
  // XOR over c in s of b_trk[k(c)]  =  XOR over the nibbles of s of smap[nibble position][nibble].
  // (Round 4.  Until round 3 this kernel combined the 64 reduced rows itself -- 4-bit tables over the selector bits, ten
  // workgroup barriers, 11.5 of the step's ~47 us on ONE CU; the update kernel's tables are now built from the raw rows and
  // every row's selector goes through the 2-KiB map first: sixteen 8-byte lookups per eight rows and lane group.)
  if (tid < 256) {
    const int g = tid >> 4, v4 = tid & 15;
    u64 x = 0;
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
      const int k = s_col2k[4 * g + b2];
      if (((v4 >> b2) & 1) && k >= 0) x ^= b_trk[k];
    }
    sm.s_smap[tid] = x;
  }
  __syncthreads();
  // the stash's candidates: rows s_ns .. s_ns + 255.  Their flags are stable (only publications write flags): read now; a row this
  // step has chosen counts as flagged
  const long long sr = (long long)sm.s_ns + tid;
  const bool scand = make_stash && tid < 256 && sr < m;
  bool sflag = true;
  if (scand) {
    sflag = elim_ld<COH>(rowflag + sr) != 0;
    const long long si = sr - scan0;
    if (si < 256) sflag |= s_chosen[si] != 0;
    else if (sm.s_far)
      for (int k = 0; k < np; ++k) sflag |= b_row[k] == sr;
  }

  if (!between(beyond || sm.s_far != 0)) return false;  // (`true`: rows beyond the priority range were read or chosen -- wait for the whole update; contains workgroup barriers; everything below writes global memory.  false: the look-ahead wait ran out -- nothing is published)
  ELIM_STAMP(4);  // every update workgroup done
  // ---- PUBLISH.  Every global load first, in one memory latency: the chosen rows AS THEY ARE NOW (every update of the previous step
  // has landed) over the block's columns [c0w, c0w+sw) and the tracking words [0, uw); for the stash, the candidates' selector word
  // and next word, and the chosen rows' words on the next column (this thread's entry of that column's slice of the tables) ----
  u64 praw[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 1024 * q, k = idx >> 6, wd = idx & 63;
    u64 v = 0;
    if (k < np) {
      const long long r = b_row[k];
      if (wd < sw) v = elim_ld<COH>(A + r * lda + c0w + wd);
      else if (wd < sw + uw) v = elim_ld<COH>(U + r * ldu + (wd - sw));
    }
    praw[q] = v;
  }
  u64 wsel = 0, wold = 0, nx = 0;
  if (make_stash && tid < 256) {  // (make_stash is uniform)
    if (scand && !sflag) {
      wsel = elim_ld<COH>(A + sr * lda + wc);
      wold = elim_ld<COH>(A + sr * lda + wc + 1);
    }
    const int g = tid >> 4, e = tid & 15;
#pragma unroll
    for (int b2 = 0; b2 < 4; ++b2) {
      const int k = g * 4 + b2;
      if (((e >> b2) & 1) && k < np) nx ^= elim_ld<COH>(A + (long long)b_row[k] * lda + wc + 1);
    }
  }
  // ... then the stores: flags, state, pivot columns
  if (prev_row >= 0) rowflag[prev_row] = 255;  // the previous step's pivots become "pivot of this block"
  if (wave == 0) {
    if (lane < np) {
      pivcols[r_cur + o_pos] = (int)(wc * 64 + o_c);
      blkpiv[jbase + o_pos] = o_row;
      st->cur_row[o_pos] = o_row;
      rowflag[o_row] = (unsigned char)(1 + o_c);
    }
    if (lane == 0) {
      st->np = np;
      st->pcmask = o_pcmask;
      st->jbase = jbase;
      st->r_cur = r_cur + np;
      st->scan = sm.s_ns;
    }
  }
  // the chosen rows, each with its own unit bit in the tracking part, AS READ (row k = insertion index k), and the selector map
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 1024 * q, k = idx >> 6, wd = idx & 63;
    u64 v = praw[q];
    if (k < np && wd >= sw && wd < sw + uw) {
      const int u = wd - sw, jj = jbase + f_pos[k];
      v ^= (jj >> 6) == u ? 1ull << (jj & 63) : 0;
    }
    ptab[idx] = v;
  }
  if (tid < 256) ptab[64 * 64 + tid] = sm.s_smap[tid];
  if (make_stash) {  // (uniform) see THE STASH above
    if (tid < 256) sm.s_nxt[tid] = nx;
    __syncthreads();
    if (tid < 256) {
      u64 v = wold;
      if (scand && !sflag) {
        const u64 sq = wsel & sm.s_pcmask;
        u64 x = 0, acc = 0;
#pragma unroll
        for (int g = 0; g < 16; ++g) x ^= sm.s_smap[g * 16 + (int)((sq >> (4 * g)) & 15)];
#pragma unroll
        for (int g = 0; g < 16; ++g) acc ^= sm.s_nxt[g * 16 + (int)((x >> (4 * g)) & 15)];
        v = wold ^ acc;
      }
      ptab[kElimStashOff + tid] = v;
      ptab[kElimStashOff + 256 + tid] = sflag ? 1 : 0;
    }
  }
  ELIM_STAMP(5);  // published (and the stash left behind)
#ifdef GF2K_DEV_VARIANTS
  if (COH && gf2k_elim_stamps && threadIdx.x == 0) gf2k_elim_stamps[8] += 1;
#endif
  return true;
}

__global__ __launch_bounds__(1024) void gf2_elim_pivot_kernel(const u64 *__restrict__ A, long long lda, int m, long long c0w,
                                                              int sw, int j, u64 colmask, const u64 *__restrict__ U,
                                                              long long ldu, int uw, gf2k_elim_state *st, int *pivcols,
                                                              u64 *__restrict__ ptab, unsigned char *rowflag,
                                                              int *__restrict__ blkpiv, int make_stash) {
  __shared__ ElimPivotShared sm;
  if (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // an earlier launch of the chain failed: touch nothing
  (void)elim_pivot_step<false>(sm, A, lda, m, c0w, sw, j, colmask, U, ldu, uw, st, pivcols, ptab, rowflag, blkpiv, [](bool) { return true; },
                               [] { return true; }, /*use_stash=*/false, make_stash != 0 && j + 1 < sw);
}

// every row adds the pivot rows selected by its bits on the pivot columns; the step's own pivot rows (row flag) are
// overwritten with their reduced form, which is the single-bit table entry of their pivot column.  Four Russians with 4-bit groups: for each nibble of
// the 64-bit selector word a 16-entry table of XOR combinations, 16 x 16 entries of 512 B (one LDS bank row each:
// lane = word, conflict-free) = 128 KiB, built once per workgroup; a row then costs 16 lookups.
constexpr int kEarlyMinRows = 192;  // rows per update workgroup from which the publication goes ahead of the update's end (see EARLY PUBLICATION)
constexpr int kUpdLds = 16 * 16 * 64 * 8 + 256 * 8 + 2048;  // the tables, the selector map, the flags of the workgroup's rows

// Bounded wait of the look-ahead workgroup for a counter the update workgroups of the same launch raise (every one of them raises
// it, whatever it did, and none of them waits for anything: they are all dispatched before or alongside this workgroup because the
// grid fits the chip -- at most 256 update workgroups of one per CU -- so the wait ends; the bound turns a scheduling surprise
// into an error flag instead of a hang).
// Returns false when the bound ran out (st->err is set then): the caller must neither search on, publish nor reset the counters --
// the update workgroups may still be running and raising them; every later launch of the chain sees err and does nothing,
// and the host reports the failure at the end of the block.
__device__ __forceinline__ bool elim_wait_count(int *cnt, int want, int *err) {
  __shared__ int wait_ok;
  if (threadIdx.x == 0) {
    int ok = 1, spins = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, the same on every part and at every shader clock
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {  // (relaxed: an acquire here is an L2 invalidate per turn)
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 255) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {  // 1 s
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    wait_ok = ok;
  }
  __syncthreads();
  // no acquire fence here (an agent-scope fence invalidates / writes back the whole L2 of the XCD): whatever this workgroup reads
  // of the launch's own stores it reads with agent-scope atomic loads (elim_ld<true>: sc1 loads, served by the memory-side of the
  // L2 hierarchy), and the update workgroups store those words with agent-scope atomic stores (sc1: write-through) and wait for
  // them with s_waitcnt vmcnt(0) BEFORE they raise the counter (`raise` below) -- the ordering the hand-off rests on; the 112 cases
  // of tests/test_gpu_elim.py run it at every size against the oracle, and the fault-injection case shows the bound's way out.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const bool ok = wait_ok != 0;
  __syncthreads();  // (wait_ok is rewritten by the next wait)
  return ok;
}

// LOOK (round 4): the launch has one workgroup more than it has update workgroups.  The extra workgroup runs the pivot SEARCH of
// step j + 1 while the others update, waits for st->cnt2 (everything updated; on large matrices for st->cntP only: the priority rows),
// and PUBLISHES step j + 1 (flags, state, raw pivot rows, selector map).  A step is then one launch, and the one-CU search runs beside the update instead of behind it.
// What the search reads -- word column j + 1 of its candidate rows as step j leaves it -- comes from THE STASH of the previous
// publication (elim_pivot_step), so it starts with the launch.  (Round 4 to mid round 5: every update workgroup rewrote that column
// for its rows first and raised st->cnt1, and the search waited for all of them: 4-5 us at the head of every update workgroup --
// three dependent memory latencies -- and 8.7 us before the search started at 65536 rows.)  A search that needs more than the
// stash's 256 rows waits for the whole update (cnt2) and reads on from memory: rare, and no slower than the two-launch form.
template <bool LOOK>
__global__ __launch_bounds__(1024) void gf2_elim_update_kernel(u64 *__restrict__ A, long long lda, int m, int full_and_flags,
                                                               long long c0w, int sw, int j, u64 *__restrict__ U,
                                                               long long ldu, int uw, gf2k_elim_state *st,
                                                               u64 *__restrict__ ptab, unsigned char *__restrict__ rowflag,
                                                               u64 colmask_next, int *pivcols, int *__restrict__ blkpiv) {
  extern __shared__ __attribute__((aligned(16))) u64 tab[];  // [group 16][entry 16][word 64]
  __shared__ ElimPivotShared sm;  // (the look-ahead workgroup's search state)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nupd = LOOK ? (int)gridDim.x - 1 : (int)gridDim.x;  // update workgroups
  if (__hip_atomic_load(&st->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // an earlier launch of the chain failed: touch nothing
  const int fault = full_and_flags >> 8;  // (test hook, M4RI_HIP_ELIM_FAULT: 1 = update workgroup 0 never raises its counters)
  const int full = full_and_flags & 1;
  if (LOOK && (int)blockIdx.x == nupd) {
#ifdef GF2K_DEV_VARIANTS
    if (gf2k_elim_stamps && tid == 0) gf2k_elim_stamps[15] = __builtin_amdgcn_s_memtime();
#endif
    // (no wait for the next column: the previous publication left the first pass's words behind -- THE STASH)
    // the counters run on through the block (gf2k_elim_begin_block zeroes them): launch j is waiting for (j + 1) x nupd.  The
    // publication waits for the PRIORITY rows only (cntP; see the update workgroups below) unless the search left them
    const int want = (j + 1) * nupd;
    const long long rows_lo_l = full ? 0 : st->r0;
    const long long per_l = ((((long long)m - rows_lo_l + nupd - 1) / nupd) + 7) & ~7ll;
    const bool early = per_l >= kEarlyMinRows && per_l <= 2048;
    (void)elim_pivot_step<true>(sm, A, lda, m, c0w, sw, j + 1, colmask_next, U, ldu, uw, st, pivcols, ptab, rowflag, blkpiv,
                                [&](bool need_all) { return elim_wait_count(need_all || !early ? &st->cnt2 : &st->cntP, want, &st->err); },
                                [&] { return elim_wait_count(&st->cnt2, want, &st->err); }, /*use_stash=*/true, /*make_stash=*/j + 2 < sw);
    return;
  }
  UPD_STAMP(0);
  const int np = st->np, r0s = st->r0;
  const u64 pcmask = st->pcmask;
  // this workgroup's (write-through, agent-scope) stores so far have completed, then the count goes up.  NOT __threadfence():
  // its L2 write-back, once per wave and raise, made a step 160 us slower at 65536 rows (measured, round 4).
  auto raise = [&](int *cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && !(fault == 1 && blockIdx.x == 0)) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  if (np == 0) {
    if (LOOK) {
      raise(&st->cnt2);
      if (tid == 0 && !(fault == 1 && blockIdx.x == 0)) __hip_atomic_fetch_add(&st->cntP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // level 0/1: entry 0 and the single-bit entries: bit i = the step's i-th chosen row as the pivot kernel read it (raw; rows past
  // np are zero).  The selector map (see the pivot kernel) goes into LDS behind the tables.
  u64 *smap = tab + 16 * 16 * 64;
  if (tid < 256) smap[tid] = ptab[64 * 64 + tid];
  const int rows_lo = full ? 0 : r0s;
  const long long wc = c0w + j;
  // this workgroup's rows: one contiguous piece, a multiple of the 8 rows a wave takes per pass (65536 rows over 255 workgroups:
  // 264 rows each, i.e. two passes of the 16 waves and a third of one wave -- 128-row pieces dealt round-robin gave two workgroups
  // three full passes)
  const long long per = ((((long long)m - rows_lo + nupd - 1) / nupd) + 7) & ~7ll;
  const long long row_b = rows_lo + (long long)blockIdx.x * per, row_e = min((long long)m, row_b + per);
  // EARLY PUBLICATION (round 5, last change).  The look-ahead workgroup needs, of all rows, only the ones its search chose and the next
  // search's candidates: rows [scan, scan + 512) as long as the search stays inside its first pass.  It used to wait until EVERY update
  // workgroup had finished and its stores had landed (~5 us between a workgroup's last store and the counter being seen) and published
  // behind that: at 65536 rows, where the update is the longer side, 8 of a step's 22 us.  Now (a) every update workgroup copies what it
  // will ever read of the shared state -- the step's record, the raw pivot rows, the selector map, THE FLAGS OF ITS ROWS -- into LDS and
  // registers up front; (b) those 512 priority rows are dealt out over all workgroups, a few each, and ONE wave of every workgroup
  // updates its few first and raises st->cntP when they have landed; (c) the publication waits for cntP only and then overwrites record,
  // rows and flags while the update of all other rows is still running.  Workgroups with more than 2048 rows (> 522,000 rows on 255
  // workgroups) keep the old hand-off: their flags do not fit the 2 KiB set aside here; so do workgroups with fewer than 192 rows (below
  // ~49,000 rows the search is the longer side of a step and the extra pass only costs: 4096^2 1.14 -> 1.18 ms with it).
  const bool pre = LOOK && per >= kEarlyMinRows && per <= 2048;
  unsigned char *const sfl = reinterpret_cast<unsigned char *>(smap + 256);
  long long P0 = 0, P1 = 0;
  if (pre) {
    P0 = st->scan;
    P1 = min((long long)m, P0 + 512);
    for (long long i = tid; i < row_e - row_b; i += 1024) sfl[i] = rowflag[row_b + i];
  }
  const int nS = sw - j;
  // tracking words past the one that holds this step's last pivot (index jbase + np - 1 of the block) are zero in every row and
  // in every table entry: those lanes neither load nor store (on average a third of the step's traffic)
  const int uw_live = min(uw, ((st->jbase + np - 1) >> 6) + 1);
  // The tables (round 5): thread (g = wave, wd = lane) owns word wd of group g -- it fetches the group's FOUR raw rows' words straight from
  // ptab and writes all sixteen XOR combinations itself, walking them in Gray-code order (one XOR and one conflict-free 8-byte
  // LDS write per entry), ONE barrier behind it.  (Until round 5: the single-bit entries through LDS first, then three levels of
  // combinations, each behind a barrier and each done by the few waves whose entry index had the level's popcount -- 6, 4 and 1 of
  // the 16 waves, sixteen turns each: ~3.5 us of a step, and at 65536 rows the update is what a step waits for.)
  {
    const int g = wave, wd = lane;
    const u64 b0 = ptab[(g * 4 + 0) * 64 + wd], b1 = ptab[(g * 4 + 1) * 64 + wd], b2 = ptab[(g * 4 + 2) * 64 + wd],
              b3 = ptab[(g * 4 + 3) * 64 + wd];
    u64 *const tg = tab + (g * 16) * 64 + wd;
    u64 v = 0;
    tg[0 * 64] = v;
    v ^= b0, tg[1 * 64] = v;
    v ^= b1, tg[3 * 64] = v;
    v ^= b0, tg[2 * 64] = v;
    v ^= b2, tg[6 * 64] = v;
    v ^= b0, tg[7 * 64] = v;
    v ^= b1, tg[5 * 64] = v;
    v ^= b0, tg[4 * 64] = v;
    v ^= b3, tg[12 * 64] = v;
    v ^= b0, tg[13 * 64] = v;
    v ^= b1, tg[15 * 64] = v;
    v ^= b0, tg[14 * 64] = v;
    v ^= b2, tg[10 * 64] = v;
    v ^= b0, tg[11 * 64] = v;
    v ^= b1, tg[9 * 64] = v;
    v ^= b0, tg[8 * 64] = v;
  }
  __syncthreads();
  UPD_STAMP(3);  // tables built
  const bool isS = lane < nS, act = lane < nS + uw_live;
  const int tword = isS ? j + lane : (act ? sw + (lane - nS) : 0);
  u64 *const base = isS ? A + wc + lane : U + (lane - nS);
  const long long ld = isS ? lda : ldu;
  constexpr int RG = 8;
  // every wave takes ONE contiguous run of the workgroup's rows, all runs the same length: 65536 rows over 255 workgroups are 264 rows
  // each = 16.5 per wave -- dealt out in passes of 16 x 8 rows, one wave had a third pass of its own to make while fifteen waited
  // (round 5: at 65536 rows the update is what a step waits for)
  const long long wrun = (row_e - row_b + 15) / 16;
  const long long wave_b = row_b + wave * wrun, wave_e = min(row_e, wave_b + wrun);
  // a lane group of 8 maps one row's selector: two nibbles per lane through the selector map, folded over the eight lanes by DPP
  auto map_selector = [&](u64 sq) -> u64 {
    const int h2 = (lane & 7) * 2;
    const u64 x = smap[h2 * 16 + (int)((sq >> (4 * h2)) & 15)] ^ smap[(h2 + 1) * 16 + (int)((sq >> (4 * h2 + 4)) & 15)];
    unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
    lo ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]: lane ^ 1
    hi ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0xB1, 0xf, 0xf, true);
    lo ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]: lane ^ 2
    hi ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x4E, 0xf, 0xf, true);
    lo ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x141, 0xf, 0xf, true);  // row_half_mirror: lane ^ 7 (the quads are uniform by now)
    hi ^= (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x141, 0xf, 0xf, true);
    return (u64)lo | ((u64)hi << 32);
  };
  auto put = [&](u64 *p, u64 v) {
    if constexpr (LOOK) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the look-ahead workgroup reads the rows it chose)
    else *p = v;
  };
  // rows [rb, re); SKIP (a compile-time switch: the plain form must not pay for it): without [s0, s1), flags from this workgroup's LDS copy
  auto run = [&](auto skip_c, const long long rb, const long long re, const long long s0, const long long s1) {
  constexpr bool SKIP = decltype(skip_c)::value;
  auto valid = [&](long long r) { return SKIP ? (r < re && !(r >= s0 && r < s1)) : r < re; };
  auto flag_of = [&](long long r) -> int { return SKIP ? (int)sfl[r - row_b] : (int)rowflag[r]; };
  if (np == 64 && nS - 1 + uw_live <= 32) {
    // TWO ROWS PER LOOKUP (round 5).  The update is bound by its LDS reads -- sixteen 8-byte reads per row and wave, which cost the
    // same 4+ clocks with 33 lanes active as with 64 (tools/lds_exec_bench) -- and on a step that found all 64 pivots of its word
    // column a row has at most 32 live words besides that column: sw - j - 1 words right of it and uw_live <= j + 1 tracking words.
    // The column itself needs no lookup then (every one of its 64 columns is a pivot column: a pivot row keeps its unit bit, every
    // other row becomes zero on it).  So lanes 0..31 take one row and lanes 32..63 another, each half with its own table entries:
    // half the LDS time per row.  Steps with fewer pivots (the matrix's last columns, rank-deficient input) take the loop below.
    const int h = lane >> 5, l32 = lane & 31, nS1 = nS - 1;
    const bool isS2 = l32 < nS1, act2 = l32 < nS1 + uw_live;
    const int tword2 = isS2 ? j + 1 + l32 : (act2 ? sw + (l32 - nS1) : 0);
    u64 *const base2 = isS2 ? A + wc + 1 + l32 : U + (l32 - nS1);
    const long long ld2 = isS2 ? lda : ldu;
    const u32 tw8 = (u32)tword2 * 8;
    for (long long r0 = rb; r0 < re; r0 += 16) {
      const long long rf = r0 + (lane & 15);
      const int flv = valid(rf) ? flag_of(rf) : 0;  // lanes 0..15: flags of the pass's rows
      const bool pvl = flv >= 1 && flv <= 64;
      // the selectors of rows r0 + q (lane group q, first round) and r0 + 8 + q (second round)
      u64 xs[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const long long rq = r0 + 8 * t + (lane >> 3);
        const int flq = valid(rq) ? flag_of(rq) : 0;
        u64 sq = valid(rq) ? A[rq * lda + wc] : 0;
        if (flq >= 1 && flq <= 64) sq = 1ull << (flq - 1);
        xs[t] = map_selector(sq);
      }
      u64 old[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long long r = r0 + q + 8 * h;
        old[q] = (act2 && valid(r)) ? base2[r * ld2] : 0;
      }
      const unsigned pivm = (unsigned)__ballot(pvl && lane < 16) >> (8 * h);  // bit q: this half's row of pair q is a pivot of this step
      // a pivot of this step becomes its reduced form (the XOR of the raw rows in its b_trk: what it held does not count); every other
      // row adds what its word selects.  (Here and not at the store: the first unconditional use of the loaded words is where the
      // compiler waits for them -- once; inside the skippable turns below it waited in every turn, with vmcnt(0), i.e. for the
      // previous turn's STORE as well.)
      u32 ol[8], oh[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        ol[q] = ((pivm >> q) & 1) ? 0 : (u32)old[q], oh[q] = ((pivm >> q) & 1) ? 0 : (u32)(old[q] >> 32);
        asm volatile("" : "+v"(ol[q]), "+v"(oh[q]));  // (pins the wait and the select here)
      }
      if (lane < 16 && valid(rf)) put(A + rf * lda + wc, pvl ? 1ull << (flv - 1) : 0);  // the step's own word column
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned loa = __builtin_amdgcn_readlane((unsigned)xs[0], 8 * q), hia = __builtin_amdgcn_readlane((unsigned)(xs[0] >> 32), 8 * q);
        const unsigned lob = __builtin_amdgcn_readlane((unsigned)xs[1], 8 * q), hib = __builtin_amdgcn_readlane((unsigned)(xs[1] >> 32), 8 * q);
        if ((loa | hia | lob | hib) == 0) continue;
        const unsigned lo = h ? lob : loa, hi = h ? hib : hia;
        u32 al = ol[q], ah = oh[q];
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
          // per-lane entries: byte address = group * 8192 (an instruction offset) + nibble * 512 + word * 8 -- the nibble shifted into
          // place and masked, OR-ed onto the word's offset (disjoint bits: one shift and one v_and_or_b32 per lookup)
          const u32 sel = g < 8 ? lo : hi;
          const int sh0 = 4 * (g & 7) - 9, sh1 = sh0 + 4;
          const u32 a0 = ((sh0 < 0 ? sel << -sh0 : sel >> sh0) & 0x1e00u) | tw8, a1 = ((sh1 < 0 ? sel << -sh1 : sel >> sh1) & 0x1e00u) | tw8;
          const u64 t0 = *(const u64 *)((const char *)tab + a0 + g * 8192), t1 = *(const u64 *)((const char *)tab + a1 + (g + 1) * 8192);
          al = __builtin_amdgcn_bitop3_b32(al, (u32)t0, (u32)t1, 0x96);
          ah = __builtin_amdgcn_bitop3_b32(ah, (u32)(t0 >> 32), (u32)(t1 >> 32), 0x96);
        }
        const long long r = r0 + q + 8 * h;
        if (act2 && valid(r)) put(base2 + r * ld2, (u64)al | ((u64)ah << 32));
      }
    }
  } else
  for (long long r0 = rb; r0 < re; r0 += RG) {
    u64 old[RG];
    const long long rf = r0 + (lane & 7);
    const int flv = valid(rf) ? flag_of(rf) : 0;  // lanes 0..7: flags of the pass's rows
    // lane group q = lane / 8 maps the selector of row r0 + q: its word's bits on the pivot columns (a pivot row of this step: the
    // single bit of its pivot column, i.e. its reduced form)
    const long long rq = r0 + (lane >> 3);
    const int flq = valid(rq) ? flag_of(rq) : 0;
    u64 sq = valid(rq) ? A[rq * lda + wc] & pcmask : 0;
    if (flq >= 1 && flq <= 64) sq = 1ull << (flq - 1);
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const long long r = r0 + q;
      old[q] = (act && valid(r)) ? base[r * ld] : 0;
    }
    const u64 x = map_selector(sq);
#pragma unroll
    for (int q = 0; q < RG; ++q) {
      const int fl = __builtin_amdgcn_readlane(flv, q);
      const unsigned lo = __builtin_amdgcn_readlane((unsigned)x, 8 * q), hi = __builtin_amdgcn_readlane((unsigned)(x >> 32), 8 * q);
      if ((lo | hi) == 0) continue;
      u32 al = 0, ah = 0;
#pragma unroll
      for (int g = 0; g < 16; g += 2) {  // (three-input XORs: one VALU instruction per half and two lookups)
        const unsigned i0 = ((g < 8 ? lo : hi) >> (4 * (g & 7))) & 15u, i1 = ((g < 8 ? lo : hi) >> (4 * (g & 7) + 4)) & 15u;
        const u64 t0 = tab[(g * 16 + i0) * 64 + tword], t1 = tab[((g + 1) * 16 + i1) * 64 + tword];
        al = __builtin_amdgcn_bitop3_b32(al, (u32)t0, (u32)t1, 0x96);
        ah = __builtin_amdgcn_bitop3_b32(ah, (u32)(t0 >> 32), (u32)(t1 >> 32), 0x96);
      }
      const u64 acc = (u64)al | ((u64)ah << 32);
      // a pivot of this step becomes its reduced form (the XOR of the raw rows in its b_trk); every other row adds what its word selects
      const bool piv = fl >= 1 && fl <= 64;
      if (act && valid(r0 + q)) put(base + (r0 + q) * ld, piv ? acc : old[q] ^ acc);
    }
  }
  };
  if (pre && wave == 15) {  // this workgroup's few of the priority rows first, by one wave; the other fifteen are on their runs already
    const long long pr = (P1 - P0 + nupd - 1) / nupd;
    const long long pb = P0 + (long long)blockIdx.x * pr, pe = min(P1, pb + pr);
    run(std::false_type{}, pb, pe, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (they have landed)
    if (lane == 0 && !(fault == 1 && blockIdx.x == 0)) __hip_atomic_fetch_add(&st->cntP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (pre) run(std::true_type{}, wave_b, wave_e, P0, P1);
  else run(std::false_type{}, wave_b, wave_e, 0, 0);
  UPD_STAMP(4);  // wave 0's run of rows done (stores issued)
  if constexpr (LOOK) raise(&st->cnt2);
  UPD_STAMP(5);  // every wave's run done and landed, count raised
}

// ---- end of a block: block pivot j (row blkpiv[j]) goes to row r0 + j; the non-pivot rows that sit inside
// [r0, r0 + rp) trade places with the pivot rows below that range ----
__global__ __launch_bounds__(1024) void gf2_elim_plan_kernel(gf2k_elim_state *st, unsigned char *rowflag,
                                                             const int *__restrict__ blkpiv, int *__restrict__ moves) {
  __shared__ int s_wcnt[2][16], s_run[2];
  __shared__ int s_vac[GF2K_ELIM_BLOCK_PIVOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = st->r0, rp = st->r_cur - r0;
  int *mv_src = moves, *mv_dst = moves + 2 * GF2K_ELIM_BLOCK_PIVOTS;
  // two elements per thread: t = tid and tid + 1024 (rp <= 2048)
  int prow[2], dpos[2], vpos[2];
  bool isdisp[2], isvac[2];
  for (int h = 0; h < 2; ++h) {
    const int t = tid + h * 1024;
    prow[h] = t < rp ? blkpiv[t] : -1;
    isvac[h] = t < rp && prow[h] >= r0 + rp;     // block pivot t comes from below the target range
    isdisp[h] = t < rp && rowflag[r0 + t] == 0;  // target slot r0 + t holds an ordinary row
  }
  if (tid < 2) s_run[tid] = 0;
  __syncthreads();
  for (int h = 0; h < 2; ++h) {  // ranks in increasing t: ballot inside a wave, wave counts through LDS
    const u64 dm = __ballot(isdisp[h]), vm = __ballot(isvac[h]);
    if (lane == 0) {
      s_wcnt[0][wave] = __popcll(dm);
      s_wcnt[1][wave] = __popcll(vm);
    }
    __syncthreads();
    int doff = s_run[0], voff = s_run[1];
    for (int w = 0; w < wave; ++w) {
      doff += s_wcnt[0][w];
      voff += s_wcnt[1][w];
    }
    dpos[h] = doff + __popcll(dm & ((1ull << lane) - 1));
    vpos[h] = voff + __popcll(vm & ((1ull << lane) - 1));
    __syncthreads();
    if (tid == 0)
      for (int w = 0; w < 16; ++w) {
        s_run[0] += s_wcnt[0][w];
        s_run[1] += s_wcnt[1][w];
      }
    __syncthreads();
  }
  for (int h = 0; h < 2; ++h)
    if (isvac[h]) s_vac[vpos[h]] = prow[h];
  __syncthreads();  // all flags have been read, all vacated rows are listed
  for (int h = 0; h < 2; ++h) {
    const int t = tid + h * 1024;
    if (t < rp) {
      mv_src[t] = prow[h];
      mv_dst[t] = r0 + t;
      rowflag[prow[h]] = 0;  // flags are per block
    }
    if (isdisp[h]) {  // the k-th ordinary row of the target range takes the place of the k-th pivot from below
      mv_src[rp + dpos[h]] = r0 + t;
      mv_dst[rp + dpos[h]] = s_vac[dpos[h]];
    }
  }
  if (tid == 0) {
    st->nmoves = rp + s_run[0];
    st->np = 0;
  }
}

__global__ __launch_bounds__(256) void gf2_elim_gather_kernel(const u64 *__restrict__ A, long long lda, long long aw,
                                                              long long c0w, const u64 *__restrict__ U, long long ldu, int uw,
                                                              const gf2k_elim_state *st, const int *__restrict__ moves,
                                                              u64 *__restrict__ tmp, long long tld) {
  const int q = blockIdx.x;
  if (q >= st->nmoves) return;
  const long long src = moves[q];
  const long long nA = aw - c0w;
  for (long long w = threadIdx.x; w < nA + uw; w += 256)
    tmp[q * tld + w] = w < nA ? A[src * lda + c0w + w] : U[src * ldu + (w - nA)];
}

__global__ __launch_bounds__(256) void gf2_elim_scatter_kernel(u64 *__restrict__ A, long long lda, long long aw,
                                                               long long c0w, u64 *__restrict__ U, long long ldu, int uw,
                                                               const gf2k_elim_state *st, const int *__restrict__ moves,
                                                               const u64 *__restrict__ tmp, long long tld) {
  const int q = blockIdx.x;
  if (q >= st->nmoves) return;
  const long long dst = moves[2 * GF2K_ELIM_BLOCK_PIVOTS + q];
  const long long nA = aw - c0w;
  for (long long w = threadIdx.x; w < nA + uw; w += 256) {
    const u64 v = tmp[q * tld + w];
    if (w < nA)
      A[dst * lda + c0w + w] = v;
    else
      U[dst * ldu + (w - nA)] = v;
  }
}

__global__ void gf2_elim_begin_block_kernel(gf2k_elim_state *st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st->r0 = st->r_cur;
    st->scan = st->r_cur;
    st->np = 0;
    st->lastword = -1;
    st->cntP = 0;
    st->cnt2 = 0;
  }
}

// last non-zero word (absolute index, words [w_lo, aw)) over the block's pivot rows [r0, r_cur) after the permutation: the
// trailing product only has to cover the columns up to it (an augmented identity is mostly zero columns for a long time).
// One wave per row, scanning backwards 64 words at a time: a dense row is done after one load.
__global__ __launch_bounds__(256) void gf2_elim_lastword_kernel(const u64 *__restrict__ A, long long lda, long long aw,
                                                                long long w_lo, gf2k_elim_state *st) {
  const int r0 = st->r0, rp = st->r_cur - r0;
  const int lane = threadIdx.x & 63;
  int best = -1;
  for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rp; r += gridDim.x * 4) {
    const u64 *row = A + (long long)(r0 + r) * lda;
    for (long long hi = aw; hi > w_lo; hi -= 64) {  // words [hi-64, hi)
      const long long w = hi - 64 + lane;
      const u64 nz = __ballot(w >= w_lo && row[w] != 0);
      if (nz) {
        best = max(best, (int)(hi - 64 + (63 - __builtin_clzll(nz))));
        break;
      }
      if (hi - 64 <= (long long)best) break;  // nothing beyond what another row of this wave already reached
    }
  }
  // most waves find the same last word: look before taking the atomic (2048 atomics on one address cost ~20 us)
  if (lane == 0 && best >= 0 && best > *reinterpret_cast<volatile int *>(&st->lastword)) atomicMax(&st->lastword, best);
}

// pivot row j of the block holds its own original content on the right-hand columns: U'[r0+j][j] ^= 1 turns
// "A ^= U' * P" into "A = U * P" for those rows
__global__ __launch_bounds__(256) void gf2_elim_toggle_kernel(u64 *__restrict__ U, long long ldu, const gf2k_elim_state *st) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int r0 = st->r0;
  if (j < st->r_cur - r0) U[(long long)(r0 + j) * ldu + (j >> 6)] ^= 1ull << (j & 63);
}

// M[i][col0 + i] = 1 for i < n (identity block of an augmented matrix); col0 + i addressed in bits
__global__ __launch_bounds__(256) void gf2_set_diag_kernel(u64 *__restrict__ M, long long ld, int n, long long col0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const long long c = col0 + i;
    M[(long long)i * ld + (c >> 6)] |= 1ull << (c & 63);
  }
}

// X[pivcols[k]] = R[k] for k < rank (rows of the solution addressed by pivot column); X is zeroed by the caller
__global__ __launch_bounds__(256) void gf2_scatter_rows_kernel(u64 *__restrict__ X, long long ldx, const u64 *__restrict__ R,
                                                               long long ldr, int words, const int *__restrict__ pivcols,
                                                               int rank) {
  const int k = blockIdx.x;
  if (k >= rank) return;
  const long long d = pivcols[k];
  for (int w = threadIdx.x; w < words; w += 256) X[d * ldx + w] = R[(long long)k * ldr + w];
}

// flag = 1 if any word of rows [row_lo, rows) x words [0, words) is non-zero
__global__ __launch_bounds__(256) void gf2_any_nonzero_kernel(const u64 *__restrict__ M, long long ld, int row_lo, int rows,
                                                              int words, int *flag) {
  const long long total = (long long)(rows - row_lo) * words;
  bool nz = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
    nz |= M[(row_lo + i / words) * ld + i % words] != 0;
  if (__ballot(nz) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------------
// small matrices: the whole matrix in LDS, textbook Gauss(-Jordan) by one workgroup
// ---------------------------------------------------------------------------------------------
// For matrices of at most 1024 rows that fit into LDS (rows x (words | 1) <= 19000 words) the blocked algorithm above is
// all latency (two launches and ~50 us per 64 columns); here one workgroup keeps the matrix in LDS (odd row stride: a
// column of words spreads over the banks) and eliminates column by column: ballot for the first row with a 1, swap it up,
// every other row with a 1 adds the pivot row (one thread per row, the pivot row is a broadcast read).
__global__ __launch_bounds__(1024) void gf2_elim_small_kernel(u64 *__restrict__ A, long long lda, int m, int ncols, int limit,
                                                              int full, int *__restrict__ rank_out, int *__restrict__ pivcols) {
  extern __shared__ __attribute__((aligned(16))) u64 M[];
  __shared__ int s_first[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int aw = (ncols + 63) >> 6, stride = aw | 1;
  for (int idx = tid; idx < m * aw; idx += 1024) M[(idx / aw) * stride + idx % aw] = A[(long long)(idx / aw) * lda + idx % aw];
  __syncthreads();
  int rank = 0;
  for (int c = 0; c < limit && rank < m; ++c) {
    const int cw = c >> 6;
    const u64 bit = 1ull << (c & 63);
    const bool mine = tid < m && (M[tid * stride + cw] & bit);
    const u64 cand = __ballot(mine && tid >= rank);
    if (lane == 0) s_first[wave] = cand ? wave * 64 + __builtin_ctzll(cand) : 0x7fffffff;
    __syncthreads();
    int p = 0x7fffffff;
#pragma unroll
    for (int w = 0; w < 16; ++w) p = min(p, s_first[w]);
    if (p == 0x7fffffff) {  // no pivot in this column (uniform)
      __syncthreads();      // s_first is rewritten by the next column
      continue;
    }
    if (p != rank) {
      for (int w = cw + tid; w < aw; w += 1024) {  // both rows are zero left of the column; rows may be wider than 1024 words
        const u64 t = M[rank * stride + w];
        M[rank * stride + w] = M[p * stride + w];
        M[p * stride + w] = t;
      }
    }
    __syncthreads();
    // after the swap row `rank` holds the pivot (row p holds what was in row `rank`): every row looks at its bit again
    const bool add = tid < m && tid != rank && (M[tid * stride + cw] & bit);
    if (add && (full || tid > rank)) {
      for (int w = cw; w < aw; ++w) M[tid * stride + w] ^= M[rank * stride + w];
    }
    if (tid == 0) pivcols[rank] = c;
    ++rank;
    __syncthreads();
  }
  for (int idx = tid; idx < m * aw; idx += 1024) A[(long long)(idx / aw) * lda + idx % aw] = M[(idx / aw) * stride + idx % aw];
  if (tid == 0) *rank_out = rank;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
extern "C" hipError_t gf2k_elim_begin_block(gf2k_elim_state *st, hipStream_t s) {
  hipLaunchKernelGGL(gf2_elim_begin_block_kernel, dim3(1), dim3(64), 0, s, st);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_elim_step(u64 *A, long long lda, int m, long long c0w, int sw, int j, u64 colmask, int full,
                                     u64 *U, long long ldu, int uw, gf2k_elim_state *st, int *pivcols, u64 *ptab,
                                     unsigned char *rowflag, int *blkpiv, u64 colmask_next, int lookahead, hipStream_t s) {
  if (sw + uw > 64 || j >= sw || sw * 64 > GF2K_ELIM_BLOCK_PIVOTS) return hipErrorInvalidValue;
  // dynamic LDS limits are per device: set them once per device, not per step (the call costs host time that a chain
  // of 50 us steps notices)
  // (concurrent host threads: the flags are atomics, setting the attribute twice is harmless; keyed by the real device ordinal)
  constexpr int kMaxDev = 1024;
  static std::atomic<int> cu_of[kMaxDev];  // 0 = not set up yet, else the device's compute-unit count
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
  int cus = cu_of[dev].load(std::memory_order_acquire);
  if (!cus) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gf2_elim_update_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kUpdLds);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gf2_elim_update_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kUpdLds);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (cus < 1) cus = 1;
    cu_of[dev].store(cus, std::memory_order_release);
  }
  // 8 rows per wave and pass, 16 waves per workgroup, one workgroup per CU (128 KiB of LDS tables each).  The look-ahead form needs
  // its WHOLE grid resident at once (the extra workgroup waits for the others): update workgroups <= CUs - 1 -- 255 on an MI355X,
  // fewer on a partition -- and the two-launch form on a device with fewer than two CUs
  if (lookahead && cus < 2) lookahead = 0;
  const int cap = lookahead ? cus - 1 : cus;
  int grid = (m + 127) / 128;
  if (grid > cap) grid = cap;
  if (grid < 1) grid = 1;
  if (!lookahead) {
    hipLaunchKernelGGL(gf2_elim_pivot_kernel, dim3(1), dim3(1024), 0, s, A, lda, m, c0w, sw, j, colmask, U, ldu, uw, st, pivcols, ptab,
                       rowflag, blkpiv, 0);
    hipLaunchKernelGGL(gf2_elim_update_kernel<false>, dim3(grid), dim3(1024), kUpdLds, s, A, lda, m, full, c0w, sw, j, U, ldu, uw, st,
                       ptab, rowflag, colmask_next, pivcols, blkpiv);
    return hipGetLastError();
  }
  // look-ahead: the search of step j + 1 runs inside the update launch of step j; the block's first step is searched on its own
  if (j == 0)
    hipLaunchKernelGGL(gf2_elim_pivot_kernel, dim3(1), dim3(1024), 0, s, A, lda, m, c0w, sw, j, colmask, U, ldu, uw, st, pivcols, ptab,
                       rowflag, blkpiv, /*make_stash=*/1);
  if (j + 1 < sw)
    hipLaunchKernelGGL(gf2_elim_update_kernel<true>, dim3(grid + 1), dim3(1024), kUpdLds, s, A, lda, m, full, c0w, sw, j, U, ldu, uw, st,
                       ptab, rowflag, colmask_next, pivcols, blkpiv);
  else
    hipLaunchKernelGGL(gf2_elim_update_kernel<false>, dim3(grid), dim3(1024), kUpdLds, s, A, lda, m, full, c0w, sw, j, U, ldu, uw, st,
                       ptab, rowflag, colmask_next, pivcols, blkpiv);
  return hipGetLastError();
}

// rows <= 1024 and rows * (words | 1) <= 19000: everything in the LDS of one workgroup (returns hipErrorInvalidValue otherwise)
extern "C" hipError_t gf2k_elim_small(u64 *A, long long lda, int m, int ncols, int limit, int full, int *rank_out, int *pivcols,
                                      hipStream_t s) {
  const int aw = (ncols + 63) / 64, stride = aw | 1;
  if (m < 1 || m > 1024 || (long long)m * stride > 19000) return hipErrorInvalidValue;
  const int lds = m * stride * 8;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&gf2_elim_small_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gf2_elim_small_kernel, dim3(1), dim3(1024), lds, s, A, lda, m, ncols, limit, full, rank_out, pivcols);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_elim_end_block(u64 *A, long long lda, long long aw, long long c0w, u64 *U, long long ldu, int uw,
                                          gf2k_elim_state *st, unsigned char *rowflag, const int *blkpiv, int *moves, u64 *tmp,
                                          long long tld, long long w_right, hipStream_t s) {
  hipLaunchKernelGGL(gf2_elim_plan_kernel, dim3(1), dim3(1024), 0, s, st, rowflag, blkpiv, moves);
  hipLaunchKernelGGL(gf2_elim_gather_kernel, dim3(2 * GF2K_ELIM_BLOCK_PIVOTS), dim3(256), 0, s, A, lda, aw, c0w, U, ldu, uw,
                     st, moves, tmp, tld);
  hipLaunchKernelGGL(gf2_elim_scatter_kernel, dim3(2 * GF2K_ELIM_BLOCK_PIVOTS), dim3(256), 0, s, A, lda, aw, c0w, U, ldu, uw,
                     st, moves, tmp, tld);
  hipLaunchKernelGGL(gf2_elim_toggle_kernel, dim3(GF2K_ELIM_BLOCK_PIVOTS / 256), dim3(256), 0, s, U, ldu, st);
  if (w_right < aw) hipLaunchKernelGGL(gf2_elim_lastword_kernel, dim3(512), dim3(256), 0, s, A, lda, aw, w_right, st);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_set_diag(u64 *M, long long ld, int n, long long col0, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(gf2_set_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, M, ld, n, col0);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_scatter_rows(u64 *X, long long ldx, const u64 *R, long long ldr, int words, const int *pivcols,
                                        int rank, hipStream_t s) {
  if (rank <= 0 || words <= 0) return hipSuccess;
  hipLaunchKernelGGL(gf2_scatter_rows_kernel, dim3(rank), dim3(256), 0, s, X, ldx, R, ldr, words, pivcols, rank);
  return hipGetLastError();
}

extern "C" hipError_t gf2k_any_nonzero(const u64 *M, long long ld, int row_lo, int rows, int words, int *flag, hipStream_t s) {
  if (rows <= row_lo || words <= 0) return hipSuccess;
  long long total = (long long)(rows - row_lo) * words;
  long long grid = (total + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(gf2_any_nonzero_kernel, dim3((unsigned)grid), dim3(256), 0, s, M, ld, row_lo, rows, words, flag);
  return hipGetLastError();
}
