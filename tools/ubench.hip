// ubench -- instruction issue-rate calibration on gfx950 (development tool).
// Each kernel runs an unrolled body REP times per wave and reports cycles per body instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, u32 *sink, int iters) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  u32 a[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) a[i] = threadIdx.x * 2654435761u + i;
  u32 t0 = lane * 16 + (threadIdx.x >> 6) * 2048, t1 = t0 + 1024, sel = 0x0c020400;
  u32x4 r0 = {1, 2, 3, 4}, r1 = {5, 6, 7, 8};
  u32x4 q[6] = {r0, r1, r0, r1, r0, r1};
  asm volatile("s_mov_b32 m0, 0" ::: "memory");
  __syncthreads();
  unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {  // 64 independent v_xor
      asm volatile(
#define X8(b) "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %9\n v_xor_b32 %2, %2, %10\n v_xor_b32 %3, %3, %11\n" \
              "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %10\n v_xor_b32 %7, %7, %11\n"
          X8(0) X8(0) X8(0) X8(0) X8(0) X8(0) X8(0) X8(0)
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
          : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]));
    } else if constexpr (MODE == 1) {  // 8 x (perm + ds_read_b128 + 4 xor), waits batched: 48 instr + 2 waits
      asm volatile(
#define L1(acc0, acc1, acc2, acc3) \
          "v_perm_b32 %12, %8, %10, %11\n ds_read_b128 %13, %12\n" \
          "v_perm_b32 %12, %9, %10, %11\n ds_read_b128 %14, %12\n" \
          "s_waitcnt lgkmcnt(1)\n" \
          "v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n" \
          "s_waitcnt lgkmcnt(0)\n" \
          "v_xor_b32 %4, %4, %9\n v_xor_b32 %5, %5, %9\n v_xor_b32 %6, %6, %9\n v_xor_b32 %7, %7, %9\n"
          L1(0,0,0,0) L1(0,0,0,0) L1(0,0,0,0) L1(0,0,0,0)
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
          : "v"(a[8] & 0xff), "v"(a[9] & 0xff), "v"(t0), "s"(sel), "v"(a[12]), "v"(r0), "v"(r1) : "memory");
    } else if constexpr (MODE == 2) {  // ds_read_b128 only, 16 per body, one wait at the end
      asm volatile(
#define R4 "ds_read_b128 %0, %2\n ds_read_b128 %1, %3\n ds_read_b128 %0, %2 offset:4096\n ds_read_b128 %1, %3 offset:4096\n"
          R4 R4 R4 R4 "s_waitcnt lgkmcnt(0)\n"
          : "+v"(r0), "+v"(r1) : "v"(t0), "v"(t1));
    } else if constexpr (MODE == 3) {  // 32 x (xor, s_nop) : does a scalar/nop take a slot?
      asm volatile(
#define XN "v_xor_b32 %0, %0, %4\n s_nop 0\n v_xor_b32 %1, %1, %5\n s_nop 0\n v_xor_b32 %2, %2, %6\n s_nop 0\n v_xor_b32 %3, %3, %7\n s_nop 0\n"
          XN XN XN XN XN XN XN XN
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]));
    } else if constexpr (MODE == 4) {  // 32 x (xor, s_waitcnt lgkmcnt(0)) with nothing outstanding
      asm volatile(
#define XW "v_xor_b32 %0, %0, %4\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %1, %1, %5\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %2, %2, %6\n s_waitcnt lgkmcnt(0)\n v_xor_b32 %3, %3, %7\n s_waitcnt lgkmcnt(0)\n"
          XW XW XW XW XW XW XW XW
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]));
    } else if constexpr (MODE == 5) {  // 32 x (xor, s_mov)
      asm volatile(
#define XS "v_xor_b32 %0, %0, %4\n s_mov_b32 s40, 1\n v_xor_b32 %1, %1, %5\n s_mov_b32 s41, 2\n v_xor_b32 %2, %2, %6\n s_mov_b32 s42, 3\n v_xor_b32 %3, %3, %7\n s_mov_b32 s43, 4\n"
          XS XS XS XS XS XS XS XS
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]) : "s40", "s41", "s42", "s43");
    } else if constexpr (MODE == 6) {  // v_perm only x64
      asm volatile(
#define P8 "v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %6\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %4, %5, %6\n" \
           "v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %6\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %4, %5, %6\n"
          P8 P8 P8 P8 P8 P8 P8 P8
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]), "s"(sel));
    } else if constexpr (MODE == 7) {  // ds_write_addtid_b32 x32 with m0 set once
      asm volatile("s_mov_b32 m0, 0\n s_nop 0\n"
#define W8 "ds_write_addtid_b32 %0 offset:0\n ds_write_addtid_b32 %0 offset:256\n ds_write_addtid_b32 %0 offset:512\n ds_write_addtid_b32 %0 offset:768\n" \
           "ds_write_addtid_b32 %0 offset:1024\n ds_write_addtid_b32 %0 offset:1280\n ds_write_addtid_b32 %0 offset:1536\n ds_write_addtid_b32 %0 offset:1792\n"
          W8 W8 W8 W8 "s_waitcnt lgkmcnt(0)\n" :: "v"(a[0]) : "memory");
    } else if constexpr (MODE == 8) {  // ds_write_b64 x32
      asm volatile(
#define V8 "ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:512\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:1536\n" \
           "ds_write_b64 %0, %1 offset:2048\n ds_write_b64 %0, %1 offset:2560\n ds_write_b64 %0, %1 offset:3072\n ds_write_b64 %0, %1 offset:3584\n"
          V8 V8 V8 V8 "s_waitcnt lgkmcnt(0)\n" :: "v"(lane * 8), "v"(*(unsigned long long *)&a[0]) : "memory");
    } else if constexpr (MODE == 9) {  // 64 v_mov_b32_sdwa (byte insert)
      asm volatile(
#define S8 "v_mov_b32_sdwa %0, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_mov_b32_sdwa %1, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n" \
           "v_mov_b32_sdwa %2, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n v_mov_b32_sdwa %3, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n" \
           "v_mov_b32_sdwa %0, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_mov_b32_sdwa %1, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n" \
           "v_mov_b32_sdwa %2, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n v_mov_b32_sdwa %3, %5 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
          S8 S8 S8 S8 S8 S8 S8 S8
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]));
    } else if constexpr (MODE == 10) {  // 64 v_and_or_b32
      asm volatile(
#define A8 "v_and_or_b32 %0, %4, %5, %0\n v_and_or_b32 %1, %4, %5, %1\n v_and_or_b32 %2, %4, %5, %2\n v_and_or_b32 %3, %4, %5, %3\n" \
           "v_and_or_b32 %0, %4, %5, %0\n v_and_or_b32 %1, %4, %5, %1\n v_and_or_b32 %2, %4, %5, %2\n v_and_or_b32 %3, %4, %5, %3\n"
          A8 A8 A8 A8 A8 A8 A8 A8
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]));
    } else if constexpr (MODE == 11) {
      asm volatile("s_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %4, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %5, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %6, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %7, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %8, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %9, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %4, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %5, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %6, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %7, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %8, %11\ns_waitcnt lgkmcnt(5)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %9, %11"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(a[8]), "+v"(t0)
          : "v"(a[9]) : "memory");
    } else if constexpr (MODE == 12) {
      asm volatile("s_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %4, %11\nds_write_addtid_b32 %12 offset:4096\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %5, %11\nds_write_addtid_b32 %12 offset:4352\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %6, %11\nds_write_addtid_b32 %12 offset:4608\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %7, %11\nds_write_addtid_b32 %12 offset:4864\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %8, %11\nds_write_addtid_b32 %12 offset:5120\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %9, %11\nds_write_addtid_b32 %12 offset:5376\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %4, %11\nds_write_addtid_b32 %12 offset:5632\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %5, %11\nds_write_addtid_b32 %12 offset:5888\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %6, %11\nds_write_addtid_b32 %12 offset:6144\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %7, %11\nds_write_addtid_b32 %12 offset:6400\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %8, %11\nds_write_addtid_b32 %12 offset:6656\ns_waitcnt lgkmcnt(11)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %9, %11\nds_write_addtid_b32 %12 offset:6912"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(a[8]), "+v"(t0)
          : "v"(a[9]) : "memory");
    } else if constexpr (MODE == 13) {
      asm volatile("s_waitcnt lgkmcnt(10)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %4, %11\nds_write_addtid_b32 %12 offset:4096\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %5, %11\nds_write_addtid_b32 %12 offset:4352\ns_waitcnt lgkmcnt(10)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %6, %11\nds_write_addtid_b32 %12 offset:4608\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %7, %11\nds_write_addtid_b32 %12 offset:4864\ns_waitcnt lgkmcnt(10)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %8, %11\nds_write_addtid_b32 %12 offset:5120\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %9, %11\nds_write_addtid_b32 %12 offset:5376\ns_waitcnt lgkmcnt(10)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %4, %11\nds_write_addtid_b32 %12 offset:5632\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %5, %11\nds_write_addtid_b32 %12 offset:5888\ns_waitcnt lgkmcnt(10)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %6, %11\nds_write_addtid_b32 %12 offset:6144\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %7, %11\nds_write_addtid_b32 %12 offset:6400\ns_waitcnt lgkmcnt(10)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %8, %11\nds_write_addtid_b32 %12 offset:6656\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %9, %11\nds_write_addtid_b32 %12 offset:6912"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(a[8]), "+v"(t0)
          : "v"(a[9]) : "memory");
    } else if constexpr (MODE == 14) {
      asm volatile("s_waitcnt lgkmcnt(4)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %4, %11\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %5, %11\ns_waitcnt lgkmcnt(4)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %6, %11\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %7, %11\ns_waitcnt lgkmcnt(4)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %8, %11\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %9, %11\ns_waitcnt lgkmcnt(4)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %4, %11\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %5, %11\ns_waitcnt lgkmcnt(4)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\nds_read_b128 %6, %11\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\nds_read_b128 %7, %11\ns_waitcnt lgkmcnt(4)\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\nds_read_b128 %8, %11\nv_xor_b32 %0, %0, %10\nv_xor_b32 %1, %1, %10\nv_xor_b32 %2, %2, %10\nv_xor_b32 %3, %3, %10\nv_mov_b32_sdwa %11, %10 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\nds_read_b128 %9, %11"
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(a[8]), "+v"(t0)
          : "v"(a[9]) : "memory");
    } else if constexpr (MODE == 18) {  // 64 v_bitop3_b32 (three-input XOR)
      asm volatile(
#define B8 "v_bitop3_b32 %0, %0, %4, %5 bitop3:0x96\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x96\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x96\n" \
           "v_bitop3_b32 %0, %0, %5, %4 bitop3:0x96\n v_bitop3_b32 %1, %1, %5, %4 bitop3:0x96\n v_bitop3_b32 %2, %2, %5, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %5, %4 bitop3:0x96\n"
          B8 B8 B8 B8 B8 B8 B8 B8
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(a[8]), "v"(a[9]));
    } else if constexpr (MODE == 19 || MODE == 20 || MODE == 21) {
      // 16 ds_read_b128 with a random 256-byte LDS row per 8-lane group.  19: the two groups of 16 consecutive lanes read
      // different 128-byte halves; 20: the same half (two-way conflicts expected); 21: 16 lanes share one row (today's tile kernel)
      const u32 grp = MODE == 21 ? (threadIdx.x >> 4) : (threadIdx.x >> 3);
      const u32 e0 = (grp * 2654435761u >> 13) & 0xff, e1 = (grp * 40503u + 977u) & 0xff;
      const u32 half = MODE == 19 ? ((lane >> 3) & 1) * 128u : 0u;
      const u32 slot = MODE == 21 ? (lane & 15) * 16u : (lane & 7) * 16u + half;
      const u32 ad0 = e0 * 256u + slot, ad1 = e1 * 256u + (MODE == 19 ? (slot ^ 128u) : slot);
      asm volatile(R4 R4 R4 R4 "s_waitcnt lgkmcnt(0)\n" : "+v"(r0), "+v"(r1) : "v"(ad0), "v"(ad1));
    } else if constexpr (MODE == 22) {  // 8 paired lookups: 2 perm, 2 read, 4 bitop3 each (64 instr + 8 waits)
      asm volatile(
#define L2 "v_perm_b32 %12, %8, %10, %11\n ds_read_b128 %13, %12\n" \
           "v_perm_b32 %12, %9, %10, %11\n ds_read_b128 %14, %12\n" \
           "s_waitcnt lgkmcnt(0)\n" \
           "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x96\n"
          L2 L2 L2 L2 L2 L2 L2 L2
          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
          : "v"(a[8] & 0xff), "v"(a[9] & 0xff), "v"(t0), "s"(sel), "v"(a[12]), "v"(r0), "v"(r1) : "memory");
    } else if constexpr (MODE == 15 || MODE == 16 || MODE == 17) {
      // 56 v_xor with 8 loads interleaved (one per 7 xor): MODE 15 dword/lane coalesced, 16 dwordx4/lane, 17 no loads (8 more xor)
      const unsigned *gp = (const unsigned *)sink + 1024 + (threadIdx.x & 63) * (MODE == 16 ? 4 : 1);
      u32 l0, l1, l2, l3, l4, l5, l6, l7;
      u32x4 w0, w1;
      (void)w0; (void)w1; (void)l0;
#define X7 "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n"
      if constexpr (MODE == 15) {
        asm volatile(X7 "global_load_dword %5, %13, off\n" X7 "global_load_dword %6, %13, off offset:256\n" X7 "global_load_dword %7, %13, off offset:512\n" X7 "global_load_dword %8, %13, off offset:768\n"
                     X7 "global_load_dword %9, %13, off offset:1024\n" X7 "global_load_dword %10, %13, off offset:1280\n" X7 "global_load_dword %11, %13, off offset:1536\n" X7 "global_load_dword %12, %13, off offset:1792\n s_waitcnt vmcnt(0)\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[8]), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3), "=&v"(l4), "=&v"(l5), "=&v"(l6), "=&v"(l7)
                     : "v"(gp) : "memory");
        a[9] ^= l0 ^ l1 ^ l2 ^ l3 ^ l4 ^ l5 ^ l6 ^ l7;
      } else if constexpr (MODE == 16) {
        asm volatile(X7 X7 X7 X7 "global_load_dwordx4 %5, %7, off\n" X7 X7 X7 X7 "global_load_dwordx4 %6, %7, off offset:1024\n s_waitcnt vmcnt(0)\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[8]), "=&v"(w0), "=&v"(w1) : "v"(gp) : "memory");
        a[9] ^= w0.x ^ w1.y;
      } else {
        asm volatile(X7 X7 X7 X7 X7 X7 X7 X7 "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[8]) :: "memory");
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime();
  u32 s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s ^= a[i];
  s ^= r0.x ^ r1.y ^ q[0].x ^ q[1].x ^ q[2].x ^ q[3].x ^ q[4].x ^ q[5].x ^ t0;
  if (s == 0x12345) sink[0] = s;
  if (lane == 0) atomicMax(out, c1 - c0);  // slowest wave of the whole grid
}

template <int MODE>
void run(const char *name, int ninstr, int threads) {
  unsigned long long *out; u32 *sink;
  CK(hipMalloc(&out, 8)); CK(hipMalloc(&sink, 1 << 20)); CK(hipMemset(sink, 0, 1 << 20));
  const int iters = getenv("UB_ITERS") ? atoi(getenv("UB_ITERS")) : 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, out, sink, iters);
  CK(hipDeviceSynchronize());
  CK(hipMemset(out, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, out, sink, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost));
  const double per_body = (double)c / iters;
  const int wps = threads / 256;  // waves per SIMD
  printf("%-34s waves/SIMD=%d  cycles/body=%8.1f  cycles/instr/wave=%6.2f  cycles/instr/SIMD=%6.2f  clock~%.2f GHz\n", name, wps ? wps : 1,
         per_body, per_body / ninstr, per_body / ninstr / (wps ? wps : 1), (double)c / (ms * 1e6));
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<0>("64 v_xor", 64, threads);
    run<6>("64 v_perm", 64, threads);
    run<3>("32 (v_xor + s_nop)", 64, threads);
    run<4>("32 (v_xor + s_waitcnt)", 64, threads);
    run<5>("32 (v_xor + s_mov)", 64, threads);
    run<1>("8 lookups (perm,read,4xor)+8 waits", 56, threads);
    run<2>("16 ds_read_b128", 16, threads);
    run<7>("32 ds_write_addtid_b32", 32, threads);
    run<8>("32 ds_write_b64", 32, threads);
    run<9>("64 v_mov_b32_sdwa", 64, threads);
    run<10>("64 v_and_or_b32", 64, threads);
    run<11>("12 lookups sdwa G=6 (84 instr)", 84, threads);
    run<14>("12 lookups sdwa, wait/2 (78)", 78, threads);
    run<12>("12 lookups + 12 addtid (96)", 96, threads);
    run<13>("12 lookups + 12 addtid wait/2 (90)", 90, threads);
    run<18>("64 v_bitop3_b32 (xor3)", 64, threads);
    run<21>("16 ds_read_b128, 16 lanes per row", 16, threads);
    run<19>("16 ds_read_b128, 8 lanes/row, halves", 16, threads);
    run<20>("16 ds_read_b128, 8 lanes/row, same half", 16, threads);
    run<22>("8 x (2 perm, 2 read, 4 xor3) + 8 waits", 72, threads);
    run<17>("64 v_xor (ref)", 64, threads);
    run<15>("56 v_xor + 8 global_load_dword", 64, threads);
    run<16>("56 v_xor + 2 global_load_dwordx4", 58, threads);
  }
  return 0;
}
