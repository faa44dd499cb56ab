// api_internal.h -- shared between mzd_host.cpp and m4ri_hip_api.cpp
#pragma once
[[noreturn]] void gf2_die(const char *msg);
#include "../../include/m4ri_hip.h"
mzd_t *gf2_mzd_init_uncleared(rci_t r, rci_t c);  // mzd_init without the memset (callers overwrite every word)
// GPU-backed transpose of a host matrix (upload, 64x64-block kernel, download); 0 on success.  Used by mzd_transpose
// for large matrices; the caller falls back to the host routine if it fails (this is not the multiply path).
int gf2_host_transpose_gpu(mzd_t *dst, mzd_t const *src);
int gf2_device_count(void);
// drop the device copy kept for M by gf2_mzd_cache_on_device, if any (mzd_free and every in-place writer call this)
void gf2_cache_forget(mzd_t const *M);
// size dispatch of the drop-in entry points (gf2_small_host.cpp): M4RI_HIP_HOST_SMALL_WORK word operations, 0 = never
long long gf2_small_work_limit();
bool gf2_small_product(long long m, long long l, long long n);
// pinned scratch blocks from the pool of mzd_host.cpp (null without a device)
void *gf2_pinned_alloc(size_t bytes);
void gf2_pinned_free(void *p, size_t bytes);
// mzd_transpose(DST, A) served from the packed side copy a fresh thin product carries (m4ri_hip_api.cpp, ResultSide): the
// destination (allocated when DST is NULL), or nullptr when A has no side copy
mzd_t *gf2_transpose_from_side_copy(mzd_t *DST, mzd_t const *A);
bool gf2_mzd_block_is_pinned(mzd_t const *M);
