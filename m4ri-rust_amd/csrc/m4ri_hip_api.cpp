// m4ri_hip_api.cpp -- C-ABI of libm4ri_hip.so: the multiply family of M4RI on the GPU.
//
// Entry points replace (paths relative to /root/reference):
//   mzd_mul_m4rm / mzd_addmul_m4rm   m4ri-sys/src/brilliantrussian.rs:210-224
//   mzd_mul / mzd_addmul             m4ri-sys/src/strassen.rs:8-31
//   mzd_mul_naive / mzd_addmul_naive / _mzd_mul_naive / _mzd_mul_va   m4ri-sys/src/mzd.rs:150-181
// Ownership and error behaviour follow the callers in m4ri-rust/src/friendly/binary_matrix.rs:
// C == NULL -> allocate (line 465), NULL return only on failure of the product (lines 467-469),
// dimension mismatch aborts like m4ri_die.
//
// There is no CPU fallback in this file: every product is a HIP kernel launch.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <string>
#include <tuple>
#include <unordered_map>
#include <array>
#include <vector>

#include "../../include/m4ri_hip.h"
#include "api_internal.h"
#include "gf2_kernels.h"

typedef uint64_t u64;

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------

static thread_local std::string tls_error;

static int fail(hipError_t e, const char *what) {
  tls_error = std::string(what) + ": " + hipGetErrorString(e);
  (void)hipGetLastError();
  return (int)e ? (int)e : -1;
}
static int fail_msg(const char *what) {
  tls_error = what;
  return -1;
}
#define HIP_TRY(expr)                                 \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return fail(_e, #expr);     \
  } while (0)

extern "C" const char *gf2_last_error(void) { return tls_error.c_str(); }

extern "C" int gf2_device_count(void) {
  static int n = [] {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
      (void)hipGetLastError();
      c = 0;
    }
    return c;
  }();
  return n;
}

static int require_device() {
  if (gf2_device_count() <= 0)
    return fail_msg("no usable HIP device: libm4ri_hip has no CPU fallback for the multiply path");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// launch census: how often each kernel of the library has been launched by this process
// ---------------------------------------------------------------------------------------------

namespace {
// open-addressing table keyed by the kernel's host-side handle; lock-free (a launch pays a hash and one atomic increment)
struct CensusSlot {
  std::atomic<const void *> key{nullptr};
  std::atomic<unsigned long long> count{0};
};
constexpr unsigned kCensusSlots = 1024;  // the library has ~130 kernels
CensusSlot g_census[kCensusSlots];

std::string census_text() {
  std::string out;
  for (unsigned i = 0; i < kCensusSlots; ++i) {
    const void *k = g_census[i].key.load(std::memory_order_acquire);
    if (!k) continue;
    const char *nm = hipKernelNameRefByPtr(k, nullptr);
    out += std::to_string(g_census[i].count.load(std::memory_order_relaxed));
    out += ' ';
    out += nm ? nm : "?";
    out += '\n';
  }
  return out;
}

// M4RI_HIP_KERNEL_CENSUS_FILE=<path>: the counts of this process are APPENDED to the file when the library is unloaded (test suites
// that launch kernels from child processes: tests/conftest.py sets it for the whole session)
struct CensusDump {
  ~CensusDump() {
    const char *path = std::getenv("M4RI_HIP_KERNEL_CENSUS_FILE");
    if (!path || !*path) return;
    const std::string t = census_text();
    if (t.empty()) return;
    if (FILE *f = std::fopen(path, "a")) {
      std::fwrite(t.data(), 1, t.size(), f);
      std::fclose(f);
    }
  }
} g_census_dump;
}  // namespace

void gf2k_note_launch(const void *kernel) {
  unsigned i = (unsigned)((reinterpret_cast<uintptr_t>(kernel) >> 3) * 2654435761u) % kCensusSlots;
  for (unsigned probe = 0; probe < kCensusSlots; ++probe, i = (i + 1) % kCensusSlots) {
    const void *k = g_census[i].key.load(std::memory_order_acquire);
    if (k == kernel) break;
    if (!k) {
      const void *expect = nullptr;
      if (g_census[i].key.compare_exchange_strong(expect, kernel, std::memory_order_acq_rel) || expect == kernel) break;
    }
  }
  g_census[i].count.fetch_add(1, std::memory_order_relaxed);
}

// "<count> <mangled kernel name>\n" for every kernel launched so far; returns the length of the whole text (without the
// terminator), of which at most cap - 1 bytes are written to buf
extern "C" size_t gf2_kernel_census(char *buf, size_t cap) {
  const std::string t = census_text();
  if (buf && cap) {
    const size_t n = t.size() < cap - 1 ? t.size() : cap - 1;
    std::memcpy(buf, t.data(), n);
    buf[n] = 0;
  }
  return t.size();
}

// ---------------------------------------------------------------------------------------------
// device memory: small caching allocator (hipMalloc is slow and synchronising)
// ---------------------------------------------------------------------------------------------

namespace {
struct DevPool {
  std::mutex mu;
  std::multimap<size_t, void *> free_;
  size_t cached = 0;
};
DevPool g_pools[16];

size_t round_size(size_t b) {
  const size_t g = b < ((size_t)64 << 20) ? ((size_t)1 << 20) : ((size_t)64 << 20);
  return ((b + g - 1) / g) * g;
}

// every block remembers the device it was allocated on: a free (possibly deferred, possibly issued while another device
// is current) files it under THAT device's pool
std::mutex g_owner_mu;
std::map<void *, int> g_owner;

void remember_owner(void *p, int dev) {
  std::lock_guard<std::mutex> lk(g_owner_mu);
  g_owner[p] = dev;
}
int owner_of(void *p, bool forget) {
  std::lock_guard<std::mutex> lk(g_owner_mu);
  auto it = g_owner.find(p);
  if (it == g_owner.end()) return -1;
  const int d = it->second;
  if (forget) g_owner.erase(it);
  return d;
}

// hipFree acts on the pointer's own device, but wants that device's context alive: keep the caller's device current
void raw_free(void *p) {
  (void)owner_of(p, true);
  (void)hipFree(p);
}

int dev_alloc(void **p, size_t bytes) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  bytes = round_size(bytes ? bytes : 1);
  DevPool &pool = g_pools[dev & 15];
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.free_.lower_bound(bytes);
    if (it != pool.free_.end() && it->first <= bytes + bytes / 4) {
      *p = it->second;
      pool.cached -= it->first;
      pool.free_.erase(it);
      return 0;
    }
  }
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    // drop the cache and retry once
    std::lock_guard<std::mutex> lk(pool.mu);
    for (auto &kv : pool.free_) raw_free(kv.second);
    pool.free_.clear();
    pool.cached = 0;
    (void)hipGetLastError();
    e = hipMalloc(p, bytes);
  }
  if (e != hipSuccess) return fail(e, "hipMalloc");
  remember_owner(*p, dev);
  return 0;
}

// Hands a block back to the pool of the device that owns it.  NOT stream-ordered: the caller guarantees that no queued
// work still touches the block (it synchronised its stream, or it goes through free_after / gf2_dmat_free).
void dev_free(void *p, size_t bytes) {
  if (!p) return;
  int dev = owner_of(p, false);
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return;
  bytes = round_size(bytes ? bytes : 1);
  DevPool &pool = g_pools[dev & 15];
  std::lock_guard<std::mutex> lk(pool.mu);
  static const size_t kMaxCached = (size_t)32 << 30;
  if (pool.cached + bytes > kMaxCached) {
    raw_free(p);
    return;
  }
  pool.free_.emplace(bytes, p);
  pool.cached += bytes;
}

// private per-thread streams of the host (mzd_t) entry points, one per device: concurrent calls from several host threads
// (BinMatrix is Send + Sync) never serialise on, or race through, a shared stream.  A thread that alternates between
// devices (a pinned multiply, then an elimination on the original device) gets the SAME stream back for each of them,
// so the per-stream arenas (g_ws) are reused instead of being stranded behind a replaced stream.
thread_local hipStream_t tls_streams[16] = {};

int get_private_stream(hipStream_t *out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail_msg("device ordinal out of range (0..15)");
  if (!tls_streams[dev]) HIP_TRY(hipStreamCreateWithFlags(&tls_streams[dev], hipStreamNonBlocking));
  *out = tls_streams[dev];
  return 0;
}

// device-resident API: the caller's stream; NULL is the (legacy, synchronising) default stream
int get_stream(void *user, hipStream_t *out) {
  *out = static_cast<hipStream_t>(user);
  return 0;
}

// deferred frees for asynchronous device-API calls: buffers used by work queued on a stream are
// handed back to the pool only after an event recorded behind that work has completed.
struct Deferred {
  hipEvent_t ev;
  void *p;
  size_t bytes;
};
std::mutex g_deferred_mu;
std::vector<Deferred> g_deferred;

void reap_deferred(bool wait) {
  std::lock_guard<std::mutex> lk(g_deferred_mu);
  size_t k = 0;
  for (size_t i = 0; i < g_deferred.size(); ++i) {
    Deferred &d = g_deferred[i];
    hipError_t q = wait ? hipEventSynchronize(d.ev) : hipEventQuery(d.ev);
    if (q == hipSuccess) {
      (void)hipEventDestroy(d.ev);
      dev_free(d.p, d.bytes);
    } else {
      (void)hipGetLastError();
      g_deferred[k++] = d;
    }
  }
  g_deferred.resize(k);
}

int free_after(hipStream_t s, void *p, size_t bytes) {
  hipEvent_t ev;
  HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  if (hipError_t e = hipEventRecord(ev, s); e != hipSuccess) {  // e.g. a destroyed stream or one of another device
    (void)hipEventDestroy(ev);
    return fail(e, "hipEventRecord");
  }
  std::lock_guard<std::mutex> lk(g_deferred_mu);
  g_deferred.push_back({ev, p, bytes});
  return 0;
}

// Per-stream scratch arena: work queued on one stream is serialised, so consecutive products on the same
// stream can share one workspace without waiting for each other.  It only ever grows.
struct StreamWs {
  void *p = nullptr;
  size_t bytes = 0;
};
std::mutex g_ws_mu;
std::map<std::tuple<int, hipStream_t, int>, StreamWs> g_ws;

// slot 0: Strassen operand arena / transposed operand of the naive entry; slot 1: split-K partial products; slot 2: packed A
int stream_workspace(hipStream_t s, size_t bytes, void **out, int slot = 0) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_ws_mu);
  StreamWs &w = g_ws[std::make_tuple(dev, s, slot)];
  if (w.bytes < bytes) {
    if (w.p) {  // still referenced by queued work: hand it back once the stream has drained past this point
      if (free_after(s, w.p, w.bytes) != 0) {
        (void)hipStreamSynchronize(s);
        dev_free(w.p, w.bytes);
      }
      w.p = nullptr;
      w.bytes = 0;
    }
    reap_deferred(false);
    void *p = nullptr;
    if (int rc = dev_alloc(&p, bytes)) return rc;
    w.p = p;
    w.bytes = bytes;
  }
  *out = w.p;
  return 0;
}

}  // namespace

// Give cached device memory back to the driver: waits for the device, then frees the per-stream scratch arenas (a
// 131072^3 product leaves a 141 GiB Strassen arena behind), the deferred frees and the block cache of the current
// device.  Safe at any quiet point; the next product allocates again.
extern "C" int gf2_trim(void) {
  if (int rc = require_device()) return rc;
  HIP_TRY(hipDeviceSynchronize());
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (auto it = g_ws.begin(); it != g_ws.end();) {
      if (std::get<0>(it->first) == dev) {
        dev_free(it->second.p, it->second.bytes);
        it = g_ws.erase(it);
      } else {
        ++it;
      }
    }
  }
  reap_deferred(true);
  DevPool &pool = g_pools[dev & 15];
  std::lock_guard<std::mutex> lk(pool.mu);
  for (auto &kv : pool.free_) raw_free(kv.second);
  pool.free_.clear();
  pool.cached = 0;
  return 0;
}

namespace {
// Side stream + events for one main stream: the leaf products of a Strassen product run there, chunk by chunk, while
// the main stream streams the operands of the next chunk / folds the previous chunk's products (HBM-bound passes under
// an LDS-bound kernel).  Cached per (device, stream); never destroyed (a handful per process).
struct SideStream {
  hipStream_t s2 = nullptr;
  hipStream_t s3 = nullptr;  // second copy stream of the host pipeline (downloads; s2 carries the uploads)
  std::vector<hipEvent_t> ev;
};
std::map<std::pair<int, hipStream_t>, SideStream> g_side;

int side_stream(hipStream_t s, int nevents, SideStream **out, bool want_s3 = false) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_ws_mu);
  SideStream &sd = g_side[{dev, s}];
  if (!sd.s2) HIP_TRY(hipStreamCreateWithFlags(&sd.s2, hipStreamNonBlocking));
  if (want_s3 && !sd.s3) HIP_TRY(hipStreamCreateWithFlags(&sd.s3, hipStreamNonBlocking));
  while ((int)sd.ev.size() < nevents) {
    hipEvent_t e;
    HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    sd.ev.push_back(e);
  }
  *out = &sd;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// kernel timing (bench.py roofline): events around the dominant multiply kernel
// ---------------------------------------------------------------------------------------------

std::mutex g_prof_mu;
bool g_prof_on = false;
struct ProfPair {
  hipEvent_t a, b;
};
std::vector<ProfPair> g_prof;

}  // namespace

extern "C" void gf2_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
}

extern "C" int gf2_prof_read(int *launches, double *ms, int reset) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double total = 0;
  for (auto &pp : g_prof) {
    HIP_TRY(hipEventSynchronize(pp.b));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, pp.a, pp.b));
    total += t;
  }
  if (launches) *launches = (int)g_prof.size();
  if (ms) *ms = total;
  if (reset) {
    for (auto &pp : g_prof) {
      (void)hipEventDestroy(pp.a);
      (void)hipEventDestroy(pp.b);
    }
    g_prof.clear();
  }
  return 0;
}

// bench.py's roofline: HIP events on the launch stream around the tile-kernel launches of one (batched) product
struct ProfScope {
  ProfPair pp{};
  bool on = false;
  hipStream_t s;
  explicit ProfScope(hipStream_t s_) : s(s_) {
    {
      std::lock_guard<std::mutex> lk(g_prof_mu);
      on = g_prof_on;
    }
    if (on && (hipEventCreate(&pp.a) != hipSuccess || hipEventCreate(&pp.b) != hipSuccess || hipEventRecord(pp.a, s) != hipSuccess)) {
      (void)hipGetLastError();
      on = false;
    }
  }
  ~ProfScope() {
    if (!on) return;
    if (hipEventRecord(pp.b, s) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(pp);
  }
};

static int launch_m4rm(gf2k_mul_args a, int cfg, hipStream_t s) {
  HIP_TRY(gf2k_m4rm(a, cfg, s));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// device products
// ---------------------------------------------------------------------------------------------

static inline int words_of(int bits) { return (bits + 63) >> 6; }

static int env_int(const char *name, int dflt) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) : dflt;
}
// Fitted model constants and A/B switches: read from the environment in development builds only (tools/libm4ri_hip_dev.so,
// built with -DGF2K_DEV_VARIANTS; the A/B scripts under tools/ load it through AB_LIB).  The shipped library uses the default:
// INTEGRATION.md section 6 lists which variables it still reads.
#ifdef GF2K_DEV_VARIANTS
#define dev_env_int(name, dflt) env_int(name, dflt)
#else
#define dev_env_int(name, dflt) (dflt)
#endif

// ---- launch geometry of the tile kernel (shared by the launcher and the level chooser) ----
// tile geometry and measured cost of the OLDER kernel variants (gf2_kernels.hip): 8 = v6 2048 x 1024 tile, two chunks per lookup step, one
// row per lane; 7 = v3 1024 x 2048 tile; 20 = v3 256 x 2048 tile (4 waves).  (The v8 family, 9-12, is priced by v8_model below;
// 90-99 are the legacy v7 of development builds.)  M4RI_HIP_M4RM_CFG overrides (shipped variants only)
struct TileGeom {
  int rows, cols;
  double cyc_per_chunk;  // measured cycles per 8 bits of the inner dimension and tile, 2.4 GHz
};
static TileGeom tile_geom(int cfg, bool packed = false) {
  if (cfg == 9 || (cfg >= 90 && cfg < 100)) return {4096, 512, packed ? 1520.0 : 1800.0};
  if (cfg == 8 || (cfg >= 80 && cfg < 90)) return {2048, 1024, packed ? 1620.0 : 1790.0};
  if (cfg == 20 || cfg == 1) return {256, 2048, 1300.0};
  return {1024, 2048, 2350.0};
}
static long long tiles_of(const TileGeom &g, int m, int n) {
  return (long long)((m + g.rows - 1) / g.rows) * ((n + g.cols - 1) / g.cols);
}

static bool cfg_reads_packed(int cfg) { return cfg == 8 || (cfg >= 9 && cfg <= 12) || (cfg >= 90 && cfg < 100); }
static int cfg_v8_rg(int cfg) { return cfg == 9 ? 8 : cfg == 10 ? 4 : cfg == 11 ? 2 : cfg == 12 ? 1 : 0; }

// split-K factor of the older kernels (v3, v6): when a product has too few tiles to fill 256 CUs, the inner dimension is cut into
// slices of at least 128 bits; the slices' partial products are combined by a second kernel
static int m4rm_ksplit_for(int m, int l, int n, int batch, int cfg) {
  static const int forced = dev_env_int("M4RI_HIP_M4RM_KSPLIT", 0);
  if (forced > 0) return forced;
  const long long wg = tiles_of(tile_geom(cfg), m, n) * batch;
  const int nw32 = (l + 31) / 32;
  if (wg >= 192) {
    // All tiles of a launch take the same time, so 520 workgroups cost three rounds of 256 where 2.03 would do.  A single plain
    // product may cut the inner dimension into a few slices to even the rounds out (the slices' partial tiles cost one write and
    // one read of C per slice): rounds(ks) / ks tile-times + ks passes over C, minimised over ks <= 8.
    static const int balance = dev_env_int("M4RI_HIP_SPLITK_BALANCE", 1);
    static const long long cap = (long long)env_int("M4RI_HIP_SPLITK_WS_MIB", 2048) << 20;
    if (!balance || batch != 1 || wg > 2048) return 1;
    const TileGeom g = tile_geom(cfg, cfg == 8);
    const double tile_cyc = (double)nw32 * 4.0 * g.cyc_per_chunk;                       // one whole-k tile
    const double pass_cyc = 2.0 * (double)m * (double)n / 8.0 / 5.0e12 * 2.4e9;         // write + read of one slice's partial C
    int best = 1;
    double best_c = std::ceil(wg / 256.0) * (tile_cyc + 20000.0);
    for (int ks = 2; ks <= 8; ++ks) {
      if (nw32 / ks < 16) break;                                                        // slices of at least 512 bits
      if ((long long)ks * m * ((words_of(n) + 1) & ~1) * 8 > cap) break;
      const double c = std::ceil(wg * ks / 256.0) * (tile_cyc / ks + 20000.0) + ks * pass_cyc;
      if (c < 0.93 * best_c) best = ks, best_c = c;
    }
    return best;
  }
  long long ks = 256 / wg;  // one round of workgroups: 256 long slices beat 512 short ones (8192x65536x16384: 2.25 vs 2.52 ms)
  if (ks > nw32 / 4) ks = nw32 / 4;  // slices of at least 128 bits
  return ks < 1 ? 1 : (int)ks;
}

// ---- which tile kernel, and how its launch is cut (shared by the launchers and the level chooser) ----
// Candidates: the v8 family (512 RG rows x 512 columns, RG = 8 / 4 / 2 / 1: variants 9 / 10 / 11 / 12) with whole tiles, or with
// the tiles of the last (incomplete) round of 256 workgroups cut into stream-K segments (everything, when there are fewer than
// 256 tiles); v6 (8) and v3 (7, and 20 for m <= 256) with their uniform split-K.  Chosen by modelled time; the constants are
// measured (tools/kbench, profiles/r03_tile_variants.txt).  `packed`: A is handed over row-group packed (Strassen leaves, or a
// plain product that packs A itself): only 8 and 9-12 read that layout.
struct TilePlan {
  int cfg = 7;
  int ksplit = 1;           // v3 / v6: uniform slices of the inner dimension
  int n_rem = 0, nseg = 0;  // v8: tiles cut into stream-K segments, number of segments
  bool packed = false;
  double t = 0;             // modelled seconds of the launch (reduction of partial tiles included)
  size_t ws_bytes = 0;      // scratch for partial tiles
  // a batched launch may be cut in two: the first `batch - tail_batch` products as planned above (whole rounds of 256 tiles), the
  // last tail_batch products in a launch of their own with its own variant and split (a short tile height, every tile cut)
  int tail_batch = 0, tail_cfg = 0, tail_n_rem = 0, tail_nseg = 0;
  size_t tail_ws_bytes = 0;
  // the rows below the last whole tile row of the main launch may run as a launch of their own with a shorter tile (a "row
  // band": 17000 rows = four tile rows of 4096 + 616 rows in 1024-row tiles instead of a fifth tile row that is 15 % full)
  int band_rows = 0, band_cfg = 0, band_n_rem = 0, band_nseg = 0;
  size_t band_ws_bytes = 0;
  size_t scratch() const { return std::max(ws_bytes, std::max(tail_ws_bytes, band_ws_bytes)); }  // the launches run one after the other
};

// microseconds per quad (32 bits of the inner dimension) of a v8 tile: table generation (256 entry writes, barrier) + RG x 1024
// lookups; an unpacked A costs 64 scattered 8-byte loads per wave and row group.  Measured on 343 leaves of 4096^3
// (profiles/r03_tile_variants.txt): packed 0.71 / 0.96 / 1.46 / 2.55 us per quad and tile for RG = 1 / 2 / 4 / 8 INCLUDING the
// tile's prologue / epilogue / hand-over, which v8_model adds separately ((3.4 + 0.6 RG) us per 128 quads there): the loop
// itself takes 0.679 / 0.924 / 1.415 / 2.486 us (M4RI_HIP_V8_QUAD_NS overrides the four; checked against 49 leaves of
// 5632^3 in 2048-row tiles: 1617 tiles x 176 quads in 1.63 ms = 1.47 us per quad and tile all in)
// Unpacked A with LONG rows: a lane's 8-byte loads walk its own row, and from ~8192 bits on the lines the 4096 rows of a
// tile keep open (64 bytes each) no longer survive in L2 until their next word is wanted -- every load fetches a line from
// HBM (65536 x 65536 x 512: 1.14 ms unpacked, 0.78 ms with the packing pass; 65536 x 8192 x 512: 0.190 against 0.101 ms;
// up to 7000 bits packing loses, 0.092 against 0.156 ms): the surcharge is five times as high from 8192 bits on.
static double v8_quad_us(int RG, bool packed, int l = 0) {
  static const double unp0 = dev_env_int("M4RI_HIP_V8_UNPACKED_BASE_NS", 100) * 1e-3, unp = dev_env_int("M4RI_HIP_V8_UNPACKED_NS", 70) * 1e-3;
  static const std::array<double, 4> loop_us = [] {
    std::array<double, 4> t{0.679, 0.924, 1.415, 2.486};
#ifdef GF2K_DEV_VARIANTS  // (M4RI_HIP_V8_QUAD_NS=a,b,c,d: the loop's nanoseconds per quad at 512 / 1024 / 2048 / 4096 rows, for model fits)
    if (const char *e = getenv("M4RI_HIP_V8_QUAD_NS")) {
      int v[4];
      if (std::sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4)
        for (int i = 0; i < 4; ++i) t[i] = v[i] * 1e-3;
    }
#endif
    return t;
  }();
  const int i = RG >= 8 ? 3 : RG >= 4 ? 2 : RG >= 2 ? 1 : 0;
  static const double long_rows = dev_env_int("M4RI_HIP_V8_UNPACKED_LONG_PCT", 500) * 1e-2;
  const double stretch = l >= 8192 ? long_rows : 1.0;  // a cliff, not a slope: 65536 x l x 512 unpacked takes 0.092 ms at l = 7000 and 0.188 ms at 8192
  return loop_us[i] + (packed ? 0.0 : (unp0 + unp * RG) * stretch);
}

// One v8 launch: `batch` products, variant cfg, the last n_rem tiles cut into about `want` segments (n_rem = 0: whole tiles only).
// Fills c (split fields normalised the way the launcher will normalise them) and returns false if the split is void or its
// scratch exceeds the cap.
static bool v8_model(int m, int l, int n, int batch, bool packed, int cfg, long long n_rem, long long want, TilePlan &c) {
  static const long long cap = (long long)env_int("M4RI_HIP_SPLITK_WS_MIB", 2048) << 20;
  static const double seg_fix = dev_env_int("M4RI_HIP_V8_SEG_FIX_NS", 3500) * 1e-9, seg_rg = dev_env_int("M4RI_HIP_V8_SEG_RG_NS", 2600) * 1e-9,
                      red_bw = dev_env_int("M4RI_HIP_V8_REDUCE_GBS", 2500) * 1e9, two_part = dev_env_int("M4RI_HIP_V8_TWO_PART_PCT", 180) * 1e-2;
  const int RG = cfg_v8_rg(cfg), R = 512 * RG, nw32 = (l + 31) / 32, Q = (nw32 + 1) / 2;
  const long long T = (long long)((m + R - 1) / R) * ((n + 511) / 512) * batch;
  const double tq = v8_quad_us(RG, packed, l) * 1e-6, tile_bytes = R * 64.0;
  const double tile_t = 2.0 * Q * tq + (3.4 + 0.6 * RG) * 1e-6;  // + prologue, epilogue (LDS transpose, stores), hand-over to the next workgroup
  c = TilePlan();
  c.cfg = cfg;
  c.packed = packed;
  if (n_rem <= 0 || Q < 1) {
    c.t = std::ceil(T / 256.0) * tile_t + 2e-6;
    return true;
  }
  if (n_rem > T) n_rem = T;
  if (want < 1) want = 256;
  const long long gtot = n_rem * Q;
  long long seg = (gtot + want - 1) / want;
  if (seg > Q) seg = Q;
  if (seg < 1) seg = 1;
  const long long ns = (gtot + seg - 1) / seg;
  if (ns <= n_rem && seg == Q) return false;  // nothing is cut
  c.n_rem = (int)n_rem;
  c.nseg = (int)ns;
  c.ws_bytes = (size_t)(2.0 * ns * tile_bytes);
  if ((long long)c.ws_bytes > cap || ns > (1 << 22)) return false;
  // partial tiles: written at the end of the segment phase (every workgroup at once: ~2.6 us per row group on top of the
  // prologue; a segment that spans two tiles pays prologue and stores twice), read back and folded into C by the reduction
  // kernel (launch gap + bytes at ~2.5 TB/s)
  const bool spans = (Q % seg) != 0;
  const double slots = (double)ns + (spans ? (double)std::min(ns, n_rem) : 0.0);
  const double reduce_t = (slots + (double)n_rem) * tile_bytes / red_bw + 2.5e-6;
  const double seg_t = 2.0 * seg * tq + (seg_fix + seg_rg * RG) * (spans ? two_part : 1.0);
  c.t = std::ceil((T - n_rem) / 256.0) * tile_t + std::ceil(ns / 256.0) * seg_t + reduce_t + 2e-6;
  return true;
}

static bool older_model(int m, int l, int n, int batch, bool packed, int cfg, TilePlan &c) {
  static const long long cap = (long long)env_int("M4RI_HIP_SPLITK_WS_MIB", 2048) << 20;
  if (packed && !cfg_reads_packed(cfg)) return false;
  const int nw32 = (l + 31) / 32;
  const TileGeom g = tile_geom(cfg, packed);
  const int ks = m4rm_ksplit_for(m, l, n, batch, cfg);
  const double wg = (double)tiles_of(g, m, n) * batch * ks;
  const double chunks = std::ceil(nw32 / (double)ks) * 4.0;
  c = TilePlan();
  c.cfg = cfg;
  c.ksplit = ks;
  c.packed = packed;
  c.t = std::ceil(wg / 256.0) * (chunks * g.cyc_per_chunk + 6000.0) / 2.4e9 + 3e-6;
  // measured against the v8 plans on narrow products with long rows (2048 x 33000 x 600: modelled 44 us, 83 measured with 256
  // slices; 9000 x 33000 x 300: 82 / 127 with 51; 33000 x 9000 x 300: 77 / 111 with 15; the same shapes through v8 on packed A:
  // 39-74 us): unpacked rows of 8192 bits and more cost these kernels 1.4x (they too read a row per lane or lane group), and
  // every slice of the inner dimension another 0.2 %
  static const double older_long = dev_env_int("M4RI_HIP_OLDER_LONG_PCT", 140) * 1e-2, older_ks = dev_env_int("M4RI_HIP_OLDER_KSPLIT_PPM", 2000) * 1e-6;
  if (!packed && l >= 8192) c.t *= older_long;
  c.t *= 1.0 + older_ks * ks;
  if (ks > 1) {
    const double part = (double)m * (double)(((words_of(n) + 1) & ~1) * 8) * batch;
    c.ws_bytes = (size_t)(part * ks);
    c.t += (part * ks * 2.0 + part) / 4.0e12 + 2.5e-6;
    if ((long long)c.ws_bytes > cap) return false;
  }
  return true;
}

enum { PLAN_TAIL = 2, PLAN_BAND = 4, PLAN_V8_ONLY = 8 };  // what a plan may contain (PlanKey::flags; bit 0 = packed A)
static TilePlan plan_tiles_uncached(int m, int l, int n, int batch, bool packed, int mode);
// Planning is pure arithmetic on the shape (and the environment, read once), but a level choice evaluates dozens of candidate
// launches: 15-55 us of host time per product, which a 0.2 ms product notices.  Every thread keeps what it has planned.
struct PlanKey {
  int m, l, n, batch, flags;
  bool operator==(const PlanKey &o) const { return m == o.m && l == o.l && n == o.n && batch == o.batch && flags == o.flags; }
};
struct PlanKeyHash {
  size_t operator()(const PlanKey &k) const {
    size_t h = (size_t)k.m * 0x9E3779B97F4A7C15ull;
    h ^= ((size_t)k.l + 0x7F4A7C15u) * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= ((size_t)k.n + 0x165667B1u) * 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h ^= ((size_t)k.batch * 31 + (size_t)k.flags) * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    return h;
  }
};
static TilePlan plan_tiles(int m, int l, int n, int batch, bool packed, int mode = PLAN_TAIL | PLAN_BAND) {
  thread_local std::unordered_map<PlanKey, TilePlan, PlanKeyHash> memo;
  const PlanKey key{m, l, n, batch, (packed ? 1 : 0) | mode};
  auto it = memo.find(key);
  if (it != memo.end()) return it->second;
  if (memo.size() > 8192) memo.clear();
  const TilePlan tp = plan_tiles_uncached(m, l, n, batch, packed, mode);
  memo.emplace(key, tp);
  return tp;
}

static TilePlan plan_tiles_uncached(int m, int l, int n, int batch, bool packed, int mode) {
  const bool allow_tail = (mode & PLAN_TAIL) != 0;
  // A/B override, restricted to variants that compute the product (the timing-only ablations exist only in development
  // builds of the kernels and would be hipErrorInvalidValue here anyway)
  static const int forced = [] {
    const int f = env_int("M4RI_HIP_M4RM_CFG", -1);
#ifdef GF2K_DEV_VARIANTS
    return f;
#else
    if (f < 0 || f == 7 || f == 8 || (f >= 9 && f <= 12) || f == 20 || f == 81 || f == 82) return f;
    std::fprintf(stderr, "m4ri_hip: M4RI_HIP_M4RM_CFG=%d ignored (accepted: 7, 8, 9, 10, 11, 12, 20, 81, 82)\n", f);
    return -1;
#endif
  }();
  static const int forced_ks = dev_env_int("M4RI_HIP_M4RM_KSPLIT", 0);
  static const int streamk = dev_env_int("M4RI_HIP_STREAMK", 1);      // 0: whole tiles only
  static const int tails = dev_env_int("M4RI_HIP_TAIL_LAUNCH", 1);    // 0: never cut a batched launch in two
  TilePlan best, c;
  bool have = false;
  auto consider = [&](const TilePlan &x) {
    if (!have || x.t < best.t) best = x, have = true;
  };
  auto v8 = [&](int cfg) {
    const int R = 512 * cfg_v8_rg(cfg);
    const long long tp = (long long)((m + R - 1) / R) * ((n + 511) / 512), T = tp * batch;
    if (forced_ks > 0) {
      if (v8_model(m, l, n, batch, packed, cfg, T, T * forced_ks, c) || v8_model(m, l, n, batch, packed, cfg, 0, 0, c)) consider(c);
      return;
    }
    if (v8_model(m, l, n, batch, packed, cfg, 0, 0, c)) consider(c);
    if (T < 1) return;
    const long long rem = T < 256 ? T : T % 256;
    if (rem == 0) return;
    if (streamk && v8_model(m, l, n, batch, packed, cfg, rem, 256, c)) consider(c);
    // the last products in a launch of their own: the first launch keeps whole rounds of 256 tiles
    if (tails && allow_tail && forced < 0 && batch > 1 && T > 256) {
      const long long b1 = (T - rem) / tp;  // products whose tiles all lie in the whole rounds
      if (b1 >= 1 && b1 < batch) {
        TilePlan head;
        if (v8_model(m, l, n, (int)b1, packed, cfg, 0, 0, head)) {
          const TilePlan tail = plan_tiles(m, l, n, batch - (int)b1, packed, 0);
          head.t += tail.t + 1.5e-6;
          head.tail_batch = batch - (int)b1;
          head.tail_cfg = tail.cfg;
          head.tail_n_rem = tail.n_rem;
          head.tail_nseg = tail.nseg;
          head.tail_ws_bytes = tail.ws_bytes;
          if (cfg_v8_rg(tail.cfg)) consider(head);
        }
      }
    }
  };
  if (forced >= 0) {
    if (cfg_v8_rg(forced)) v8(forced);
    else if (older_model(m, l, n, batch, packed, forced, c)) consider(c);
    if (!have) {  // a forced variant that cannot read this operand layout: the caller asks again unpacked
      best.cfg = forced;
      best.packed = false;
      best.ksplit = m4rm_ksplit_for(m, l, n, batch, forced);
      TilePlan est;  // (its time: the estimate of whole 512-row tiles, so that the level / shape planners never see a free launch)
      if (v8_model(m, l, n, batch, false, 12, 0, 0, est)) best.t = est.t;
    }
    return best;
  }
  const bool v8_only = (mode & PLAN_V8_ONLY) != 0;
  if (m <= 256 && !packed && !v8_only) {
    if (older_model(m, l, n, batch, packed, 20, c)) consider(c);
    if (have) return best;  // (rejected only when its split-K scratch exceeds M4RI_HIP_SPLITK_WS_MIB: the general candidates follow)
  }
  if (!v8_only) {
    if (!packed && older_model(m, l, n, batch, packed, 7, c)) consider(c);
    if (older_model(m, l, n, batch, packed, 8, c)) consider(c);
  }
  for (int cfg = 9; cfg <= 12; ++cfg) v8(cfg);
  // a row band: whole tile rows of 4096 (2048) in the main launch, the rows below them in a launch of their own with the
  // tile height that suits them (the same B, disjoint rows of A and C; the band's partial tiles reuse the scratch)
  static const int bands = dev_env_int("M4RI_HIP_ROW_BANDS", 1);
  static const double band_gain = dev_env_int("M4RI_HIP_ROW_BAND_MIN_GAIN_PCT", 3) * 1e-2;
  if (bands && (mode & PLAN_BAND) && have) {
    const TilePlan uniform = best;
    for (int R = 4096; R >= 2048; R >>= 1) {
      const int m1 = m / R * R, m2 = m - m1;
      if (m1 < R || m2 <= 0) continue;
      TilePlan head = plan_tiles(m1, l, n, batch, packed, (mode & PLAN_TAIL) | PLAN_V8_ONLY);
      const TilePlan band = plan_tiles(m2, l, n, batch, packed, PLAN_V8_ONLY);
      if (!cfg_v8_rg(head.cfg) || !cfg_v8_rg(band.cfg)) continue;
      // + the launch boundary between the two; batched leaves: priced 3 % up (measured at 40000^3 and 52000^3, where padded plans
      // with banded leaves were modelled 0.5 % ahead of the peeled plans and ran 3 % behind them; plain products match the model:
      // 17000^3 -9 %, 12700 x 1024 x 40000 -26 %, 70000^3's bottom strip 40.4 -> 39.1 ms)
      head.t = (head.t + band.t) * (batch > 1 ? 1.03 : 1.0) + 3e-6;
      if (head.t >= uniform.t * (1.0 - band_gain)) continue;
      head.band_rows = m2;
      head.band_cfg = band.cfg;
      head.band_n_rem = band.n_rem;
      head.band_nseg = band.nseg;
      head.band_ws_bytes = band.ws_bytes;
      consider(head);
    }
  }
  // nothing accepted (every candidate wanted more scratch than the cap allows): whole, unsplit 512-row tiles need none, so the
  // plan always carries a real estimate (a default-constructed plan would price the launch as free for pick_levels / plan_shape)
  if (!have && v8_model(m, l, n, batch, packed, 12, 0, 0, c)) consider(c);
  return best;
}

// modelled duration of one (batched) tile-kernel launch
static double m4rm_time_model(int m, int l, int n, int batch, bool packed) { return plan_tiles(m, l, n, batch, packed).t; }
// ... of the leaf launch of a Strassen product (the last split pass may write the A leaves packed: mul_strassen takes the better plan)
static double leaf_time_model(int m, int l, int n, int batch, bool may_pack) {
  const double t = m4rm_time_model(m, l, n, batch, false);
  return may_pack ? std::min(t, m4rm_time_model(m, l, n, batch, true)) : t;
}
// fixed cost of the packing pass of a plain product: its launch and the dependent-kernel boundary behind it (4.8 us measured for
// the 2 MiB of a 4096^2 operand, profiles/r03_config2_kernel_stats.csv, of which 0.8 us are the bytes)
static const double kPackLaunch = 5.0e-6;
// ... of a plain product, which may pack A itself first (mul_m4rm_plain makes the same comparison)
static double plain_time_model(int m, int l, int n) {
  static const double bw = (double)dev_env_int("M4RI_HIP_STREAM_GBS", 5000) * 1e9;
  static const int plain_pack = env_int("M4RI_HIP_PLAIN_APACK", 1);
  double t = m4rm_time_model(m, l, n, 1, false);
  const long long wp = (words_of(l) + 1) & ~1ll, prow = ((long long)m + 63) & ~63ll;
  if (plain_pack && m >= 512 && prow * wp * 8 < (1ll << 32)) {
    const TilePlan pk = plan_tiles(m, l, n, 1, true);
    if (cfg_reads_packed(pk.cfg)) t = std::min(t, pk.t + 2.0 * (double)prow * (double)wp * 8.0 / bw + kPackLaunch);
  }
  return t;
}

// one planned (batched) product: the launch, and the launch of the tail products if the plan cuts the batch in two
static int apply_tile_plan(gf2k_mul_args &a, const TilePlan &tp, hipStream_t s);
static int launch_planned(gf2k_mul_args a, const TilePlan &tp, hipStream_t s) {
  ProfScope prof(s);
  const int total = a.batch;
  const int band = tp.band_rows > 0 && tp.band_rows < a.m ? tp.band_rows : 0;
  const gf2k_mul_args whole = a;
  a.m -= band;
  const bool cut = tp.tail_batch > 0 && tp.tail_batch < total;
  if (cut) a.batch = total - tp.tail_batch;
  if (int rc = apply_tile_plan(a, tp, s)) return rc;
  if (int rc = launch_m4rm(a, tp.cfg, s)) return rc;
  if (cut) {
    gf2k_mul_args b = a;
    const long long b1 = a.batch;
    b.A += b1 * a.sA;
    b.B += b1 * a.sB;
    b.C += b1 * a.sC;
    b.batch = tp.tail_batch;
    b.ksplit = 1;
    b.n_rem = b.P ? tp.tail_n_rem : 0;
    b.nseg = b.P ? tp.tail_nseg : 0;
    if (int rc = launch_m4rm(b, tp.tail_cfg, s)) return rc;
  }
  if (!band) return 0;
  // the row band: rows [m - band, m) of every product (m - band is a multiple of 2048, so the offset is the same
  // expression for row-major and row-group-packed A)
  gf2k_mul_args r = whole;
  r.A += (long long)a.m * whole.lda;
  r.C += (long long)a.m * whole.ldc;
  r.m = band;
  r.ksplit = 1;
  r.P = a.P;
  r.p_words = a.p_words;
  r.n_rem = r.P ? tp.band_n_rem : 0;
  r.nseg = r.P ? tp.band_nseg : 0;
  return launch_m4rm(r, tp.band_cfg, s);
}

// the plan of a plain product: A unpacked, or packed by a pass of its own when the model says that pays
static TilePlan plain_plan(int m, int l, int n, bool *pack) {
  static const double bw = (double)dev_env_int("M4RI_HIP_STREAM_GBS", 5000) * 1e9;
  static const int plain_pack = env_int("M4RI_HIP_PLAIN_APACK", 1);
  TilePlan tp = plan_tiles(m, l, n, 1, false);
  *pack = false;
  const long long wp = (words_of(l) + 1) & ~1ll, prow = ((long long)m + 63) & ~63ll;
  if (plain_pack && m >= 512 && prow * wp * 8 < (1ll << 32)) {
    const TilePlan pk = plan_tiles(m, l, n, 1, true);
    if (cfg_reads_packed(pk.cfg) && pk.t + 2.0 * (double)prow * (double)wp * 8.0 / bw + kPackLaunch < tp.t) tp = pk, *pack = true;
  }
  return tp;
}

// fills in the launch fields of `a` from a plan (scratch for partial tiles from the stream's workspace slot 1); falls back to an
// unsplit launch when the scratch cannot be had
static int apply_tile_plan(gf2k_mul_args &a, const TilePlan &tp, hipStream_t s) {
  a.ksplit = 1;
  a.n_rem = a.nseg = 0;
  a.P = nullptr;
  a.p_words = 0;
  const size_t want = tp.scratch();  // (the launches of a plan run one after the other)
  if (want == 0) return 0;
  void *ws = nullptr;
  if (stream_workspace(s, want, &ws, 1) != 0) return 0;
  a.P = static_cast<u64 *>(ws);
  a.p_words = (long long)(want / sizeof(u64));
  if (tp.ws_bytes == 0) return 0;
  if (cfg_v8_rg(tp.cfg)) {
    a.n_rem = tp.n_rem;
    a.nseg = tp.nseg;
  } else {
    a.ksplit = tp.ksplit;
    a.ldp = (words_of(a.n) + 1) & ~1ll;
    a.sP = (long long)a.m * a.ldp;
  }
  return 0;
}

// The level plan: a product with L Strassen levels runs L levels of operand splits, ONE batched leaf launch and L levels of
// product merges.  The passes fuse levels so that intermediate operands are never written:
//   step {k, virt}: k (1..3) levels fused in one kernel, operands materialised at the step's end; virt: one more level on
//   top that is never materialised -- the kernel reads the one or two quadrants of the grandparent whose XOR is its source
//   (on the product side the 7 parents of that level ARE materialised: merge3 into them, then a single-level merge).
//   L: 1 {1}  2 {2}  3 {3}  4 {3 + virtual}  5 {2, 3}  6 {3, 3}          (M4RI_HIP_STRASSEN_FUSE3=0: pairs only, round 1's plan)
struct PlanStep {
  int k;
  bool virt;
  int levels() const { return k + (virt ? 1 : 0); }
};
static std::vector<PlanStep> strassen_plan(int L) {
  const int fuse3 = env_int("M4RI_HIP_STRASSEN_FUSE3", 1);  // read per call: the tests switch plans
  std::vector<PlanStep> p;
  if (L <= 0) return p;
  if (!fuse3) {
    if (L & 1) p.push_back({1, false});
    for (int lv = L & 1; lv < L; lv += 2) p.push_back({2, false});
    return p;
  }
  switch (L) {
    case 1: p = {{1, false}}; break;
    case 2: p = {{2, false}}; break;
    case 3: p = {{3, false}}; break;
    case 4: p = {{3, true}}; break;
    case 5: p = {{2, false}, {3, false}}; break;
    default: p = {{3, false}, {3, false}}; break;  // 6
  }
  return p;
}

// Does the last split pass of an L-level product write the A leaves row-group packed?  (Every split kernel has a packed form --
// the single-level one since round 4 --; the leaf rows must be a multiple of 64; M4RI_HIP_APACK=0 switches the layout off.)
static bool strassen_packs_a(int m, int L) {
  static const int apack_on = env_int("M4RI_HIP_APACK", 1);
  if (!apack_on || L < 1) return false;
  return ((m >> L) & 63) == 0;
}

static size_t pow7(int i) {
  size_t p = 1;
  while (i-- > 0) p *= 7;
  return p;
}

// bytes the split / merge passes of an L-level product move (every kernel reads its sources once and writes its
// destinations once; a virtual level reads 12 quadrants instead of 4)
static double strassen_pass_bytes(double m, double l, double n, int L) {
  double bytes = 0;
  int prev = 0;
  for (const PlanStep &st : strassen_plan(L)) {
    const int i = prev + st.levels();
    const double p7 = (double)pow7(prev);
    const double a_i = (m * l) / std::pow(4.0, i) / 8.0, b_i = (l * n) / std::pow(4.0, i) / 8.0, c_i = (m * n) / std::pow(4.0, i) / 8.0;
    const double rd = st.virt ? 12.0 * std::pow(4.0, st.k) : std::pow(4.0, st.k), wr = std::pow(7.0, st.levels());
    bytes += p7 * (rd + wr) * (a_i + b_i);  // operand sides
    if (st.virt)  // products: merge3 into the 7 level-(prev+1) parents, then a single-level merge
      bytes += p7 * c_i * (wr + 7.0 * std::pow(4.0, st.k) + 7.0 * std::pow(4.0, st.k) + std::pow(4.0, st.k + 1));
    else
      bytes += p7 * c_i * (wr + std::pow(4.0, st.k));
    prev = i;
  }
  return bytes;
}

// number of Strassen levels: `req` > 0 explicit; 0 automatic = the level count with the smallest modelled time
//   t(L) = modelled time of the batched leaf launch (rounds of 256 workgroups x chunks x measured cycles per
//          chunk, split-K included)  +  bytes moved by the split / merge passes / bw
// `leaf_min` bounds the leaf dimensions from below (mzd_mul's cutoff argument, strassen.rs:8-18).
// modelled seconds of the product with exactly L levels on the shape as given (-1: L levels do not divide it)
// fixed cost of one split / merge pass besides its bytes: the dependent-kernel boundary and the ramp of a launch (three passes
// per plan step).  5 us: with 3 us the model preferred two levels at 8192^3 (133 against 138 us), which measures 0.123 against 0.115 ms
static const double kPassLaunch = 5.0e-6;
static double level_time_model(int m, int l, int n, int L) {
  static const double bw = (double)dev_env_int("M4RI_HIP_STREAM_GBS", 5000) * 1e9;        // streaming B/s
  if (L <= 0) return plain_time_model(m, l, n);
  const int d = 1 << L;
  if (L > 6 || m % d || l % (128 * d) || n % (128 * d)) return -1.0;  // leaf rows integral, leaf widths an even word count
  return leaf_time_model(m >> L, l >> L, n >> L, (int)pow7(L), strassen_packs_a(m, L)) + strassen_pass_bytes(m, l, n, L) / bw +
         3 * kPassLaunch * (double)strassen_plan(L).size();
}
extern "C" double gf2_model_time(int m, int l, int n, int levels) { return level_time_model(m, l, n, levels); }

static int pick_levels_uncached(int m, int l, int n, int req, int leaf_min, double *t_out);
static int pick_levels(int m, int l, int n, int req, int leaf_min, double *t_out = nullptr) {
  struct Res {
    int L;
    double t;
  };
  thread_local std::unordered_map<PlanKey, Res, PlanKeyHash> memo;
  const PlanKey key{m, l, n, req, leaf_min * 2 + (env_int("M4RI_HIP_STRASSEN_FUSE3", 1) ? 1 : 0)};  // (the tests switch level plans)
  auto it = memo.find(key);
  if (it == memo.end()) {
    if (memo.size() > 8192) memo.clear();
    Res r{0, 0.0};
    r.L = pick_levels_uncached(m, l, n, req, leaf_min, &r.t);
    it = memo.emplace(key, r).first;
  }
  if (t_out) *t_out = it->second.t;
  return it->second.L;
}
static int pick_levels_uncached(int m, int l, int n, int req, int leaf_min, double *t_out) {
  static const int max_auto = env_int("M4RI_HIP_STRASSEN_MAX_LEVELS", 5);
  const int cap = req > 0 ? (req > 6 ? 6 : req) : max_auto;
  int best = 0;
  double best_t = 0;
  for (int L = 0; L <= cap; ++L) {
    if (L > 0) {
      const int d = 1 << L;
      if (m % d || l % (128 * d) || n % (128 * d)) break;  // leaf rows integral, leaf widths an even word count
      if (req <= 0 && ((m >> L) < (leaf_min < 1024 ? leaf_min : 1024) || (l >> L) < leaf_min || (n >> L) < leaf_min)) break;
    }
    if (req > 0) {
      best = L;
      continue;
    }
    double t = level_time_model(m, l, n, L);
    // tools/levels_sweep.py (profiles/r03_levels_sweep.txt): the model is 3-10 % pessimistic for 0 and 2 levels and within 3 % for
    // 3 and more, so a further level must promise 1.5 % (up to two levels) / 3 % (beyond) over the best count below it
    if (L == 0 || t < best_t * (L >= 3 ? 0.97 : 0.985)) {
      best = L;
      best_t = t;
    }
  }
  if (t_out) *t_out = best_t;
  return best;
}

// 64-bit words of the operand arena: operands and products at the end of every step, plus the parents of a virtual level
static size_t strassen_ws_words(int m, int l, int n, int L) {
  size_t total = 0;
  int prev = 0;
  for (const PlanStep &st : strassen_plan(L)) {
    const int i = prev + st.levels();
    const size_t mi = (size_t)m >> i, li = (size_t)l >> i, ni = (size_t)n >> i;
    total += pow7(i) * (mi * (li / 64) + li * (ni / 64) + mi * (ni / 64));
    if (st.virt) total += pow7(prev + 1) * ((size_t)m >> (prev + 1)) * (((size_t)n >> (prev + 1)) / 64);
    prev = i;
  }
  return total;
}

// Few columns against a long inner dimension (`&A * &v` with a large square A; a block of up to 64 vectors): B is transposed
// (n rows of l bits, a few KiB) and a wave per row streams A once (gf2_widevec_kernel), 32 vectors per pass.  The tile kernel
// would compute 512 columns to deliver n (65536^2 times 64 vectors: 1.18 ms against 0.3), the lane-per-row kernels read an
// 8-KiB row 8 bytes at a time per lane (65536^2 times one vector: 0.30 ms against 0.1).
static bool widevec_shape(int m, int l, int n) {
  static const int on = dev_env_int("M4RI_HIP_WIDEVEC", 1);
  if (!on || n > 64 || m < 1) return false;
  // Where it wins, from an A/B grid against the older paths on one box (tools/ab_widevec.sh, profiles/r03_widevec_ab.txt;
  // time of the wave-per-row kernel / time of what ran before, at m = 65536 and m = 1000):
  //   n <= 16: l = 2048 0.28-0.48 / 0.8-1.1, 4096 0.37-0.63 / 0.6-0.9, 20000 0.08-0.16 / 0.4-0.6, 65536 0.08-0.16 / 0.4-0.55
  //   n  = 32: l = 4096 1.27 / 0.74, 20000 0.31 / 0.59, 65536 0.30 / 0.97        n = 64: l = 20000 0.63 / 1.04, 65536 0.61 / 1.84
  // (three instructions per vector and word bound it from 9 vectors on; a row per wave needs rows to fill the chip)
  if (m <= 8) return l >= 8192 && (n <= 32 || m >= 4);  // against the v*A kernel: 8 x 65536 x 1 74 -> 9 us, x 64 120 -> 62 us
  static const int minl16 = dev_env_int("M4RI_HIP_WIDEVEC_MINL16", 2048), minl32 = dev_env_int("M4RI_HIP_WIDEVEC_MINL32", 8192);
  // rows of 768 ... 2047 bits: up to 8 vectors and not too many rows (20000 x 1000 x 8: 19 us against 30 for the tile kernel and 44
  // for the generation table kernel; at 2^20 rows the table kernel wins: 86 us against 220)
  if (n <= 8 && l >= 768 && m <= 131072) return true;
  if (n <= 16) return l >= minl16;
  if (n <= 32) return l >= minl32;
  return l >= 16384 && m >= 8192;
}
// up to 64 vectors against a long inner dimension through 4-bit tables rebuilt per 512-bit slab (gf2_tallskinny7_kernel)
static bool ts_long_shape(int m, int l, int n) {
  static const int mode = dev_env_int("M4RI_HIP_TS7", 1);  // 0 off, 1 by rule, 2 whenever it can (A/B)
  if (!mode || n > 64 || l <= 256 || m < 1) return false;
  if (mode == 2) return true;
  // Where it wins (tools/ab_ts7.sh, profiles/r03_ts7_ab.txt: its time does not depend on n -- 65536^2: 0.21 ms, 2^20 x 4096: 0.18 ms --
  // while the wave-per-row kernel costs three instructions per vector and word and the tile kernel computes 512 columns):
  //   17-64 vectors: from 2048-bit rows on, or from 16384 rows on (65536^2 x 64: 0.72 -> 0.21 ms; 4096 x 65536 x 64: 92 -> 27 us)
  //    9-16 vectors: many rows and rows of at most 8192 bits (65536 x 4096 x 16: 32 -> 19 us); longer rows stream faster per wave
  //    1-8  vectors: many short rows (2^20 x 1000 x 8: 88 -> 49 us, where the generation table kernel used to run)
  if (n > 16) return l >= 2048 || m >= 16384;
  if (n > 8) return m >= 65536 && l <= 8192;
  return m >= 65536 && l <= 2048;
}
// 9-128 rows against a B much taller than wide: computed transposed (see mul_m4rm_plain)
static bool few_rows_t_shape(int m, int l, int n) {
  static const int few = dev_env_int("M4RI_HIP_FEW_ROWS_T", 1);
  const bool tall = (n >= 1024 && l >= 8 * (long long)n && (long long)l * n >= (1ll << 26) && m <= 64) ||
                    (n > 64 && n <= 1024 && l >= 16384 && l >= 8 * (long long)n && (long long)l * n >= (1ll << 21));
  return few && m > 8 && m <= 128 && tall;
}
static size_t few_rows_t_bytes(int m, int l, int n, int accumulate) {
  const long long ldl = (words_of(l) + 1) & ~1ll, ldn = (words_of(n) + 1) & ~1ll, ldct = ((m + 63) / 64 + 1) & ~1ll;
  return ((size_t)n * ldl + (size_t)l * 2 + (size_t)n * ldct + (accumulate ? (size_t)m * ldn : 0)) * sizeof(u64);
}
static int mul_widevec(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, hipStream_t s) {
  const int m = A->nrows, l = A->ncols, n = B->ncols;
  const long long ldbt = (words_of(l) + 1) & ~1ll;
  void *bt = nullptr;
  if (int rc = stream_workspace(s, (size_t)n * ldbt * sizeof(u64), &bt)) return rc;
  HIP_TRY(gf2k_transpose(static_cast<u64 *>(bt), ldbt, B->data, B->ld, l, n, s));
  const u64 *Bt = static_cast<const u64 *>(bt);
  const int n0 = n < 32 ? n : 32;
  HIP_TRY(gf2k_widevec(A->data, A->ld, Bt, ldbt, C->data, C->ld, m, l, n0, accumulate, 0, s));
  if (n > 32) HIP_TRY(gf2k_widevec(A->data, A->ld, Bt + 32 * ldbt, ldbt, C->data, C->ld, m, l, n - 32, 1, 32, s));
  return 0;
}

// Which kernel family a plain (no Strassen) product takes -- ONE decision, used by mul_m4rm_plain and by
// gf2_mul_workspace_bytes (the query had drifted from the dispatch order).
enum PlainPath {
  kPathNothing,      // an empty operand
  kPathZeroInner,    // l == 0: C is zero
  kPathSlabTables,   // gf2k_tallskinny_long: up to 128 columns against a long inner dimension, no scratch
  kPathSlabPasses,   // the same in passes of 128 columns
  kPathWideVec,      // a wave per row: scratch = the transposed vectors
  kPathTallSkinny,   // tables over all of B in LDS (l <= 1024): no scratch
  kPathFewRows,      // m <= 8: v*A kernel, no scratch
  kPathFewRowsT,     // 9-128 rows against a tall B, computed transposed: scratch = few_rows_t_bytes
  kPathTiles         // the planned tile kernels (maybe a packed copy of A and stream-K / split-K partial tiles)
};
static PlainPath plain_path(int m, int l, int n) {
  if (m == 0 || n == 0) return kPathNothing;
  if (l == 0) return kPathZeroInner;
  if (ts_long_shape(m, l, n)) return kPathSlabTables;
  {
    static const int mp = dev_env_int("M4RI_HIP_TS7_MULTIPASS", 1);
    static const int maxn = dev_env_int("M4RI_HIP_TS7_MAXN", 256);
    // (129-192 columns would pay a whole second pass for at most 64 of them: 20000 x 40000 x 160 148 -> 165 us)
    static const int minl128 = dev_env_int("M4RI_HIP_TS7_MINL128", 1000);  // (65536 x 1000 x 128: 45 -> 17 us, 262144 x 4096 x 128: 327 -> 82 us)
    if (mp && n > 64 && n <= maxn && (n <= 128 ? m >= 256 && l >= minl128 : n > 192 && m >= 4096 && l >= 32768) && ts_long_shape(m, l, 64))
      return kPathSlabPasses;
  }
  if (widevec_shape(m, l, n)) return kPathWideVec;
  // the table kernels for 256 < l <= 1024 give a workgroup 4096 rows: below 2^19 rows they leave most of the chip idle
  // (65536 x 1000 x 64: 44 us whatever the row count, against 15-45 us through the tile kernel; tools/ab_ts_long.sh)
  static const int ts_long_min_rows = dev_env_int("M4RI_HIP_TS_LONG_MIN_ROWS", 524288);
#ifdef GF2K_DEV_VARIANTS
  constexpr bool has_generation_kernel = true;
#else
  // the shipped gf2k_tallskinny has no kernel for n <= 64 with 256 < l <= 1024 (the slab kernel above takes those shapes under its
  // present thresholds; gf2_tallskinny4_kernel lives in development builds only): should a retuned threshold ever let one through,
  // it goes to the tile kernel instead of failing
  constexpr bool has_generation_kernel = false;
#endif
  if (n <= 256 && m >= (l > 256 ? ts_long_min_rows : 2048) && (n > 64 || l > 64) && l <= 1024 && (has_generation_kernel || n > 64 || l <= 256))
    return kPathTallSkinny;
  if (m <= 8) return kPathFewRows;
  if (few_rows_t_shape(m, l, n)) return kPathFewRowsT;
  return kPathTiles;
}

static int mul_m4rm_plain(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, hipStream_t s) {
  const int m = A->nrows, l = A->ncols, n = B->ncols;
  const PlainPath path = plain_path(m, l, n);
  if (path == kPathNothing) return 0;
  if (path == kPathZeroInner) {
    if (!accumulate) HIP_TRY(gf2k_xor2d(C->data, C->ld, nullptr, 0, nullptr, 0, m, words_of(n), s));
    return 0;
  }
  if (path == kPathSlabTables) {
    HIP_TRY(gf2k_tallskinny_long(A->data, A->ld, B->data, B->ld, C->data, C->ld, m, l, n, accumulate, s));
    return 0;
  }
  // 65-128 columns against a long inner dimension: that kernel with 16-byte entries, where the tile kernel finds a single column
  // tile and a handful of row tiles (65536^2 x 128: 0.78 -> 0.27 ms; 20000^2 x 128: 143 -> 46 us; 9000 x 33000 x 100: 126 -> 37 us).
  // A second pass for 129-256 columns pays from 32768-bit rows on (65536^2 x 256: 0.78 -> 0.53 ms; 65536 x 8192 x 256: 98 -> 125 us).
  if (path == kPathSlabPasses) {
    for (int c0 = 0; c0 < n; c0 += 128)
      HIP_TRY(gf2k_tallskinny_long(A->data, A->ld, B->data + c0 / 64, B->ld, C->data + c0 / 64, C->ld, m, l, std::min(128, n - c0), accumulate, s));
    return 0;
  }
  if (path == kPathWideVec) return mul_widevec(C, A, B, accumulate, s);  // few columns, long rows: a wave per row
  // tall and skinny: tables over ALL of B, A streamed once.  Built for short inner dimensions (a batch of LPN samples: l = 256);
  // with a long one the tables are rebuilt every 256 bits and the tile kernel with split-K is ~10x faster (65536 x 65600 x 64:
  // 6.4 ms here), so the border strips of peeled products do not come this way
  if (path == kPathTallSkinny) {
    HIP_TRY(gf2k_tallskinny(A->data, A->ld, B->data, B->ld, C->data, C->ld, m, l, n, accumulate, s));
    return 0;
  }
  if (path == kPathFewRows) {  // a handful of rows: stream B once (v*A path, binary_matrix.rs:552-563)
    if (!accumulate) HIP_TRY(gf2k_xor2d(C->data, C->ld, nullptr, 0, nullptr, 0, m, words_of(n), s));
    HIP_TRY(gf2k_va(A->data, A->ld, B->data, B->ld, C->data, C->ld, m, l, n, s));
    return 0;
  }
  // 9 to 128 rows against a tall B: the tile kernel would build its 256-entry tables for a handful of rows (64 x 65536 x 4096:
  // 90 us; 16 x 200000 x 600: 144 us for 15 MB of B).  Transposed, the product is n rows of l bits times at most 64 vectors per
  // pass -- the slab table kernel's shape: C^T = B^T A^T, with B transposed once (one more pass over B) and the small operands
  // transposed in and out.  Only for a B much taller than wide: the transposition of B runs at 1.7-1.9 TB/s (64 x 20000 x 20000:
  // 93 -> 151 us, 64 x 65536 x 65536: 0.86 -> 0.81 ms).
  {
    if (path == kPathFewRowsT) {
      const int passes = (m + 63) / 64;
      const long long ldl = (words_of(l) + 1) & ~1ll, wn = words_of(n), ldn = (wn + 1) & ~1ll, ldct = (passes + 1) & ~1ll;
      const size_t wBt = (size_t)n * ldl, wAt = (size_t)l * 2, wCt = (size_t)n * ldct, wTmp = accumulate ? (size_t)m * ldn : 0;
      void *ws = nullptr;
      if (stream_workspace(s, (wBt + wAt + wCt + wTmp) * sizeof(u64), &ws) == 0) {
        u64 *Bt = static_cast<u64 *>(ws), *At = Bt + wBt, *Ct = At + wAt, *Tmp = Ct + wCt;
        HIP_TRY(gf2k_transpose(Bt, ldl, B->data, B->ld, l, n, s));  // n x l
        for (int p = 0; p < passes; ++p) {
          const int mp = std::min(64, m - 64 * p);
          HIP_TRY(gf2k_transpose(At, 2, A->data + (long long)64 * p * A->ld, A->ld, mp, l, s));  // l x mp (one word per row)
          HIP_TRY(gf2k_tallskinny_long(Bt, ldl, At, 2, Ct + p, ldct, n, l, mp, 0, s));           // word p of the n rows of C^T
        }
        if (accumulate) {
          HIP_TRY(gf2k_transpose(Tmp, ldn, Ct, ldct, n, m, s));  // m x n
          HIP_TRY(gf2k_xor2d(C->data, C->ld, C->data, C->ld, Tmp, ldn, m, (int)wn, s));
        } else {
          HIP_TRY(gf2k_transpose(C->data, C->ld, Ct, ldct, n, m, s));
        }
        return 0;
      }
    }
  }
  // A tall product may first copy A into the row-group-packed layout (one extra pass over A, ~0.2 ms per GiB) so that the
  // paired tile kernels fetch it with contiguous loads: taken when the modelled launch gains more than the pass costs
  const long long wp = (words_of(l) + 1) & ~1ll, prow = ((long long)m + 63) & ~63ll;
  bool packed = false;
  TilePlan tp = plain_plan(m, l, n, &packed);
  const u64 *Aptr = A->data;
  long long lda = A->ld;
  if (packed) {
    void *pa = nullptr;
    if (stream_workspace(s, (size_t)(prow * wp * 8), &pa, 2) == 0) {
      HIP_TRY(gf2k_packA(static_cast<u64 *>(pa), wp, A->data, A->ld, m, words_of(l), s));
      Aptr = static_cast<const u64 *>(pa);
      lda = wp;
    } else {
      packed = false;
      tp = plan_tiles(m, l, n, 1, false);
    }
  }
  // buffer descriptors of the tile kernel carry 32-bit byte counts: one tile of A rows must stay below 4 GiB
  if ((!packed && (long long)A->ld * 8 * 4096 >= (1ll << 32)) || (long long)B->ld * 8 * 32 >= (1ll << 31))
    return fail_msg("gf2_mul_dev: row stride too large for the tile kernel (more than ~8 million columns)");
  gf2k_mul_args a{};
  a.A = Aptr;
  a.a_packed = packed ? 1 : 0;
  a.B = B->data;
  a.C = C->data;
  a.lda = lda;
  a.ldb = B->ld;
  a.ldc = C->ld;
  a.m = m;
  a.l = l;
  a.n = n;
  a.batch = 1;
  a.accumulate = accumulate;
  return launch_planned(a, tp, s);
}

// quadrants (0 = X11, 1 = X12, 2 = X21, 3 = X22) that combination q of a side adds up (second entry -1: a plain copy);
// the device copy of this table lives in gf2_kernels.hip (kStrassenSupp)
static const int kSupp[2][7][2] = {{{0, 3}, {2, 3}, {0, -1}, {3, -1}, {0, 1}, {2, 0}, {1, 3}},
                                   {{0, 3}, {0, -1}, {1, 3}, {2, 0}, {3, -1}, {0, 1}, {2, 3}}};

static int mul_strassen(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, int L, hipStream_t s,
                        bool sync_free) {
  const int m = A->nrows, l = A->ncols, n = B->ncols;
  // the level passes use 16-byte accesses: row strides must be even and the bases 16-byte aligned
  if ((A->ld | B->ld | C->ld) & 1) L = 0;
  if ((reinterpret_cast<uintptr_t>(A->data) | reinterpret_cast<uintptr_t>(B->data) | reinterpret_cast<uintptr_t>(C->data)) & 15) L = 0;
  if (L <= 0) return mul_m4rm_plain(C, A, B, accumulate, s);
  const size_t ws_bytes = strassen_ws_words(m, l, n, L) * sizeof(u64);
  void *ws = nullptr;
  if (int rc = stream_workspace(s, ws_bytes, &ws)) return rc;
  u64 *cur = static_cast<u64 *>(ws);
  const std::vector<PlanStep> plan = strassen_plan(L);
  std::vector<u64 *> Aop(L + 1, nullptr), Bop(L + 1, nullptr), Pop(L + 1, nullptr);
  {
    int prev = 0;
    for (const PlanStep &st : plan) {
      const int i = prev + st.levels();
      const size_t mi = (size_t)m >> i, li = (size_t)l >> i, ni = (size_t)n >> i, p7 = pow7(i);
      Aop[i] = cur;
      cur += p7 * mi * (li / 64);
      Bop[i] = cur;
      cur += p7 * li * (ni / 64);
      Pop[i] = cur;
      cur += p7 * mi * (ni / 64);
      if (st.virt) {  // the parents of the virtual level exist on the product side only
        Pop[prev + 1] = cur;
        cur += pow7(prev + 1) * ((size_t)m >> (prev + 1)) * (((size_t)n >> (prev + 1)) / 64);
      }
      prev = i;
    }
  }
  // leaf operands of A in the row-group-packed layout of the paired tile kernel (its A loads become contiguous): written by
  // the last split pass when that pass is a fused one
  TilePlan leaf_plan = plan_tiles(m >> L, l >> L, n >> L, (int)pow7(L), false);
  bool a_packed = false;
  if (strassen_packs_a(m, L)) {
    const TilePlan pk = plan_tiles(m >> L, l >> L, n >> L, (int)pow7(L), true);
    if (cfg_reads_packed(pk.cfg) && pk.t <= leaf_plan.t) leaf_plan = pk, a_packed = true;
  }

  // one split step on one side: operands of level `prev` (7^prev of them, or the caller's matrix) -> level i
  auto split_step = [&](const PlanStep &st, int prev, int i, int side, bool pack) -> int {
    const bool isA = side == 0;
    const int rows_prev = (isA ? m : l) >> prev, rows_i = (isA ? m : l) >> i;
    const int words_prev = ((isA ? l : n) >> prev) / 64, words_i = ((isA ? l : n) >> i) / 64;
    const int batch = (int)pow7(prev);
    const gf2_dmat *top = isA ? A : B;
    const u64 *src = prev ? (isA ? Aop[prev] : Bop[prev]) : top->data;
    const long long lds_ = prev ? (long long)words_prev : top->ld;
    const long long srcStride = prev ? (long long)rows_prev * lds_ : 0;
    u64 *dst = isA ? Aop[i] : Bop[i];
    const long long dstStride = (long long)rows_i * words_i;
    const int kside = pack ? 2 : side;
    if (st.k == 1) return (int)gf2k_strassen_split(dst, words_i, dstStride, src, lds_, srcStride, rows_i, words_i, kside, batch, s);
    if (st.k == 2) return (int)gf2k_strassen_split2(dst, words_i, dstStride, src, lds_, srcStride, rows_i, words_i, kside, batch, s);
    const u64 *s0[7], *s1[7];
    int groups = 1;
    s0[0] = src;
    s1[0] = nullptr;
    if (st.virt) {
      groups = 7;
      const long long hq = rows_prev / 2, wq = words_prev / 2;  // quadrants of the source operand
      auto quad = [&](int q) { return src + (long long)(q >> 1) * hq * lds_ + (long long)(q & 1) * wq; };
      for (int g = 0; g < 7; ++g) {
        s0[g] = quad(kSupp[side][g][0]);
        s1[g] = kSupp[side][g][1] >= 0 ? quad(kSupp[side][g][1]) : nullptr;
      }
    }
    return (int)gf2k_strassen_split3(dst, words_i, dstStride, s0, s1, groups, lds_, srcStride, rows_i, words_i, kside, batch, s);
  };

  auto run = [&]() -> int {
    int prev = 0;
    for (size_t k = 0; k < plan.size(); ++k) {
      const int i = prev + plan[k].levels();
      const bool last = k + 1 == plan.size();
      HIP_TRY((hipError_t)split_step(plan[k], prev, i, 0, last && a_packed));
      HIP_TRY((hipError_t)split_step(plan[k], prev, i, 1, false));
      prev = i;
    }
    {  // all 7^L leaf products in one batched launch
      const int mL = m >> L, lL = l >> L, nL = n >> L;
      gf2k_mul_args a{};
      a.lda = lL / 64;
      a.ldb = nL / 64;
      a.ldc = nL / 64;
      a.sA = (long long)mL * a.lda;
      a.sB = (long long)lL * a.ldb;
      a.sC = (long long)mL * a.ldc;
      a.A = Aop[L];
      a.B = Bop[L];
      a.C = Pop[L];
      a.m = mL;
      a.l = lL;
      a.n = nL;
      a.batch = (int)pow7(L);
      a.accumulate = 0;
      a.a_packed = a_packed ? 1 : 0;
      if (int r = launch_planned(a, leaf_plan, s)) return r;
    }
    // fold the products back up
    int i = L;
    for (int k = (int)plan.size() - 1; k >= 0; --k) {
      const PlanStep &st = plan[k];
      const int up = i - st.levels();  // products of level i -> level `up`
      const int mi = m >> i, wi = (n >> i) / 64, batch = (int)pow7(up);
      u64 *dst = up ? Pop[up] : C->data;
      const long long ldd = up ? (long long)((n >> up) / 64) : C->ld;
      const long long strD = up ? (long long)(m >> up) * ldd : 0;
      const int acc = up ? 0 : accumulate;
      const long long dP = (long long)mi * wi;
      if (st.k == 1) {
        HIP_TRY(gf2k_strassen_merge(dst, ldd, strD, Pop[i], wi, dP, mi, wi, acc, batch, s));
      } else if (st.k == 2) {
        HIP_TRY(gf2k_strassen_merge2(dst, ldd, strD, Pop[i], wi, dP, mi, wi, acc, batch, s));
      } else if (!st.virt) {
        HIP_TRY(gf2k_strassen_merge3(dst, ldd, strD, Pop[i], wi, dP, mi, wi, acc, 1, batch, s));
      } else {
        const int m1 = m >> (up + 1), w1 = (n >> (up + 1)) / 64;  // the 7 parents of the virtual level, dense
        HIP_TRY(gf2k_strassen_merge3(Pop[up + 1], w1, (long long)m1 * w1, Pop[i], wi, dP, mi, wi, 0, 7, batch, s));
        HIP_TRY(gf2k_strassen_merge(dst, ldd, strD, Pop[up + 1], w1, (long long)m1 * w1, m1, w1, acc, batch, s));
      }
      i = up;
    }
    return 0;
  };
  int rc = run();
  if (sync_free && rc == 0 && hipStreamSynchronize(s) != hipSuccess) rc = fail(hipGetLastError(), "hipStreamSynchronize");
  return rc;
}

// Dimensions that do not divide by the level plan (leaf rows integral and a multiple of 64 for the packed layout, leaf widths
// an even word count) keep their Strassen levels in one of two ways (the cliff VERDICT r1 item 4 names: 60000^3 or 65600^3
// would otherwise run as plain M4RM):
//   pad   A and B are copied into zero-padded buffers whose dimensions are rounded UP (one extra pass over each), the product
//         runs on the padded shape and its top left m x n corner is copied / added into C -- right when the dimensions fall
//         a little short of a multiple (60000 -> 61440);
//   peel  the largest dividing core (dimensions rounded DOWN) goes through Strassen in place, the three border strips
//         (bottom rows, right columns, the tail of the inner dimension) through the plain kernels -- right when the dimensions
//         are a little above a multiple (65600 = 65536 + 64: padding would push the 4096-row leaves to two row tiles each).
// The choice is by modelled time against plain M4RM on the given shape.
struct ShapePlan {
  int kind = 0;  // 0 plain, 1 pad, 2 peel
  int L = 0, mp = 0, lp = 0, np = 0;  // padded or core dimensions
  double t = 0;
};
static double plain_model(int m, int l, int n) {
  if (m <= 0 || l <= 0 || n <= 0) return 0.0;
  return plain_time_model(m, l, n);
}

// a border strip runs through the automatic choice among the level counts that divide it (no further padding / peeling)
static double strip_model(int m, int l, int n, int leaf_min) {
  if (m <= 0 || l <= 0 || n <= 0) return 0.0;
  double t = 0;
  (void)pick_levels(m, l, n, 0, leaf_min, &t);
  return t + 3e-6;
}

static ShapePlan plan_shape_uncached(int m, int l, int n, int req, int leaf_min);
static ShapePlan plan_shape(int m, int l, int n, int req, int leaf_min) {
  thread_local std::unordered_map<PlanKey, ShapePlan, PlanKeyHash> memo;
  const PlanKey key{m, l, n, req, leaf_min * 2 + (env_int("M4RI_HIP_STRASSEN_FUSE3", 1) ? 1 : 0)};
  auto it = memo.find(key);
  if (it != memo.end()) return it->second;
  if (memo.size() > 8192) memo.clear();
  const ShapePlan sp = plan_shape_uncached(m, l, n, req, leaf_min);
  memo.emplace(key, sp);
  return sp;
}
static ShapePlan plan_shape_uncached(int m, int l, int n, int req, int leaf_min) {
  static const double bw = (double)dev_env_int("M4RI_HIP_STREAM_GBS", 5000) * 1e9;
  static const int max_auto = env_int("M4RI_HIP_STRASSEN_MAX_LEVELS", 5);
  static const int debug = dev_env_int("M4RI_HIP_DEBUG_PLAN", 0);
  ShapePlan best;
  const double plain = plain_model(m, l, n);
  best.t = plain;
  if (debug) std::fprintf(stderr, "m4ri_hip plan %d x %d x %d: plain %.3f ms\n", m, l, n, best.t * 1e3);
  const int lo = req > 0 ? (req > 6 ? 6 : req) : 1, hi = req > 0 ? lo : max_auto;
  bool forced_done = false;
  static const int only_kind = dev_env_int("M4RI_HIP_SHAPE_KIND", 0);  // A/B measurements: 1 = padded plans only, 2 = peeled plans only
  auto consider = [&](int kind, int L, long long mm, long long ll, long long nn, double t) {
    if (only_kind && kind != only_kind) return;
    if (debug) std::fprintf(stderr, "  L=%d %s %lld x %lld x %lld: %.3f ms\n", L, kind == 1 ? "pad " : "peel", mm, ll, nn, t * 1e3);
    // (not forced:) the model is coarse: a plan must promise 5 % over plain M4RM to be taken (8 % until the tile model stopped counting
    // the per-tile overhead twice; 30000^3 then sat exactly on the threshold: padded 3.71 ms, plain 4.06 ms measured)
    static const double min_gain = dev_env_int("M4RI_HIP_SHAPE_MIN_GAIN_PCT", 5) * 1e-2;
    if ((req > 0 && !forced_done) || (t < best.t && (req > 0 || t < (1.0 - min_gain) * plain))) best = {kind, L, (int)mm, (int)ll, (int)nn, t}, forced_done = true;
  };
  for (int L = lo; L <= hi; ++L) {
    const long long um = 64ll << L, uw = 128ll << L;
    auto core_time = [&](long long mm, long long ll, long long nn) {
      return leaf_time_model((int)(mm >> L), (int)(ll >> L), (int)(nn >> L), (int)pow7(L), strassen_packs_a((int)mm, L)) +
             strassen_pass_bytes((double)mm, (double)ll, (double)nn, L) / bw + 3 * kPassLaunch * (double)strassen_plan(L).size();
    };
    auto leaves_ok = [&](long long mm, long long ll, long long nn) {
      return req > 0 || ((mm >> L) >= 1024 && (ll >> L) >= leaf_min && (nn >> L) >= leaf_min);
    };
    // pad: round up
    const long long mu = ((long long)m + um - 1) / um * um, lu = ((long long)l + uw - 1) / uw * uw, nu = ((long long)n + uw - 1) / uw * uw;
    if (mu <= 0x7fffffff && lu <= 0x7fffffff && nu <= 0x7fffffff && leaves_ok(mu, lu, nu))
      consider(1, L, mu, lu, nu, core_time(mu, lu, nu) + 2.0 * ((double)mu * lu + (double)lu * nu + (double)mu * nu) / 8.0 / bw + 3 * 3e-6);
    // peel: round down -- to the plan's unit, and to whole tiles of the leaf kernel (4096 rows, 512 columns per leaf: a 4352-row
    // leaf occupies two row tiles, so 70000 is better served by a 65536-row core than by a 69632-row one)
    auto peel = [&](long long md, long long ld, long long nd) {
      if (md <= 0 || ld <= 0 || nd <= 0 || !leaves_ok(md, ld, nd)) return;
      if (md == m && ld == l && nd == n) return;  // the shape as given: pick_levels' business
      // x 1.05: measured against padded plans of the same shapes (60000^3: peeled 26.6 ms at a model of 26.6, padded 25.2 at a model
      // of 26.9; 70000^3: 40.5 / 39.8 and 47.0 / 48.7), the peeled plans run 5-6 % above their model relative to the padded ones
      consider(2, L, md, ld, nd,
               1.05 * (core_time(md, ld, nd) + strip_model((int)md, l - (int)ld, (int)nd, leaf_min) + strip_model((int)md, l, n - (int)nd, leaf_min) +
                       strip_model(m - (int)md, l, n, leaf_min)));
    };
    const long long md = (long long)m / um * um, ld = (long long)l / uw * uw, nd = (long long)n / uw * uw;
    peel(md, ld, nd);
    const long long tm = 4096ll << L, tn = 512ll << L;
    const long long md2 = (long long)m / tm * tm, nd2 = (long long)n / tn * tn;
    if (md2 != md || nd2 != nd) {
      peel(md2 ? md2 : md, ld, nd2 ? nd2 : nd);
      if (md2 && md2 != md && nd2 != nd) peel(md2, ld, nd);
    }
  }
  if (debug) std::fprintf(stderr, "  -> kind %d L=%d (%d x %d x %d) %.3f ms\n", best.kind, best.L, best.mp, best.lp, best.np, best.t * 1e3);
  return best;
}

static int mul_strassen(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, int L, hipStream_t s, bool sync_free);
static int cap_levels_by_memory(int m, int l, int n, int L, hipStream_t s, size_t extra_bytes = 0);

static int mul_strassen_padded(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, const ShapePlan &pp, hipStream_t s) {
  const int m = A->nrows, l = A->ncols, n = B->ncols;
  const long long wa = pp.lp / 64, wb = pp.np / 64;
  const size_t wordsA = (size_t)pp.mp * wa, wordsB = (size_t)pp.lp * wb, wordsC = (size_t)pp.mp * wb;
  void *ws = nullptr;
  if (int rc = stream_workspace(s, (wordsA + wordsB + wordsC) * sizeof(u64), &ws, 3)) return rc;
  u64 *pa = static_cast<u64 *>(ws), *pb = pa + wordsA, *pc = pb + wordsB;
  HIP_TRY(gf2k_padcopy(pa, wa, pp.mp, (int)wa, A->data, A->ld, m, words_of(l), s));
  HIP_TRY(gf2k_padcopy(pb, wb, pp.lp, (int)wb, B->data, B->ld, l, words_of(n), s));
  gf2_dmat Ap{pa, wa, pp.mp, pp.lp}, Bp{pb, wb, pp.lp, pp.np}, Cp{pc, wb, pp.mp, pp.np};
  if (int rc = mul_strassen(&Cp, &Ap, &Bp, 0, pp.L, s, false)) return rc;
  // rows / columns past the operands are zero in the padded product, so whole words of the corner are exact
  HIP_TRY(gf2k_xor2d(C->data, C->ld, pc, wb, accumulate ? C->data : nullptr, C->ld, m, words_of(n), s));
  return 0;
}

// core through Strassen in place (views of the caller's buffers: the core's column offsets are multiples of 128 bits), borders plain
static int mul_strassen_peeled(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, const ShapePlan &pp, hipStream_t s) {
  const int m = A->nrows, l = A->ncols, n = B->ncols;
  const int mc = pp.mp, lc = pp.lp, nc = pp.np;
  gf2_dmat Ac{A->data, A->ld, mc, lc}, Bc{B->data, B->ld, lc, nc}, Cc{C->data, C->ld, mc, nc};
  if (int rc = mul_strassen(&Cc, &Ac, &Bc, accumulate, pp.L, s, false)) return rc;
  // a border strip: the level count among those that divide it (no further padding / peeling), capped by memory
  auto strip = [&](gf2_dmat *Cs, const gf2_dmat *As, const gf2_dmat *Bs, int acc) -> int {
    static const int leaf_min = env_int("M4RI_HIP_STRASSEN_LEAF_MIN", 2048);
    int Ls = pick_levels(As->nrows, As->ncols, Bs->ncols, 0, leaf_min);
    Ls = cap_levels_by_memory(As->nrows, As->ncols, Bs->ncols, Ls, s);
    return mul_strassen(Cs, As, Bs, acc, Ls, s, false);
  };
  if (l > lc) {  // tail of the inner dimension: core block of C ^= A[0:mc, lc:l] * B[lc:l, 0:nc]
    gf2_dmat At{A->data + lc / 64, A->ld, mc, l - lc}, Bt{B->data + (long long)lc * B->ld, B->ld, l - lc, nc};
    if (int rc = strip(&Cc, &At, &Bt, 1)) return rc;
  }
  if (n > nc) {  // right columns
    gf2_dmat Ar{A->data, A->ld, mc, l}, Br{B->data + nc / 64, B->ld, l, n - nc}, Cr{C->data + nc / 64, C->ld, mc, n - nc};
    if (int rc = strip(&Cr, &Ar, &Br, accumulate)) return rc;
  }
  if (m > mc) {  // bottom rows
    gf2_dmat Ab{A->data + (long long)mc * A->ld, A->ld, m - mc, l}, Cb{C->data + (long long)mc * C->ld, C->ld, m - mc, n};
    if (int rc = strip(&Cb, &Ab, B, accumulate)) return rc;
  }
  return 0;
}

static int mul_naive_dev(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, hipStream_t s,
                         bool sync_free) {
  // mzd_mul_naive (mzd.rs:150-152) = transpose B, then the row-parity product (mzd.rs:154-168).
  // For wide B the table kernel computes the same bits far faster, so only narrow products
  // (C one word wide: the matrix x vector path of mul_slice, binary_matrix.rs:416-431) take this route.
  const int m = A->nrows, l = A->ncols, n = B->ncols;
  if (n > 64 || l == 0) return mul_m4rm_plain(C, A, B, accumulate, s);
  if (m == 0 || n == 0) return 0;
  if (ts_long_shape(m, l, n)) {
    HIP_TRY(gf2k_tallskinny_long(A->data, A->ld, B->data, B->ld, C->data, C->ld, m, l, n, accumulate, s));
    if (sync_free && hipStreamSynchronize(s) != hipSuccess) return fail(hipGetLastError(), "hipStreamSynchronize");
    return 0;
  }
  if (widevec_shape(m, l, n)) {  // long rows: a wave per row
    if (int rc = mul_widevec(C, A, B, accumulate, s)) return rc;
    if (sync_free && hipStreamSynchronize(s) != hipSuccess) return fail(hipGetLastError(), "hipStreamSynchronize");
    return 0;
  }
  if (n > 8 && m >= 2048) return mul_m4rm_plain(C, A, B, accumulate, s);  // batch of vectors: table kernel (see there)
  // one to eight vectors against MANY short rows (`&A * &v` on 2^20 LPN samples): the 8-bit table kernel of gf2_lpn.inc streams A
  // with wave-contiguous non-temporal loads and costs the same whatever n <= 64 is (2^20 x 256 x 1 cold: 9.4 us through the
  // AND / popcount kernel below, 8.7-8.9 through the tables); with fewer rows the popcount kernel's small workgroups start faster
  static const int narrow_lpn_rows = dev_env_int("M4RI_HIP_NARROW_LPN_ROWS", 262144);
  if (narrow_lpn_rows > 0 && m >= narrow_lpn_rows && l <= 256 && l > 64) {
    HIP_TRY(gf2k_tallskinny(A->data, A->ld, B->data, B->ld, C->data, C->ld, m, l, n, accumulate, s));
    if (sync_free && hipStreamSynchronize(s) != hipSuccess) return fail(hipGetLastError(), "hipStreamSynchronize");
    return 0;
  }
  if ((size_t)n * words_of(l) * 8 <= 65536) {  // one launch: B is transposed into LDS by every block
    hipError_t e1 = gf2k_narrow(A->data, A->ld, B->data, B->ld, C->data, C->ld, m, l, n, accumulate, s);
    if (e1 != hipSuccess) return fail(e1, "gf2k_narrow");
    if (sync_free && hipStreamSynchronize(s) != hipSuccess) return fail(hipGetLastError(), "hipStreamSynchronize");
    return 0;
  }
  const long long ldbt = (words_of(l) + 1) & ~1ll;
  const size_t bytes = (size_t)n * ldbt * sizeof(u64);
  void *bt = nullptr;
  if (int rc = stream_workspace(s, bytes, &bt)) return rc;
  hipError_t e = gf2k_transpose(static_cast<u64 *>(bt), ldbt, B->data, B->ld, l, n, s);
  if (e != hipSuccess) return fail(e, "gf2k_transpose");
  e = gf2k_rowparity(A->data, A->ld, static_cast<u64 *>(bt), ldbt, C->data, C->ld, m, l, n, accumulate, s);
  if (e != hipSuccess) return fail(e, "gf2k_rowparity");
  if (sync_free && hipStreamSynchronize(s) != hipSuccess) return fail(hipGetLastError(), "hipStreamSynchronize");
  return 0;
}

static int check_mul_dims(const gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B) {
  if (!C || !A || !B || !C->data || !A->data || !B->data) return fail_msg("gf2_mul_dev: null operand");
  if (A->ncols != B->nrows || C->nrows != A->nrows || C->ncols != B->ncols)
    return fail_msg("gf2_mul_dev: dimension mismatch");
  if (A->ld < words_of(A->ncols) || B->ld < words_of(B->ncols) || C->ld < words_of(C->ncols))
    return fail_msg("gf2_mul_dev: row stride smaller than row width");
  return 0;
}

// Products that use the per-stream workspace enqueue several kernels that must stay contiguous on the stream
// (another host thread enqueueing on the SAME stream in between would reuse the arena under them).
static std::mutex g_enqueue_mu;

// The operand arena of L levels must fit: what the driver reports free plus what this library already holds (its block
// cache and this stream's current arena are handed back before a larger one is allocated).
// `extra_bytes`: what the caller allocates besides the arena (the zero-padded copies of a padded product).
static int cap_levels_by_memory(int m, int l, int n, int L, hipStream_t s, size_t extra_bytes) {
  if (L <= 0) return L;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return L;
  }
  size_t mine = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    auto it = g_ws.find(std::make_tuple(dev, s, 0));
    if (it != g_ws.end()) mine += it->second.bytes;
  }
  {
    DevPool &pool = g_pools[dev & 15];
    std::lock_guard<std::mutex> lk(pool.mu);
    mine += pool.cached;
  }
  const size_t avail = (size_t)((free_b + mine) * 0.95);
  while (L > 0 && strassen_ws_words(m, l, n, L) * sizeof(u64) + extra_bytes > avail) --L;
  return L;
}

static int mul_dispatch(gf2_dmat *C, const gf2_dmat *A, const gf2_dmat *B, int accumulate, int algo, int param,
                        hipStream_t s, bool sync_free) {
  std::unique_lock<std::mutex> lk(g_enqueue_mu, std::defer_lock);
  if (!sync_free) lk.lock();  // host-path calls own a private (thread-local) stream
  switch (algo) {
    case GF2_ALGO_NAIVE:
      return mul_naive_dev(C, A, B, accumulate, s, sync_free);
    case GF2_ALGO_M4RM: {
      int rc = mul_m4rm_plain(C, A, B, accumulate, s);
      if (rc == 0 && sync_free && hipStreamSynchronize(s) != hipSuccess)
        rc = fail(hipGetLastError(), "hipStreamSynchronize");
      return rc;
    }
    case GF2_ALGO_AUTO:
    case GF2_ALGO_STRASSEN: {
      static const int leaf_min = env_int("M4RI_HIP_STRASSEN_LEAF_MIN", 2048);
      const int m = A->nrows, l = A->ncols, n = B->ncols;
      int L = pick_levels(m, l, n, param, leaf_min);
      static const int pad_on = env_int("M4RI_HIP_STRASSEN_PAD", 1);
      const bool even = ((A->ld | B->ld | C->ld) & 1) == 0;
      const bool divides = L > 0 && (param <= 0 || L == (param > 6 ? 6 : param)) && even;
      if (pad_on && !divides && m >= 1024 && (long long)l * n >= (1ll << 22)) {  // lost levels to divisibility (or to odd strides)?
        ShapePlan pp = plan_shape(m, l, n, param, leaf_min);
        if (pp.kind == 2 && !even) pp.kind = 0;  // peeling works on views of the caller's buffers
        // a padded product also holds the three padded copies (slot 3) next to the arena
        const size_t pad_bytes = pp.kind == 1 ? ((size_t)pp.mp * (pp.lp / 64) + (size_t)pp.lp * (pp.np / 64) + (size_t)pp.mp * (pp.np / 64)) * sizeof(u64) : 0;
        if (pp.kind && pp.L > L && cap_levels_by_memory(pp.mp, pp.lp, pp.np, pp.L, s, pad_bytes) == pp.L) {
          int rc = pp.kind == 1 ? mul_strassen_padded(C, A, B, accumulate, pp, s) : mul_strassen_peeled(C, A, B, accumulate, pp, s);
          if (rc == 0 && sync_free && hipStreamSynchronize(s) != hipSuccess) rc = fail(hipGetLastError(), "hipStreamSynchronize");
          return rc;
        }
      }
      L = cap_levels_by_memory(m, l, n, L, s);
      return mul_strassen(C, A, B, accumulate, L, s, sync_free);
    }
    default:
      return fail_msg("gf2_mul_dev: unknown algorithm");
  }
}

extern "C" int gf2_mul_dev(gf2_dmat *C, gf2_dmat const *A, gf2_dmat const *B, int accumulate, int algo, int param,
                           void *stream) {
  if (int rc = require_device()) return rc;
  if (int rc = check_mul_dims(C, A, B)) return rc;
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  return mul_dispatch(C, A, B, accumulate, algo, param, s, /*sync_free=*/false);
}

extern "C" int gf2_mul_nt_dev(gf2_dmat *C, gf2_dmat const *A, gf2_dmat const *Bt, int accumulate, void *stream) {
  if (int rc = require_device()) return rc;
  if (!C || !A || !Bt || !C->data || !A->data || !Bt->data) return fail_msg("gf2_mul_nt_dev: null operand");
  if (A->ncols != Bt->ncols || C->nrows != A->nrows || C->ncols != Bt->nrows)
    return fail_msg("gf2_mul_nt_dev: dimension mismatch");
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  const int m = A->nrows, l = A->ncols, n = Bt->nrows;
  if (widevec_shape(m, l, n)) {  // long rows, at most 64 vectors: a wave per row (Bt is already what that kernel reads)
    HIP_TRY(gf2k_widevec(A->data, A->ld, Bt->data, Bt->ld, C->data, C->ld, m, l, n < 32 ? n : 32, accumulate, 0, s));
    if (n > 32) HIP_TRY(gf2k_widevec(A->data, A->ld, Bt->data + 32 * Bt->ld, Bt->ld, C->data, C->ld, m, l, n - 32, 1, 32, s));
    return 0;
  }
  HIP_TRY(gf2k_rowparity(A->data, A->ld, Bt->data, Bt->ld, C->data, C->ld, m, l, n, accumulate, s));
  return 0;
}

extern "C" size_t gf2_mul_workspace_bytes(int m, int l, int n, int algo, int param) {
  // the scratch of a plain product follows the path mul_m4rm_plain takes (plain_path): a packed copy of A and partial tiles for
  // the planned tile kernels, the transposed vectors for the wave-per-row kernel, nothing for the table kernels
  auto plain_bytes = [&]() -> size_t {
    switch (plain_path(m, l, n)) {
      case kPathWideVec: return (size_t)n * ((words_of(l) + 1) & ~1) * 8;
      case kPathFewRowsT: return few_rows_t_bytes(m, l, n, 1);
      case kPathTiles: {
        bool pack = false;
        const TilePlan tp = plain_plan(m, l, n, &pack);
        return tp.scratch() + (pack ? (size_t)((m + 63) & ~63) * (size_t)((words_of(l) + 1) & ~1) * 8 : 0);
      }
      default: return 0;
    }
  };
  if (algo == GF2_ALGO_NAIVE) {
    // mul_naive_dev: wide products are forwarded to mul_m4rm_plain; up to 64 columns take the table / wave-per-row kernels or the
    // AND / popcount kernels, which need at most the transposed B
    if (n > 64 || l == 0) return plain_bytes();
    if (m > 0 && ts_long_shape(m, l, n)) return 0;
    if (m > 0 && widevec_shape(m, l, n)) return (size_t)n * ((words_of(l) + 1) & ~1) * 8;
    if (n > 8 && m >= 2048) return plain_bytes();
    return (size_t)n * ((words_of(l) + 1) & ~1) * 8;
  }
  const size_t plain_ws = plain_bytes();
  if (algo == GF2_ALGO_M4RM) return plain_ws;
  static const int leaf_min = env_int("M4RI_HIP_STRASSEN_LEAF_MIN", 2048);
  const int L = pick_levels(m, l, n, param, leaf_min);
  if (L <= 0) return plain_ws;
  const bool pk = strassen_packs_a(m, L);
  const TilePlan a = plan_tiles(m >> L, l >> L, n >> L, (int)pow7(L), false), b = pk ? plan_tiles(m >> L, l >> L, n >> L, (int)pow7(L), true) : a;
  const TilePlan &lp = (pk && cfg_reads_packed(b.cfg) && b.t <= a.t) ? b : a;
  return strassen_ws_words(m, l, n, L) * sizeof(u64) + lp.scratch();
}

extern "C" int gf2_add_dev(gf2_dmat *C, gf2_dmat const *A, gf2_dmat const *B, void *stream) {
  if (int rc = require_device()) return rc;
  if (A->nrows != B->nrows || A->ncols != B->ncols || C->nrows != A->nrows || C->ncols != A->ncols)
    return fail_msg("gf2_add_dev: dimension mismatch");
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  HIP_TRY(gf2k_xor2d(C->data, C->ld, A->data, A->ld, B->data, B->ld, A->nrows, words_of(A->ncols), s));
  return 0;
}

extern "C" int gf2_transpose_dev(gf2_dmat *D, gf2_dmat const *S, void *stream) {
  if (int rc = require_device()) return rc;
  if (D->nrows != S->ncols || D->ncols != S->nrows) return fail_msg("gf2_transpose_dev: dimension mismatch");
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  HIP_TRY(gf2k_transpose(D->data, D->ld, S->data, S->ld, S->nrows, S->ncols, s));
  return 0;
}

extern "C" int gf2_equal_dev(gf2_dmat const *A, gf2_dmat const *B, int *equal, void *stream) {
  if (int rc = require_device()) return rc;
  if (A->nrows != B->nrows || A->ncols != B->ncols) {
    *equal = 0;
    return 0;
  }
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  void *flag = nullptr;
  if (int rc = dev_alloc(&flag, sizeof(int))) return rc;
  int host = 0, rc = 0;
  do {
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), s);
    if (e == hipSuccess) e = gf2k_diff(A->data, A->ld, B->data, B->ld, A->nrows, A->ncols, static_cast<int *>(flag), s);
    if (e == hipSuccess) e = hipMemcpyAsync(&host, flag, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = fail(e, "gf2_equal_dev");
  } while (0);
  dev_free(flag, sizeof(int));
  *equal = host ? 0 : 1;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// device matrices <-> host mzd_t
// ---------------------------------------------------------------------------------------------

static inline long long dev_ld_for(int ncols) {
  const long long w = words_of(ncols);
  return w <= 1 ? (w ? w : 1) : ((w + 1) & ~1ll);
}

extern "C" int gf2_dmat_alloc(gf2_dmat *M, int nrows, int ncols) {
  if (int rc = require_device()) return rc;
  if (!M || nrows < 0 || ncols < 0) return fail_msg("gf2_dmat_alloc: bad arguments");
  M->nrows = nrows;
  M->ncols = ncols;
  M->ld = dev_ld_for(ncols);
  void *p = nullptr;
  if (int rc = dev_alloc(&p, (size_t)(nrows ? nrows : 1) * M->ld * sizeof(u64))) return rc;
  M->data = static_cast<u64 *>(p);
  return 0;
}

// internal: the caller has already waited for every stream that touched M
static void dmat_release(gf2_dmat *M) {
  if (!M || !M->data) return;
  dev_free(M->data, (size_t)(M->nrows ? M->nrows : 1) * M->ld * sizeof(u64));
  M->data = nullptr;
}

// Public free = hipFree semantics: the device API is asynchronous, so the block may still be read or written by queued
// work on any stream; wait for the owning device before the block can be handed to another caller.
extern "C" void gf2_dmat_free(gf2_dmat *M) {
  if (!M || !M->data) return;
  int cur = 0, own = owner_of(M->data, false);
  if (hipGetDevice(&cur) != hipSuccess) cur = -1;
  if (own >= 0 && cur >= 0 && own != cur) (void)hipSetDevice(own);
  if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
  if (own >= 0 && cur >= 0 && own != cur) (void)hipSetDevice(cur);
  dmat_release(M);
}

// Stream-ordered free: the block returns to the pool once everything queued on `stream` so far has completed.  For
// matrices that were only ever used on that one stream (temporaries of a chain of device products).
extern "C" int gf2_dmat_free_async(gf2_dmat *M, void *stream) {
  if (!M || !M->data) return 0;
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  const size_t bytes = (size_t)(M->nrows ? M->nrows : 1) * M->ld * sizeof(u64);
  if (free_after(s, M->data, bytes) != 0) {  // could not record an event: fall back to waiting
    gf2_dmat_free(M);
    return 0;
  }
  M->data = nullptr;
  reap_deferred(false);
  return 0;
}

extern "C" int gf2_dmat_fill_random_block(gf2_dmat *M, uint64_t seed, int64_t row0, int64_t col_word0, int full_ncols,
                                          void *stream) {
  if (int rc = require_device()) return rc;
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  HIP_TRY(gf2k_fill_random(M->data, M->ld, M->nrows, M->ncols, seed, row0, full_ncols > 0 ? words_of(full_ncols) : 0,
                           col_word0, s));
  return 0;
}

extern "C" int gf2_dmat_fill_random_rows(gf2_dmat *M, uint64_t seed, int64_t row0, void *stream) {
  return gf2_dmat_fill_random_block(M, seed, row0, 0, 0, stream);
}

extern "C" int gf2_dmat_fill_random(gf2_dmat *M, uint64_t seed, void *stream) {
  return gf2_dmat_fill_random_rows(M, seed, 0, stream);
}

extern "C" double gf2_strassen_pass_bytes(int m, int l, int n, int levels) {
  return levels > 0 ? strassen_pass_bytes(m, l, n, levels) : 0.0;
}

// How a device product of this shape would run (no device needed: the cost model's answer, before the memory cap):
// *kind = 0 as given (levels Strassen levels, 0 = plain M4RM), 1 zero-padded to dims[0..2], 2 peeled to the core dims[0..2]
// with the border strips through the plain kernels.  Returns the number of Strassen levels.
extern "C" double gf2_tile_plan(int m, int l, int n, int batch, int packed, long long out[9]) {
  const TilePlan tp = plan_tiles(m, l, n, batch < 1 ? 1 : batch, packed != 0);
  if (out) {
    out[0] = tp.cfg;
    out[1] = tp.ksplit;
    out[2] = tp.n_rem;
    out[3] = tp.nseg;
    out[4] = (long long)tp.scratch();
    out[5] = tp.tail_batch;
    out[6] = tp.tail_cfg;
    out[7] = tp.tail_n_rem;
    out[8] = tp.tail_nseg;
  }
  return tp.t;
}

// the row band of the same plan: out = {rows of the band (0: none), its variant, its stream-K cut (tiles, segments), its scratch}
extern "C" void gf2_tile_plan_band(int m, int l, int n, int batch, int packed, long long out[5]) {
  if (!out) return;
  const TilePlan tp = plan_tiles(m, l, n, batch < 1 ? 1 : batch, packed != 0);
  out[0] = tp.band_rows;
  out[1] = tp.band_cfg;
  out[2] = tp.band_n_rem;
  out[3] = tp.band_nseg;
  out[4] = (long long)tp.band_ws_bytes;
}

extern "C" int gf2_mul_plan(int m, int l, int n, int algo, int param, int *kind, int dims[3]) {
  int k = 0, L = 0, d[3] = {m, l, n};
  if (algo == GF2_ALGO_AUTO || algo == GF2_ALGO_STRASSEN) {
    static const int leaf_min = env_int("M4RI_HIP_STRASSEN_LEAF_MIN", 2048);
    static const int pad_on = env_int("M4RI_HIP_STRASSEN_PAD", 1);
    L = pick_levels(m, l, n, param, leaf_min);
    const bool divides = L > 0 && (param <= 0 || L == (param > 6 ? 6 : param));
    if (pad_on && !divides && m >= 1024 && (long long)l * n >= (1ll << 22)) {
      const ShapePlan pp = plan_shape(m, l, n, param, leaf_min);
      if (pp.kind && pp.L > L) {
        k = pp.kind;
        L = pp.L;
        d[0] = pp.mp;
        d[1] = pp.lp;
        d[2] = pp.np;
      }
    }
  }
  if (kind) *kind = k;
  if (dims) dims[0] = d[0], dims[1] = d[1], dims[2] = d[2];
  return L;
}

extern "C" int gf2_strassen_levels(int m, int l, int n, int algo, int param) {
  if (algo != GF2_ALGO_AUTO && algo != GF2_ALGO_STRASSEN) return 0;
  static const int leaf_min = env_int("M4RI_HIP_STRASSEN_LEAF_MIN", 2048);
  int L = pick_levels(m, l, n, param, leaf_min);
  int ndev = 0;
  if (L > 0 && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) L = cap_levels_by_memory(m, l, n, L, nullptr);
  else (void)hipGetLastError();
  return L;
}

// host rows -> device. Our mzd_t are single-block with a constant rowstride (mzd_host.cpp), windows included, so rows
// [r0, r0 + dst->nrows) of `src` are one strided region starting at src->rows[r0].
// Asynchronous form (internal): `src` must stay untouched until the stream has passed the copy.
static int upload_rows_async(gf2_dmat *dst, mzd_t const *src, int r0, hipStream_t s) {
  if (r0 < 0 || r0 + dst->nrows > src->nrows || dst->ncols != src->ncols) return fail_msg("gf2_dmat_upload: dimension mismatch");
  if (dst->nrows == 0 || src->ncols == 0) return 0;
  const size_t wbytes = (size_t)src->width * sizeof(word);
  if (dst->ld == src->rowstride)
    HIP_TRY(hipMemcpyAsync(dst->data, src->rows[r0], ((size_t)(dst->nrows - 1) * src->rowstride + src->width) * sizeof(word),
                           hipMemcpyHostToDevice, s));
  else
    HIP_TRY(hipMemcpy2DAsync(dst->data, (size_t)dst->ld * sizeof(u64), src->rows[r0], (size_t)src->rowstride * sizeof(word),
                             wbytes, dst->nrows, hipMemcpyHostToDevice, s));
  return 0;
}
static int upload_async(gf2_dmat *dst, mzd_t const *src, hipStream_t s) {
  if (dst->nrows != src->nrows) return fail_msg("gf2_dmat_upload: dimension mismatch");
  return upload_rows_async(dst, src, 0, s);
}

// public form: returns once the host rows have been consumed (the caller may free or modify `src` right away;
// pinned blocks make the copy itself truly asynchronous, so this has to wait for it)
extern "C" int gf2_dmat_upload(gf2_dmat *dst, mzd_t const *src, void *stream) {
  if (int rc = require_device()) return rc;
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  if (int rc = upload_async(dst, src, s)) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

// device -> rows [r0, r0 + src->nrows) of a host matrix; returns when they are complete in host memory
static int download_rows(mzd_t *dst, int r0, gf2_dmat const *src, hipStream_t s) {
  if (r0 < 0 || r0 + src->nrows > dst->nrows || dst->ncols != src->ncols) return fail_msg("gf2_dmat_download: dimension mismatch");
  const int nrows = src->nrows;
  if (nrows == 0 || dst->ncols == 0) return 0;
  const size_t wbytes = (size_t)dst->width * sizeof(word);
  const bool windowed = (dst->flags & mzd_flag_windowed_zerooffset) != 0;
  if (windowed && dst->high_bitmask != m4ri_ffff) {
    // the last word of each row is shared with the parent matrix: merge under the mask
    std::vector<word> tmp((size_t)nrows * dst->width);
    HIP_TRY(hipMemcpy2DAsync(tmp.data(), wbytes, src->data, (size_t)src->ld * sizeof(u64), wbytes, nrows,
                             hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (rci_t i = 0; i < nrows; ++i) {
      word *d = dst->rows[r0 + i];
      const word *t = tmp.data() + (size_t)i * dst->width;
      for (wi_t j = 0; j + 1 < dst->width; ++j) d[j] = t[j];
      d[dst->width - 1] = (d[dst->width - 1] & ~dst->high_bitmask) | (t[dst->width - 1] & dst->high_bitmask);
    }
    return 0;
  }
  if (!windowed && src->ld == dst->rowstride)
    HIP_TRY(hipMemcpyAsync(dst->rows[r0], src->data, ((size_t)(nrows - 1) * dst->rowstride + dst->width) * sizeof(word),
                           hipMemcpyDeviceToHost, s));
  else
    HIP_TRY(hipMemcpy2DAsync(dst->rows[r0], (size_t)dst->rowstride * sizeof(word), src->data,
                             (size_t)src->ld * sizeof(u64), wbytes, nrows, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}

extern "C" int gf2_dmat_download(mzd_t *dst, gf2_dmat const *src, void *stream) {
  if (int rc = require_device()) return rc;
  if (dst->nrows != src->nrows || dst->ncols != src->ncols) return fail_msg("gf2_dmat_download: dimension mismatch");
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  return download_rows(dst, 0, src, s);
}

// ---------------------------------------------------------------------------------------------
// M4RI entry points on host mzd_t: upload, multiply on the device, download
// ---------------------------------------------------------------------------------------------

namespace {
// Operand cache: device copies of host matrices that the caller declared constant (gf2_mzd_cache_on_device).  A product
// whose A or B is cached skips that upload -- the drop-in path of repeated A*v with a fixed A (mul_slice,
// binary_matrix.rs:416-431) is otherwise bound by moving A over PCIe every call.
// Entries are keyed by the matrix' BLOCK (mzd_t::blocks, which windows share with their parent, mzd.rs:323-346): a library
// call that writes through a window therefore drops the parent's copy too.  An entry is handed out as a shared reference
// that a product holds until its stream has drained, so gf2_mzd_uncache from another thread never frees a copy in use.
struct CachedOperand {
  gf2_dmat d{};
  int dev = 0;
  int nrows = 0, ncols = 0, rowstride = 0;
  const word *row0 = nullptr;
  ~CachedOperand() {
    if (d.data) dev_free(d.data, (size_t)(d.nrows ? d.nrows : 1) * d.ld * sizeof(u64));
  }
};
std::mutex g_cache_mu;
std::map<const void *, std::shared_ptr<CachedOperand>> g_cache;

// Result side copies: the PACKED TRANSPOSED form of a fresh thin product, kept on the host next to the product.  The friendly layer
// turns every `&A * &v` into mzd_mul_naive(NULL, A, v^T) followed by mzd_transpose(NULL, result) (binary_matrix.rs:416-431,
// :332-361, :528-542): in M4RI's layout the 2^20 x 1 result is one 64-bit word per row, 8 MiB for 128 KiB of bits, and gathering
// bit 0 of 2^20 words on the host cost 204 us of a 565-us call (profiles/r05_av_breakdown.txt).  The device has those bits in a
// register file anyway: a product into a library-allocated (NULL) destination with at most M4RI_HIP_RESULT_SIDE_COLS columns also
// transposes C on the device (one launch) and brings the n x m form down beside C; mzd_transpose of that matrix is then a copy.
// Keyed like the operand cache by the matrix' block and dropped by the same gf2_cache_forget calls (every library routine that
// writes a matrix, and mzd_free); stores through rows[] are invisible to the library, as for the operand cache.
struct ResultSide {
  word *buf = nullptr;  // pinned; ncols rows of ld words
  size_t bytes = 0;
  size_t ld = 0;
  int nrows = 0, ncols = 0, rowstride = 0;  // of the product
  const word *row0 = nullptr;
  ~ResultSide() { gf2_pinned_free(buf, bytes); }
};
std::map<const void *, std::shared_ptr<ResultSide>> g_result_side;

struct DMatOwner {
  gf2_dmat d{};
  std::shared_ptr<CachedOperand> borrowed;  // d belongs to the operand cache, kept alive by this reference
  bool released = false;                    // ownership moved elsewhere
  ~DMatOwner() {
    if (!borrowed && !released) dmat_release(&d);  // every user synchronises its stream before the owner goes out of scope
  }
};

static inline const void *cache_key(const mzd_t *M) { return M->blocks ? static_cast<const void *>(M->blocks) : M; }

std::shared_ptr<CachedOperand> cache_lookup(const mzd_t *M) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  auto it = g_cache.find(cache_key(M));
  if (it == g_cache.end()) return nullptr;
  const CachedOperand &c = *it->second;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev != c.dev) return nullptr;
  // the same view of the block (a window of a cached parent, or the parent of a cached window, is a different operand)
  if (c.nrows != M->nrows || c.ncols != M->ncols || c.rowstride != M->rowstride || (M->nrows && c.row0 != M->rows[0])) return nullptr;
  return it->second;
}

// device copy of rows [r0, r1) of M whose row stride equals the host row stride when the host block is contiguous, so
// that the transfer is one linear DMA
int to_device_rows(DMatOwner &o, const mzd_t *M, int r0, int r1, hipStream_t s, bool copy) {
  if (copy && r0 == 0 && r1 == M->nrows) {
    if (auto hit = cache_lookup(M)) {  // a read-only operand that already lives on the device
      o.d = hit->d;
      o.borrowed = std::move(hit);
      return 0;
    }
  }
  o.d.nrows = r1 - r0;
  o.d.ncols = M->ncols;
  const bool windowed = (M->flags & mzd_flag_windowed_zerooffset) != 0;
  o.d.ld = (!windowed && M->rowstride >= 1) ? M->rowstride : dev_ld_for(M->ncols);
  void *p = nullptr;
  if (int rc = dev_alloc(&p, (size_t)(o.d.nrows ? o.d.nrows : 1) * o.d.ld * sizeof(u64))) return rc;
  o.d.data = static_cast<u64 *>(p);
  if (copy) return upload_rows_async(&o.d, M, r0, s);  // the callers synchronise before they return
  return 0;
}
int to_device(DMatOwner &o, const mzd_t *M, hipStream_t s, bool copy) { return to_device_rows(o, M, 0, M->nrows, s, copy); }

// Streams for the worker threads of a multi-device product: leased from a per-device pool for the duration of a call
// (a thread_local stream per short-lived worker would leak one stream, and its scratch arenas, per call).
std::mutex g_lease_mu;
std::vector<hipStream_t> g_lease[16];
int lease_stream(int dev, hipStream_t *out) {
  {
    std::lock_guard<std::mutex> lk(g_lease_mu);
    auto &v = g_lease[dev & 15];
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return 0;
    }
  }
  HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));  // on the current device: the caller has set `dev`
  return 0;
}
void unlease_stream(int dev, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_lease_mu);
  g_lease[dev & 15].push_back(s);
}

// One device's share of a host product: C[r0:r1, :] (+)= A[r0:r1, :] * B on the CURRENT device, stream s.  Returns when
// those rows of C are complete in host memory.
// The side copy of a fresh thin product (see ResultSide): C^T into a device scratch, brought down into `side->buf` on stream s.
// The caller synchronises s before it looks at the buffer or lets `dT` go.
// ---- schedules of a large host product (round 5) ----
// Modelled device time of one (sub-)product as the host path would run it.
double host_product_model(int m, int l, int n, int algo, int param) {
  if (algo == GF2_ALGO_M4RM || algo == GF2_ALGO_NAIVE) return plain_time_model(m, l, n);
  static const int leaf_min = env_int("M4RI_HIP_STRASSEN_LEAF_MIN", 2048);
  double t = 0;
  (void)pick_levels(m, l, n, param, leaf_min, &t);
  return t > 0 ? t : plain_time_model(m, l, n);
}
// A host product is three queues -- uploads, products, downloads -- and a plan is the order and size of their pieces.  Two families:
//   row blocks   C_i = A_i B: blocks of A and C (the first one against the two halves of B, so that it can start early); every
//                block's rows of C leave as soon as they exist, but a block product runs below the whole product's efficiency
//                (8192-row blocks of 32768^3: 1.45 ms against 1.0 for a quarter of the whole);
//   slabs        C ^= A[:, K] B[K, :] over slabs K of the INNER dimension: every product keeps all rows (65536 x 16384 x 65536
//                accumulate: 8.4 ms, a quarter of the whole product is 7.2), only the first slab's upload is exposed -- A's
//                slab is a 2-D copy of 1-2 KiB pieces, which runs at the linear rate --; the last slab is multiplied in row
//                blocks so that C does not leave all at once behind it.
// Which one ends first depends on how compute and PCIe compare, so each candidate is played through with the planner's own time
// model and the copy rate of the link (profiles/r05_slab_prices.txt has the measured sub-products).
struct HostPlan {
  // slab plans: gslabs[g] = the sizes (bits) of the slabs of the inner dimension through which row group g of A and C runs, one group
  // after the other (a finished group's rows of C leave while the next group is multiplied; half-height sub-products are no less
  // efficient: 32768^3 16384 x 8192 x 32768 0.64 ms against 1.24 for twice the rows).  The FIRST group's slabs also bring B in, so
  // they start small (a short lead-in) and grow; later groups find B resident and take few, large slabs.  Empty: the row-block plan.
  std::vector<std::vector<int>> gslabs;
  double t_end = 0;
};
constexpr int kHostPlanCandidates = 12;
HostPlan plan_host_product(int rows, int l, int n, int algo, int param, bool b_resident, size_t a_row_bytes, size_t b_row_bytes, size_t c_row_bytes,
                           double *all_t_end = nullptr, int *chosen = nullptr) {
  static const double rate = (double)dev_env_int("M4RI_HIP_PCIE_GBS", 55) * 1e9;
  // read per call (tests and A/B runs): the number of a candidate below (1 row blocks ... 12); 0 = by the model
  const int forced = env_int("M4RI_HIP_HOST_PLAN", 0);
  auto T = [&](int m_, int l_) { return host_product_model(m_, l_, n, algo, param); };
  const int NB = 4;
  // row blocks: quarter, half, quarter when a quarter keeps 16384 rows (see host_mul_range), B in two halves for the first block
  HostPlan rb;
  {
    const int q = rows / 4;
    std::vector<int> bnd = q >= 16384 ? std::vector<int>{0, q, 3 * q, rows} : std::vector<int>{0, q, 2 * q, 3 * q, rows};
    const bool halves = !b_resident && l >= 8192 && l % 256 == 0;
    double up = 0, tc = 0, td = 0;
    std::vector<double> arrA;
    up += (double)bnd[1] * a_row_bytes / rate;
    arrA.push_back(up);
    double bt = up, bb = up;
    if (!b_resident) {
      up += (double)l / 2 * b_row_bytes / rate;
      bt = up;
      up += (double)l / 2 * b_row_bytes / rate;
      bb = up;
    }
    for (size_t i = 1; i + 1 < bnd.size(); ++i) {
      up += (double)(bnd[i + 1] - bnd[i]) * a_row_bytes / rate;
      arrA.push_back(up);
    }
    for (size_t i = 0; i + 1 < bnd.size(); ++i) {
      const int R = bnd[i + 1] - bnd[i];
      if (i == 0 && halves) {
        tc = std::max(tc, std::max(arrA[0], bt)) + T(R, l / 2);
        tc = std::max(tc, bb) + T(R, l / 2);
      } else {
        tc = std::max(tc, std::max(arrA[i], bb)) + T(R, l);
      }
      td = std::max(td, tc) + (double)R * c_row_bytes / rate;
    }
    rb.t_end = td;
  }
  auto slab_plan = [&](const std::vector<std::vector<int>> &gs) {
    HostPlan hp;
    hp.gslabs = gs;
    const int NR = (int)gs.size(), rg = rows / NR;
    double up = 0, tc = 0, td = 0;
    for (int g = 0; g < NR; ++g) {
      const std::vector<int> &ks = gs[g];
      for (size_t si = 0; si < ks.size(); ++si) {
        up += ((double)rg * ks[si] / 8.0 + ((b_resident || g > 0) ? 0.0 : (double)ks[si] * b_row_bytes)) / rate;
        if (si + 1 < ks.size() || g + 1 < NR) {
          tc = std::max(tc, up) + T(rg, ks[si]);
          if (si + 1 == ks.size()) td = std::max(td, tc) + (double)rg * c_row_bytes / rate;
        } else {
          for (int b = 0; b < NB; ++b) {
            tc = std::max(tc, up) + T(rg / NB, ks[si]);
            td = std::max(td, tc) + (double)(rg / NB) * c_row_bytes / rate;
          }
        }
      }
    }
    hp.t_end = td;
    return hp;
  };
  // the candidates, in the numbering of M4RI_HIP_HOST_PLAN (a candidate that does not apply to the shape keeps its number, t_end < 0)
  const bool slabs_ok = l % (8 * 1024) == 0 && l >= 16384 && rows % (NB * 64) == 0;
  const bool two_ok = slabs_ok && rows % (2 * NB * 64) == 0 && rows / 2 >= 8192;
  const bool fine_ok = l % (16 * 1024) == 0;  // sixteenths of the inner dimension stay multiples of 1024 bits
  const std::vector<int> E4{l / 4, l / 4, l / 4, l / 4}, H2{l / 2, l / 2}, W1{l}, G8{l / 8, l / 8, l / 4, l / 2},
      G16{l / 16, l / 16, l / 8, l / 4, l / 4, l / 4};
  HostPlan none;
  none.t_end = -1.0;
  std::vector<HostPlan> cands{rb};
  cands.push_back(slabs_ok ? slab_plan({E4}) : none);                       // 2
  cands.push_back(slabs_ok ? slab_plan({G8}) : none);                       // 3
  cands.push_back(slabs_ok ? slab_plan({H2}) : none);                       // 4
  cands.push_back(two_ok ? slab_plan({E4, E4}) : none);                     // 5
  cands.push_back(two_ok ? slab_plan({H2, H2}) : none);                     // 6
  cands.push_back(two_ok && fine_ok ? slab_plan({G16, H2}) : none);         // 7
  cands.push_back(two_ok && fine_ok ? slab_plan({G16, E4}) : none);         // 8
  cands.push_back(two_ok ? slab_plan({G8, H2}) : none);                     // 9
  cands.push_back(two_ok ? slab_plan({E4, H2}) : none);                     // 10
  cands.push_back(slabs_ok && fine_ok ? slab_plan({G16}) : none);           // 11
  cands.push_back(two_ok && fine_ok ? slab_plan({G16, W1}) : none);         // 12
  static_assert(kHostPlanCandidates == 12, "candidate list");
  if (all_t_end)
    for (size_t i = 0; i < (size_t)kHostPlanCandidates; ++i) all_t_end[i] = i < cands.size() ? cands[i].t_end : -1.0;
  // the fastest slab plan by the model -- within half a percent the LATER candidate wins: two row groups measured 0.2-0.3 ms ahead
  // of their one-group twins where the model has them level (32768^3 7.9 against 8.2 ms, 65536^3 41.0 against 41.4; profiles/
  // r05_host_plan_ab.txt) --, taken if it promises 3 % over the row blocks
  size_t best = 0, bs = 0;
  for (size_t i = 1; i < cands.size(); ++i)
    if (cands[i].t_end > 0 && (!bs || cands[i].t_end <= 1.005 * cands[bs].t_end)) bs = i;
  if (bs && cands[bs].t_end < 0.97 * cands[0].t_end) best = bs;
  if (forced >= 1 && forced <= (int)cands.size() && cands[(size_t)forced - 1].t_end > 0) best = (size_t)forced - 1;
  if (chosen) *chosen = (int)best + 1;
  return cands[best];
}

// Rows [r0, r0 + c.nrows) of a fresh thin product C (r0 a multiple of 64) into their words of every row of the side copy.  The
// transposition kernel stores straight into the pinned host buffer (device-visible like every hipHostMalloc block; 128 KiB for
// 2^20 x 1): no scratch and no second download queued behind C's on the copy engine.  Complete once stream s has been synchronised.
int result_side_rows(ResultSide *side, const gf2_dmat &c, int r0, hipStream_t s) {
  const hipError_t e = gf2k_transpose(reinterpret_cast<u64 *>(side->buf) + r0 / 64, (long long)side->ld, c.data, c.ld, c.nrows, c.ncols, s);
  return e == hipSuccess ? 0 : fail(e, "gf2k_transpose");
}

// One to four vectors against many rows of 65..256 bits: the table-free kernel packs the side copy itself (a ballot per vector and 64
// rows), so product and transposed form are ONE launch.  1 = done, 0 = not this shape (the caller multiplies and transposes), < 0 error.
bool thin_vector_shape(int m, int l, int n) { return n >= 1 && n <= 4 && l > 64 && l <= 256 && m >= 262144; }
int thin_product_with_side(ResultSide *side, u64 *c, long long ldc, const u64 *a, long long lda, const gf2_dmat &b, int m, int l, hipStream_t s) {
  if (!side || !thin_vector_shape(m, l, b.ncols)) return 0;
  const hipError_t e = gf2k_tallskinny_side(a, lda, b.data, b.ld, c, ldc, m, l, b.ncols, reinterpret_cast<u64 *>(side->buf), (long long)side->ld, s);
  if (e == hipErrorNotSupported) return 0;
  return e == hipSuccess ? 1 : fail(e, "gf2k_tallskinny_side");
}

// One device's share of a host product: C[r0:r1, :] (+)= A[r0:r1, :] * B on the CURRENT device, stream s.  Four ways to run it,
// each a function below; host_mul_range picks.
struct HostMulArgs {
  mzd_t *C;
  const mzd_t *A, *B;
  int r0, r1, accumulate, algo, param;
  hipStream_t s;
  ResultSide *side;  // not null: the product is fresh (library-allocated) and thin: leave its packed transposed form here
};

// (a) Slabs of the inner dimension: C (+)= A[G, K_s] B[K_s, :] for every row group G in turn (plan_host_product chose the slab lists); a
// finished group's rows of C leave while the next group is multiplied, the LAST group's last slab runs in four row blocks whose rows
// leave one by one.
int host_mul_slabs(const HostMulArgs &h, const HostPlan &hplan) {
  mzd_t *const C = h.C;
  const mzd_t *const A = h.A, *const B = h.B;
  const int r0 = h.r0, r1 = h.r1, rows = h.r1 - h.r0, accumulate = h.accumulate, algo = h.algo, param = h.param;
  const hipStream_t s = h.s;
  ResultSide *const side = h.side;
  int rc = 0;
  (void)r0, (void)r1, (void)side, (void)accumulate, (void)algo, (void)param;
  // ---- slabs of the inner dimension: C (+)= A[G, K_s] B[K_s, :] for every row group G in turn; a finished group's rows of C leave
  // while the next group is multiplied, the LAST group's last slab runs in four row blocks whose rows leave one by one ----
  const int NR = (int)hplan.gslabs.size(), NBL = 4, RGr = rows / NR;
  int nslabs = 0;
  for (const auto &g : hplan.gslabs) nslabs += (int)g.size();
  SideStream *sd = nullptr;
  rc = side_stream(s, nslabs + NR + NBL, &sd, /*want_s3=*/true);
  DMatOwner dA, dB, dC;
  const bool bcached = (bool)cache_lookup(B);
  if (!rc) rc = to_device(dB, B, sd->s2, bcached);  // a cached B is borrowed (nothing is copied); otherwise allocated here, uploaded by slabs
  if (!rc) rc = to_device_rows(dA, A, r0, r1, s, false);
  if (!rc) rc = to_device_rows(dC, C, r0, r1, s, false);
  if (!rc && (dA.d.ld != A->rowstride || dC.d.ld != C->rowstride || (!bcached && dB.d.ld != B->rowstride)))
    rc = fail_msg("host pipeline: unexpected device stride");
  hipEvent_t *evU = sd ? sd->ev.data() : nullptr, *evC = evU + nslabs;
  auto rows_bytes = [](const mzd_t *M, int nr) { return ((size_t)(nr - 1) * M->rowstride + M->width) * sizeof(word); };
  for (int g = 0, ev = 0; !rc && g < NR; ++g) {  // every upload is queued at once: the next piece travels while this one is multiplied
    int k0 = 0;
    for (size_t si = 0; !rc && si < hplan.gslabs[g].size(); ++si, ++ev) {
      const int ks = hplan.gslabs[g][si];
      if (hipMemcpy2DAsync(dA.d.data + (size_t)g * RGr * dA.d.ld + k0 / 64, (size_t)dA.d.ld * sizeof(u64), A->rows[r0 + g * RGr] + k0 / 64,
                           (size_t)A->rowstride * sizeof(word), (size_t)ks / 8, (size_t)RGr, hipMemcpyHostToDevice, sd->s2) != hipSuccess ||
          (!bcached && g == 0 &&
           hipMemcpyAsync(dB.d.data + (size_t)k0 * dB.d.ld, B->rows[k0], rows_bytes(B, ks), hipMemcpyHostToDevice, sd->s2) != hipSuccess) ||
          hipEventRecord(evU[ev], sd->s2) != hipSuccess)
        rc = fail(hipGetLastError(), "host pipeline: upload of a slab");
      k0 += ks;
    }
  }
  int nev_c = 0;
  auto download = [&](int row0, int nr, const gf2_dmat &c) {
    if (hipEventRecord(evC[nev_c], s) != hipSuccess || hipStreamWaitEvent(sd->s3, evC[nev_c], 0) != hipSuccess ||
        hipMemcpyAsync(C->rows[r0 + row0], c.data, rows_bytes(C, nr), hipMemcpyDeviceToHost, sd->s3) != hipSuccess)
      rc = fail(hipGetLastError(), "host pipeline: download");
    ++nev_c;
  };
  for (int g = 0, ev = 0; !rc && g < NR; ++g) {
    int k0 = 0;
    const int S = (int)hplan.gslabs[g].size();
    for (int si = 0; !rc && si < S; ++si, ++ev) {
      const int ks = hplan.gslabs[g][si];
      if (hipStreamWaitEvent(s, evU[ev], 0) != hipSuccess) rc = fail(hipGetLastError(), "host pipeline: wait");
      gf2_dmat a = dA.d, b = dB.d, c = dC.d;
      a.data += (size_t)g * RGr * a.ld + k0 / 64;
      a.nrows = RGr;
      a.ncols = ks;
      b.data += (size_t)k0 * b.ld;
      b.nrows = ks;
      c.data += (size_t)g * RGr * c.ld;
      c.nrows = RGr;
      if (si + 1 < S || g + 1 < NR) {
        if (!rc) rc = mul_dispatch(&c, &a, &b, si > 0, algo, param, s, /*sync_free=*/false);
        if (!rc && si + 1 == S) download(g * RGr, RGr, c);
      } else {
        for (int bl = 0; !rc && bl < NBL; ++bl) {
          const int R = RGr / NBL;
          gf2_dmat ab = a, cb = c;
          ab.data += (size_t)bl * R * ab.ld;
          ab.nrows = R;
          cb.data += (size_t)bl * R * cb.ld;
          cb.nrows = R;
          rc = mul_dispatch(&cb, &ab, &b, si > 0, algo, param, s, /*sync_free=*/false);
          if (!rc) download(g * RGr + bl * R, R, cb);
        }
      }
      k0 += ks;
    }
  }
  if (sd && hipStreamSynchronize(sd->s2) != hipSuccess && !rc) rc = fail(hipGetLastError(), "host pipeline: upload stream");
  if (sd && sd->s3 && hipStreamSynchronize(sd->s3) != hipSuccess && !rc) rc = fail(hipGetLastError(), "host pipeline: download stream");
  if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail(hipGetLastError(), "host pipeline: compute stream");
  return rc;
}

// (b) Row blocks of A and C (contiguous rows) x halves of the inner dimension for the first block (contiguous rows of B):
//   C_i = A_i[:, 0:l/2] * B[0:l/2, :]  ^  A_i[:, l/2:l] * B[l/2:l, :]
// so that the first product can start after ONE block of A and HALF of B have arrived (7.3 ms of PCIe at n = 65536 instead of 12.2
// with all of B first) and every transfer is one linear copy.  Upload order: A_0, B top, B bottom, A_1, ...  `thin`: the LPN shapes,
// whose kernels stream A at HBM rate (the call IS the upload of A): two blocks, the second short.
int host_mul_row_blocks(const HostMulArgs &h, bool thin, int pipe_blocks) {
  mzd_t *const C = h.C;
  const mzd_t *const A = h.A, *const B = h.B;
  const int r0 = h.r0, r1 = h.r1, rows = h.r1 - h.r0, accumulate = h.accumulate, algo = h.algo, param = h.param;
  const hipStream_t s = h.s;
  ResultSide *const side = h.side;
  int rc = 0;
  (void)r0, (void)r1, (void)side, (void)accumulate, (void)algo, (void)param;
  // Units: row blocks of A and C (contiguous rows) x halves of the inner dimension (contiguous rows of B):
  //   C_i = A_i[:, 0:l/2] * B[0:l/2, :]  ^  A_i[:, l/2:l] * B[l/2:l, :]
  // so that the first product can start after ONE block of A and HALF of B have arrived (7.3 ms of PCIe at n = 65536
  // instead of 12.2 with all of B first) and every transfer is one linear copy.  Upload order: A_0, B top, B bottom, A_1, ...
  const int l = A->ncols;
  const bool bcached = (bool)cache_lookup(B);
  const int K = (!bcached && l >= 8192 && l % 256 == 0 && B->rowstride >= 1 && !(B->flags & mzd_flag_windowed_zerooffset)) ? 2 : 1;
  SideStream *sd = nullptr;
  rc = side_stream(s, 2 * pipe_blocks + 3, &sd, /*want_s3=*/true);
  DMatOwner dA, dB, dC;
  if (!rc) rc = to_device(dB, B, sd->s2, K == 1);  // K == 2: allocated here, uploaded in halves below
  if (!rc) rc = to_device_rows(dA, A, r0, r1, s, false);
  if (!rc) rc = to_device_rows(dC, C, r0, r1, s, false);
  // block boundaries: four equal blocks, or -- when a quarter still has 16384 rows -- a quarter, a half and a quarter: the
  // half-size block multiplies at the rate of the big tiles (32768 x 32768 x 65536: 8.2 ms against 2 x 4.5 ms for two quarters)
  // while the first and the last block stay short (quick start, short tail of the download)
  std::vector<int> bnd;
  {
    const int q = rows / pipe_blocks;
    if (pipe_blocks == 4 && q >= 16384 && !thin) bnd = {0, q, 3 * q, rows};
    // a thin product is nothing but its copies: every block boundary costs ~20 us between two uploads, and what follows the last
    // upload (~100 us of fixed latencies: event -> kernel 26, kernel -> copy 35, the copies' own ~9 each) is exposed -- so two blocks,
    // the second short: 3/16 of the rows put the end of the first block's download where the second block's kernel ends
    // (2^20 x 256 x 1 under the profiler: 793 us with four equal blocks, 763 with these two; unpipelined 800; profiles/r05_av_timeline.txt)
    else if (thin && pipe_blocks == 4) bnd = {0, (int)((long long)rows * 13 / 16) & ~63, rows};
    else
      for (int i = 0; i <= pipe_blocks; ++i) bnd.push_back(i * q);
  }
  const int NBLK = (int)bnd.size() - 1;
  auto rows_bytes = [](const mzd_t *M, int nr) { return ((size_t)(nr - 1) * M->rowstride + M->width) * sizeof(word); };
  if (!rc && (dA.d.ld != A->rowstride || dC.d.ld != C->rowstride || (K == 2 && dB.d.ld != B->rowstride)))
    rc = fail_msg("host pipeline: unexpected device stride");
  hipEvent_t *evA = sd ? sd->ev.data() : nullptr, *evC = evA + pipe_blocks, *evB = evC + pipe_blocks;
  auto upload_a = [&](int i) {
    if (hipMemcpyAsync(dA.d.data + (size_t)bnd[i] * dA.d.ld, A->rows[r0 + bnd[i]], rows_bytes(A, bnd[i + 1] - bnd[i]), hipMemcpyHostToDevice,
                       sd->s2) != hipSuccess ||
        hipEventRecord(evA[i], sd->s2) != hipSuccess)
      rc = fail(hipGetLastError(), "host pipeline: upload of A");
  };
  if (!rc) upload_a(0);
  for (int k = 0; !rc && k < K; ++k) {  // K == 1: B went up whole above (or lives in the operand cache)
    if (K == 2 && hipMemcpyAsync(dB.d.data + (size_t)k * (l / 2) * dB.d.ld, B->rows[k * (l / 2)], rows_bytes(B, l / 2), hipMemcpyHostToDevice,
                                 sd->s2) != hipSuccess)
      rc = fail(hipGetLastError(), "host pipeline: upload of B");
    if (!rc && hipEventRecord(evB[k], sd->s2) != hipSuccess) rc = fail(hipGetLastError(), "host pipeline: event");
  }
  for (int i = 1; !rc && i < NBLK; ++i) upload_a(i);
  for (int i = 0; !rc && i < NBLK; ++i) {
    const int R = bnd[i + 1] - bnd[i];
    gf2_dmat c = dC.d;
    c.data += (size_t)bnd[i] * c.ld;
    c.nrows = R;
    if (hipStreamWaitEvent(s, evA[i], 0) != hipSuccess) rc = fail(hipGetLastError(), "host pipeline: wait");
    // only the FIRST block is multiplied in halves of the inner dimension (it starts while the bottom half of B is still on the
    // wire); by the time a later block has arrived all of B is resident, and one product over the whole inner dimension is the
    // more efficient launch (65536^3: 2 x 16384 x 32768 x 65536 take 9.2 ms, 16384 x 65536 x 65536 takes 8.2)
    const int Ki = i == 0 ? K : 1;
    for (int k = 0; !rc && k < Ki; ++k) {
      gf2_dmat a = dA.d, b = dB.d;
      a.data += (size_t)bnd[i] * a.ld + (size_t)k * (l / Ki) / 64;
      a.nrows = R;
      a.ncols = l / Ki;
      b.data += (size_t)k * (l / Ki) * b.ld;
      b.nrows = l / Ki;
      for (int kk = (Ki == 1 ? 0 : k); !rc && kk < (Ki == 1 ? K : k + 1); ++kk)
        if (hipStreamWaitEvent(s, evB[kk], 0) != hipSuccess) rc = fail(hipGetLastError(), "host pipeline: wait");
      if (!rc) rc = mul_dispatch(&c, &a, &b, k > 0, algo, param, s, /*sync_free=*/false);
    }
    if (!rc && (hipEventRecord(evC[i], s) != hipSuccess || hipStreamWaitEvent(sd->s3, evC[i], 0) != hipSuccess ||
                hipMemcpyAsync(C->rows[r0 + bnd[i]], c.data, rows_bytes(C, R), hipMemcpyDeviceToHost, sd->s3) != hipSuccess))
      rc = fail(hipGetLastError(), "host pipeline: download");
    if (!rc && side) rc = result_side_rows(side, c, bnd[i], s);  // beside the block's download
  }
  if (sd && hipStreamSynchronize(sd->s2) != hipSuccess && !rc) rc = fail(hipGetLastError(), "host pipeline: upload stream");
  if (sd && sd->s3 && hipStreamSynchronize(sd->s3) != hipSuccess && !rc) rc = fail(hipGetLastError(), "host pipeline: download stream");
  if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail(hipGetLastError(), "host pipeline: compute stream");
  return rc;
}

// (c) Everything up, one product, everything down (small products, windows, accumulating calls, operands the pipelines do not take).
int host_mul_plain(const HostMulArgs &h) {
  mzd_t *const C = h.C;
  const mzd_t *const A = h.A, *const B = h.B;
  const int r0 = h.r0, r1 = h.r1, rows = h.r1 - h.r0, accumulate = h.accumulate, algo = h.algo, param = h.param;
  const hipStream_t s = h.s;
  ResultSide *const side = h.side;
  int rc = 0;
  (void)r0, (void)r1, (void)side, (void)accumulate, (void)algo, (void)param;
  DMatOwner dA, dB, dC;
  rc = to_device_rows(dA, A, r0, r1, s, true);
  if (!rc) rc = to_device(dB, B, s, true);
  if (!rc) rc = to_device_rows(dC, C, r0, r1, s, accumulate != 0);
  // (thin products: no wait between the kernel and the download -- the download's own launch latency would be exposed behind it)
  int fused = 0;  // product and side copy in one launch (one to four vectors)
  if (!rc && side && !accumulate) {
    fused = thin_product_with_side(side, dC.d.data, dC.d.ld, dA.d.data, dA.d.ld, dB.d, rows, A->ncols, s);
    if (fused < 0) rc = fused;
  }
  if (!rc && fused != 1) rc = mul_dispatch(&dC.d, &dA.d, &dB.d, accumulate, algo, param, s, /*sync_free=*/!(side || B->ncols <= 64));
  if (!rc && side && fused == 1) {
    rc = download_rows(C, r0, &dC.d, s);  // (syncs: the side copy is complete with it)
  } else if (!rc && side) {
    // C comes down on the download stream while the compute stream transposes it and brings the small form down
    SideStream *sd = nullptr;
    rc = side_stream(s, 1, &sd, /*want_s3=*/true);
    if (!rc && (hipEventRecord(sd->ev[0], s) != hipSuccess || hipStreamWaitEvent(sd->s3, sd->ev[0], 0) != hipSuccess ||
                hipMemcpyAsync(C->rows[r0], dC.d.data, ((size_t)(rows - 1) * C->rowstride + C->width) * sizeof(word), hipMemcpyDeviceToHost,
                               sd->s3) != hipSuccess))
      rc = fail(hipGetLastError(), "thin product: download");
    if (!rc) rc = result_side_rows(side, dC.d, 0, s);
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail(hipGetLastError(), "thin product: compute stream");
    if (sd && sd->s3 && hipStreamSynchronize(sd->s3) != hipSuccess && !rc) rc = fail(hipGetLastError(), "thin product: download stream");
  } else if (!rc) rc = download_rows(C, r0, &dC.d, s);
  if (rc) (void)hipStreamSynchronize(s);
  return rc;
}

// (d) `&A * &v` with A on the host (round 5): the vector kernel reads A from, and writes C and the side copy into, the PINNED host
// blocks themselves -- the launch IS the transfer (32 MiB in at the rate a kernel pulls over PCIe, 8 MiB out beside it), with no copy
// queue between its pieces: 2^20 x 256 x 1 0.69 ms for the product against 0.77 through uploads, kernel and downloads in two row
// blocks (profiles/r05_zero_copy_probe.txt; the side copy is what made it worth having: it used to need C on the device).
// 1 = done, 0 = not this case, < 0 error.
int host_mul_zero_copy(const HostMulArgs &h) {
  mzd_t *const C = h.C;
  const mzd_t *const A = h.A, *const B = h.B;
  const int r0 = h.r0, r1 = h.r1, rows = h.r1 - h.r0, accumulate = h.accumulate, algo = h.algo, param = h.param;
  const hipStream_t s = h.s;
  ResultSide *const side = h.side;
  int rc = 0;
  (void)r0, (void)r1, (void)side, (void)accumulate, (void)algo, (void)param;
  DMatOwner dB;
  rc = to_device(dB, B, s, true);
  int done = 0;
  if (!rc) done = thin_product_with_side(side, C->rows[0], C->rowstride, A->rows[0], A->rowstride, dB.d, rows, A->ncols, s);
  if (done < 0) rc = done;
  if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail(hipGetLastError(), "thin product: stream");
  return rc ? rc : done;
}

// Returns when rows [r0, r1) of C are complete in host memory.
int host_mul_range(mzd_t *C, const mzd_t *A, const mzd_t *B, int r0, int r1, int accumulate, int algo, int param, hipStream_t s,
                   ResultSide *side = nullptr) {
  const int rows = r1 - r0;
  if (rows <= 0) return 0;
  const HostMulArgs h{C, A, B, r0, r1, accumulate, algo, param, s, side};
  // Large products are pipelined: an upload stream brings the operands in, a download stream takes C out (PCIe is full duplex: the two
  // directions get a stream each), the compute stream multiplies a piece as soon as it has arrived -- PCIe is most of a host call:
  // 1.5 GiB at n = 65536.
  static const int pipe_blocks = env_int("M4RI_HIP_HOST_PIPELINE_BLOCKS", 4);  // (2 blocks at 32768 / 16384 rows measured slower: 9.7 / 2.1 against 9.0 / 1.9 ms)
  const bool plain_layout = !(A->flags & mzd_flag_windowed_zerooffset) && !(C->flags & mzd_flag_windowed_zerooffset) &&
                            A->rowstride >= 1 && C->rowstride >= 1;
  const bool whole = r0 == 0 && r1 == A->nrows;
  // (Round 4 measured a 2 x 2 plan for mid-sized products -- the four quadrants of C as units, A in two row blocks, B in two
  // column panels by 2-D copies, upload order A_0, B_0, B_1, A_1 -- against the row blocks at 32768^3: 8.6 against 8.8 ms.
  // The 2-D copies run at the linear rate (hipMemcpy2DAsync, 2 KiB rows: 54 GB/s), but a 16384 x 32768 x 16384 quadrant takes
  // 1.33-1.40 ms -- 49 leaves = 392 tiles = 1.5 rounds, run as a whole round plus a tail launch -- where a quarter of the whole
  // product's time would be 1.1: four of them are 5.5 ms of device work behind the 2.4 ms the first two pieces take to arrive.
  // Not kept: profiles/r04_host_path_timeline.txt.)
  static const int zero_copy = dev_env_int("M4RI_HIP_THIN_ZERO_COPY", 1);
  if (zero_copy && side && whole && !accumulate && plain_layout && thin_vector_shape(rows, A->ncols, B->ncols) && !cache_lookup(A) &&
      gf2_mzd_block_is_pinned(A) && gf2_mzd_block_is_pinned(C) && (size_t)rows * A->rowstride * sizeof(word) >= ((size_t)8 << 20)) {
    const int done = host_mul_zero_copy(h);
    if (done != 0) return done < 0 ? done : 0;
  }
  // thin products (the LPN shape, 2^20 x 256 times a few vectors) are pipelined too
  const bool thin = B->ncols <= 256 && A->ncols <= 1024 && (size_t)rows * A->rowstride * sizeof(word) >= ((size_t)8 << 20);
  const bool pipelined = pipe_blocks >= 2 && !accumulate && plain_layout && rows >= 16384 && rows % (pipe_blocks * 64) == 0 &&
                         !(whole && cache_lookup(A));
  const bool big = (long long)A->ncols * B->ncols >= (1ll << 28);
  if (pipelined && big && pipe_blocks == 4 && B->rowstride >= 1 && !(B->flags & mzd_flag_windowed_zerooffset)) {
    const HostPlan hplan = plan_host_product(rows, A->ncols, B->ncols, algo, param, (bool)cache_lookup(B), (size_t)A->rowstride * sizeof(word),
                                             (size_t)B->rowstride * sizeof(word), (size_t)C->rowstride * sizeof(word));
    if (!hplan.gslabs.empty()) return host_mul_slabs(h, hplan);
  }
  if (pipelined && (big || thin)) return host_mul_row_blocks(h, thin, pipe_blocks);
  return host_mul_plain(h);
}

// Devices a host product of this shape is spread over.  M4RI_HIP_DEVICES: unset = the current device only (the fan-out is
// opt-in: it has not been measured on a multi-GPU box yet, and under torchrun every rank sees every GPU); "auto" = every
// visible device once the product is large enough to pay for a copy of B per device -- ignored inside a torch.distributed
// job (WORLD_SIZE > 1), where the ranks already own a device each; "all"; or a comma-separated list of device ordinals
// (an ordinal may repeat: two shares on one device -- how the one-GPU test box exercises this path; ONE ordinal pins every
// host entry point -- products, elimination, transpose, operand cache -- to that device).
std::vector<int> parse_device_list(const char *e) {
  std::vector<int> out;
  const int nvis = gf2_device_count();
  for (const char *p = e; p && *p;) {
    char *end = nullptr;
    const long d = std::strtol(p, &end, 10);
    if (end == p) break;
    if (d >= 0 && d < nvis) out.push_back((int)d);
    p = (*end == ',') ? end + 1 : end;
    if (*end && *end != ',') break;
  }
  return out;
}

std::vector<int> pick_devices(long long m, long long l, long long n) {
  const char *e = std::getenv("M4RI_HIP_DEVICES");
  std::vector<int> out;
  if (!e || !*e) return out;
  const int nvis = gf2_device_count();
  const bool autom = std::strcmp(e, "auto") == 0, all = std::strcmp(e, "all") == 0;
  if (!autom && !all) return parse_device_list(e);
  if (autom) {
    const char *ws = std::getenv("WORLD_SIZE");
    if (ws && std::atoi(ws) > 1) return out;  // one process per GPU already
  }
  // automatic: a share should keep >= 4096 rows (tall tiles) and the product should outweigh moving B once more per device
  if (nvis > 1 && (all || (2.0 * (double)m * (double)l * (double)n >= 7.0e13 && m >= 8192))) {
    int k = nvis;
    while (k > 1 && m / k < 4096 && !all) --k;
    for (int d = 0; d < k; ++d) out.push_back(d);
  }
  return out;
}

// M4RI_HIP_DEVICES = one ordinal: every host entry point runs on that device.  RAII: sets it, restores the caller's.
struct PinnedDevice {
  int prev = -1;
  bool switched = false;
  PinnedDevice() {
    const char *e = std::getenv("M4RI_HIP_DEVICES");
    if (!e || !*e || std::strcmp(e, "auto") == 0 || std::strcmp(e, "all") == 0) return;
    const std::vector<int> d = parse_device_list(e);
    if (d.size() != 1) return;
    if (hipGetDevice(&prev) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    if (prev != d[0] && hipSetDevice(d[0]) == hipSuccess) switched = true;
  }
  ~PinnedDevice() {
    if (switched) (void)hipSetDevice(prev);
  }
  PinnedDevice(const PinnedDevice &) = delete;
  PinnedDevice &operator=(const PinnedDevice &) = delete;
};

// C (+)= A*B with the rows of A and C divided among `devs` (one worker thread per share; each uploads its rows of A and
// its own copy of B over its own PCIe link, multiplies, and downloads its rows of C).  Shares are independent: the inner
// dimension is never split, so there is no reduction.
int host_mul_multi(mzd_t *C, const mzd_t *A, const mzd_t *B, int accumulate, int algo, int param, const std::vector<int> &devs) {
  const int k = (int)devs.size(), m = A->nrows;
  // shares: equal, rounded up to a multiple of 1024 rows (whole tiles / Strassen-divisible blocks), the last one takes the rest
  long long per = ((long long)m + k - 1) / k;
  per = (per + 1023) / 1024 * 1024;
  std::vector<int> rcs(k, 0);
  std::vector<std::string> errs(k);
  std::vector<std::thread> th;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int t = 0; t < k; ++t) {
    const int r0 = (int)std::min<long long>(m, per * t), r1 = (int)std::min<long long>(m, per * (t + 1));
    if (r1 <= r0) continue;
    th.emplace_back([&, t, r0, r1] {
      const int dev = devs[t];
      if (hipSetDevice(dev) != hipSuccess) {
        rcs[t] = fail(hipGetLastError(), "hipSetDevice");
        errs[t] = gf2_last_error();
        return;
      }
      hipStream_t s = nullptr;
      rcs[t] = lease_stream(dev, &s);
      if (!rcs[t]) {
        rcs[t] = host_mul_range(C, A, B, r0, r1, accumulate, algo, param, s);
        if (rcs[t]) (void)hipStreamSynchronize(s);
        unlease_stream(dev, s);
      }
      if (rcs[t]) errs[t] = gf2_last_error();
    });
  }
  for (auto &x : th) x.join();
  (void)hipSetDevice(cur);
  for (int t = 0; t < k; ++t)
    if (rcs[t]) {
      tls_error = "device " + std::to_string(devs[t]) + ": " + errs[t];
      return rcs[t];
    }
  return 0;
}

mzd_t *host_mul_on(mzd_t *C, const mzd_t *A, const mzd_t *B, int accumulate, int algo, int param, const char *name,
                   const int *devices, int ndev) {
  if (A->ncols != B->nrows) gf2_die((std::string(name) + ": A ncols need to match B nrows.").c_str());
  if (accumulate && !C) gf2_die((std::string(name) + ": C must not be NULL.").c_str());
  const bool allocated = (C == nullptr);
  if (C && (C->nrows != A->nrows || C->ncols != B->ncols))
    gf2_die((std::string(name) + ": C (ret) has wrong dimensions.").c_str());
  auto bail = [&](const char *why) -> mzd_t * {
    std::fprintf(stderr, "m4ri_hip: %s failed: %s (%s)\n", name, why, gf2_last_error());
    return nullptr;
  };
  if (require_device()) return bail("no device");
  // size dispatch: a product of a few thousand word operations is done on the host by the time a
  // device call would have uploaded its operands (gf2_small_host.cpp); operands the caller pinned to the device stay there
  const bool small = !devices && gf2_small_product(A->nrows, A->ncols, B->ncols) && !cache_lookup(A) && !cache_lookup(B);
  if (!C) C = (small || A->nrows == 0 || B->ncols == 0) ? mzd_init(A->nrows, B->ncols) : gf2_mzd_init_uncleared(A->nrows, B->ncols);
  else gf2_cache_forget(C);  // about to be overwritten
  if (A->nrows == 0 || B->ncols == 0) return C;
  if (small) {
    if (gf2_mul_host_small(C, A, B, accumulate) == 0) return C;
    if (allocated) mzd_free(C);
    return bail("host product");
  }
  std::vector<int> devs;
  if (devices) {
    for (int i = 0; i < ndev; ++i) {
      if (devices[i] < 0 || devices[i] >= gf2_device_count()) {
        if (allocated) mzd_free(C);
        fail_msg("device ordinal out of range");
        return bail("device list");
      }
      devs.push_back(devices[i]);
    }
  } else {
    devs = pick_devices(A->nrows, A->ncols, B->ncols);
  }
  const bool windows = ((A->flags | C->flags) & mzd_flag_windowed_zerooffset) != 0;
  int rc;
  if (devs.size() > 1 && !windows) {
    rc = host_mul_multi(C, A, B, accumulate, algo, param, devs);
  } else {
    int cur = 0, want = devs.size() == 1 ? devs[0] : -1;
    if (want >= 0 && (hipGetDevice(&cur) != hipSuccess || hipSetDevice(want) != hipSuccess)) want = -1;
    hipStream_t s;
    if (get_private_stream(&s)) {
      if (allocated) mzd_free(C);
      return bail("stream");
    }
    // a fresh thin product also comes back in its packed transposed form (see ResultSide)
    static const int side_cols = env_int("M4RI_HIP_RESULT_SIDE_COLS", 8);
    std::shared_ptr<ResultSide> side;
    if (allocated && !windows && C->ncols <= side_cols && C->blocks && C->blocks[0].size >= ((size_t)1 << 20)) {
      side = std::make_shared<ResultSide>();
      side->ld = ((size_t)C->nrows + 63) / 64;
      side->bytes = (size_t)C->ncols * side->ld * sizeof(word);
      side->buf = static_cast<word *>(gf2_pinned_alloc(side->bytes));
      if (!side->buf) side.reset();  // pinning failed: the product does not depend on it
    }
    rc = host_mul_range(C, A, B, 0, A->nrows, accumulate, algo, param, s, side.get());
    if (want >= 0) (void)hipSetDevice(cur);
    if (!rc && side) {
      side->nrows = C->nrows;
      side->ncols = C->ncols;
      side->rowstride = C->rowstride;
      side->row0 = C->rows[0];
      std::lock_guard<std::mutex> lk(g_cache_mu);
      g_result_side[cache_key(C)] = std::move(side);
    }
  }
  if (rc) {
    if (allocated) mzd_free(C);
    return bail("device product");
  }
  return C;
}

mzd_t *host_mul(mzd_t *C, const mzd_t *A, const mzd_t *B, int accumulate, int algo, int param, const char *name) {
  return host_mul_on(C, A, B, accumulate, algo, param, name, nullptr, 0);
}
}  // namespace

// One process, several devices: C = A*B (C NULL: allocated) with the rows of A and C divided among `devices` (ordinals of
// hipGetDeviceCount's numbering; an ordinal may repeat).  mzd_mul / mzd_mul_m4rm / mzd_mul_naive do the same by
// themselves for large products when more than one device is visible (M4RI_HIP_DEVICES).
extern "C" mzd_t *gf2_mul_multi(mzd_t *C, mzd_t const *A, mzd_t const *B, int algo, int param, const int *devices, int ndev) {
  if (!A || !B || !devices || ndev < 1) {
    fail_msg("gf2_mul_multi: bad arguments");
    return nullptr;
  }
  return host_mul_on(C, A, B, 0, algo, param, "gf2_mul_multi", devices, ndev);
}

void gf2_cache_forget(mzd_t const *M) {
  std::shared_ptr<CachedOperand> c;
  std::shared_ptr<ResultSide> sd;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (!g_result_side.empty()) {
      auto it = g_result_side.find(cache_key(M));
      if (it != g_result_side.end()) {
        sd = std::move(it->second);
        g_result_side.erase(it);
      }
    }
    if (g_cache.empty()) return;
    auto it = g_cache.find(cache_key(M));
    if (it == g_cache.end()) return;
    c = std::move(it->second);
    g_cache.erase(it);
  }
  // products that looked the copy up hold their own reference until their stream has drained; the block goes back to the
  // pool when the last reference is dropped (here, if nobody is using it)
}

extern "C" int gf2_mzd_cache_on_device(mzd_t const *M) {
  if (int rc = require_device()) return rc;
  if (!M || M->nrows == 0 || M->ncols == 0) return fail_msg("gf2_mzd_cache_on_device: empty matrix");
  gf2_cache_forget(M);
  PinnedDevice pin;  // M4RI_HIP_DEVICES = one ordinal: run there
  hipStream_t s;
  if (int rc = get_private_stream(&s)) return rc;
  DMatOwner o;
  if (int rc = to_device(o, M, s, true)) return rc;
  HIP_TRY(hipStreamSynchronize(s));
  auto c = std::make_shared<CachedOperand>();
  HIP_TRY(hipGetDevice(&c->dev));
  c->d = o.d;
  c->nrows = M->nrows;
  c->ncols = M->ncols;
  c->rowstride = M->rowstride;
  c->row0 = M->rows[0];
  o.released = true;  // ownership moves to the cache
  std::lock_guard<std::mutex> lk(g_cache_mu);
  g_cache[cache_key(M)] = std::move(c);
  return 0;
}

extern "C" void gf2_mzd_uncache(mzd_t const *M) { gf2_cache_forget(M); }

// The schedules a large product on HOST matrices can take, as the library's time model plays them through (plan_host_product):
// t_end[i] = modelled seconds of schedule i + 1 in M4RI_HIP_HOST_PLAN's numbering (-1: not applicable to this shape); returns the
// number of the schedule the host path takes (0: the product is not pipelined at all).  Plain row-major operands of the natural strides.
extern "C" int gf2_host_plan_model(int m, int l, int n, int algo, int param, double t_end[12]) {
  for (int i = 0; i < kHostPlanCandidates; ++i) t_end[i] = -1.0;
  if (m < 16384 || m % 256 || (long long)l * n < (1ll << 28)) return 0;
  auto stride = [](int c) { const size_t w = (size_t)(c + 63) / 64; return (w < 3 || (w & 1) == 0 ? w : w + 1) * sizeof(word); };
  int chosen = 0;
  (void)plan_host_product(m, l, n, algo, param, false, stride(l), stride(n), stride(n), t_end, &chosen);
  return chosen;
}

// mzd_transpose(DST, A) from the side copy of A, if A is a fresh thin product that still has one: DST (allocated when NULL) or nullptr.
mzd_t *gf2_transpose_from_side_copy(mzd_t *DST, mzd_t const *A) {
  std::shared_ptr<ResultSide> sd;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    if (g_result_side.empty()) return nullptr;
    auto it = g_result_side.find(cache_key(A));
    if (it == g_result_side.end()) return nullptr;
    sd = it->second;
  }
  // the same view of the block (a window of the product is a different matrix)
  if (sd->nrows != A->nrows || sd->ncols != A->ncols || sd->rowstride != A->rowstride || sd->row0 != A->rows[0] ||
      (A->flags & mzd_flag_windowed_zerooffset))
    return nullptr;
  const bool fresh = DST == nullptr;
  if (fresh) DST = gf2_mzd_init_uncleared(A->ncols, A->nrows);
  const wi_t w = DST->width;
  for (rci_t i = 0; i < DST->nrows; ++i) {
    word *d = DST->rows[i];
    const word *t = sd->buf + (size_t)i * sd->ld;
    if (w > 1) std::memcpy(d, t, (size_t)(w - 1) * sizeof(word));
    d[w - 1] = fresh ? (t[w - 1] & DST->high_bitmask) : ((d[w - 1] & ~DST->high_bitmask) | (t[w - 1] & DST->high_bitmask));
  }
  return DST;
}

int gf2_host_transpose_gpu(mzd_t *dst, mzd_t const *src) {
  if (require_device()) return -1;
  PinnedDevice pin;  // M4RI_HIP_DEVICES = one ordinal: run there
  hipStream_t s;
  if (get_private_stream(&s)) return -1;
  DMatOwner dS, dD;
  int rc = 0;
  // Large matrices in four or eight row blocks of the source: block i goes up on one copy stream, is transposed into ITS words of every row
  // of the destination, and those words come down through a 2-D copy on the other copy stream while block i + 1 goes up (PCIe is
  // full duplex; 2-D copies of 2-KiB pieces run at the linear rate, see host_mul_range).  65536^2: 19.3 -> 12.3 ms (9.4 ms per direction).
  static const int pipe_on = dev_env_int("M4RI_HIP_TRANSPOSE_PIPELINE", 1);  // 0 off, 1 by rule, n >= 2: n blocks (A/B)
  // (65536^2 on one box: 2 / 4 / 8 / 16 blocks 15.1 / 13.2 / 12.3 / 16.7 ms, unpipelined 19.3; 32768 x 65536: 4 blocks 6.6, 8 blocks 8.3 --
  // the 2-D copy wants pieces of at least 1 KiB)
  const int NB = pipe_on >= 2 ? pipe_on : (src->nrows >= 65536 ? 8 : 4);
  const bool plain = !(src->flags & mzd_flag_windowed_zerooffset) && !(dst->flags & mzd_flag_windowed_zerooffset) &&
                     src->rowstride >= 1 && dst->rowstride >= 1 && src->nrows > 0 && src->ncols > 0 &&
                     src->rows[src->nrows - 1] == src->rows[0] + (size_t)(src->nrows - 1) * src->rowstride &&
                     dst->rows[dst->nrows - 1] == dst->rows[0] + (size_t)(dst->nrows - 1) * dst->rowstride;
  if (pipe_on && plain && src->nrows % (NB * 512) == 0 && src->nrows / NB >= (pipe_on >= 2 ? 2048 : 8192) /* pieces of >= 1 KiB in the 2-D copies */ &&
      (long long)src->nrows * src->ncols >= (1ll << 31) && !cache_lookup(src)) {
    SideStream *side = nullptr;
    rc = side_stream(s, 2 * NB, &side, /*want_s3=*/true);
    if (!rc) rc = to_device(dS, src, s, false);
    if (!rc) rc = to_device(dD, dst, s, false);
    if (!rc && (dS.d.ld != src->rowstride || dD.d.ld != dst->rowstride)) rc = fail_msg("transpose pipeline: unexpected device stride");
    const int R = src->nrows / NB;
    hipEvent_t *evU = side ? side->ev.data() : nullptr, *evT = evU + NB;
    for (int i = 0; !rc && i < NB; ++i) {
      const size_t up = ((size_t)(R - 1) * src->rowstride + src->width) * sizeof(word);
      if (hipMemcpyAsync(dS.d.data + (size_t)i * R * dS.d.ld, src->rows[(size_t)i * R], up, hipMemcpyHostToDevice, side->s2) != hipSuccess ||
          hipEventRecord(evU[i], side->s2) != hipSuccess || hipStreamWaitEvent(s, evU[i], 0) != hipSuccess) {
        rc = fail(hipGetLastError(), "transpose pipeline: upload");
        break;
      }
      const hipError_t e = gf2k_transpose(dD.d.data + (size_t)i * R / 64, dD.d.ld, dS.d.data + (size_t)i * R * dS.d.ld, dS.d.ld, R, src->ncols, s);
      if (e != hipSuccess) {
        rc = fail(e, "gf2k_transpose");
        break;
      }
      if (hipEventRecord(evT[i], s) != hipSuccess || hipStreamWaitEvent(side->s3, evT[i], 0) != hipSuccess ||
          hipMemcpy2DAsync(dst->rows[0] + (size_t)i * R / 64, (size_t)dst->rowstride * sizeof(word), dD.d.data + (size_t)i * R / 64,
                           (size_t)dD.d.ld * sizeof(u64), (size_t)R / 8, (size_t)dst->nrows, hipMemcpyDeviceToHost, side->s3) != hipSuccess)
        rc = fail(hipGetLastError(), "transpose pipeline: download");
    }
    if (side && hipStreamSynchronize(side->s2) != hipSuccess && !rc) rc = fail(hipGetLastError(), "transpose pipeline: upload stream");
    if (hipStreamSynchronize(s) != hipSuccess && !rc) rc = fail(hipGetLastError(), "transpose pipeline: compute stream");
    if (side && side->s3 && hipStreamSynchronize(side->s3) != hipSuccess && !rc) rc = fail(hipGetLastError(), "transpose pipeline: download stream");
    return rc;
  }
  rc = to_device(dS, src, s, true);
  if (!rc) rc = to_device(dD, dst, s, false);
  if (!rc) {
    hipError_t e = gf2k_transpose(dD.d.data, dD.d.ld, dS.d.data, dS.d.ld, src->nrows, src->ncols, s);
    if (e != hipSuccess) rc = fail(e, "gf2k_transpose");
  }
  if (!rc) rc = gf2_dmat_download(dst, &dD.d, s);
  if (rc) (void)hipStreamSynchronize(s);
  return rc;
}

// Strassen levels from M4RI's cutoff argument: recursion continues while the halved dimension stays
// >= cutoff (strassen.rs:8-18: "Minimal dimension for Strassen recursion"); 0 = library default.
static int levels_from_cutoff(const mzd_t *A, const mzd_t *B, int cutoff) {
  if (cutoff <= 0) return 0;  // automatic
  const int mn = A->nrows < A->ncols ? (A->nrows < B->ncols ? A->nrows : B->ncols)
                                     : (A->ncols < B->ncols ? A->ncols : B->ncols);
  int L = 0;
  while (L < 6 && (mn >> (L + 1)) >= cutoff) ++L;
  return L ? L : -1;  // -1: explicit "no recursion"
}

extern "C" mzd_t *mzd_mul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k) {
  (void)k;  // table size hint; the kernel's tables are fixed at 8 bits (LDS bank-row geometry)
  return host_mul(C, A, B, 0, GF2_ALGO_M4RM, 0, "mzd_mul_m4rm");
}
extern "C" mzd_t *mzd_addmul_m4rm(mzd_t *C, mzd_t const *A, mzd_t const *B, int k) {
  (void)k;
  return host_mul(C, A, B, 1, GF2_ALGO_M4RM, 0, "mzd_addmul_m4rm");
}
extern "C" mzd_t *mzd_mul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) {
  const int L = levels_from_cutoff(A, B, cutoff);
  if (L < 0) return host_mul(C, A, B, 0, GF2_ALGO_M4RM, 0, "mzd_mul");
  return host_mul(C, A, B, 0, GF2_ALGO_STRASSEN, L, "mzd_mul");
}
extern "C" mzd_t *mzd_addmul(mzd_t *C, mzd_t const *A, mzd_t const *B, int cutoff) {
  const int L = levels_from_cutoff(A, B, cutoff);
  if (L < 0) return host_mul(C, A, B, 1, GF2_ALGO_M4RM, 0, "mzd_addmul");
  return host_mul(C, A, B, 1, GF2_ALGO_STRASSEN, L, "mzd_addmul");
}
extern "C" mzd_t *mzd_mul_naive(mzd_t *C, mzd_t const *A, mzd_t const *B) {
  return host_mul(C, A, B, 0, GF2_ALGO_NAIVE, 0, "mzd_mul_naive");
}
extern "C" mzd_t *mzd_addmul_naive(mzd_t *C, mzd_t const *A, mzd_t const *B) {
  return host_mul(C, A, B, 1, GF2_ALGO_NAIVE, 0, "mzd_addmul_naive");
}

extern "C" mzd_t *_mzd_mul_naive(mzd_t *C, mzd_t const *A, mzd_t const *Bt, int clear) {
  // C (+)= A * Bt^T, Bt pre-transposed (mzd.rs:154-168); C is "preallocated" upstream
  if (!C) gf2_die("_mzd_mul_naive: C must be preallocated.");
  if (A->ncols != Bt->ncols || C->nrows != A->nrows || C->ncols != Bt->nrows)
    gf2_die("_mzd_mul_naive: dimension mismatch.");
  if (require_device()) {
    std::fprintf(stderr, "m4ri_hip: _mzd_mul_naive failed: %s\n", gf2_last_error());
    return nullptr;
  }
  if (A->nrows == 0 || Bt->nrows == 0) return C;
  gf2_cache_forget(C);  // about to be overwritten
  {
    const long long lim = gf2_small_work_limit();
    if (lim > 0 && (long long)A->nrows * Bt->nrows * A->width <= lim && !cache_lookup(A) && !cache_lookup(Bt))
      return gf2_mul_nt_host_small(C, A, Bt, clear == 0) == 0 ? C : nullptr;  // size dispatch, see host_mul_on
  }
  PinnedDevice pin;  // M4RI_HIP_DEVICES = one ordinal: run there
  hipStream_t s;
  if (get_private_stream(&s)) return nullptr;
  int rc;
  {
    DMatOwner dA, dB, dC;
    rc = to_device(dA, A, s, true);
    if (!rc) rc = to_device(dB, Bt, s, true);
    if (!rc) rc = to_device(dC, C, s, clear == 0);
    if (!rc) rc = gf2_mul_nt_dev(&dC.d, &dA.d, &dB.d, clear == 0, s);
    if (!rc) rc = gf2_dmat_download(C, &dC.d, s);
    if (rc) (void)hipStreamSynchronize(s);
  }
  if (rc) {
    std::fprintf(stderr, "m4ri_hip: _mzd_mul_naive failed: %s\n", gf2_last_error());
    return nullptr;
  }
  return C;
}

extern "C" mzd_t *_mzd_mul_va(mzd_t *C, mzd_t const *v, mzd_t const *A, int clear) {
  if (!C) gf2_die("_mzd_mul_va: C must be preallocated.");
  return host_mul(C, v, A, clear == 0, GF2_ALGO_M4RM, 0, "_mzd_mul_va");
}

// ---------------------------------------------------------------------------------------------
// elimination: echelon forms, inverse, linear systems (kernels in gf2_elim.hip)
//   mzd_echelonize / _m4ri / _pluq   m4ri-sys/src/echelonform.rs:16-37, caller binary_matrix.rs:258-261
//   mzd_inv_m4ri                     m4ri-sys/src/brilliantrussian.rs:201-208, caller binary_matrix.rs:265-268
//   mzd_solve_left                   m4ri-sys/src/solve.rs:12-29, caller binary_matrix.rs:582-586
// ---------------------------------------------------------------------------------------------

namespace {
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int alloc(size_t b) {
    bytes = b ? b : 8;
    return dev_alloc(&p, bytes);
  }
  ~DevBuf() { dev_free(p, bytes); }
  template <class T>
  T *as() const { return static_cast<T *>(p); }
};

// In-place echelon form of the first `col_limit` columns of A (0 = all); row operations act on whole rows, so the
// columns beyond the limit carry an augmented right-hand side along.  full != 0: reduced row echelon form (unique);
// full == 0: pivot rows are only cleared below their column block.  Synchronous.  pivcols_dev (optional) receives a
// device array of the pivot columns that stays valid until `keep` is destroyed.
int echelonize_dev(gf2_dmat *A, int full, int col_limit, int *rank_out, int *pivcols_host, DevBuf *pivcols_keep,
                   hipStream_t s) {
  const int m = A->nrows, ncols = A->ncols;
  const int limit = (col_limit > 0 && col_limit < ncols) ? col_limit : ncols;
  *rank_out = 0;
  if (m == 0 || limit == 0) return 0;
  const int kbw_env = env_int("M4RI_HIP_ELIM_BLOCK_WORDS", 32);  // read per call: tests shrink it
  const int KBW = kbw_env < 1 ? 1 : (kbw_env > 32 ? 32 : kbw_env);
  const long long aw = words_of(ncols), lw = words_of(limit), lda = A->ld;
  const int uw = KBW;
  const int max_rank = m < limit ? m : limit;
  const long long pld = (aw + 1) & ~1ll, tld = aw + uw;
  const int prow_max = m < KBW * 64 ? m : KBW * 64;

  DevBuf st, pivs, U, ptab, tmp, P, flags, blkpiv, moves;
  DevBuf &pv = pivcols_keep ? *pivcols_keep : pivs;
  if (int rc = st.alloc(sizeof(gf2k_elim_state))) return rc;
  if (int rc = pv.alloc((size_t)(max_rank + 64) * sizeof(int))) return rc;
  // small problems: one workgroup, the whole matrix in LDS.  It costs ~1 us per column it has to look at (it stops when
  // the rank reaches the row count) against ~0.8 us per column plus ~50 us fixed for the blocked algorithm below, so it
  // is taken when at most 256 columns can matter (1000 x 64: 94 us against 139; 10 x 10: 60 against 109)
  const int cols_to_visit = limit < m + 64 ? limit : m + 64;
  // (cols_to_visit is an estimate: a rank-deficient input walks all `limit` columns serially, hence the second bound)
  if (m <= 1024 && (long long)m * (aw | 1) <= 19000 && cols_to_visit <= 256 && limit <= 4096 && dev_env_int("M4RI_HIP_ELIM_SMALL", 1)) {
    HIP_TRY(gf2k_elim_small(A->data, lda, m, ncols, limit, full, reinterpret_cast<int *>(st.p), pv.as<int>(), s));
    HIP_TRY(hipMemcpyAsync(rank_out, st.p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (pivcols_host && *rank_out > 0)
      HIP_TRY(hipMemcpyAsync(pivcols_host, pv.p, (size_t)*rank_out * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
  }
  if (int rc = U.alloc((size_t)m * uw * sizeof(u64))) return rc;
  // the step's 64 pivot rows over the block + the 16 x 16 selector map + the stash of the next search (256 words, 256 flags: gf2_elim.hip)
  if (int rc = ptab.alloc((64 * 64 + 256 + 512) * sizeof(u64))) return rc;
  if (int rc = tmp.alloc((size_t)2 * GF2K_ELIM_BLOCK_PIVOTS * tld * sizeof(u64))) return rc;
  if (int rc = P.alloc((size_t)prow_max * pld * sizeof(u64))) return rc;
  if (int rc = flags.alloc((size_t)m)) return rc;
  if (int rc = blkpiv.alloc(GF2K_ELIM_BLOCK_PIVOTS * sizeof(int))) return rc;
  if (int rc = moves.alloc(4 * GF2K_ELIM_BLOCK_PIVOTS * sizeof(int))) return rc;
  gf2k_elim_state *dst = st.as<gf2k_elim_state>();
  HIP_TRY(hipMemsetAsync(dst, 0, sizeof(gf2k_elim_state), s));
  HIP_TRY(hipMemsetAsync(flags.p, 0, (size_t)m, s));

  // the pivot search of step j + 1 inside the update launch of step j (gf2_elim.hip, round 4); 0: two launches per step as before
  // (both switches are read by the shipped library too: a part or a partition on which the look-ahead launch cannot be resident as
  // a whole must be able to turn it off)
  static const int lookahead = env_int("M4RI_HIP_ELIM_LOOKAHEAD", 1);
  // test hook: 1 = update workgroup 0 of every look-ahead launch never raises its counters, so the look-ahead workgroup's bounded
  // wait runs out (tests/test_gpu_elim.py::test_lookahead_failure_is_reported_not_hung)
  static const int fault = env_int("M4RI_HIP_ELIM_FAULT", 0);
  const int full_and_flags = (full ? 1 : 0) | (fault << 8);
  // Without augmented columns (limit == ncols) the trailing product of a block does not wait for the block's result: it is
  // enqueued with what the host knows BEFORE the block -- the rank so far = the block's first pivot row r0 -- and with the largest
  // pivot count the block can have (the tracking columns of pivots that were not found are zero, so the rows of P they meet do
  // not matter); the record of the block comes back through pinned memory while the product runs.  The ~70 us per block the
  // device used to idle between the block's last kernel and the product's first (copy back, wake-up, a dozen launches) are gone:
  // 65536^2 59.1 -> 57.5 ms, 16384^2 8.2 -> 7.8, 4096^2 1.87 -> 1.81 (same box).  A block without pivots costs one wasted product, after which the next block
  // takes the waiting form; augmented systems (inverse, solve) keep it always: their product is cut at the last non-zero word.
  static const int spec_on = env_int("M4RI_HIP_ELIM_SPECULATE", 1);
  const bool spec = spec_on && limit == ncols;
  thread_local gf2k_elim_state *hpin = nullptr;
  thread_local std::map<int, hipEvent_t> hevs;  // one per device ORDINAL: an event belongs to the device that was current when it was created
  hipEvent_t hev = nullptr;
  if (spec) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (!hpin) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&hpin), sizeof(gf2k_elim_state), hipHostMallocPortable));
    hipEvent_t &slot = hevs[dev];
    if (!slot) HIP_TRY(hipEventCreateWithFlags(&slot, hipEventDisableTiming));
    hev = slot;
  }
  // Whatever way this function is left, the stream has drained before the scratch buffers above are handed back to the pool: an
  // error return behind an enqueued trailing product must not free U / P / tmp under it.  Declared after the buffers,
  // so destroyed before them.
  struct DrainOnExit {
    hipStream_t s;
    ~DrainOnExit() {
      if (hipStreamSynchronize(s) != hipSuccess) (void)hipGetLastError();
    }
  } drain{s};
  bool prev_empty = false;
  int r_cur = 0;
  for (long long c0w = 0; c0w < lw && r_cur < m; c0w += KBW) {
    const int sw = (int)(lw - c0w < KBW ? lw - c0w : KBW);
    const bool spec_now = spec && !prev_empty;
    HIP_TRY(hipMemsetAsync(U.p, 0, (size_t)m * uw * sizeof(u64), s));
    HIP_TRY(gf2k_elim_begin_block(dst, s));
    for (int j = 0; j < sw; ++j) {
      const bool last = (c0w + j == lw - 1) && (limit & 63);
      const u64 colmask = last ? ((1ull << (limit & 63)) - 1) : ~0ull;
      const bool last_next = (c0w + j + 1 == lw - 1) && (limit & 63);
      const u64 colmask_next = last_next ? ((1ull << (limit & 63)) - 1) : ~0ull;
      HIP_TRY(gf2k_elim_step(A->data, lda, m, c0w, sw, j, colmask, full_and_flags, U.as<u64>(), uw, uw, dst, pv.as<int>(),
                             ptab.as<u64>(), flags.as<unsigned char>(), blkpiv.as<int>(), colmask_next, lookahead, s));
    }
    HIP_TRY(gf2k_elim_end_block(A->data, lda, aw, c0w, U.as<u64>(), uw, uw, dst, flags.as<unsigned char>(), blkpiv.as<int>(),
                                moves.as<int>(), tmp.as<u64>(), tld, spec_now ? aw /* (no last-word scan) */ : c0w + sw, s));
    if (spec_now) {
      HIP_TRY(hipMemcpyAsync(hpin, dst, sizeof(gf2k_elim_state), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipEventRecord(hev, s));
      const int r0 = r_cur, rpmax = std::min(sw * 64, m - r0);
      const long long cR = c0w + sw;
      if (rpmax > 0 && cR < aw) {
        const int rows_lo = full ? 0 : r0;
        const int nright = ncols - (int)(cR * 64);
        HIP_TRY(hipMemcpy2DAsync(P.p, (size_t)pld * sizeof(u64), A->data + (long long)r0 * lda + cR, (size_t)lda * sizeof(u64),
                                 (size_t)(aw - cR) * sizeof(u64), rpmax, hipMemcpyDeviceToDevice, s));
        gf2_dmat Cw{A->data + (long long)rows_lo * lda + cR, lda, m - rows_lo, nright};
        gf2_dmat Uw{U.as<u64>() + (long long)rows_lo * uw, uw, m - rows_lo, rpmax};
        gf2_dmat Pw{P.as<u64>(), pld, rpmax, nright};
        if (int rc = mul_m4rm_plain(&Cw, &Uw, &Pw, 1, s)) return rc;
      }
      HIP_TRY(hipEventSynchronize(hev));  // the block's record (the product is running or queued behind it)
      if (hpin->err) return fail_msg("gf2 elimination: the look-ahead workgroup's wait for the update workgroups ran out");
      if (hpin->r0 != r0) return fail_msg("gf2 elimination: the device's rank record disagrees with the host's");
      prev_empty = hpin->r_cur == hpin->r0;
      r_cur = hpin->r_cur;
      continue;
    }
    gf2k_elim_state hst;
    HIP_TRY(hipMemcpyAsync(&hst, dst, sizeof(hst), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (hst.err) return fail_msg("gf2 elimination: the look-ahead workgroup's wait for the update workgroups ran out");
    const int head[7] = {hst.r0, hst.r_cur, hst.np, hst.nmoves, hst.jbase, hst.scan, hst.lastword};
    const int r0 = head[0], rp = head[1] - head[0];
    r_cur = head[1];
    prev_empty = rp == 0;
    const long long cR = c0w + sw;
    const long long wlast = head[6];  // the block's pivot rows are zero beyond this word: so is their contribution
    if (rp > 0 && cR < aw && wlast >= cR) {
      // everything right of the block in one product: A[rows, right] ^= U'[rows, 0:rp] * (pivot rows of the block)
      const int rows_lo = full ? 0 : r0;
      const int nright = wlast == aw - 1 ? ncols - (int)(cR * 64) : (int)((wlast + 1 - cR) * 64);
      HIP_TRY(hipMemcpy2DAsync(P.p, (size_t)pld * sizeof(u64), A->data + (long long)r0 * lda + cR, (size_t)lda * sizeof(u64),
                               (size_t)(wlast + 1 - cR) * sizeof(u64), rp, hipMemcpyDeviceToDevice, s));
      gf2_dmat Cw{A->data + (long long)rows_lo * lda + cR, lda, m - rows_lo, nright};
      gf2_dmat Uw{U.as<u64>() + (long long)rows_lo * uw, uw, m - rows_lo, rp};
      gf2_dmat Pw{P.as<u64>(), pld, rp, nright};
      // plain M4RM on purpose: (a) the callers hold g_enqueue_mu, which mul_dispatch takes itself; (b) Strassen levels over
      // this shape (m x 2048 x n, through mul_strassen directly) were measured neutral: 85.7 against 85.3 ms at 65536^2 (round 3);
      // through the planner (GF2_ALGO_AUTO) again neutral in round 4: 58.58 against 58.56 ms, 16384^2 inverse 10.5 against 8.8
      if (int rc = mul_m4rm_plain(&Cw, &Uw, &Pw, 1, s)) return rc;
    }
  }
  *rank_out = r_cur;
  if (pivcols_host && r_cur > 0)
    HIP_TRY(hipMemcpyAsync(pivcols_host, pv.p, (size_t)r_cur * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return 0;
}
}  // namespace

extern "C" int gf2_echelonize_dev(gf2_dmat *A, int full, int ncols_limit, int *rank, int *pivot_cols, void *stream) {
  if (int rc = require_device()) return rc;
  if (!A || !A->data || !rank) return fail_msg("gf2_echelonize_dev: null argument");
  if (A->ld < words_of(A->ncols)) return fail_msg("gf2_echelonize_dev: row stride smaller than row width");
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  std::lock_guard<std::mutex> lk(g_enqueue_mu);
  return echelonize_dev(A, full, ncols_limit, rank, pivot_cols, nullptr, s);
}

// [ A | 0-pad to a word boundary | I ] -> reduced echelon form of the left part; singular unless rank == n
static int inverse_dev(gf2_dmat *Ainv, const u64 *Adata, long long lda, bool a_on_host, int n, int *singular, hipStream_t s) {
  const int nw = words_of(n);
  gf2_dmat T{nullptr, dev_ld_for(nw * 64 + n), n, nw * 64 + n};
  DevBuf tb;
  if (int rc = tb.alloc((size_t)n * T.ld * sizeof(u64))) return rc;
  T.data = tb.as<u64>();
  HIP_TRY(hipMemsetAsync(T.data, 0, (size_t)n * T.ld * sizeof(u64), s));
  HIP_TRY(hipMemcpy2DAsync(T.data, (size_t)T.ld * sizeof(u64), Adata, (size_t)lda * sizeof(u64), (size_t)nw * sizeof(u64), n,
                           a_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
  HIP_TRY(gf2k_set_diag(T.data, T.ld, n, (long long)nw * 64, s));
  int rank = 0;
  if (int rc = echelonize_dev(&T, 1, n, &rank, nullptr, nullptr, s)) return rc;
  *singular = rank < n;
  if (rank == n) {
    HIP_TRY(hipMemcpy2DAsync(Ainv->data, (size_t)Ainv->ld * sizeof(u64), T.data + nw, (size_t)T.ld * sizeof(u64),
                             (size_t)nw * sizeof(u64), n, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return 0;
}

extern "C" int gf2_inverse_dev(gf2_dmat *Ainv, gf2_dmat const *A, int *singular, void *stream) {
  if (int rc = require_device()) return rc;
  if (!Ainv || !A || !singular || !A->data || !Ainv->data) return fail_msg("gf2_inverse_dev: null argument");
  if (A->nrows != A->ncols || Ainv->nrows != A->nrows || Ainv->ncols != A->ncols)
    return fail_msg("gf2_inverse_dev: matrices must be square and of equal size");
  hipStream_t s;
  if (int rc = get_stream(stream, &s)) return rc;
  *singular = 0;
  if (A->nrows == 0) return 0;
  std::lock_guard<std::mutex> lk(g_enqueue_mu);
  return inverse_dev(Ainv, A->data, A->ld, false, A->nrows, singular, s);
}

static int host_echelonize(mzd_t *A, int full, const char *name) {
  if (A->nrows == 0 || A->ncols == 0) return 0;
  gf2_cache_forget(A);  // modified in place
  auto bail = [&](const char *why) {
    std::fprintf(stderr, "m4ri_hip: %s failed: %s (%s)\n", name, why, gf2_last_error());
    std::abort();  // the M4RI signature has no error channel (returns the rank)
    return 0;
  };
  if (require_device()) return bail("no device");
  {
    const long long lim = gf2_small_work_limit();  // size dispatch, see host_mul_on
    if (lim > 0 && (long long)A->nrows * A->width * (A->nrows < A->ncols ? A->nrows : A->ncols) <= lim)
      return gf2_echelonize_host_small(A, full);
  }
  PinnedDevice pin;  // M4RI_HIP_DEVICES = one ordinal: run there
  hipStream_t s;
  if (get_private_stream(&s)) return bail("stream");
  int rank = 0, rc;
  {
    DMatOwner dA;
    rc = to_device(dA, A, s, true);
    if (!rc) rc = echelonize_dev(&dA.d, full, 0, &rank, nullptr, nullptr, s);
    if (!rc) rc = gf2_dmat_download(A, &dA.d, s);
    if (rc) (void)hipStreamSynchronize(s);
  }
  if (rc) return bail("device elimination");
  return rank;
}

extern "C" rci_t mzd_echelonize(mzd_t *A, int full) { return host_echelonize(A, full, "mzd_echelonize"); }
extern "C" rci_t mzd_echelonize_m4ri(mzd_t *A, int full, int k) {
  (void)k;  // table size hint of the CPU algorithm
  return host_echelonize(A, full, "mzd_echelonize_m4ri");
}
extern "C" rci_t mzd_echelonize_pluq(mzd_t *A, int full) { return host_echelonize(A, full, "mzd_echelonize_pluq"); }

extern "C" mzd_t *mzd_inv_m4ri(mzd_t *dst, mzd_t const *src, int k) {
  (void)k;
  if (src->nrows != src->ncols) gf2_die("mzd_inv_m4ri: matrix must be square.");
  if (dst && (dst->nrows != src->nrows || dst->ncols != src->ncols)) gf2_die("mzd_inv_m4ri: dst has wrong dimensions.");
  auto bail = [&](const char *why) -> mzd_t * {
    std::fprintf(stderr, "m4ri_hip: mzd_inv_m4ri failed: %s (%s)\n", why, gf2_last_error());
    return nullptr;
  };
  if (require_device()) return bail("no device");
  const int n = src->nrows;
  if (n == 0) return dst ? dst : mzd_init(0, 0);
  PinnedDevice pin;  // M4RI_HIP_DEVICES = one ordinal: run there
  hipStream_t s;
  if (get_private_stream(&s)) return bail("stream");
  int rc, singular = 0;
  const bool allocated = dst == nullptr;
  if (!dst) dst = gf2_mzd_init_uncleared(n, n);
  else gf2_cache_forget(dst);
  {
    DMatOwner dI;
    rc = to_device(dI, dst, s, false);
    if (!rc) rc = inverse_dev(&dI.d, src->rows[0], src->rowstride, true, n, &singular, s);
    if (!rc && !singular) rc = gf2_dmat_download(dst, &dI.d, s);
    if (rc) (void)hipStreamSynchronize(s);
  }
  if (rc || singular) {
    if (allocated) mzd_free(dst);
    if (rc) return bail("device elimination");
    return nullptr;  // not invertible: no inverse to return (callers see NULL; BinMatrix::inverted panics "Can't be NULL")
  }
  return dst;
}

extern "C" int mzd_solve_left(mzd_t *A, mzd_t *B, int cutoff, int inconsistency_check) {
  (void)cutoff;
  if (A->ncols > B->nrows) gf2_die("mzd_solve_left: A ncols must be smaller than B nrows.");
  if (A->nrows > B->nrows) gf2_die("mzd_solve_left: A nrows must be smaller than B nrows.");
  const int m = A->nrows, n = A->ncols, kb = B->ncols;
  if (m == 0 || n == 0 || kb == 0) return 0;
  gf2_cache_forget(A);  // both are overwritten
  gf2_cache_forget(B);
  auto bail = [&](const char *why) {
    std::fprintf(stderr, "m4ri_hip: mzd_solve_left failed: %s (%s)\n", why, gf2_last_error());
    std::abort();  // -1 means "inconsistent" in this signature; a device failure is not that
    return -1;
  };
  if (require_device()) return bail("no device");
  PinnedDevice pin;  // M4RI_HIP_DEVICES = one ordinal: run there
  hipStream_t s;
  if (get_private_stream(&s)) return bail("stream");
  const int nw = words_of(n), bw = words_of(kb);
  int rc = 0, inconsistent = 0;
  do {
    // T = [ A | pad | B[0:m] ]
    gf2_dmat T{nullptr, dev_ld_for(nw * 64 + kb), m, nw * 64 + kb};
    DevBuf tb, xb, flag, pivs;
    if ((rc = tb.alloc((size_t)m * T.ld * sizeof(u64)))) break;
    T.data = tb.as<u64>();
    hipError_t e = hipMemsetAsync(T.data, 0, (size_t)m * T.ld * sizeof(u64), s);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(T.data, (size_t)T.ld * sizeof(u64), A->rows[0], (size_t)A->rowstride * sizeof(word),
                           (size_t)nw * sizeof(u64), m, hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(T.data + nw, (size_t)T.ld * sizeof(u64), B->rows[0], (size_t)B->rowstride * sizeof(word),
                           (size_t)bw * sizeof(u64), m, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) {
      rc = fail(e, "mzd_solve_left: upload");
      break;
    }
    int rank = 0;
    if ((rc = echelonize_dev(&T, 1, n, &rank, nullptr, &pivs, s))) break;
    if (inconsistency_check && rank < m) {
      if ((rc = flag.alloc(sizeof(int)))) break;
      e = hipMemsetAsync(flag.p, 0, sizeof(int), s);
      if (e == hipSuccess) e = gf2k_any_nonzero(T.data + nw, T.ld, rank, m, bw, flag.as<int>(), s);
      if (e == hipSuccess) e = hipMemcpyAsync(&inconsistent, flag.p, sizeof(int), hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        rc = fail(e, "mzd_solve_left: consistency check");
        break;
      }
    }
    // X[pivot column k] = reduced right-hand side row k; free variables are 0; rows n.. of B are cleared
    gf2_dmat X{nullptr, dev_ld_for(kb), B->nrows, kb};
    if ((rc = xb.alloc((size_t)B->nrows * X.ld * sizeof(u64)))) break;
    X.data = xb.as<u64>();
    e = hipMemsetAsync(X.data, 0, (size_t)B->nrows * X.ld * sizeof(u64), s);
    if (e == hipSuccess) e = gf2k_scatter_rows(X.data, X.ld, T.data + nw, T.ld, bw, pivs.as<int>(), rank, s);
    if (e != hipSuccess) {
      rc = fail(e, "mzd_solve_left: solution rows");
      break;
    }
    if ((rc = gf2_dmat_download(B, &X, s))) break;
    gf2_dmat Ared{T.data, T.ld, m, n};  // "A Input matrix (overwritten)": left holding its reduced echelon form
    if ((rc = gf2_dmat_download(A, &Ared, s))) break;
  } while (0);
  if (rc) {
    (void)hipStreamSynchronize(s);
    return bail("device elimination");
  }
  return inconsistent ? -1 : 0;
}
