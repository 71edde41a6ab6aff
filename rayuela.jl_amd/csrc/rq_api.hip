// rq_api.hip -- the C ABI of librayuela_hip.so (include/rayuela_hip.h): argument checks,
// device buffers, H2D/D2H for the host-pointer entry points, the legacy linscan_aqd_query symbol.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <string>
#include <thread>

#include <vector>
#include "rq_internal.h"

namespace rq {

static thread_local char g_err[512] = "";
static thread_local unsigned g_err_seq = 0;     // errors raised on this thread so far (DevBuf: "is my call unwinding?")
static thread_local double g_t_total = 0, g_t_h2d = 0, g_t_kernel = 0, g_t_d2h = 0;

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  ++g_err_seq;
  return code;
}

void set_timing(double total_ms, double h2d_ms, double kernel_ms, double d2h_ms) {
  g_t_total = total_ms; g_t_h2d = h2d_ms; g_t_kernel = kernel_ms; g_t_d2h = d2h_ms;
}

int fail_hip(hipError_t e, const char *what, const char *file, int line) {
  snprintf(g_err, sizeof(g_err), "HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file,
           line, what);
  ++g_err_seq;
  (void)hipGetLastError();
  return (int)e > 0 ? (int)e : 1;
}

void forgive_oom() {
  g_err[0] = 0;
  (void)hipGetLastError();
}

// ---- tuning knobs: env RQ_<KEY>, or rq_set_tuning() ---------------------------------------------
struct Knob { char key[32]; int value; };
static Knob g_knobs[64];
static int g_nknobs = 0;
static std::mutex g_mu;

int tuning(const char *key, int dflt) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < g_nknobs; ++i)
      if (!strcmp(g_knobs[i].key, key)) return g_knobs[i].value;
  }
  std::string env = std::string("RQ_") + key;
  const char *v = getenv(env.c_str());
  return v ? atoi(v) : dflt;
}

// ---- device context -----------------------------------------------------------------------------
// Scratch is keyed by (device, stream): two streams of one device never share candidate buffers or
// work counters, so concurrent rq_dev_* calls on different streams cannot corrupt each other.  The
// launch mutex serialises the memset + launch sequences of one device across host threads (a second
// thread's counter reset must not slip between another call's reset and its kernel).
constexpr int MAX_STREAM_WS = 8;
struct StreamWs {
  hipStream_t stream = nullptr;
  bool used = false;
  void *ws[WS_SLOTS] = {nullptr};
  size_t ws_bytes[WS_SLOTS] = {0};
};
struct DevCtx {
  bool inited = false;
  DeviceInfo info;
  StreamWs sw[MAX_STREAM_WS];
  std::recursive_mutex launch_mu;
  hipStream_t aux[2] = {nullptr, nullptr};   // compute / transfer streams of the host-pointer calls
  // device buffers of finished host-pointer calls, kept for the next one (hipMalloc + hipFree of the 170 MB a
  // linscan_pq call stages cost 1.5-2.5 ms, a third of the call); rq_release_workspaces frees them
  struct Cached { void *p; size_t bytes; };
  std::vector<Cached> pool;
  size_t pool_bytes = 0;
};
static DevCtx g_dev[16];

int device_info(DeviceInfo *out) {
  int dev = 0;
  RQ_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail(RQ_ENODEVICE, "device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lk(g_mu);
  DevCtx &c = g_dev[dev];
  if (!c.inited) {
    hipDeviceProp_t prop;
    RQ_HIP(hipGetDeviceProperties(&prop, dev));
    c.info.device = dev;
    c.info.num_cu = prop.multiProcessorCount;
    strncpy(c.info.arch, prop.gcnArchName, sizeof(c.info.arch) - 1);
    c.info.arch[sizeof(c.info.arch) - 1] = 0;
    if (strncmp(c.info.arch, "gfx950", 6) != 0)
      return fail(RQ_ENODEVICE, "librayuela_hip is built for gfx950 (MI355X); device %d is %s", dev, c.info.arch);
    c.inited = true;
  }
  *out = c.info;
  return RQ_OK;
}

// The device's persistent (compute, transfer) stream pair; callers hold the DeviceLock.
int aux_streams(hipStream_t *cs, hipStream_t *xs) {
  int dev = 0;
  RQ_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  DevCtx &c = g_dev[dev];
  for (int i = 0; i < 2; ++i)
    if (!c.aux[i]) RQ_HIP(hipStreamCreateWithFlags(&c.aux[i], hipStreamNonBlocking));
  *cs = c.aux[0];
  *xs = c.aux[1];
  return RQ_OK;
}

DeviceLock::DeviceLock() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { (void)hipGetLastError(); dev = 0; }
  mu_ = &g_dev[dev].launch_mu;
  static_cast<std::recursive_mutex *>(mu_)->lock();
}
DeviceLock::~DeviceLock() { static_cast<std::recursive_mutex *>(mu_)->unlock(); }

int workspace(int slot, size_t bytes, void **ptr, hipStream_t stream) {
  int dev = 0;
  RQ_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  DevCtx &c = g_dev[dev];
  StreamWs *w = nullptr;
  for (int i = 0; i < MAX_STREAM_WS && !w; ++i)
    if (c.sw[i].used && c.sw[i].stream == stream) w = &c.sw[i];
  for (int i = 0; i < MAX_STREAM_WS && !w; ++i)
    if (!c.sw[i].used) { w = &c.sw[i]; w->used = true; w->stream = stream; }
  if (!w)
    return fail(RQ_EUNSUPPORTED, "more than %d distinct streams have used device %d (rq_release_workspaces frees them)",
                MAX_STREAM_WS, dev);
  if (w->ws_bytes[slot] < bytes) {
    if (w->ws[slot]) {
      RQ_HIP(hipDeviceSynchronize());
      RQ_HIP(hipFree(w->ws[slot]));
      w->ws[slot] = nullptr;
      w->ws_bytes[slot] = 0;
    }
    size_t want = bytes + bytes / 4;
    want = (want + 255) & ~(size_t)255;
    RQ_HIP(hipMalloc(&w->ws[slot], want));
    w->ws_bytes[slot] = want;
  }
  *ptr = w->ws[slot];
  return RQ_OK;
}

// Free the scratch of one stream of the current device (called before the stream is destroyed).
int release_stream_workspace(hipStream_t stream) {
  int dev = 0;
  RQ_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  DevCtx &c = g_dev[dev];
  for (int i = 0; i < MAX_STREAM_WS; ++i) {
    if (!c.sw[i].used || c.sw[i].stream != stream) continue;
    RQ_HIP(hipStreamSynchronize(stream));
    for (int s = 0; s < WS_SLOTS; ++s) {
      if (c.sw[i].ws[s]) (void)hipFree(c.sw[i].ws[s]);
      c.sw[i].ws[s] = nullptr;
      c.sw[i].ws_bytes[s] = 0;
    }
    c.sw[i].used = false;
    c.sw[i].stream = nullptr;
  }
  return RQ_OK;
}

// Free every scratch buffer of the current device (streams that were destroyed leave theirs behind).
int release_workspaces() {
  int dev = 0;
  RQ_HIP(hipGetDevice(&dev));
  DeviceLock launch_lock;   // no other host thread is between "look up scratch" and "launch" on this device
  RQ_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_mu);
  DevCtx &c = g_dev[dev];
  for (int i = 0; i < MAX_STREAM_WS; ++i) {
    for (int s = 0; s < WS_SLOTS; ++s) {
      if (c.sw[i].ws[s]) (void)hipFree(c.sw[i].ws[s]);
      c.sw[i].ws[s] = nullptr;
      c.sw[i].ws_bytes[s] = 0;
    }
    c.sw[i].used = false;
    c.sw[i].stream = nullptr;
  }
  for (auto &b : c.pool) (void)hipFree(b.p);
  c.pool.clear();
  c.pool_bytes = 0;
  return RQ_OK;
}

// ---- page-locked result buffers for the language shims --------------------------------------------------------
// A linscan call returns 8 * k * nq bytes (80 MB at SIFT1M shape, k = 1000).  Into a fresh pageable array (Julia
// `zeros`, numpy `empty`: untouched pages) the device-to-host copies run at the speed of first-touch page faults --
// 18 GB/s on the bench box, 4.4 ms, more than the scan itself; 16 host threads copying into fresh pages reach the same
// 18 GB/s (tools/micro/pagefault_copy.cpp), so it is the kernel's fault path, not the copy.  rq_host_alloc hands the
// shim a page-locked buffer instead (hipHostMalloc, ~50 GB/s, no faults) that the shim wraps as the result array and
// gives back with rq_host_free when the array is collected; freed buffers are pooled (hipHostMalloc costs
// milliseconds).  Limits: HOST_PIN_MAX_MB (4096) handed out at any time -- beyond it rq_host_alloc returns NULL and
// the shim uses an ordinary array -- and HOST_PIN_POOL_MB (1024) kept idle.
struct HostPool {
  struct Buf { void *p; size_t bytes; };
  std::vector<Buf> idle, live;
  size_t idle_bytes = 0, live_bytes = 0;
};
static HostPool g_hp;

void *host_pool_alloc(size_t bytes) {
  if (!bytes || !tuning("HOST_PIN", 1)) return nullptr;
  const size_t want = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
  int max_mb = tuning("HOST_PIN_MAX_MB", 0);
  if (max_mb <= 0) max_mb = 4096;
  const size_t max_live = (size_t)max_mb << 20;
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_hp.live_bytes + want > max_live) return nullptr;
  int best = -1;
  for (size_t i = 0; i < g_hp.idle.size(); ++i)
    if (g_hp.idle[i].bytes >= want && g_hp.idle[i].bytes <= 2 * want &&
        (best < 0 || g_hp.idle[i].bytes < g_hp.idle[best].bytes)) best = (int)i;
  HostPool::Buf b{nullptr, 0};
  if (best >= 0) {
    b = g_hp.idle[best];
    g_hp.idle_bytes -= b.bytes;
    g_hp.idle.erase(g_hp.idle.begin() + best);
  } else {
    if (hipHostMalloc(&b.p, want, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    b.bytes = want;
  }
  g_hp.live.push_back(b);
  g_hp.live_bytes += b.bytes;
  return b.p;
}

void host_pool_free(void *p) {
  if (!p) return;
  int keep_mb = tuning("HOST_PIN_POOL_MB", 0);
  if (keep_mb <= 0) keep_mb = 1024;
  const size_t keep = (size_t)keep_mb << 20;
  HostPool::Buf b{nullptr, 0};
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_hp.live.size(); ++i)
      if (g_hp.live[i].p == p) { b = g_hp.live[i]; g_hp.live.erase(g_hp.live.begin() + i); break; }
    if (!b.p) return;                       // not ours
    g_hp.live_bytes -= b.bytes;
    if (g_hp.idle_bytes + b.bytes <= keep && g_hp.idle.size() < 64) {
      g_hp.idle.push_back(b);
      g_hp.idle_bytes += b.bytes;
      return;
    }
  }
  (void)hipHostFree(b.p);
}

// true when [p, p + bytes) lies inside a buffer rq_host_alloc handed out (page-locked and mapped: the device can write it)
bool host_pool_owns(const void *p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &b : g_hp.live)
    if ((const char *)p >= (const char *)b.p && (const char *)p + bytes <= (const char *)b.p + b.bytes) return true;
  return false;
}

void host_pool_trim() {
  std::vector<HostPool::Buf> drop;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    drop.swap(g_hp.idle);
    g_hp.idle_bytes = 0;
  }
  for (auto &b : drop) (void)hipHostFree(b.p);
}

// RAII device buffer for the host-pointer entry points.  Buffers up to HOST_CACHE_MAX_MB (256) each go back to a
// per-device pool of at most HOST_CACHE_MB (2048) instead of hipFree: every host-pointer call synchronises before it
// returns, so a pooled buffer is idle.
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int dev = -1;
  unsigned err_seq = 0;   // g_err_seq when the buffer was taken
  ~DevBuf() {
    if (!p) return;
    // A call that returns normally has synchronised its streams.  One that unwinds early (RQ_TRY / RQ_HIP return with
    // kernels or async copies still queued on the aux streams) has not: drain the device before the buffer can be
    // handed to the next call.  Every error path goes through fail() / fail_hip(), which bump the thread's counter.
    if (err_seq != g_err_seq) {
      int cur = -1;
      if (hipGetDevice(&cur) == hipSuccess && dev >= 0 && cur != dev) (void)hipSetDevice(dev);
      (void)hipDeviceSynchronize();
      if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
    }
    const size_t each = (size_t)std::max(0, tuning("HOST_CACHE_MAX_MB", 256)) << 20;
    const size_t total = (size_t)std::max(0, tuning("HOST_CACHE_MB", 2048)) << 20;
    if (dev >= 0 && dev < 16 && bytes <= each) {
      std::lock_guard<std::mutex> lk(g_mu);
      DevCtx &c = g_dev[dev];
      if (c.pool_bytes + bytes <= total && c.pool.size() < 64) {
        c.pool.push_back({p, bytes});
        c.pool_bytes += bytes;
        return;
      }
    }
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev && dev >= 0) (void)hipSetDevice(dev);
    (void)hipFree(p);
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
  }
  int alloc(size_t want) {
    want = ((want ? want : 16) + 255) & ~(size_t)255;
    err_seq = g_err_seq;
    RQ_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < 16) {
      std::lock_guard<std::mutex> lk(g_mu);
      DevCtx &c = g_dev[dev];
      int best = -1;       // smallest pooled buffer that fits and is not more than twice too large
      for (size_t i = 0; i < c.pool.size(); ++i)
        if (c.pool[i].bytes >= want && c.pool[i].bytes <= 2 * want + (1u << 20) &&
            (best < 0 || c.pool[i].bytes < c.pool[best].bytes)) best = (int)i;
      if (best >= 0) {
        p = c.pool[best].p;
        bytes = c.pool[best].bytes;
        c.pool_bytes -= bytes;
        c.pool.erase(c.pool.begin() + best);
        return RQ_OK;
      }
    }
    RQ_HIP(hipMalloc(&p, want));
    bytes = want;
    return RQ_OK;
  }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

struct Timer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
};

// ---- bank-aware row order of a resident base (rq_order.hip) ---------------------------------------------
// Does ordering a scratch copy of the base INSIDE a call pay?  From ORDER_MIN_NQ queries on (the four small kernels cost
// ~0.1 ms per 1e6 rows, the scan gains ~0.15 us per query group and 1e6 rows) and below ORDER_MAX_K neighbours: at
// k = 10000 a fifth of the rows reach the exact evaluation and 44 % of the time is the per-query sample sort -- the table
// gathers no longer bound the kernel (measured: 5.73 ms in arrival order, 5.74 prepared, 5.84 with the ordering inside).
// k <= 0: a base prepared once (index handles, rq_dev_order_rows) -- ordered whenever it is large enough.
bool order_pays(int64_t n, int64_t nq, int k) {
  const int mode = tuning("SCAN_ORDER", 1);
  if (mode <= 0 || n < tuning("ORDER_MIN_ROWS", 65536)) return false;
  if (mode > 1 || k <= 0) return true;
  return nq >= tuning("ORDER_MIN_NQ", 2048) && k < tuning("ORDER_MAX_K", 8192);
}

size_t order_base_bytes(int64_t n, int mp) { return (((size_t)n * mp + 255) & ~(size_t)255) + (size_t)n * 4; }

// codes [n][mp] (already padded to a tiled width) -> *out_codes [n][mp] in bank-aware order + *out_perm [n], both inside
// `dst` (order_base_bytes).  The key scratch comes from the (device, stream) workspace.
int order_base(const uint8_t **out_codes, const uint32_t **out_perm, void *dst, const uint8_t *codes, int64_t n, int mp,
               hipStream_t stream) {
  int nb[8];
  OrderTiling ot;
  scan_order_tiling(mp, &ot);
  const int bits = order_key_bits(n, mp, ot, nb);
  if (bits <= 0) return RQ_OK;          // tiny base: stays as it is (*out_perm untouched = nullptr)
  void *tmp = nullptr;
  RQ_TRY(workspace(WS_ORDER_TMP, order_scratch_bytes(n, bits), &tmp, stream));
  uint32_t *pm = reinterpret_cast<uint32_t *>((uint8_t *)dst + (order_base_bytes(n, mp) - (size_t)n * 4));
  RQ_TRY(order_rows_launch((uint8_t *)dst, pm, codes, n, mp, tmp, ot, stream));
  *out_codes = (const uint8_t *)dst;
  *out_perm = pm;
  return RQ_OK;
}

// ---- shared implementation of the scan on device pointers --------------------------------------------
int dev_linscan(float *dists, uint32_t *ids, uint64_t *keys, const uint8_t *codes, const float *centers,
                const float *queries, int64_t n, int64_t nq, int m, int d, int k, uint32_t id_offset,
                int id_base, hipStream_t stream, int lut_mode, const float *row_bias, const ScanBase *base) {
  const uint32_t *perm = base ? base->perm : nullptr;
  uint8_t *norm_prepared = base ? base->norm_prepared : nullptr;
  const bool padded = base && (base->padded || base->perm);
  if (nq <= 0) return RQ_OK;
  if (n < 1 || n >= (1LL << 31)) return fail(RQ_EINVAL, "n=%lld must be in [1, 2^31)", (long long)n);
  if (lut_mode < LUT_PQ || lut_mode > LUT_CQ) return fail(RQ_EINVAL, "lut_mode=%d", lut_mode);
  if (m < 1 || d < 1 || (lut_mode == LUT_PQ && (d < m || d % m != 0)))
    return fail(RQ_EINVAL, "scan needs d %% m == 0 (src/Linscan.jl:23 Cint(d/m)); got d=%d m=%d", d, m);
  if (k < 1 || k > RQ_MAX_K) return fail(RQ_EUNSUPPORTED, "k=%d outside [1, %d]", k, RQ_MAX_K);
  if (k > n) return fail(RQ_EINVAL, "k=%d > n=%lld (undefined in the reference, deps/src/linscan_aqd.cpp:91)", k, (long long)n);
  if ((uint64_t)id_offset + (uint64_t)n > 0xFFFFFFFFull) return fail(RQ_EINVAL, "row ids overflow uint32");
  if (id_base != 0 && id_base != 1) return fail(RQ_EINVAL, "id_base must be 0 or 1");
  if (!keys && (!dists || !ids)) return fail(RQ_EINVAL, "need dists+ids or keys");
  if (((uintptr_t)codes & 15) != 0) return fail(RQ_EINVAL, "codes pointer must be 16-byte aligned");
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock launch_lock;   // workspace lookup + counter reset + launches of this device: one thread at a time
  const int mp = scan_padded_m(m);
  if (mp < 0) return fail(RQ_EUNSUPPORTED, "the ADC scan kernels cover 1 <= m <= 64 sub-quantizers; got m=%d", m);
  if (mp != m && !padded) {
    // row width not one of the tiled ones: zero-pad the rows (padding tables are all zero, so the
    // sequential sum is unchanged bit for bit)
    void *padded = nullptr;
    RQ_TRY(workspace(WS_PAD, (size_t)n * mp, &padded, stream));
    RQ_TRY(pad_codes_launch((uint8_t *)padded, codes, n, m, mp, stream));
    codes = (const uint8_t *)padded;
  }
  // Bank-aware row order (rq_order.hip).  `perm` given: `codes` is an ordered base already, rows padded to mp bytes
  // (order_base: index handles, rq_dev_order_rows, the host-pointer calls).  Otherwise the call orders a copy itself when
  // that pays: the four small kernels cost ~40 us at 1e6 rows, a scan of nq queries gains ~15 % of its time -- from
  // ORDER_MIN_NQ (2048) queries on.  LSQ scans index row_bias / norm bytes by position and keep the arrival order.
  // The in-call order is an optimisation on hidden scratch (n * (mp + 8) bytes per device and stream, kept until
  // rq_release_workspaces): it is skipped above ORDER_MAX_SCRATCH_MB (default 2048 MiB = a 1.25e8-row m = 8 shard; bigger
  // bases belong in an index handle or rq_dev_order_rows, which order ONCE), and a scratch allocation that fails (out of memory)
  // means "scan in arrival order" -- the round-3 path, which needs no extra memory -- never a failed search; any other error of
  // the ordering kernels is returned.
  if (!perm && !row_bias && order_pays(n, nq, k) &&
      order_base_bytes(n, mp) + (size_t)n * 4 <= (size_t)std::max(0, tuning("ORDER_MAX_SCRATCH_MB", 2048)) * 1048576ull) {
    void *ord = nullptr;
    const uint8_t *ocodes = codes;
    const uint32_t *operm = nullptr;
    int orc = workspace(WS_ORDER, order_base_bytes(n, mp), &ord, stream);
    order_set_call_queries(nq);      // (the greedy balance of the order costs more than the sort: only for batches it pays for)
    if (orc == RQ_OK) orc = order_base(&ocodes, &operm, ord, codes, n, mp, stream);
    order_set_call_queries(0);
    if (orc == RQ_OK) {
      codes = ocodes;
      perm = operm;
    } else if (is_oom(orc)) {
      forgive_oom();                // no memory for the scratch: not an error of the search, it runs in arrival order
    } else {
      return orc;                   // a launch error of the ordering kernels is an error
    }
  }
  ScanPlan pl;
  RQ_TRY(scan_plan(pl, n, nq, m, d, k, di.num_cu, tuning("SCAN_SLICES", 0)));
  void *cand = nullptr, *counter = nullptr;
  RQ_TRY(workspace(WS_CAND, pl.cand_bytes, &cand, stream));
  RQ_TRY(workspace(WS_COUNTER, WS_COUNTER_BYTES, &counter, stream));
  // whole items write the answer (or final keys) themselves; the sliced tail leaves per-slice key lists
  // that merge_topk turns into the same outputs for those queries
  const int64_t q_tail = std::min<int64_t>(nq, (int64_t)pl.whole * pl.qg);
  const bool sliced = pl.nslices > 1 && q_tail < nq;
  const bool both = keys && (dists || ids);      // keys requested together with dists/ids: unpack at the end
  void *part = nullptr;
  if (sliced) RQ_TRY(workspace(WS_KEYS, (size_t)(nq - q_tail) * pl.nslices * k * sizeof(uint64_t), &part, stream));
  void *normb = norm_prepared;     // LSQ pre-filter: one byte per row, 32 bytes of min / step, |c|^2 tables, f32 residual norms
  if (lut_mode == LUT_LSQ && row_bias && !normb)
    RQ_TRY(workspace(WS_NORMB, lsq_norm_bytes(n), &normb, stream));
  RQ_TRY(scan_launch(pl, both ? nullptr : dists, both ? nullptr : ids, keys, (uint64_t *)part, codes, centers, queries,
                     n, nq, m, d, k, id_offset, id_base, (uint32_t *)counter, (uint64_t *)cand, stream, lut_mode,
                     row_bias, (uint8_t *)normb, perm, norm_prepared != nullptr));
  if (sliced) {
    const size_t off = (size_t)q_tail * k;
    RQ_TRY(merge_launch(both || !dists ? nullptr : dists + off, both || !ids ? nullptr : ids + off,
                        keys ? keys + off : nullptr, (const uint64_t *)part, nq - q_tail, (int)pl.nslices, k, id_base,
                        stream));
  }
  if (both) return merge_launch(dists, ids, nullptr, keys, nq, 1, k, id_base, stream);
  return RQ_OK;
}

// Scan `nq` resident queries and bring the [nq][k] results to host memory.  Large batches are scanned in
// chunks of 4096 queries (one full round of work items) on a compute stream while a second stream copies
// the previous chunk's results: 80 MB of results at k = 1000 cost 4.6-6 ms over PCIe into pageable memory,
// against 7 ms of kernel.  scan(q0, nqc, stream) launches the scan of queries [q0, q0+nqc) into dd/di.
template <class ScanFn>
static int scan_and_fetch(float *dists, uint32_t *ids, float *dd, uint32_t *di, int64_t nq, int k, ScanFn scan) {
  const int64_t hc = std::max(256, tuning("HOST_CHUNK", 4096));
  const int64_t chunk = (nq >= 2 * hc && tuning("HOST_OVERLAP", 1)) ? hc : nq;
  const size_t row = (size_t)k * 4;
  if (!dd) {
    // the caller's result arrays are page-locked buffers of the library (rq_host_alloc): the scan writes its answer
    // straight into them over PCIe -- one launch over all queries, no copy back
    Timer t2;
    RQ_TRY(scan(0, nq, nullptr));
    RQ_HIP(hipDeviceSynchronize());
    g_t_kernel = t2.ms();
    return RQ_OK;
  }
  if (chunk >= nq) {
    Timer t2;
    RQ_TRY(scan(0, nq, nullptr));
    RQ_HIP(hipDeviceSynchronize());
    g_t_kernel = t2.ms();
    Timer t3;
    RQ_HIP(hipMemcpy(dists, dd, (size_t)nq * row, hipMemcpyDeviceToHost));
    RQ_HIP(hipMemcpy(ids, di, (size_t)nq * row, hipMemcpyDeviceToHost));
    g_t_d2h = t3.ms();
    return RQ_OK;
  }
  struct Streams {
    hipStream_t cs = nullptr, xs = nullptr;   // the device's cached pair (compute, transfer): not owned
    hipEvent_t ev[2] = {nullptr, nullptr};
    ~Streams() {
      if (ev[0]) (void)hipEventDestroy(ev[0]);
      if (ev[1]) (void)hipEventDestroy(ev[1]);
    }
  } st;
  RQ_HIP(hipDeviceSynchronize());   // the uploads on the null stream are done
  RQ_TRY(aux_streams(&st.cs, &st.xs));
  RQ_HIP(hipEventCreateWithFlags(&st.ev[0], hipEventDisableTiming));
  RQ_HIP(hipEventCreateWithFlags(&st.ev[1], hipEventDisableTiming));
  Timer t2;
  int64_t prev_q0 = -1, prev_n = 0;
  int c = 0;
  auto fetch = [&](int64_t q0, int64_t nqc, hipEvent_t ev) -> int {
    RQ_HIP(hipStreamWaitEvent(st.xs, ev, 0));
    RQ_HIP(hipMemcpyAsync(dists + (size_t)q0 * k, dd + (size_t)q0 * k, (size_t)nqc * row, hipMemcpyDeviceToHost, st.xs));
    RQ_HIP(hipMemcpyAsync(ids + (size_t)q0 * k, di + (size_t)q0 * k, (size_t)nqc * row, hipMemcpyDeviceToHost, st.xs));
    return RQ_OK;
  };
  for (int64_t q0 = 0; q0 < nq; q0 += chunk, ++c) {
    const int64_t nqc = std::min(chunk, nq - q0);
    RQ_TRY(scan(q0, nqc, st.cs));
    RQ_HIP(hipEventRecord(st.ev[c & 1], st.cs));
    if (prev_q0 >= 0) RQ_TRY(fetch(prev_q0, prev_n, st.ev[(c - 1) & 1]));   // overlaps with the scan just queued
    prev_q0 = q0; prev_n = nqc;
  }
  RQ_HIP(hipStreamSynchronize(st.cs));
  g_t_kernel = t2.ms();
  Timer t3;
  RQ_TRY(fetch(prev_q0, prev_n, st.ev[(c - 1) & 1]));
  RQ_HIP(hipStreamSynchronize(st.xs));
  g_t_d2h = t3.ms();
  return RQ_OK;
}

// Results in the library's page-locked arrays (rq_host_alloc): the kernel can store them over PCIe itself (one launch, no
// copy back: 3.6 vs 3.8 ms at k = 1000).  Kernel stores cross PCIe at ~36 GB/s, the copy engine at ~45: from
// HOST_DIRECT_MAX_MB (256) of results on -- k = 10000: 800 MB -- chunked scans with copy-engine transfers behind them win
// (22.2 -> 20.0 ms; the floor is 800 MB / ~52 GB/s = 15.4 ms of PCIe plus the first chunk's scan).
static bool use_direct_results(const float *dists, const uint32_t *ids, int64_t nq, int k) {
  const size_t res_bytes = (size_t)nq * k * 8;
  const size_t direct_max = (size_t)(tuning("HOST_DIRECT_MAX_MB", 0) > 0 ? tuning("HOST_DIRECT_MAX_MB", 0) : 256) << 20;
  return tuning("HOST_DIRECT", 1) && res_bytes <= direct_max && host_pool_owns(dists, (size_t)nq * k * 4) &&
         host_pool_owns(ids, (size_t)nq * k * 4);
}

static int host_linscan(float *dists, uint32_t *ids, const uint8_t *codes, const float *centers,
                        const float *queries, const float *R, int64_t n, int64_t nq, int m, int d, int k,
                        int id_base) {
  Timer tt;
  g_t_h2d = g_t_kernel = g_t_d2h = 0;
  if (nq <= 0) return RQ_OK;
  if (n < 1 || m < 1 || d < m || d % m) return fail(RQ_EINVAL, "bad shape n=%lld m=%d d=%d", (long long)n, m, d);
  if (k < 1 || k > n) return fail(RQ_EINVAL, "k=%d must be in [1, n=%lld]", k, (long long)n);
  SavedDevice saved;      // RAYUELA_HIP_DEVICES may move this call to another device: the caller's device comes back
  {
    // RAYUELA_HIP_DEVICES lists more than one entry: row-shard the base over those devices (rq_index.hip)
    int devs[64];
    const int nd = env_devices(devs, 64);
    if (nd > 1) {
      const int rc = host_linscan_sharded(dists, ids, codes, centers, queries, R, n, nq, m, d, k, id_base, devs, nd);
      double tot, a, b, c;
      rq_last_timing(&tot, &a, &b, &c);
      set_timing(tt.ms(), a, b, c);
      return rc;
    }
    if (nd == 1) RQ_HIP(hipSetDevice(devs[0]));      // `saved` below puts the caller's device back on every exit path
  }
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;   // host-pointer calls on one device run one at a time (shared scratch + streams)
  DevBuf dcodes, dcent, dq, dr, drq, dd, di_;
  const size_t cb = (size_t)n * m, ce = (size_t)m * 256 * (d / m) * 4, qb = (size_t)nq * d * 4;
  RQ_TRY(dcodes.alloc(cb)); RQ_TRY(dcent.alloc(ce)); RQ_TRY(dq.alloc(qb));
  const bool direct = use_direct_results(dists, ids, nq, k);
  if (!direct) { RQ_TRY(dd.alloc((size_t)nq * k * 4)); RQ_TRY(di_.alloc((size_t)nq * k * 4)); }
  Timer t1;
  RQ_HIP(hipMemcpy(dcodes.p, codes, cb, hipMemcpyHostToDevice));
  RQ_HIP(hipMemcpy(dcent.p, centers, ce, hipMemcpyHostToDevice));
  RQ_HIP(hipMemcpy(dq.p, queries, qb, hipMemcpyHostToDevice));
  const float *qdev = dq.as<float>();
  if (R) {
    RQ_TRY(dr.alloc((size_t)d * d * 4)); RQ_TRY(drq.alloc(qb));
    RQ_HIP(hipMemcpy(dr.p, R, (size_t)d * d * 4, hipMemcpyHostToDevice));
  }
  g_t_h2d = t1.ms();
  if (R) {
    RQ_TRY(rotate_launch(drq.as<float>(), dr.as<float>(), dq.as<float>(), d, nq, di.num_cu, nullptr));
    qdev = drq.as<float>();
  }
  float *ddp = direct ? dists : dd.as<float>();       // direct: the kernel's own stores land in the caller's arrays
  uint32_t *dip = direct ? ids : di_.as<uint32_t>();
  const uint8_t *cdev = dcodes.as<uint8_t>();
  const float *cen = dcent.as<float>();
  // the query chunks below scan the same base: order it once for the whole call (rows of a tiled width only -- other
  // widths are padded per chunk scan and ordered there)
  const uint32_t *perm = nullptr;
  DevBuf dord;
  if (scan_padded_m(m) == m && order_pays(n, nq, k)) {
    RQ_TRY(dord.alloc(order_base_bytes(n, m)));
    RQ_TRY(order_base(&cdev, &perm, dord.p, cdev, n, m, nullptr));
  }
  ScanBase sbase;
  sbase.perm = perm;
  RQ_TRY(scan_and_fetch(dists, ids, direct ? nullptr : ddp, dip, nq, k, [&](int64_t q0, int64_t nqc, hipStream_t stream) {
    return dev_linscan(ddp + (size_t)q0 * k, dip + (size_t)q0 * k, nullptr, cdev, cen, qdev + (size_t)q0 * d, n, nqc, m,
                       d, k, 0, id_base, stream, LUT_PQ, nullptr, &sbase);
  }));
  g_t_total = tt.ms();
  return RQ_OK;
}

// linscan_lsq / linscan_cq on host pointers (src/Linscan.jl:118-193): codebooks [m*h][d], h = 256
static int host_linscan_aq(float *dists, uint32_t *ids, const uint8_t *codes, const float *queries,
                           const float *codebooks, const float *dbnorms, const float *R, int64_t n, int64_t nq,
                           int m, int h, int d, int k, int id_base, int lut_mode) {
  Timer tt;
  g_t_h2d = g_t_kernel = g_t_d2h = 0;
  if (nq <= 0) return RQ_OK;
  if (h != 256) return fail(RQ_EUNSUPPORTED, "the scan kernels cover h = 256 (uint8 codes); got h=%d", h);
  if (n < 1 || m < 1 || d < 1) return fail(RQ_EINVAL, "bad shape n=%lld m=%d d=%d", (long long)n, m, d);
  if (k < 1 || k > n) return fail(RQ_EINVAL, "k=%d must be in [1, n=%lld]", k, (long long)n);
  if (lut_mode == LUT_LSQ && !dbnorms) return fail(RQ_EINVAL, "dbnorms is NULL");
  SavedDevice saved;      // RAYUELA_HIP_DEVICES may move this call to another device: the caller's device comes back
  {
    // the additive-quantizer scans run on ONE device: the first entry of RAYUELA_HIP_DEVICES (row sharding over several
    // devices covers linscan_pq / linscan_opq, rq_index.hip)
    int devs[64];
    if (env_devices(devs, 64) >= 1) RQ_HIP(hipSetDevice(devs[0]));
  }
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;   // host-pointer calls on one device run one at a time (shared scratch + streams)
  DevBuf dcodes, dcb, dq, dn, dr, drq, dd, di_;
  const size_t cb = (size_t)n * m, ce = (size_t)m * 256 * d * 4, qb = (size_t)nq * d * 4;
  RQ_TRY(dcodes.alloc(cb)); RQ_TRY(dcb.alloc(ce)); RQ_TRY(dq.alloc(qb));
  const bool direct = use_direct_results(dists, ids, nq, k);
  if (!direct) { RQ_TRY(dd.alloc((size_t)nq * k * 4)); RQ_TRY(di_.alloc((size_t)nq * k * 4)); }
  Timer t1;
  RQ_HIP(hipMemcpy(dcodes.p, codes, cb, hipMemcpyHostToDevice));
  RQ_HIP(hipMemcpy(dcb.p, codebooks, ce, hipMemcpyHostToDevice));
  RQ_HIP(hipMemcpy(dq.p, queries, qb, hipMemcpyHostToDevice));
  if (dbnorms) {
    RQ_TRY(dn.alloc((size_t)n * 4));
    RQ_HIP(hipMemcpy(dn.p, dbnorms, (size_t)n * 4, hipMemcpyHostToDevice));
  }
  const float *qdev = dq.as<float>();
  if (R) {
    RQ_TRY(dr.alloc((size_t)d * d * 4)); RQ_TRY(drq.alloc(qb));
    RQ_HIP(hipMemcpy(dr.p, R, (size_t)d * d * 4, hipMemcpyHostToDevice));
  }
  g_t_h2d = t1.ms();
  if (R) {
    RQ_TRY(rotate_launch(drq.as<float>(), dr.as<float>(), dq.as<float>(), d, nq, di.num_cu, nullptr));
    qdev = drq.as<float>();
  }
  float *ddp = direct ? dists : dd.as<float>();
  uint32_t *dip = direct ? ids : di_.as<uint32_t>();
  const uint8_t *cdev = dcodes.as<uint8_t>();
  const float *cbk = dcb.as<float>();
  const float *nrm = dbnorms ? dn.as<float>() : nullptr;
  RQ_TRY(scan_and_fetch(dists, ids, direct ? nullptr : ddp, dip, nq, k, [&](int64_t q0, int64_t nqc, hipStream_t stream) {
    return dev_linscan(ddp + (size_t)q0 * k, dip + (size_t)q0 * k, nullptr, cdev, cbk, qdev + (size_t)q0 * d, n, nqc, m,
                       d, k, 0, id_base, stream, lut_mode, nrm);
  }));
  g_t_total = tt.ms();
  return RQ_OK;
}

// quantize_pq / quantize_opq of rows [0, n) of a HOST matrix on the current device.  The rows are uploaded in
// chunks of ~128 MB on a transfer stream while the previous chunk is rotated / encoded on the compute stream
// (two device buffers; the codes stay on the device and come back in ONE copy at the end -- a pageable D2H per
// chunk would make the host wait for each kernel and serialise the pipeline).  PCIe is the bound of this call
// (57 GB/s from pageable memory): the kernels now hide behind the uploads instead of adding to them.
static int encode_host_rows(uint8_t *codes, int16_t *codes1, const float *X, const float *R, const float *C,
                            int64_t n, int d, int m, int h, double *t_h2d, double *t_tail) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;   // host-pointer calls on one device run one at a time (shared scratch + streams)
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, std::max<int64_t>(32768, (1LL << 27) / ((int64_t)d * 4))));
  const bool piped = tuning("HOST_OVERLAP", 1) && n > chunk;
  DevBuf dX[2], dRX, dR, dC, dcodes, d16;
  RQ_TRY(dX[0].alloc((size_t)chunk * d * 4));
  if (piped) RQ_TRY(dX[1].alloc((size_t)chunk * d * 4));
  RQ_TRY(dC.alloc((size_t)h * d * 4));
  RQ_TRY(dcodes.alloc((size_t)n * m));
  if (codes1) RQ_TRY(d16.alloc((size_t)n * m * 2));
  RQ_HIP(hipMemcpy(dC.p, C, (size_t)h * d * 4, hipMemcpyHostToDevice));
  if (R) {
    RQ_TRY(dR.alloc((size_t)d * d * 4));
    RQ_TRY(dRX.alloc((size_t)chunk * d * 4));
    RQ_HIP(hipMemcpy(dR.p, R, (size_t)d * d * 4, hipMemcpyHostToDevice));
  }
  hipStream_t cs = nullptr, xs = nullptr;
  RQ_TRY(aux_streams(&cs, &xs));
  struct Events {
    hipEvent_t up[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    ~Events() { for (int i = 0; i < 2; ++i) { if (up[i]) (void)hipEventDestroy(up[i]); if (done[i]) (void)hipEventDestroy(done[i]); } }
  } ev;
  for (int i = 0; i < 2; ++i) {
    RQ_HIP(hipEventCreateWithFlags(&ev.up[i], hipEventDisableTiming));
    RQ_HIP(hipEventCreateWithFlags(&ev.done[i], hipEventDisableTiming));
  }
  RQ_HIP(hipDeviceSynchronize());      // the small uploads on the null stream are done
  int k = 0;
  for (int64_t r0 = 0; r0 < n; r0 += chunk, ++k) {
    const int64_t nr = std::min(chunk, n - r0);
    const int b = piped ? (k & 1) : 0;
    if (k >= (piped ? 2 : 1)) RQ_HIP(hipEventSynchronize(ev.done[b]));   // the kernels that read this buffer are done
    Timer t1;
    RQ_HIP(hipMemcpyAsync(dX[b].p, X + (size_t)r0 * d, (size_t)nr * d * 4, hipMemcpyHostToDevice, xs));
    RQ_HIP(hipEventRecord(ev.up[b], xs));
    *t_h2d += t1.ms();
    RQ_HIP(hipStreamWaitEvent(cs, ev.up[b], 0));
    const float *src = dX[b].as<float>();
    if (R) {
      RQ_TRY(rotate_launch(dRX.as<float>(), dR.as<float>(), dX[b].as<float>(), d, nr, di.num_cu, cs));
      src = dRX.as<float>();
    }
    RQ_TRY(encode_launch(dcodes.as<uint8_t>() + (size_t)r0 * m, src, dC.as<float>(), nr, d, m, h, di.num_cu, cs));
    RQ_HIP(hipEventRecord(ev.done[b], cs));
  }
  Timer t2;
  if (codes1) RQ_TRY(widen_codes_launch(d16.as<int16_t>(), dcodes.as<uint8_t>(), n * m, cs));
  RQ_HIP(hipStreamSynchronize(cs));
  if (codes1) RQ_HIP(hipMemcpy(codes1, d16.p, (size_t)n * m * 2, hipMemcpyDeviceToHost));
  else RQ_HIP(hipMemcpy(codes, dcodes.p, (size_t)n * m, hipMemcpyDeviceToHost));
  *t_tail += t2.ms();
  return RQ_OK;
}

static int host_encode(uint8_t *codes, int16_t *codes1, const float *X, const float *R, const float *C,
                       int64_t n, int d, int m, int h) {
  Timer tt;
  g_t_h2d = g_t_kernel = g_t_d2h = 0;
  if (n <= 0) return RQ_OK;
  if (d < 1 || m < 1 || h < 1) return fail(RQ_EINVAL, "bad shape d=%d m=%d h=%d", d, m, h);
  int devs[64];
  const int nd = env_devices(devs, 64);
  if (nd > 1 && n >= 2 * nd) {
    // RAYUELA_HIP_DEVICES lists several devices: the rows are independent, so every device encodes its own
    // contiguous share.  One host thread per device -- pageable uploads block the issuing thread, and every GPU
    // has its own PCIe link, so the host-to-device rate scales with the devices.
    std::vector<std::thread> th;
    std::vector<int> rc(nd, RQ_OK);
    std::vector<std::string> msg(nd);
    std::vector<double> h2d(nd, 0.0), tail(nd, 0.0);
    const int64_t per = n / nd, extra = n % nd;
    int64_t row = 0;
    for (int i = 0; i < nd; ++i) {
      const int64_t cnt = per + (i < extra ? 1 : 0), r0 = row;
      row += cnt;
      th.emplace_back([&, i, r0, cnt]() {
        if (hipSetDevice(devs[i]) != hipSuccess) { rc[i] = RQ_ENODEVICE; msg[i] = "hipSetDevice failed"; return; }
        rc[i] = encode_host_rows(codes ? codes + (size_t)r0 * m : nullptr, codes1 ? codes1 + (size_t)r0 * m : nullptr,
                                 X + (size_t)r0 * d, R, C, cnt, d, m, h, &h2d[i], &tail[i]);
        if (rc[i] != RQ_OK) msg[i] = g_err;
      });
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < nd; ++i)
      if (rc[i] != RQ_OK) return fail(rc[i], "device %d: %s", devs[i], msg[i].c_str());
    g_t_h2d = *std::max_element(h2d.begin(), h2d.end());
    g_t_d2h = *std::max_element(tail.begin(), tail.end());
    g_t_total = tt.ms();
    g_t_kernel = 0;
    return RQ_OK;
  }
  SavedDevice saved;
  if (nd == 1) RQ_HIP(hipSetDevice(devs[0]));
  double h2d = 0, tail = 0;
  RQ_TRY(encode_host_rows(codes, codes1, X, R, C, n, d, m, h, &h2d, &tail));
  g_t_h2d = h2d;
  g_t_d2h = tail;          // what is left after the last upload: last chunk's kernels + the one copy back
  g_t_total = tt.ms();
  g_t_kernel = std::max(0.0, g_t_total - h2d - tail);
  return RQ_OK;
}

}  // namespace rq

using namespace rq;

extern "C" {

#ifndef RQ_BUILD_ID
#define RQ_BUILD_ID "unknown"
#endif
const char *rq_version(void) { return "rayuela-hip 0.1 (gfx950) build " RQ_BUILD_ID; }
const char *rq_last_scan_kernel(void) { return last_scan_kernel_name(); }
const char *rq_last_error(void) { return g_err; }

int rq_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

int rq_set_device(int device) {
  RQ_HIP(hipSetDevice(device));
  return RQ_OK;
}

int rq_set_tuning(const char *key, int value) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < g_nknobs; ++i)
    if (!strcmp(g_knobs[i].key, key)) { g_knobs[i].value = value; return RQ_OK; }
  if (g_nknobs >= 64) return fail(RQ_EINVAL, "too many tuning keys");
  strncpy(g_knobs[g_nknobs].key, key, 31);
  g_knobs[g_nknobs].key[31] = 0;
  g_knobs[g_nknobs++].value = value;
  return RQ_OK;
}

int rq_scan_plan(int64_t n, int64_t nq, int m, int d, int k, int num_cu, int64_t *out8) {
  if (!out8 || n < 1 || nq < 1 || k < 1 || num_cu < 1) return fail(RQ_EINVAL, "rq_scan_plan: bad arguments");
  ScanPlan pl;
  RQ_TRY(scan_plan(pl, n, nq, m, d, k, num_cu, tuning("SCAN_SLICES", 0)));
  out8[0] = pl.qg; out8[1] = pl.ngroups; out8[2] = pl.whole; out8[3] = pl.nslices; out8[4] = pl.rows_per_slice;
  out8[5] = pl.grid; out8[6] = pl.cap; out8[7] = (pl.bigk ? 1 : 0) | (pl.xcd ? 2 : 0);
  return RQ_OK;
}

int rq_scan_orders_in_call(int64_t n, int64_t nq, int k) {
  // would a raw-pointer scan of this shape (rq_dev_linscan / rq_linscan_pq, PQ tables) order a scratch copy of the base itself?
  // 2: ... and balance it (greedy pass of rq_order.hip: from ORDER_GREEDY_MIN_NQ queries on)
  if (!(order_pays(n, nq, k) &&
        order_base_bytes(n, 8) + (size_t)n * 4 <= (size_t)std::max(0, tuning("ORDER_MAX_SCRATCH_MB", 2048)) * 1048576ull))
    return 0;
  return tuning("ORDER_GREEDY", 1) && nq >= tuning("ORDER_GREEDY_MIN_NQ", 16384) ? 2 : 1;
}

int rq_scan_stats(unsigned long long *out8) {
  // diagnostics: phase cycle counters of the last scan launched with tuning SCAN_STATS=1
  void *counter = nullptr;
  RQ_TRY(workspace(WS_COUNTER, WS_COUNTER_BYTES, &counter, nullptr));
  RQ_HIP(hipDeviceSynchronize());
  RQ_HIP(hipMemcpy(out8, (char *)counter + 64, 128, hipMemcpyDeviceToHost));
  return RQ_OK;
}

int rq_scan_finish_stats(unsigned long long *out8) {
  // diagnostics: the counters behind the 16 of rq_scan_stats (same launch, same tuning)
  if (!out8) return fail(RQ_EINVAL, "rq_scan_finish_stats: null output");
  void *counter = nullptr;
  RQ_TRY(workspace(WS_COUNTER, WS_COUNTER_BYTES, &counter, nullptr));
  RQ_HIP(hipDeviceSynchronize());
  RQ_HIP(hipMemcpy(out8, (char *)counter + 64 + 128, 64, hipMemcpyDeviceToHost));
  return RQ_OK;
}

int rq_release_workspaces(void) {
  rq::sharded_cache_release();
  rq::host_pool_trim();
  return release_workspaces();
}

void *rq_host_alloc(size_t bytes) { return rq::host_pool_alloc(bytes); }
void rq_host_free(void *p) { rq::host_pool_free(p); }

int rq_last_timing(double *total_ms, double *h2d_ms, double *kernel_ms, double *d2h_ms) {
  if (total_ms) *total_ms = g_t_total;
  if (h2d_ms) *h2d_ms = g_t_h2d;
  if (kernel_ms) *kernel_ms = g_t_kernel;
  if (d2h_ms) *d2h_ms = g_t_d2h;
  return RQ_OK;
}

void linscan_aqd_query(float *dists, unsigned int *res, unsigned char *codes, float *centers, float *queries,
                       int N, unsigned int NQ, int B, int K, int dim1codes, int dim1queries, int subdim) {
  // deps/src/linscan_aqd.cpp:48 -- the reference zeroes dists first; keep that on failure too
  if (dists && K > 0) memset(dists, 0, (size_t)K * NQ * sizeof(float));
  const int m = B / 8;
  int rc;
  if (m < 1 || dim1codes != m || dim1queries != m * subdim) {
    rc = fail(RQ_EINVAL, "linscan_aqd_query: expects dim1codes == B/8 and dim1queries == (B/8)*subdim "
                         "(B=%d dim1codes=%d dim1queries=%d subdim=%d)", B, dim1codes, dim1queries, subdim);
  } else {
    rc = host_linscan(dists, res, codes, centers, queries, nullptr, N, NQ, m, dim1queries, K, 0);
  }
  if (rc != RQ_OK) fprintf(stderr, "librayuela_hip: linscan_aqd_query failed (%d): %s\n", rc, g_err);
}

void linscan_aqd_query_extra_byte(float *dists, int *idx, unsigned char *codes, float *queries, float *codebooks,
                                  float *dbnorms, int nqueries, int ncodes, int m, int h, int d, int nn) {
  const int rc = host_linscan_aq(dists, (uint32_t *)idx, codes, queries, codebooks, dbnorms, nullptr, ncodes, nqueries,
                                 m, h, d, nn, 1, LUT_LSQ);   // ids ONE-based (linscan_aqd_pairwise_byte.cpp:76)
  if (rc != RQ_OK) fprintf(stderr, "librayuela_hip: linscan_aqd_query_extra_byte failed (%d): %s\n", rc, g_err);
}

void linscan_aqd_cq_query_extra_byte(float *dists, int *idx, unsigned char *codes, float *queries, float *codebooks,
                                     int nqueries, int ncodes, int m, int h, int d, int nn) {
  const int rc = host_linscan_aq(dists, (uint32_t *)idx, codes, queries, codebooks, nullptr, nullptr, ncodes, nqueries,
                                 m, h, d, nn, 1, LUT_CQ);
  if (rc != RQ_OK) fprintf(stderr, "librayuela_hip: linscan_aqd_cq_query_extra_byte failed (%d): %s\n", rc, g_err);
}

int rq_linscan_lsq(float *dists, uint32_t *ids, const uint8_t *codes, const float *queries, const float *codebooks,
                   const float *dbnorms, const float *R, int64_t n, int64_t nq, int m, int h, int d, int k,
                   int id_base) {
  return host_linscan_aq(dists, ids, codes, queries, codebooks, dbnorms, R, n, nq, m, h, d, k, id_base, LUT_LSQ);
}

// ---- linscan_lsq over a PREPARED base: codes, norms, codebooks resident, the filter's O(n) preprocessing done once ---------
struct rq_lsq_index_impl {
  int device = 0;
  int64_t n = 0;
  int m = 0, mp = 0, d = 0;
  uint8_t *codes = nullptr;      // [n][mp] (zero-padded to a tiled row width)
  float *cb = nullptr, *norms = nullptr;
  uint8_t *normb = nullptr;      // prepared norm buffer (lsq_norm_bytes) or nullptr when the filter does not apply
  void *ordered = nullptr;       // codes in bank-aware row order + perm (order_base), norms permuted alike; else nullptr
  const uint8_t *ocodes = nullptr;
  const uint32_t *perm = nullptr;
};

static void lsq_free(rq_lsq_index_impl *ix) {
  if (!ix) return;
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  if (hipSetDevice(ix->device) == hipSuccess) {
    (void)hipDeviceSynchronize();
    if (ix->codes) (void)hipFree(ix->codes);
    if (ix->cb) (void)hipFree(ix->cb);
    if (ix->norms) (void)hipFree(ix->norms);
    if (ix->normb) (void)hipFree(ix->normb);
    if (ix->ordered) (void)hipFree(ix->ordered);
  }
  if (have) (void)hipSetDevice(cur);
  (void)hipGetLastError();
  delete ix;
}

rq_lsq_index *rq_lsq_prepare(const uint8_t *codes, const float *codebooks, const float *dbnorms, int64_t n, int m, int h,
                             int d) {
  if (!codes || !codebooks || !dbnorms) { fail(RQ_EINVAL, "rq_lsq_prepare: NULL argument"); return nullptr; }
  if (h != 256) { fail(RQ_EUNSUPPORTED, "the scan kernels cover h = 256 (uint8 codes); got h=%d", h); return nullptr; }
  if (n < 1 || n >= (1LL << 31) || m < 1 || d < 1) { fail(RQ_EINVAL, "rq_lsq_prepare: bad shape n=%lld m=%d d=%d", (long long)n, m, d); return nullptr; }
  const int mp = scan_padded_m(m);
  if (mp < 0) { fail(RQ_EUNSUPPORTED, "the ADC scan kernels cover 1 <= m <= 64 sub-quantizers; got m=%d", m); return nullptr; }
  // device placement as in the one-shot call (rq_linscan_lsq / host_linscan_aq): the first device of RAYUELA_HIP_DEVICES when the
  // list is set, else the calling thread's current device; the thread's device is restored on every exit path
  SavedDevice saved;
  int devs[64];
  const int nd = env_devices(devs, 64);
  if (nd > 0 && hipSetDevice(devs[0]) != hipSuccess) { (void)hipGetLastError(); fail(RQ_EINVAL, "rq_lsq_prepare: cannot select device %d", devs[0]); return nullptr; }
  DeviceInfo di;
  if (device_info(&di) != RQ_OK) return nullptr;
  DeviceLock lock;
  rq_lsq_index_impl *ix = new rq_lsq_index_impl;
  ix->device = di.device; ix->n = n; ix->m = m; ix->mp = mp; ix->d = d;
  auto body = [&]() -> int {
    RQ_HIP(hipMalloc((void **)&ix->codes, (size_t)n * mp));
    RQ_HIP(hipMalloc((void **)&ix->cb, (size_t)m * 256 * d * 4));
    RQ_HIP(hipMalloc((void **)&ix->norms, (size_t)n * 4));
    RQ_HIP(hipMemcpy(ix->cb, codebooks, (size_t)m * 256 * d * 4, hipMemcpyHostToDevice));
    RQ_HIP(hipMemcpy(ix->norms, dbnorms, (size_t)n * 4, hipMemcpyHostToDevice));
    if (mp == m) {
      RQ_HIP(hipMemcpy(ix->codes, codes, (size_t)n * m, hipMemcpyHostToDevice));
    } else {
      DevBuf raw;
      RQ_TRY(raw.alloc((size_t)n * m));
      RQ_HIP(hipMemcpy(raw.p, codes, (size_t)n * m, hipMemcpyHostToDevice));
      RQ_TRY(pad_codes_launch(ix->codes, raw.as<uint8_t>(), n, m, mp, nullptr));
      RQ_HIP(hipDeviceSynchronize());
    }
    // the base is resident: bank-aware row order once (rq_order.hip), norms permuted alike -- row_bias and the filter's
    // norm bytes are indexed by POSITION in the kernel, ids come from perm
    // (an ordered copy that cannot be allocated or built leaves the base in arrival order: slower gathers, same answer)
    if (tuning("INDEX_ORDER", 1) && order_pays(n, 0, 0) && hipMalloc(&ix->ordered, order_base_bytes(n, mp)) != hipSuccess) {
      (void)hipGetLastError();      // out of memory: the base stays in arrival order
      ix->ordered = nullptr;
    }
    if (ix->ordered) {
      const int orc = order_base(&ix->ocodes, &ix->perm, ix->ordered, ix->codes, n, mp, nullptr);
      if (is_oom(orc)) { forgive_oom(); ix->perm = nullptr; }      // (the key scratch)
      else if (orc != RQ_OK) return orc;                           // launch errors surface
      if (ix->perm) {
        float *pn = nullptr;
        RQ_HIP(hipMalloc((void **)&pn, (size_t)n * 4));
        RQ_TRY(gather_f32_launch(pn, ix->norms, ix->perm, n, nullptr));
        RQ_HIP(hipDeviceSynchronize());
        RQ_HIP(hipFree(ix->norms));
        ix->norms = pn;
        RQ_HIP(hipFree(ix->codes));
        ix->codes = nullptr;
      } else {
        RQ_HIP(hipFree(ix->ordered));
        ix->ordered = nullptr;
      }
    }
    const uint8_t *cur = ix->perm ? ix->ocodes : ix->codes;
    if ((mp == 8 || mp == 16) && tuning("SCAN_FILTER", 1) && tuning("SCAN_FILTER_LSQ", 1)) {
      RQ_HIP(hipMalloc((void **)&ix->normb, lsq_norm_bytes(n)));
      RQ_TRY(lsq_norm_prepare(ix->normb, cur, ix->cb, ix->norms, n, mp, m, d, nullptr));
      RQ_HIP(hipDeviceSynchronize());
    }
    return RQ_OK;
  };
  if (body() != RQ_OK) { lsq_free(ix); return nullptr; }
  return reinterpret_cast<rq_lsq_index *>(ix);
}

void rq_lsq_release(rq_lsq_index *handle) { lsq_free(reinterpret_cast<rq_lsq_index_impl *>(handle)); }

int rq_lsq_search(rq_lsq_index *handle, float *dists, uint32_t *ids, const float *queries, const float *R, int64_t nq, int k,
                  int id_base) {
  rq_lsq_index_impl *ix = reinterpret_cast<rq_lsq_index_impl *>(handle);
  if (!ix) return fail(RQ_EINVAL, "rq_lsq_search: NULL handle");
  if (nq <= 0) return RQ_OK;
  if (!dists || !ids || !queries) return fail(RQ_EINVAL, "rq_lsq_search: NULL argument");
  if (k < 1 || k > ix->n) return fail(RQ_EINVAL, "k=%d must be in [1, n=%lld]", k, (long long)ix->n);
  Timer tt;
  g_t_h2d = g_t_kernel = g_t_d2h = 0;
  SavedDevice saved;
  RQ_HIP(hipSetDevice(ix->device));
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;
  const int d = ix->d;
  DevBuf dq, dr, drq, dd, di_;
  const size_t qb = (size_t)nq * d * 4;
  RQ_TRY(dq.alloc(qb));
  const bool direct = use_direct_results(dists, ids, nq, k);
  if (!direct) { RQ_TRY(dd.alloc((size_t)nq * k * 4)); RQ_TRY(di_.alloc((size_t)nq * k * 4)); }
  Timer t1;
  RQ_HIP(hipMemcpy(dq.p, queries, qb, hipMemcpyHostToDevice));
  const float *qdev = dq.as<float>();
  if (R) {
    RQ_TRY(dr.alloc((size_t)d * d * 4)); RQ_TRY(drq.alloc(qb));
    RQ_HIP(hipMemcpy(dr.p, R, (size_t)d * d * 4, hipMemcpyHostToDevice));
  }
  g_t_h2d = t1.ms();
  if (R) {
    RQ_TRY(rotate_launch(drq.as<float>(), dr.as<float>(), dq.as<float>(), d, nq, di.num_cu, nullptr));
    qdev = drq.as<float>();
  }
  float *ddp = direct ? dists : dd.as<float>();
  uint32_t *dip = direct ? ids : di_.as<uint32_t>();
  RQ_TRY(scan_and_fetch(dists, ids, direct ? nullptr : ddp, dip, nq, k, [&](int64_t q0, int64_t nqc, hipStream_t stream) {
    ScanBase sb;
    sb.padded = true;
    sb.norm_prepared = ix->normb;
    sb.perm = ix->perm;
    return dev_linscan(ddp + (size_t)q0 * k, dip + (size_t)q0 * k, nullptr, ix->perm ? ix->ocodes : ix->codes, ix->cb, qdev + (size_t)q0 * d, ix->n, nqc,
                       ix->m, d, k, 0, id_base, stream, LUT_LSQ, ix->norms, &sb);
  }));
  g_t_total = tt.ms();
  return RQ_OK;
}

int rq_linscan_cq(float *dists, uint32_t *ids, const uint8_t *codes, const float *queries, const float *codebooks,
                  int64_t n, int64_t nq, int m, int h, int d, int k, int id_base) {
  return host_linscan_aq(dists, ids, codes, queries, codebooks, nullptr, nullptr, n, nq, m, h, d, k, id_base, LUT_CQ);
}

int rq_dev_linscan_aq(float *dists, uint32_t *ids, uint64_t *keys, const uint8_t *codes, const float *codebooks,
                      const float *queries, const float *dbnorms, int64_t n, int64_t nq, int m, int d, int k,
                      int lut_mode, uint32_t id_offset, int id_base, void *stream) {
  if (lut_mode != LUT_LSQ && lut_mode != LUT_CQ) return fail(RQ_EINVAL, "lut_mode must be 1 (LSQ) or 2 (CQ)");
  if (lut_mode == LUT_LSQ && !dbnorms) return fail(RQ_EINVAL, "dbnorms is NULL");
  return dev_linscan(dists, ids, keys, codes, codebooks, queries, n, nq, m, d, k, id_offset, id_base,
                     (hipStream_t)stream, lut_mode, lut_mode == LUT_LSQ ? dbnorms : nullptr);
}

int rq_linscan_pq(float *dists, uint32_t *ids, const uint8_t *codes, const float *centers, const float *queries,
                  int64_t n, int64_t nq, int m, int d, int k, int id_base) {
  return host_linscan(dists, ids, codes, centers, queries, nullptr, n, nq, m, d, k, id_base);
}

int rq_linscan_opq(float *dists, uint32_t *ids, const uint8_t *codes, const float *centers, const float *queries,
                   const float *R, int64_t n, int64_t nq, int m, int d, int k, int id_base) {
  if (!R) return fail(RQ_EINVAL, "R is NULL");
  return host_linscan(dists, ids, codes, centers, queries, R, n, nq, m, d, k, id_base);
}

int rq_encode_pq(uint8_t *codes, const float *X, const float *C, int64_t n, int d, int m, int h) {
  return host_encode(codes, nullptr, X, nullptr, C, n, d, m, h);
}
int rq_encode_opq(uint8_t *codes, const float *X, const float *R, const float *C, int64_t n, int d, int m, int h) {
  if (!R) return fail(RQ_EINVAL, "R is NULL");
  return host_encode(codes, nullptr, X, R, C, n, d, m, h);
}
int rq_encode_pq_i16(int16_t *codes1, const float *X, const float *C, int64_t n, int d, int m, int h) {
  return host_encode(nullptr, codes1, X, nullptr, C, n, d, m, h);
}
int rq_encode_opq_i16(int16_t *codes1, const float *X, const float *R, const float *C, int64_t n, int d, int m,
                      int h) {
  if (!R) return fail(RQ_EINVAL, "R is NULL");
  return host_encode(nullptr, codes1, X, R, C, n, d, m, h);
}

// ---- resident dataset: X uploaded once, encoded as often as needed ------------------------------------------
struct rq_dataset_impl {
  int device, d;
  int64_t n;
  float *X, *RX;
};

rq_dataset *rq_dataset_upload(const float *X, int64_t n, int d) {
  if (!X || n < 1 || d < 1) { fail(RQ_EINVAL, "rq_dataset_upload: bad arguments"); return nullptr; }
  DeviceInfo di;
  if (device_info(&di) != RQ_OK) return nullptr;
  rq_dataset_impl *ds = new rq_dataset_impl{di.device, d, n, nullptr, nullptr};
  if (hipMalloc((void **)&ds->X, (size_t)n * d * 4) != hipSuccess ||
      hipMemcpy(ds->X, X, (size_t)n * d * 4, hipMemcpyHostToDevice) != hipSuccess) {
    fail(RQ_ENODEVICE, "rq_dataset_upload: cannot place %lld x %d floats on device %d", (long long)n, d, di.device);
    (void)hipGetLastError();
    if (ds->X) (void)hipFree(ds->X);
    delete ds;
    return nullptr;
  }
  return reinterpret_cast<rq_dataset *>(ds);
}

int rq_dataset_encode(rq_dataset *handle, uint8_t *codes, int16_t *codes1, const float *R, const float *C, int m, int h) {
  rq_dataset_impl *ds = reinterpret_cast<rq_dataset_impl *>(handle);
  if (!ds || !C || (!codes && !codes1)) return fail(RQ_EINVAL, "rq_dataset_encode: bad arguments");
  if (m < 1 || h < 1 || ds->d < m) return fail(RQ_EINVAL, "bad shape d=%d m=%d h=%d", ds->d, m, h);
  Timer tt;
  int cur = 0;
  RQ_HIP(hipGetDevice(&cur));
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{cur};
  RQ_HIP(hipSetDevice(ds->device));
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;
  const int d = ds->d;
  const int64_t n = ds->n;
  DevBuf dC, dR, dcodes, d16;
  RQ_TRY(dC.alloc((size_t)h * d * 4)); RQ_TRY(dcodes.alloc((size_t)n * m));
  RQ_HIP(hipMemcpy(dC.p, C, (size_t)h * d * 4, hipMemcpyHostToDevice));
  const float *src = ds->X;
  if (R) {
    RQ_TRY(dR.alloc((size_t)d * d * 4));
    RQ_HIP(hipMemcpy(dR.p, R, (size_t)d * d * 4, hipMemcpyHostToDevice));
    if (!ds->RX) RQ_HIP(hipMalloc((void **)&ds->RX, (size_t)n * d * 4));
    RQ_TRY(rotate_launch(ds->RX, dR.as<float>(), ds->X, d, n, di.num_cu, nullptr));
    src = ds->RX;
  }
  RQ_TRY(encode_launch(dcodes.as<uint8_t>(), src, dC.as<float>(), n, d, m, h, di.num_cu, nullptr));
  if (codes1) {
    RQ_TRY(d16.alloc((size_t)n * m * 2));
    RQ_TRY(widen_codes_launch(d16.as<int16_t>(), dcodes.as<uint8_t>(), n * m, nullptr));
  }
  RQ_HIP(hipDeviceSynchronize());
  g_t_h2d = 0;
  g_t_kernel = tt.ms();
  Timer t3;
  if (codes1) RQ_HIP(hipMemcpy(codes1, d16.p, (size_t)n * m * 2, hipMemcpyDeviceToHost));
  if (codes) RQ_HIP(hipMemcpy(codes, dcodes.p, (size_t)n * m, hipMemcpyDeviceToHost));
  g_t_d2h = t3.ms();
  g_t_total = tt.ms();
  return RQ_OK;
}

void rq_dataset_free(rq_dataset *handle) {
  rq_dataset_impl *ds = reinterpret_cast<rq_dataset_impl *>(handle);
  if (!ds) return;
  int cur = 0;
  const bool have = hipGetDevice(&cur) == hipSuccess;
  if (hipSetDevice(ds->device) == hipSuccess) {
    if (ds->X) (void)hipFree(ds->X);
    if (ds->RX) (void)hipFree(ds->RX);
  }
  if (have) (void)hipSetDevice(cur);
  (void)hipGetLastError();
  delete ds;
}

int rq_rotate_T(float *RX, const float *R, const float *X, int d, int64_t n) {
  if (n <= 0) return RQ_OK;
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DevBuf dX, dR, dRX;
  RQ_TRY(dX.alloc((size_t)n * d * 4)); RQ_TRY(dRX.alloc((size_t)n * d * 4)); RQ_TRY(dR.alloc((size_t)d * d * 4));
  RQ_HIP(hipMemcpy(dX.p, X, (size_t)n * d * 4, hipMemcpyHostToDevice));
  RQ_HIP(hipMemcpy(dR.p, R, (size_t)d * d * 4, hipMemcpyHostToDevice));
  RQ_TRY(rotate_launch(dRX.as<float>(), dR.as<float>(), dX.as<float>(), d, n, di.num_cu, nullptr));
  RQ_HIP(hipMemcpy(RX, dRX.p, (size_t)n * d * 4, hipMemcpyDeviceToHost));
  return RQ_OK;
}

int rq_dev_encode_pq(uint8_t *codes, const float *X, const float *C, int64_t n, int d, int m, int h, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return encode_launch(codes, X, C, n, d, m, h, di.num_cu, (hipStream_t)stream);
}

int rq_dev_encode_pq_filter_w(uint8_t *codes, float *W, const float *X, const float *C, int64_t n, int d, int m, int h,
                              void *stream) {
  if (!W) return fail(RQ_EINVAL, "W is NULL");
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return encode_launch(codes, X, C, n, d, m, h, di.num_cu, (hipStream_t)stream, W);
}

const char *rq_last_encode_kernel(void) { return last_encode_kernel_name(); }
int rq_last_encode_stats(uint64_t *out2) {
  if (!out2) return fail(RQ_EINVAL, "rq_last_encode_stats: NULL");
  unsigned long long t[2];
  last_encode_stats(t);
  out2[0] = t[0]; out2[1] = t[1];
  return RQ_OK;
}

int rq_dev_rotate_T(float *RX, const float *R, const float *X, int d, int64_t n, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return rotate_launch(RX, R, X, d, n, di.num_cu, (hipStream_t)stream);
}

int rq_dev_encode_opq(uint8_t *codes, const float *X, const float *R, const float *C, int64_t n, int d, int m,
                      int h, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  void *tmp = nullptr;
  RQ_TRY(workspace(WS_TMP, (size_t)n * d * 4, &tmp, (hipStream_t)stream));
  RQ_TRY(rotate_launch((float *)tmp, R, X, d, n, di.num_cu, (hipStream_t)stream));
  return encode_launch(codes, (const float *)tmp, C, n, d, m, h, di.num_cu, (hipStream_t)stream);
}

int rq_dev_encode_rvq(uint8_t *codes, float *Xr, const float *codebooks, int64_t n, int d, int m, int h,
                      uint32_t *counts, void *stream) {
  if (n <= 0) return RQ_OK;
  if (d < 1 || m < 1 || h < 1 || h > 256) return fail(RQ_EINVAL, "rvq: bad shape d=%d m=%d h=%d", d, m, h);
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  void *tmp = nullptr;
  RQ_TRY(workspace(WS_TMP, (size_t)n, &tmp, (hipStream_t)stream));
  return rvq_encode_launch(codes, Xr, (uint8_t *)tmp, counts, codebooks, n, d, m, h, di.num_cu, (hipStream_t)stream);
}

static int host_encode_rvq(uint8_t *codes, int16_t *codes1, const float *X, const float *C, int64_t n, int d, int m,
                           int h, uint32_t *counts, float *Xr_out) {
  Timer tt;
  g_t_h2d = g_t_kernel = g_t_d2h = 0;
  if (counts) memset(counts, 0, (size_t)m * h * sizeof(uint32_t));
  if (n <= 0) return RQ_OK;
  if (d < 1 || m < 1 || h < 1 || h > 256) return fail(RQ_EINVAL, "rvq: bad shape d=%d m=%d h=%d", d, m, h);
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;   // host-pointer calls on one device run one at a time (shared scratch + streams)
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (1LL << 30) / ((int64_t)d * 4)));
  DevBuf dX, dC, dcodes, dstage, d16, dcnt;
  RQ_TRY(dX.alloc((size_t)chunk * d * 4));
  RQ_TRY(dC.alloc((size_t)m * h * d * 4));
  RQ_TRY(dcodes.alloc((size_t)chunk * m));
  RQ_TRY(dstage.alloc((size_t)chunk));
  RQ_TRY(dcnt.alloc((size_t)m * h * 4));
  if (codes1) RQ_TRY(d16.alloc((size_t)chunk * m * 2));
  RQ_HIP(hipMemcpy(dC.p, C, (size_t)m * h * d * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> part(counts ? (size_t)m * h : 0);
  for (int64_t r0 = 0; r0 < n; r0 += chunk) {
    const int64_t nr = std::min(chunk, n - r0);
    Timer t1;
    RQ_HIP(hipMemcpy(dX.p, X + (size_t)r0 * d, (size_t)nr * d * 4, hipMemcpyHostToDevice));
    g_t_h2d += t1.ms();
    Timer t2;
    RQ_TRY(rvq_encode_launch(dcodes.as<uint8_t>(), dX.as<float>(), dstage.as<uint8_t>(),
                             counts ? dcnt.as<unsigned int>() : nullptr, dC.as<float>(), nr, d, m, h, di.num_cu,
                             nullptr));
    if (codes1) RQ_TRY(widen_codes_launch(d16.as<int16_t>(), dcodes.as<uint8_t>(), nr * m, nullptr));
    RQ_HIP(hipDeviceSynchronize());
    g_t_kernel += t2.ms();
    Timer t3;
    if (codes1)
      RQ_HIP(hipMemcpy(codes1 + (size_t)r0 * m, d16.p, (size_t)nr * m * 2, hipMemcpyDeviceToHost));
    else
      RQ_HIP(hipMemcpy(codes + (size_t)r0 * m, dcodes.p, (size_t)nr * m, hipMemcpyDeviceToHost));
    if (Xr_out) RQ_HIP(hipMemcpy(Xr_out + (size_t)r0 * d, dX.p, (size_t)nr * d * 4, hipMemcpyDeviceToHost));
    if (counts) {
      RQ_HIP(hipMemcpy(part.data(), dcnt.p, part.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < part.size(); ++i) counts[i] += part[i];
    }
    g_t_d2h += t3.ms();
  }
  g_t_total = tt.ms();
  return RQ_OK;
}

int rq_encode_rvq(uint8_t *codes, const float *X, const float *codebooks, int64_t n, int d, int m, int h,
                  uint32_t *counts, float *Xr_out) {
  return host_encode_rvq(codes, nullptr, X, codebooks, n, d, m, h, counts, Xr_out);
}
int rq_encode_rvq_i16(int16_t *codes1, const float *X, const float *codebooks, int64_t n, int d, int m, int h,
                      uint32_t *counts, float *Xr_out) {
  return host_encode_rvq(nullptr, codes1, X, codebooks, n, d, m, h, counts, Xr_out);
}

int rq_dev_adc_lut(float *lut, const float *centers, const float *queries, int64_t nq, int m, int subdim,
                   void *stream) {
  if (nq <= 0) return RQ_OK;
  return lut_launch(lut, centers, queries, nq, m, subdim, (hipStream_t)stream);
}

int rq_dev_linscan(float *dists, uint32_t *ids, uint64_t *keys, const uint8_t *codes, const float *centers,
                   const float *queries, int64_t n, int64_t nq, int m, int d, int k, uint32_t id_offset,
                   int id_base, void *stream) {
  return dev_linscan(dists, ids, keys, codes, centers, queries, n, nq, m, d, k, id_offset, id_base,
                     (hipStream_t)stream);
}

int rq_scan_row_width(int m) { return scan_padded_m(m); }

// host-only: how a base of n rows x m bytes would be ordered -- out[0..7] key bits per leading code byte, [8] total bits,
// [9] rows per lane group, [10] rows per shuffle granule, [11] padded row width (tests; no device needed)
int rq_order_plan(int64_t n, int m, int *out, int cap) {
  if (!out || cap < 12) return fail(RQ_EINVAL, "rq_order_plan: out needs 12 ints");
  const int mp = scan_padded_m(m);
  if (mp < 0) return fail(RQ_EUNSUPPORTED, "the ADC scan kernels cover 1 <= m <= 64 sub-quantizers; got m=%d", m);
  OrderTiling ot;
  scan_order_tiling(mp, &ot);
  int nb[8];
  const int total = order_key_bits(n, mp, ot, nb);
  for (int c = 0; c < 8; ++c) out[c] = nb[c];
  out[8] = total; out[9] = ot.group; out[10] = ot.gran; out[11] = mp;
  if (cap >= 14) {          // [12] tables the greedy balance deals over (0: plain sort), [13] wavefronts per workgroup that balance
    uint32_t gp[4] = {0, 0, 0, 0};
    const int64_t ns = n - order_sample_rows(n, ot.blk, nullptr);
    const bool on = total > 0 && order_greedy_plan(ns, mp, total + 3, gp) && tuning("ORDER_BITS", 0) <= 0;
    out[12] = on ? (int)gp[0] : 0;
    out[13] = on ? (int)gp[1] : 0;
  }
  return RQ_OK;
}

int64_t rq_order_bytes(int64_t n, int m) {
  const int mp = scan_padded_m(m);
  return (mp < 0 || n < 1) ? 0 : (int64_t)order_base_bytes(n, mp);
}

int rq_dev_order_rows(void *ordered, const uint8_t **codes_out, const uint32_t **perm_out, const uint8_t *codes, int64_t n,
                      int m, void *stream) {
  if (!ordered || !codes_out || !perm_out || !codes) return fail(RQ_EINVAL, "order_rows: NULL argument");
  if (n < 1 || n >= (1LL << 31)) return fail(RQ_EINVAL, "n=%lld must be in [1, 2^31)", (long long)n);
  const int mp = scan_padded_m(m);
  if (mp < 0) return fail(RQ_EUNSUPPORTED, "the ADC scan kernels cover 1 <= m <= 64 sub-quantizers; got m=%d", m);
  if ((((uintptr_t)codes | (uintptr_t)ordered) & 15) != 0) return fail(RQ_EINVAL, "codes / ordered must be 16-byte aligned");
  DeviceLock launch_lock;
  hipStream_t st = (hipStream_t)stream;
  const uint8_t *src = codes;
  if (mp != m) {          // pad first (the ordered copy has the tiled row width)
    void *padded = nullptr;
    RQ_TRY(workspace(WS_PAD, (size_t)n * mp, &padded, st));
    RQ_TRY(pad_codes_launch((uint8_t *)padded, codes, n, m, mp, st));
    src = (const uint8_t *)padded;
  }
  *codes_out = nullptr; *perm_out = nullptr;
  RQ_TRY(order_base(codes_out, perm_out, ordered, src, n, mp, st));
  if (!*perm_out) {       // tiny base: nothing to order -- identity copy so that the pair is always usable
    RQ_HIP(hipMemcpyAsync(ordered, src, (size_t)n * mp, hipMemcpyDeviceToDevice, st));
    *codes_out = (const uint8_t *)ordered;
  }
  return RQ_OK;
}

int rq_dev_linscan_ordered(float *dists, uint32_t *ids, uint64_t *keys, const uint8_t *codes_ordered, const uint32_t *perm,
                           const float *centers, const float *queries, int64_t n, int64_t nq, int m, int d, int k,
                           uint32_t id_offset, int id_base, void *stream) {
  if (!perm) {
    // identity order (rq_dev_order_rows on a tiny base): rows are padded already, scan them as they are
    // (m = 12 rows arrive 16 bytes wide from rq_dev_order_rows: `padded` tells the scan so)
    ScanBase sb0;
    sb0.padded = scan_padded_m(m) != m;
    return dev_linscan(dists, ids, keys, codes_ordered, centers, queries, n, nq, m, d, k, id_offset, id_base, (hipStream_t)stream,
                       LUT_PQ, nullptr, &sb0);
  }
  ScanBase sb;
  sb.perm = perm;
  return dev_linscan(dists, ids, keys, codes_ordered, centers, queries, n, nq, m, d, k, id_offset, id_base,
                     (hipStream_t)stream, LUT_PQ, nullptr, &sb);
}

int rq_dev_merge_topk(float *dists, uint32_t *ids, uint64_t *keys_out, const uint64_t *keys_in, int64_t nq, int P,
                      int k, int id_base, void *stream) {
  if (nq <= 0) return RQ_OK;
  if (P < 1 || k < 1 || k > RQ_MAX_K) return fail(RQ_EINVAL, "merge: P=%d k=%d", P, k);
  return merge_launch(dists, ids, keys_out, keys_in, nq, P, k, id_base, (hipStream_t)stream);
}

int rq_dev_synth_codes(uint8_t *codes, int64_t n, int m, uint64_t seed, int64_t row0, void *stream) {
  if (n <= 0) return RQ_OK;
  return synth_codes_launch(codes, n, m, seed, row0, (hipStream_t)stream);
}

int rq_dev_update_centers(float *C, uint32_t *counts, const float *X, const uint8_t *codes, int64_t n, int d, int m,
                          int h, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return update_centers_launch(C, counts, X, codes, n, d, m, h, di.num_cu, (hipStream_t)stream);
}

int rq_dev_reconstruct(float *CB, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h, void *stream) {
  return reconstruct_launch(CB, codes, C, n, d, m, h, (hipStream_t)stream);
}

int rq_dev_qerror(double *acc, const float *X, const float *CB, int64_t n, int d, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return qerror_launch(acc, X, CB, n, d, di.num_cu, (hipStream_t)stream);
}

int rq_dev_gram(float *G, const float *X, const float *CB, int64_t n, int d, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return gram_launch(G, X, CB, n, d, di.num_cu, (hipStream_t)stream);
}

// the same two reductions with CB given as (codes, C) -- no n x d reconstruction (what the training loops run when the shape
// allows 16-byte gathers: d % 4 == 0, sub-spaces on multiples of 4, gram: d <= 256; RQ_EUNSUPPORTED otherwise)
int rq_dev_gram_codes(float *G, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return gram_codes_launch(G, X, codes, C, n, d, m, h, di.num_cu, (hipStream_t)stream);
}

int rq_dev_qerror_codes(double *acc, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h, void *stream) {
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  return qerror_codes_launch(acc, X, codes, C, n, d, m, h, di.num_cu, (hipStream_t)stream);
}

}  // extern "C"
