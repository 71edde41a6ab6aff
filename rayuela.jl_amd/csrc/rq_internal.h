// rq_internal.h -- shared host-side declarations of librayuela_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

#include "../../include/rayuela_hip.h"

namespace rq {

// Records the message for rq_last_error() (thread-local) and returns `code`.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int fail_hip(hipError_t e, const char *what, const char *file, int line);
// The only failure an OPTIONAL step (the bank-aware ordering of a base) may swallow: no memory for its scratch.  fail_hip returns
// the HIP code, so rc == hipErrorOutOfMemory identifies it; forgive_oom() then withdraws the message (the step is skipped, the
// call succeeds).  Launch / synchronisation errors are never forgiven: a device fault must surface where it happened.
inline bool is_oom(int rc) { return rc == (int)hipErrorOutOfMemory; }
void forgive_oom();

#define RQ_HIP(expr)                                                       \
  do {                                                                     \
    hipError_t _e = (expr);                                                \
    if (_e != hipSuccess) return ::rq::fail_hip(_e, #expr, __FILE__, __LINE__); \
  } while (0)

#define RQ_TRY(expr)            \
  do {                          \
    int _r = (expr);            \
    if (_r != RQ_OK) return _r; \
  } while (0)

int tuning(const char *key, int dflt);  // env RQ_<KEY> or rq_set_tuning override

struct DeviceInfo {
  int device;
  int num_cu;
  char arch[64];
};
int device_info(DeviceInfo *out);

// Grow-only scratch owned by the library, keyed by (current device, stream): launches on different
// streams never share a buffer.  At most 8 distinct streams per device (release_workspaces() resets).
int workspace(int slot, size_t bytes, void **ptr, hipStream_t stream);
int release_workspaces();
int release_stream_workspace(hipStream_t stream);
void *host_pool_alloc(size_t bytes);
void host_pool_free(void *p);
void host_pool_trim();
void sharded_cache_release();
int aux_streams(hipStream_t *compute, hipStream_t *transfer);
// WS_COUNTER: [0..63] work tickets / finished items per XCD, [64..255] diagnostics (rq_scan_stats), [256..] chunk-pacing
// counters of the big-base scan: 8 XCDs x SCAN_PACE_SLOTS x {chunks done, members}
constexpr int SCAN_PACE_SLOTS = 64;
constexpr size_t WS_COUNTER_BYTES = 256 + 8 * SCAN_PACE_SLOTS * 8;
enum { WS_CAND = 0, WS_COUNTER = 1, WS_KEYS = 2, WS_TMP = 3, WS_PAD = 4, WS_MERGE = 5, WS_NORMB = 6, WS_ORDER = 7, WS_ORDER_TMP = 8, WS_ENCFLAG = 9, WS_SLOTS = 10 };

// Per-device launch lock (recursive): held while a call looks up scratch, resets the work counter and
// launches, so two host threads cannot interleave those sequences on one device.
class DeviceLock {
 public:
  DeviceLock();
  ~DeviceLock();
  DeviceLock(const DeviceLock &) = delete;
  DeviceLock &operator=(const DeviceLock &) = delete;
 private:
  void *mu_;
};

// The calling thread's current device, restored on every exit path of a call that switches devices.
struct SavedDevice {
  int dev = -1;
  SavedDevice() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
  ~SavedDevice() { if (dev >= 0) (void)hipSetDevice(dev); }
  SavedDevice(const SavedDevice &) = delete;
  SavedDevice &operator=(const SavedDevice &) = delete;
};

// Milliseconds reported by rq_last_timing() for the calling thread.
void set_timing(double total_ms, double h2d_ms, double kernel_ms, double d2h_ms);

// ---- ADC scan -------------------------------------------------------------------------------
struct ScanPlan {
  int qg, blk;
  uint32_t ngroups, nslices, rows_per_slice, whole;
  uint32_t cap, trigger, p2, scratch_keys, grid, sample;
  size_t cand_bytes, gtab_off, bkt_off;
  bool lds_ok, bigk, spread;
  bool xcd;        // big base: short row windows handed out per XCD (see plan_for)
};
int scan_plan(ScanPlan &pl, int64_t n, int64_t nq, int m, int d, int K, int num_cu, int force_slices);
int scan_launch(const ScanPlan &pl, float *dists, uint32_t *ids, uint64_t *keys, uint64_t *part, const uint8_t *codes,
                const float *centers, const float *queries, int64_t n, int64_t nq, int m, int d, int K,
                uint32_t id_offset, int id_base, uint32_t *work_counter, uint64_t *cand,
                hipStream_t stream, int lut_mode = 0, const float *row_bias = nullptr, uint8_t *norm_buf = nullptr,
                const uint32_t *perm = nullptr, bool norm_ready = false);
const char *last_scan_kernel_name();      // which instantiation the calling thread's last scan_launch chose
size_t lsq_norm_bytes(int64_t n);      // LSQ pre-filter: bytes of a base's prepared norm buffer
int lsq_norm_prepare(uint8_t *norm_buf, const uint8_t *codes, const float *centers, const float *row_bias, int64_t n,
                     int mp, int m_real, int d, hipStream_t stream);
enum { LUT_PQ = 0, LUT_LSQ = 1, LUT_CQ = 2 };
// what a caller may know about a resident base beyond its code bytes
struct ScanBase {
  const uint32_t *perm = nullptr;     // rows are in bank-aware order (rq_order.hip): perm[position] = row; implies `padded`
  uint8_t *norm_prepared = nullptr;   // LSQ: the pre-filter's prepared norm buffer (lsq_norm_prepare), else per call
  bool padded = false;                // rows are scan_padded_m(m) bytes wide already
};
// argument checks + planner + launches of one resident shard (rq_dev_linscan's body)
int dev_linscan(float *dists, uint32_t *ids, uint64_t *keys, const uint8_t *codes, const float *centers,
                const float *queries, int64_t n, int64_t nq, int m, int d, int k, uint32_t id_offset,
                int id_base, hipStream_t stream, int lut_mode = LUT_PQ, const float *row_bias = nullptr,
                const ScanBase *base = nullptr);
bool order_pays(int64_t n, int64_t nq, int k);                          // SCAN_ORDER / ORDER_MIN_ROWS / ORDER_MIN_NQ / ORDER_MAX_K
size_t order_base_bytes(int64_t n, int mp);
int order_base(const uint8_t **out_codes, const uint32_t **out_perm, void *dst, const uint8_t *codes, int64_t n, int mp,
               hipStream_t stream);
// ---- bank-aware row order (rq_order.hip) ---------------------------------------------------------
struct OrderTiling { int rpt, gran, group, cbits, blk; };                    // see scan_order_tiling (rq_scan.hip)
void scan_order_tiling(int mp, OrderTiling *t);
void order_set_call_queries(int64_t nq);   // > 0: the ordering that follows serves ONE scan of nq queries (greedy balance only if it pays)
bool order_greedy_plan(int64_t n, int mp, int budget, uint32_t out[4]);
int order_key_bits(int64_t n, int mp, const OrderTiling &t, int nb[8]);  // key layout; returns the total bits (0: no ordering)
size_t order_scratch_bytes(int64_t n, int total_bits);
int order_sample_stride();                                               // ORDER_SAMPLE_STRIDE (16; < 2: no sample blocks)
uint32_t order_sample_rows(int64_t n, int blk, uint32_t *sgroups);   // arrival-order sample blocks of an ordered base
int gather_f32_launch(float *dst, const float *src, const uint32_t *perm, int64_t n, hipStream_t stream);
int order_rows_launch(uint8_t *dst, uint32_t *perm, const uint8_t *src, int64_t n, int mp, void *scratch,
                      const OrderTiling &t, hipStream_t stream);
// [P][nq][k] -> [nq][P][k] (lists gathered shard-major, merged query-major)
int interleave_keys_launch(uint64_t *dst, const uint64_t *src, int64_t nq, int P, int k, size_t pstride, hipStream_t stream);
int scan_padded_m(int m);   // smallest tiled row width >= m (2,4,8,16,32,64) or -1
int pad_codes_launch(uint8_t *dst, const uint8_t *src, int64_t n, int m, int mp, hipStream_t stream);
int merge_launch(float *dists, uint32_t *ids, uint64_t *keys_out, const uint64_t *keys_in, int64_t nq,
                 int P, int K, int id_base, hipStream_t stream);
int lut_launch(float *lut, const float *centers, const float *queries, int64_t nq, int m, int sub,
               hipStream_t stream);
int synth_codes_launch(uint8_t *codes, int64_t n, int m, uint64_t seed, int64_t row0, hipStream_t stream);

// ---- multi-device (rq_index.hip) ---------------------------------------------------------------
int env_devices(int *out, int cap);   // RAYUELA_HIP_DEVICES -> device list (0 entries = unset)
int host_linscan_sharded(float *dists, uint32_t *ids, const uint8_t *codes, const float *centers, const float *queries,
                         const float *R, int64_t n, int64_t nq, int m, int d, int k, int id_base, const int *devices,
                         int ndev);

// ---- encode / rotation ----------------------------------------------------------------------
int encode_launch(uint8_t *codes, const float *X, const float *C, int64_t n, int d, int m, int h,
                  int num_cu, hipStream_t stream, float *dbg_w = nullptr);
const char *last_encode_kernel_name();   // which kernel the calling thread's last encode_launch chose
void last_encode_stats(unsigned long long out[2]);   // tuning ENC_STATS = 1: {pairs, pairs that took the exact pass} of that encode
int rvq_residual_launch(float *Xr, const float *Ci, const uint8_t *stage_codes, uint8_t *codes, unsigned int *cnt,
                        int64_t n, int d, int m, int stage, hipStream_t stream);
int rvq_encode_launch(uint8_t *codes, float *Xr, uint8_t *stage_codes, unsigned int *counts, const float *C,
                      int64_t n, int d, int m, int h, int num_cu, hipStream_t stream);
int rotate_launch(float *RX, const float *R, const float *X, int d, int64_t n, int num_cu,
                  hipStream_t stream);
int widen_codes_launch(int16_t *out1, const uint8_t *codes, int64_t nelem, hipStream_t stream);

// ---- training reductions (rq_train.hip) --------------------------------------------------------
int update_centers_launch(float *C, unsigned int *counts, const float *X, const uint8_t *codes, int64_t n, int d,
                          int m, int h, int num_cu, hipStream_t stream);
int reconstruct_launch(float *CB, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h,
                       hipStream_t stream);
int qerror_launch(double *acc_dev, const float *X, const float *CB, int64_t n, int d, int num_cu,
                  hipStream_t stream);
// kmeans++ seeding of all m sub-spaces (Clustering.jl init=:kmpp): seeds [m][h] rows, C = their sub-vectors;
// mincost [n][m] floats and partial [m][1024] doubles are scratch, u [m][h] uniforms in [0,1) (device pointers)
int kmpp_init_launch(float *C, long long *seeds, float *mincost, double *partial, const double *u, const float *X,
                     int64_t n, int d, int m, int h, hipStream_t stream);
int polar_factor_launch(float *Rimg, const float *G, double *Vw, int warm, int d, int *status, double *scratch, hipStream_t stream);
bool codes_forms_ok(int d, int m, int h, bool for_gram);
int gram_codes_launch(float *G, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h, int num_cu,
                      hipStream_t stream);
int qerror_codes_launch(double *acc_dev, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h,
                        int num_cu, hipStream_t stream);
size_t polar_ns_scratch_bytes(int d, int num_cu);
int polar_ns_launch(float *Rimg, const float *G, int d, int *status, void *scratch, int num_cu, hipStream_t stream);
int codes_changed_launch(unsigned long long *out, const uint8_t *a, const uint8_t *b, size_t nbytes, hipStream_t stream);
int gram_launch(float *G, const float *X, const float *CB, int64_t n, int d, int num_cu, hipStream_t stream);

}  // namespace rq
