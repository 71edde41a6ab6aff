// rq_index.hip -- device-resident index handle of the C ABI: codes uploaded once, searched many times,
// on ONE device or row-sharded over the GPUs of a node from a single host process.
//
// The reference has no multi-device path: src/Linscan.jl:5-26 hands the whole code matrix to one
// OpenMP loop over queries (deps/src/linscan_aqd.cpp:55-61), whose only long-axis device is the 1e7-row
// chunking with a carried top-k (:52-53,78-92).  The MI355X analogue behind the same call:
//
//   shard     rows [row0_s, row0_s + n_s) of the base live on device dev_s as resident uint8 codes
//             (rq_index_set_codes splits the caller's matrix; rq_index_set_codes_synth fills the
//             SIFT1B-shape synthetic base on the devices themselves)
//   scan      every device scans its shards for ALL queries on its own stream (queries and codebooks
//             are replicated, 0.5 MB + 128 KiB) and leaves sorted packed keys
//             (ordered(dist) << 32 | GLOBAL row id) per shard                    -- no communication
//   gather    the per-shard key lists (nq*k*8 bytes each) go to the root device over xGMI: RCCL
//             send/recv inside one group on the single-process communicator clique (ncclCommInitAll,
//             /opt/rocm/include/rccl/rccl.h:236), i.e. a gather-to-root over P-1 DISTINCT links -- or
//             hipMemcpyPeerAsync pushes (SDMA engines, no CUs) when RCCL is unavailable or disabled
//             (tuning EXCHANGE_PEER=1); shards that live on the root device write in place
//   merge     merge_topk over the P lists per query on the root.  Keys are totally ordered and ids are
//             global, so the answer is bit-identical to one scan of the whole base.
//
// A device may appear several times in the device list: every occurrence is one LOGICAL shard (that is
// how the sharded path is tested on a one-GPU box, and how a base of >= 2^31 rows fits one device).
// librccl is dlopen()ed on first use, so the library has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "rq_internal.h"
#include "rq_topk.h"

namespace rq {

// ---- RCCL through dlopen ------------------------------------------------------------------------
struct RcclApi {
  void *handle = nullptr;
  bool tried = false;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return handle != nullptr; }
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static bool rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.tried) return g_rccl.ok();
  g_rccl.tried = true;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return false;
#define RQ_SYM(field, name)                                                \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) { dlclose(h); return false; }
  RQ_SYM(CommInitAll, "ncclCommInitAll")
  RQ_SYM(CommDestroy, "ncclCommDestroy")
  RQ_SYM(GroupStart, "ncclGroupStart")
  RQ_SYM(GroupEnd, "ncclGroupEnd")
  RQ_SYM(Send, "ncclSend")
  RQ_SYM(Recv, "ncclRecv")
  RQ_SYM(GetErrorString, "ncclGetErrorString")
#undef RQ_SYM
  g_rccl.handle = h;
  return true;
}

#define RQ_NCCL(expr)                                                                                 \
  do {                                                                                                \
    ncclResult_t _r = (expr);                                                                         \
    if (_r != ncclSuccess) return fail(1000 + (int)_r, "RCCL error %d (%s) in `%s`", (int)_r,         \
                                       g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?", #expr); \
  } while (0)

// ---- [P][nq][k] -> [nq][P][k] ---------------------------------------------------------------------
// pstride = keys between the blocks of two shards (nq_total * k when only a chunk of the queries is interleaved)
__global__ void interleave_keys_kernel(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, size_t nq, uint32_t P,
                                       uint32_t k, size_t pstride) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_q = (size_t)P * k;
  if (i >= nq * per_q) return;
  const size_t q = i / per_q;
  const uint32_t rem = (uint32_t)(i - q * per_q);
  const uint32_t p = rem / k, j = rem - p * k;
  dst[i] = src[(size_t)p * pstride + q * k + j];
}

int interleave_keys_launch(uint64_t *dst, const uint64_t *src, int64_t nq, int P, int k, size_t pstride, hipStream_t stream) {
  const size_t total = (size_t)nq * P * k;
  if (!total) return RQ_OK;
  hipLaunchKernelGGL(interleave_keys_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, dst, src,
                     (size_t)nq, (uint32_t)P, (uint32_t)k, pstride);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// ---- the handle -------------------------------------------------------------------------------------
constexpr int IX_MAX_CHUNKS = 8;

struct IxDev {          // one per DISTINCT device of the index
  int device = 0;
  int num_cu = 0;
  hipStream_t stream = nullptr;     // uploads, rotation, scans
  hipStream_t xstream = nullptr;    // top-k lists towards the root (SDMA peer copies / RCCL send, recv)
  hipStream_t mstream = nullptr;    // root only: interleave + merge of a query chunk
  hipEvent_t ev_scan[IX_MAX_CHUNKS] = {nullptr};   // this device's lists of chunk c are complete
  hipEvent_t ev_xfer[IX_MAX_CHUNKS] = {nullptr};   // ... and have arrived on the root
  float *centers = nullptr;
  float *queries = nullptr, *queries_rot = nullptr, *R = nullptr;
  size_t q_cap = 0;
};

struct IxShard {
  int dev = 0;          // index into devs
  int64_t row0 = 0, n = 0;
  uint8_t *codes = nullptr;   // the shard's rows -- in bank-aware order when perm != nullptr (rq_order.hip)
  const uint32_t *perm = nullptr;   // position -> row of the shard (inside the `codes` allocation)
  uint64_t *keys = nullptr;   // [nq][k] sorted keys of the last search (shards off the root device)
  size_t keys_cap = 0;
  uint64_t *tmp = nullptr;    // [nq][k_local] when the shard holds fewer than k rows
  size_t tmp_cap = 0;
};


static int grow(void **p, size_t *cap, size_t bytes) {
  if (*cap >= bytes && *p) return RQ_OK;
  if (*p) { RQ_HIP(hipDeviceSynchronize()); RQ_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
  const size_t want = (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
  RQ_HIP(hipMalloc(p, want));
  *cap = want;
  return RQ_OK;
}

struct Clock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

}  // namespace rq

using namespace rq;

struct rq_index {
  int m = 0, d = 0;
  int64_t n = 0;
  uint32_t id_offset = 0;
  std::vector<IxDev> devs;
  std::vector<IxShard> shards;
  // on the root device (devs[0])
  uint64_t *gathered = nullptr, *inter = nullptr;
  size_t gathered_cap = 0, inter_cap = 0;
  float *dd = nullptr;
  uint32_t *di = nullptr;
  size_t dd_cap = 0, di_cap = 0;
  // exchange
  ncclComm_t comms[16] = {nullptr};
  int exchange = 0;     // 0 none (one device), 1 peer copies, 2 RCCL
  bool peer_ok = false;    // every device has a direct path to the root (hipDeviceCanAccessPeer)
  bool selftest = false;   // tuning EXCHANGE_SELFTEST on ONE device: logical shards 1.. travel through RCCL send/recv to self
  std::mutex mu;
};

namespace rq {

static void index_free(rq_index *ix) {
  if (!ix) return;
  SavedDevice saved;
  if (ix->exchange == 2 && g_rccl.ok())
    for (size_t i = 0; i < ix->devs.size(); ++i)
      if (ix->comms[i]) (void)g_rccl.CommDestroy(ix->comms[i]);
  for (auto &s : ix->shards) {
    if (hipSetDevice(ix->devs[s.dev].device) != hipSuccess) continue;
    if (s.codes) (void)hipFree(s.codes);
    if (s.keys) (void)hipFree(s.keys);
    if (s.tmp) (void)hipFree(s.tmp);
  }
  for (size_t i = 0; i < ix->devs.size(); ++i) {
    IxDev &dv = ix->devs[i];
    if (hipSetDevice(dv.device) != hipSuccess) continue;
    if (i == 0) {
      if (ix->gathered) (void)hipFree(ix->gathered);
      if (ix->inter) (void)hipFree(ix->inter);
      if (ix->dd) (void)hipFree(ix->dd);
      if (ix->di) (void)hipFree(ix->di);
    }
    if (dv.centers) (void)hipFree(dv.centers);
    if (dv.queries) (void)hipFree(dv.queries);
    if (dv.queries_rot) (void)hipFree(dv.queries_rot);
    if (dv.R) (void)hipFree(dv.R);
    for (int c = 0; c < IX_MAX_CHUNKS; ++c) {
      if (dv.ev_scan[c]) (void)hipEventDestroy(dv.ev_scan[c]);
      if (dv.ev_xfer[c]) (void)hipEventDestroy(dv.ev_xfer[c]);
    }
    if (dv.xstream) (void)hipStreamDestroy(dv.xstream);
    if (dv.mstream) { (void)release_stream_workspace(dv.mstream); (void)hipStreamDestroy(dv.mstream); }
    if (dv.stream) { (void)release_stream_workspace(dv.stream); (void)hipStreamDestroy(dv.stream); }
  }
  (void)hipGetLastError();
  delete ix;
}

static int index_build(rq_index *ix, int m, int d, const float *centers_host, const int *devices, int ndev) {
  if (m < 1 || d < m || d % m) return fail(RQ_EINVAL, "index: scan needs d %% m == 0 (src/Linscan.jl:23); got d=%d m=%d", d, m);
  if (!centers_host) return fail(RQ_EINVAL, "index: centers is NULL");
  if (ndev < 1 || ndev > 64) return fail(RQ_EINVAL, "index: 1 <= number of shards <= 64, got %d", ndev);
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) { (void)hipGetLastError(); return fail(RQ_ENODEVICE, "no HIP device visible"); }
  ix->m = m; ix->d = d;
  const size_t ce = (size_t)m * 256 * (d / m) * 4;
  for (int s = 0; s < ndev; ++s) {
    const int dev = devices[s];
    if (dev < 0 || dev >= visible || dev >= 16) return fail(RQ_EINVAL, "index: device %d not in [0, %d)", dev, std::min(visible, 16));
    int di = -1;
    for (size_t i = 0; i < ix->devs.size(); ++i) if (ix->devs[i].device == dev) di = (int)i;
    if (di < 0) {
      IxDev dv;
      dv.device = dev;
      RQ_HIP(hipSetDevice(dev));
      DeviceInfo info;
      RQ_TRY(device_info(&info));
      dv.num_cu = info.num_cu;
      ix->devs.push_back(dv);
      di = (int)ix->devs.size() - 1;
      IxDev &r = ix->devs[di];
      RQ_HIP(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
      RQ_HIP(hipStreamCreateWithFlags(&r.xstream, hipStreamNonBlocking));
      if (di == 0) RQ_HIP(hipStreamCreateWithFlags(&r.mstream, hipStreamNonBlocking));
      for (int c = 0; c < IX_MAX_CHUNKS; ++c) {
        RQ_HIP(hipEventCreateWithFlags(&r.ev_scan[c], hipEventDisableTiming));
        RQ_HIP(hipEventCreateWithFlags(&r.ev_xfer[c], hipEventDisableTiming));
      }
      RQ_HIP(hipMalloc((void **)&r.centers, ce));
      RQ_HIP(hipMemcpy(r.centers, centers_host, ce, hipMemcpyHostToDevice));
    }
    IxShard sh;
    sh.dev = di;
    ix->shards.push_back(sh);
  }
  const int nd = (int)ix->devs.size();
  if (nd == 1 && ix->shards.size() > 1 && tuning("EXCHANGE_SELFTEST", 0) && rccl_load()) {
    // one-GPU boxes: run the RCCL transport anyway (a clique of one, every list but the first sent to self), so
    // that dlopen, communicator, group semantics and datatypes are exercised on real hardware
    int list[1] = {ix->devs[0].device};
    if (g_rccl.CommInitAll(ix->comms, 1, list) == ncclSuccess) { ix->exchange = 2; ix->selftest = true; }
  }
  if (nd > 1) {
    // direct xGMI paths root <-> every other device (a failure only means staged copies)
    const int root = ix->devs[0].device;
    ix->peer_ok = true;
    for (int i = 1; i < nd; ++i) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, root, ix->devs[i].device) != hipSuccess || !can) ix->peer_ok = false;
      if (can) {
        (void)hipSetDevice(root);
        (void)hipDeviceEnablePeerAccess(ix->devs[i].device, 0);
        (void)hipSetDevice(ix->devs[i].device);
        (void)hipDeviceEnablePeerAccess(root, 0);
      }
      (void)hipGetLastError();
    }
    ix->exchange = 1;
    if (!tuning("EXCHANGE_PEER", 0) && rccl_load()) {
      int list[16];
      for (int i = 0; i < nd; ++i) list[i] = ix->devs[i].device;
      const ncclResult_t r = g_rccl.CommInitAll(ix->comms, nd, list);
      if (r == ncclSuccess) ix->exchange = 2;
      else fprintf(stderr, "librayuela_hip: ncclCommInitAll failed (%s); using peer copies for the top-k exchange\n",
                   g_rccl.GetErrorString(r));
    }
  }
  return RQ_OK;
}

static void shard_bounds(rq_index *ix, int64_t n) {
  const int64_t P = (int64_t)ix->shards.size(), per = n / P, extra = n % P;
  int64_t row = 0;
  for (int64_t s = 0; s < P; ++s) {
    ix->shards[s].row0 = row;
    ix->shards[s].n = per + (s < extra ? 1 : 0);
    row += ix->shards[s].n;
  }
}

// (re)allocate every shard's code array for a base of n rows; fill(shard, stream) queues the upload / generator
template <class Fill>
static int index_set(rq_index *ix, int64_t n, uint32_t id_offset, Fill fill) {
  if (!ix) return fail(RQ_EINVAL, "index is NULL");
  if (n < 1) return fail(RQ_EINVAL, "index: n=%lld must be >= 1", (long long)n);
  // same bound as dev_linscan: the last id must stay below 0xFFFFFFFF (that id is the padding key KEY_MAX's)
  if ((uint64_t)id_offset + (uint64_t)n > 0xFFFFFFFFull) return fail(RQ_EINVAL, "index: row ids overflow uint32 (id_offset + n must be <= 2^32 - 1)");
  std::lock_guard<std::mutex> lk(ix->mu);
  SavedDevice saved;
  shard_bounds(ix, n);
  for (auto &s : ix->shards)
    if (s.n >= (1LL << 31)) return fail(RQ_EUNSUPPORTED, "index: a shard of %lld rows exceeds 2^31-1; use more shards", (long long)s.n);
  for (auto &s : ix->shards) {
    IxDev &dv = ix->devs[s.dev];
    RQ_HIP(hipSetDevice(dv.device));
    if (s.codes) { RQ_HIP(hipStreamSynchronize(dv.stream)); RQ_HIP(hipFree(s.codes)); s.codes = nullptr; }
    s.perm = nullptr;
    if (s.n == 0) continue;
    RQ_HIP(hipMalloc((void **)&s.codes, (size_t)s.n * ix->m));
    RQ_TRY(fill(s, dv.stream));
    // The base is resident for many searches: put its rows in bank-aware order ONCE (rq_order.hip; the scan's table gathers
    // then hit distinct LDS columns, keys still carry the original row numbers through perm).  Rows of a tiled width only;
    // other widths are padded and ordered per search.  Costs 4 bytes per row for perm; the arrival-order copy is freed.
    if (tuning("INDEX_ORDER", 1) && scan_padded_m(ix->m) == ix->m && order_pays(s.n, 0, 0)) {
      DeviceLock order_lock;      // scratch lookup + launches of this device, like a scan
      // (an ordered copy that cannot be ALLOCATED leaves the shard in arrival order: slower gathers, the same answer;
      // set_codes does not fail for want of memory for an optimisation)
      void *ord = nullptr;
      const uint8_t *oc = nullptr;
      const uint32_t *op = nullptr;
      int rc = RQ_OK;
      if (hipMalloc(&ord, order_base_bytes(s.n, ix->m)) != hipSuccess) { (void)hipGetLastError(); rc = (int)hipErrorOutOfMemory; ord = nullptr; }
      if (rc == RQ_OK) rc = order_base(&oc, &op, ord, s.codes, s.n, ix->m, dv.stream);
      if (rc == RQ_OK) { hipError_t e = hipStreamSynchronize(dv.stream); if (e != hipSuccess) rc = fail_hip(e, "order sync", __FILE__, __LINE__); }
      if (rc == RQ_OK && op) {
        RQ_HIP(hipFree(s.codes));
        s.codes = (uint8_t *)ord;
        s.perm = op;
      } else {
        if (ord) (void)hipFree(ord);
        // only "no memory" is forgiven; a launch or synchronisation error of the ordering kernels is a device fault and is
        // returned (it used to be swallowed and reported as RQ_OK)
        if (is_oom(rc)) forgive_oom();
        else if (rc != RQ_OK) { (void)release_stream_workspace(dv.stream); return rc; }
      }
      (void)release_stream_workspace(dv.stream);    // the key scratch (4 bytes per row + the histogram) is not needed again
    }
  }
  for (auto &dv : ix->devs) {
    RQ_HIP(hipSetDevice(dv.device));
    RQ_HIP(hipStreamSynchronize(dv.stream));
  }
  ix->n = n;
  ix->id_offset = id_offset;
  return RQ_OK;
}

int index_search(rq_index *ix, float *dists, uint32_t *ids, const float *queries_host, const float *R_host, int64_t nq,
                 int k, int id_base) {
  if (!ix || ix->n < 1) return fail(RQ_EINVAL, "index has no codes");
  if (nq <= 0) return RQ_OK;
  if (k < 1 || k > RQ_MAX_K) return fail(RQ_EUNSUPPORTED, "k=%d outside [1, %d]", k, RQ_MAX_K);
  if (k > ix->n) return fail(RQ_EINVAL, "k=%d > n=%lld (undefined in the reference, deps/src/linscan_aqd.cpp:91)", k, (long long)ix->n);
  if (id_base != 0 && id_base != 1) return fail(RQ_EINVAL, "id_base must be 0 or 1");
  if (!dists || !ids || !queries_host) return fail(RQ_EINVAL, "index search: NULL argument");
  std::lock_guard<std::mutex> lk(ix->mu);
  SavedDevice saved;
  Clock tt;
  const int m = ix->m, d = ix->d;
  const int P = (int)ix->shards.size();
  const size_t qb = (size_t)nq * d * 4, cnt = (size_t)nq * k, ob = cnt * 4;
  IxDev &root = ix->devs[0];

  // ---- queries to every device (and R'q there: linscan_opq rotates the queries first, src/Linscan.jl:102) ----
  Clock t1;
  for (auto &dv : ix->devs) {
    RQ_HIP(hipSetDevice(dv.device));
    if (dv.q_cap < qb) {
      RQ_HIP(hipStreamSynchronize(dv.stream));
      if (dv.queries) RQ_HIP(hipFree(dv.queries));
      if (dv.queries_rot) RQ_HIP(hipFree(dv.queries_rot));
      dv.queries = dv.queries_rot = nullptr;
      dv.q_cap = 0;
      RQ_HIP(hipMalloc((void **)&dv.queries, qb));
      RQ_HIP(hipMalloc((void **)&dv.queries_rot, qb));
      dv.q_cap = qb;
    }
    RQ_HIP(hipMemcpyAsync(dv.queries, queries_host, qb, hipMemcpyHostToDevice, dv.stream));
    if (R_host) {
      if (!dv.R) RQ_HIP(hipMalloc((void **)&dv.R, (size_t)d * d * 4));
      RQ_HIP(hipMemcpyAsync(dv.R, R_host, (size_t)d * d * 4, hipMemcpyHostToDevice, dv.stream));
      RQ_TRY(rotate_launch(dv.queries_rot, dv.R, dv.queries, d, nq, dv.num_cu, dv.stream));
    }
  }
  const double h2d_ms = t1.ms();
  Clock t2;

  RQ_HIP(hipSetDevice(root.device));
  RQ_TRY(grow((void **)&ix->dd, &ix->dd_cap, ob));
  RQ_TRY(grow((void **)&ix->di, &ix->di_cap, ob));
  if (P == 1) {
    IxShard &s = ix->shards[0];
    ScanBase sb0;
    sb0.perm = s.perm;
    RQ_TRY(dev_linscan(ix->dd, ix->di, nullptr, s.codes, root.centers, R_host ? root.queries_rot : root.queries, s.n, nq, m,
                       d, k, ix->id_offset, id_base, root.stream, LUT_PQ, nullptr, &sb0));
  } else {
    RQ_TRY(grow((void **)&ix->gathered, &ix->gathered_cap, (size_t)P * cnt * 8));
    RQ_TRY(grow((void **)&ix->inter, &ix->inter_cap, (size_t)P * cnt * 8));
    // Query chunks: with several devices and long lists (k x nq x 8 B per shard: 80 MB at SIFT1M shape with k = 1000,
    // 800 MB with the reference's default k = 10000) the exchange is as long as a device's scan of its 1/8 of the rows.
    // The queries are then cut into C chunks and pipelined: scan of chunk c+1 on `stream` | lists of chunk c towards
    // the root on `xstream` | merge of chunk c-1 on the root's `mstream`.  In that mode the lists travel as SDMA peer
    // copies: the persistent scan grid holds every CU and all of the LDS, an RCCL kernel would wait for the scan to
    // drain, the copy engines do not need a CU.  Every extra chunk costs a shard ~0.17 ms of fixed scan overhead
    // (tools/chunk_cost.py: tables, threshold sample, launch tails), an xGMI link moves 64 MB in ~0.42 ms, so a
    // chunk is at least 64 MB.  Unchunked (BASELINE config 5: 0.8 MB per device) the exchange is one RCCL group.
    int C = tuning("IDX_QCHUNKS", 0);
    if (C <= 0) C = (ix->devs.size() > 1) ? (int)std::min<size_t>(cnt * 8 / ((size_t)64 << 20), 4) : 1;
    C = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(C, IX_MAX_CHUNKS), nq));
    const bool use_rccl = ix->exchange == 2 && (C == 1 || ix->selftest || !ix->peer_ok || tuning("EXCHANGE_CHUNK_RCCL", 0));
    auto remote = [&](int si) { return ix->shards[si].dev != 0 || (ix->selftest && si > 0); };

    // ---- buffers first (growing one may synchronise the device) -----------------------------------------
    for (int si = 0; si < P; ++si) {
      IxShard &s = ix->shards[si];
      IxDev &dv = ix->devs[s.dev];
      RQ_HIP(hipSetDevice(dv.device));
      if (remote(si)) RQ_TRY(grow((void **)&s.keys, &s.keys_cap, cnt * 8));
      const int k_local = (int)std::min<int64_t>(k, s.n);
      if (k_local < k && k_local > 0) RQ_TRY(grow((void **)&s.tmp, &s.tmp_cap, (size_t)nq * k_local * 8));
      if (k_local < k)      // KEY_MAX padding: the shard has < k rows
        RQ_HIP(hipMemsetAsync(remote(si) ? s.keys : ix->gathered + (size_t)si * cnt, 0xFF, cnt * 8, dv.stream));
    }
    for (int c = 0; c < C; ++c) {
      const int64_t q0 = nq * c / C, nqc = nq * (c + 1) / C - q0;
      // ---- local scans of this chunk: every device works through its shards on its own stream ------------
      for (int si = 0; si < P; ++si) {
        IxShard &s = ix->shards[si];
        IxDev &dv = ix->devs[s.dev];
        const int k_local = (int)std::min<int64_t>(k, s.n);
        if (k_local == 0) continue;
        RQ_HIP(hipSetDevice(dv.device));
        uint64_t *out = (remote(si) ? s.keys : ix->gathered + (size_t)si * cnt) + (size_t)q0 * k;   // root shards: in place
        uint64_t *dst = k_local < k ? s.tmp + (size_t)q0 * k_local : out;
        const float *qs = (R_host ? dv.queries_rot : dv.queries) + (size_t)q0 * d;
        ScanBase sbs;
        sbs.perm = s.perm;
        RQ_TRY(dev_linscan(nullptr, nullptr, dst, s.codes, dv.centers, qs, s.n, nqc, m, d, k_local,
                           (uint32_t)(ix->id_offset + (uint64_t)s.row0), 0, dv.stream, LUT_PQ, nullptr, &sbs));
        if (k_local < k)
          RQ_HIP(hipMemcpy2DAsync(out, (size_t)k * 8, dst, (size_t)k_local * 8, (size_t)k_local * 8, (size_t)nqc,
                                  hipMemcpyDeviceToDevice, dv.stream));
      }
      for (auto &dv : ix->devs) {
        RQ_HIP(hipSetDevice(dv.device));
        RQ_HIP(hipEventRecord(dv.ev_scan[c], dv.stream));
      }
      // ---- the lists of the other devices travel to the root ---------------------------------------------
      RQ_HIP(hipSetDevice(root.device));
      RQ_HIP(hipStreamWaitEvent(root.mstream, root.ev_scan[c], 0));
      if (use_rccl) {
        bool any = false;
        for (size_t i = 0; i < ix->devs.size(); ++i) {
          IxDev &dv = ix->devs[i];
          RQ_HIP(hipSetDevice(dv.device));
          RQ_HIP(hipStreamWaitEvent(dv.xstream, dv.ev_scan[c], 0));
        }
        RQ_NCCL(g_rccl.GroupStart());
        for (int si = 0; si < P; ++si) {
          if (!remote(si)) continue;
          IxShard &s = ix->shards[si];
          IxDev &dv = ix->devs[s.dev];
          any = true;
          RQ_NCCL(g_rccl.Send(s.keys + (size_t)q0 * k, (size_t)nqc * k, ncclUint64, 0, ix->comms[s.dev], dv.xstream));
          RQ_NCCL(g_rccl.Recv(ix->gathered + (size_t)si * cnt + (size_t)q0 * k, (size_t)nqc * k, ncclUint64, s.dev,
                              ix->comms[0], root.xstream));
        }
        RQ_NCCL(g_rccl.GroupEnd());
        if (any) {
          RQ_HIP(hipSetDevice(root.device));
          RQ_HIP(hipEventRecord(root.ev_xfer[c], root.xstream));
          RQ_HIP(hipStreamWaitEvent(root.mstream, root.ev_xfer[c], 0));
        }
      } else {
        for (size_t i = 1; i < ix->devs.size(); ++i) {
          IxDev &dv = ix->devs[i];
          RQ_HIP(hipSetDevice(dv.device));
          RQ_HIP(hipStreamWaitEvent(dv.xstream, dv.ev_scan[c], 0));
          for (int si = 0; si < P; ++si) {
            IxShard &s = ix->shards[si];
            if (s.dev != (int)i) continue;
            RQ_HIP(hipMemcpyPeerAsync(ix->gathered + (size_t)si * cnt + (size_t)q0 * k, root.device,
                                      s.keys + (size_t)q0 * k, dv.device, (size_t)nqc * k * 8, dv.xstream));
          }
          RQ_HIP(hipEventRecord(dv.ev_xfer[c], dv.xstream));
          RQ_HIP(hipStreamWaitEvent(root.mstream, dv.ev_xfer[c], 0));
        }
      }
      // ---- merge of the chunk on the root ------------------------------------------------------------------
      RQ_HIP(hipSetDevice(root.device));
      uint64_t *inter = ix->inter + (size_t)q0 * P * k;
      RQ_TRY(interleave_keys_launch(inter, ix->gathered + (size_t)q0 * k, nqc, P, k, cnt, root.mstream));
      RQ_TRY(merge_launch(ix->dd + (size_t)q0 * k, ix->di + (size_t)q0 * k, nullptr, inter, nqc, P, k, id_base, root.mstream));
    }
  }
  for (size_t i = ix->devs.size(); i-- > 0;) {
    IxDev &dv = ix->devs[i];
    RQ_HIP(hipSetDevice(dv.device));
    RQ_HIP(hipStreamSynchronize(dv.stream));
    RQ_HIP(hipStreamSynchronize(dv.xstream));
    if (dv.mstream) RQ_HIP(hipStreamSynchronize(dv.mstream));
  }
  const double kernel_ms = t2.ms();
  Clock t3;
  RQ_HIP(hipSetDevice(root.device));
  RQ_HIP(hipMemcpy(dists, ix->dd, ob, hipMemcpyDeviceToHost));
  RQ_HIP(hipMemcpy(ids, ix->di, ob, hipMemcpyDeviceToHost));
  set_timing(tt.ms(), h2d_ms, kernel_ms, t3.ms());
  return RQ_OK;
}

// Device list of the host-pointer entry points: env RAYUELA_HIP_DEVICES = "0,1,2,3" | "all" (unset: the current
// device only).  A repeated ordinal is a logical shard.  Returns the number of entries written (0 = unset).
int env_devices(int *out, int cap) {
  const char *v = getenv("RAYUELA_HIP_DEVICES");
  if (!v || !*v) return 0;
  int n = 0;
  if (!strcmp(v, "all")) {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) { (void)hipGetLastError(); return 0; }
    for (int i = 0; i < visible && n < cap; ++i) out[n++] = i;
    return n;
  }
  const char *p = v;
  while (*p && n < cap) {
    char *end = nullptr;
    const long x = strtol(p, &end, 10);
    if (end == p) break;
    out[n++] = (int)x;
    p = end;
    while (*p == ',' || *p == ' ') ++p;
  }
  return n;
}

// linscan_pq / linscan_opq on host pointers over several devices (RAYUELA_HIP_DEVICES).  The sharded index behind these
// calls is KEPT between calls with the same device list, m and d: building one costs the RCCL communicator
// (ncclCommInitAll over the clique: hundreds of milliseconds), peer-access setup, streams and events -- a Julia session
// that calls linscan_pq in a loop would pay that every time.  Per call only the codebooks (128 KiB) and the code shards
// are uploaded again.  rq_release_workspaces() drops the cached index (its shards stay allocated until then).
static std::mutex g_shc_mu;
static rq_index *g_shc_ix = nullptr;
static std::vector<int> g_shc_devs;

void sharded_cache_release() {
  std::lock_guard<std::mutex> lk(g_shc_mu);
  if (g_shc_ix) index_free(g_shc_ix);
  g_shc_ix = nullptr;
  g_shc_devs.clear();
}

int host_linscan_sharded(float *dists, uint32_t *ids, const uint8_t *codes, const float *centers, const float *queries,
                         const float *R, int64_t n, int64_t nq, int m, int d, int k, int id_base, const int *devices,
                         int ndev) {
  std::lock_guard<std::mutex> lk(g_shc_mu);      // host-pointer calls over a device set run one at a time
  SavedDevice saved;
  const std::vector<int> devs(devices, devices + ndev);
  int rc = RQ_OK;
  if (g_shc_ix && (g_shc_devs != devs || g_shc_ix->m != m || g_shc_ix->d != d || !tuning("SHARDED_CACHE", 1))) {
    index_free(g_shc_ix);
    g_shc_ix = nullptr;
  }
  if (!g_shc_ix) {
    rq_index *ix = new rq_index();
    rc = index_build(ix, m, d, centers, devices, ndev);
    if (rc != RQ_OK) { index_free(ix); return rc; }
    g_shc_ix = ix;
    g_shc_devs = devs;
  } else {
    const size_t ce = (size_t)m * 256 * (d / m) * 4;      // same shape, new codebooks
    for (auto &dv : g_shc_ix->devs) {
      RQ_HIP(hipSetDevice(dv.device));
      RQ_HIP(hipMemcpy(dv.centers, centers, ce, hipMemcpyHostToDevice));
    }
  }
  rc = rq_index_set_codes(g_shc_ix, codes, n, 0);
  if (rc == RQ_OK) rc = index_search(g_shc_ix, dists, ids, queries, R, nq, k, id_base);
  if (rc != RQ_OK || !tuning("SHARDED_CACHE", 1)) {       // never keep an index that failed half-way
    index_free(g_shc_ix);
    g_shc_ix = nullptr;
    g_shc_devs.clear();
  }
  return rc;
}

}  // namespace rq

extern "C" {

rq_index *rq_index_create_sharded(int m, int d, const float *centers_host, const int *devices, int ndev) {
  if (!devices) { fail(RQ_EINVAL, "index: devices is NULL"); return nullptr; }
  rq_index *ix = new rq_index();
  SavedDevice saved;
  if (index_build(ix, m, d, centers_host, devices, ndev) != RQ_OK) { index_free(ix); return nullptr; }
  return ix;
}

rq_index *rq_index_create(int m, int d, const float *centers_host) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); fail(RQ_ENODEVICE, "no device"); return nullptr; }
  return rq_index_create_sharded(m, d, centers_host, &dev, 1);
}

int rq_index_set_codes(rq_index *ix, const uint8_t *codes_host, int64_t n, uint32_t id_offset) {
  if (!codes_host) return fail(RQ_EINVAL, "index: codes is NULL");
  return index_set(ix, n, id_offset, [&](IxShard &s, hipStream_t stream) -> int {
    RQ_HIP(hipMemcpyAsync(s.codes, codes_host + (size_t)s.row0 * ix->m, (size_t)s.n * ix->m, hipMemcpyHostToDevice, stream));
    return RQ_OK;
  });
}

int rq_index_set_codes_synth(rq_index *ix, int64_t n, uint64_t seed, uint32_t id_offset) {
  return index_set(ix, n, id_offset, [&](IxShard &s, hipStream_t stream) -> int {
    return synth_codes_launch(s.codes, s.n, ix->m, seed, s.row0, stream);
  });
}

int rq_index_search(rq_index *ix, float *dists, uint32_t *ids, const float *queries_host, int64_t nq, int k,
                    int id_base) {
  return index_search(ix, dists, ids, queries_host, nullptr, nq, k, id_base);
}

int rq_index_search_opq(rq_index *ix, float *dists, uint32_t *ids, const float *queries_host, const float *R_host,
                        int64_t nq, int k, int id_base) {
  if (!R_host) return fail(RQ_EINVAL, "R is NULL");
  return index_search(ix, dists, ids, queries_host, R_host, nq, k, id_base);
}

int rq_index_info(rq_index *ix, int64_t *out, int cap) {
  if (!ix || !out || cap < 4) return fail(RQ_EINVAL, "rq_index_info: bad arguments");
  out[0] = (int64_t)ix->shards.size();
  out[1] = (int64_t)ix->devs.size();
  out[2] = ix->exchange;
  out[3] = ix->n;
  for (size_t s = 0; s < ix->shards.size() && 4 + (int)s < cap; ++s) out[4 + s] = ix->shards[s].n;
  return RQ_OK;
}

void rq_index_destroy(rq_index *ix) { index_free(ix); }

}  // extern "C"
