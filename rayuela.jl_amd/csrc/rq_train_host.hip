// rq_train_host.hip -- train_pq / train_opq as C-ABI entry points.
//
// Host pointers in and out, like the Julia callers have them (src/PQ.jl:68-99, src/OPQ.jl:49-139);
// X is uploaded once and every O(n) step of every iteration runs on the device (encode, rotation and the
// reductions of rq_train.hip).  What stays on the host is what the reference keeps in scalar Julia/LAPACK:
// the d x d SVD of X CB' (a one-sided Jacobi in double here: no LAPACK dependency) and the bookkeeping.
// Initial centres of train_pq / train_rvq come from kmeans++ (D^2 sampling on the device, rq_train.hip), like the
// reference's kmeans(..., init=:kmpp); train_opq samples h rows like src/OPQ.jl:82.  Randomness (the uniforms of
// kmeans++, sampled rows, re-seeding of empty clusters, init == random) comes from a splitmix64 stream seeded by
// the caller -- the reference draws from Julia's global RNG, so runs are comparable statistically, not bit for bit.
#include <math.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "rq_internal.h"

namespace rq {

struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) / 9007199254740992.0; }
  double normal() {  // Box-Muller
    double u1 = uniform(), u2 = uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
  }
};

// h distinct indices out of n (sample(1:n, h, replace=false), src/OPQ.jl:82): partial Fisher-Yates on a
// sparse map when n is large
static void sample_distinct(Rng &rng, int64_t n, int h, std::vector<int64_t> &out) {
  out.clear();
  std::vector<std::pair<int64_t, int64_t>> swaps;  // position -> value for touched positions
  auto get = [&](int64_t pos) {
    for (auto &p : swaps) if (p.first == pos) return p.second;
    return pos;
  };
  auto set = [&](int64_t pos, int64_t val) {
    for (auto &p : swaps) if (p.first == pos) { p.second = val; return; }
    swaps.push_back({pos, val});
  };
  for (int i = 0; i < h; ++i) {
    const int64_t j = i + (int64_t)(rng.next() % (uint64_t)(n - i));
    const int64_t vi = get(i), vj = get(j);
    set(i, vj); set(j, vi);
    out.push_back(vj);
  }
}

// Polar factor U V' of the d x d matrix G (row-major) by one-sided Jacobi SVD in double.
// G = U S V'  ->  out = U V'   (src/OPQ.jl:112-113: U, S, VV = svd(X * CB'); R = U * VV')
// Vwarm (optional, d x d row-major, in/out): right singular vectors of the previous call.  Successive
// G = X CB' of an OPQ run differ little, so starting from A = G Vwarm, V = Vwarm the sweeps converge in
// 2-3 instead of ~10; the polar factor itself does not depend on the starting point.
static void polar_factor(const double *G, double *out, int d, std::vector<double> *Vwarm = nullptr) {
  // columns are stored as rows (At[p] = column p of G, Vt[p] = column p of V): every inner loop runs over
  // contiguous memory, the three dot products keep four partial sums each (4 independent fma chains
  // instead of one latency-bound chain) -- 128 x 128: 25 ms -> ~5 ms per call
  std::vector<double> At((size_t)d * d), Vt((size_t)d * d, 0.0);
  if (Vwarm && Vwarm->size() == (size_t)d * d) {
    // Vt[p] = column p of Vwarm;  At[p] = column p of G Vwarm = sum_k G[:, k] Vwarm[k][p]
    for (int k = 0; k < d; ++k)
      for (int p = 0; p < d; ++p) Vt[(size_t)p * d + k] = (*Vwarm)[(size_t)k * d + p];
    std::fill(At.begin(), At.end(), 0.0);
    for (int p = 0; p < d; ++p) {
      double *ap = &At[(size_t)p * d];
      const double *vp = &Vt[(size_t)p * d];
      for (int i = 0; i < d; ++i) {
        const double *gi = &G[(size_t)i * d];
        double acc = 0;
        for (int k = 0; k < d; ++k) acc += gi[k] * vp[k];
        ap[i] = acc;
      }
    }
  } else {
    for (int i = 0; i < d; ++i)
      for (int p = 0; p < d; ++p) At[(size_t)p * d + i] = G[(size_t)i * d + p];
    for (int i = 0; i < d; ++i) Vt[(size_t)i * d + i] = 1.0;
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < d - 1; ++p) {
      double *ap = &At[(size_t)p * d], *vp = &Vt[(size_t)p * d];
      for (int q = p + 1; q < d; ++q) {
        double *aq = &At[(size_t)q * d], *vq = &Vt[(size_t)q * d];
        double al[4] = {0, 0, 0, 0}, be[4] = {0, 0, 0, 0}, ga[4] = {0, 0, 0, 0};
        int i = 0;
        for (; i + 4 <= d; i += 4)
          for (int u = 0; u < 4; ++u) {
            al[u] += ap[i + u] * ap[i + u];
            be[u] += aq[i + u] * aq[i + u];
            ga[u] += ap[i + u] * aq[i + u];
          }
        for (; i < d; ++i) { al[0] += ap[i] * ap[i]; be[0] += aq[i] * aq[i]; ga[0] += ap[i] * aq[i]; }
        const double alpha = (al[0] + al[1]) + (al[2] + al[3]), beta = (be[0] + be[1]) + (be[2] + be[3]);
        const double gamma = (ga[0] + ga[1]) + (ga[2] + ga[3]);
        if (alpha == 0.0 || beta == 0.0) continue;
        const double lim = fabs(gamma) / sqrt(alpha * beta);
        if (lim > off) off = lim;
        if (lim < 1e-15) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
        for (int k = 0; k < d; ++k) {
          const double x = ap[k], y = aq[k];
          ap[k] = c * x - sn * y;
          aq[k] = sn * x + c * y;
        }
        for (int k = 0; k < d; ++k) {
          const double x = vp[k], y = vq[k];
          vp[k] = c * x - sn * y;
          vq[k] = sn * x + c * y;
        }
      }
    }
    if (off < 1e-14) break;
  }
  if (Vwarm) {   // keep V for the next call
    Vwarm->assign((size_t)d * d, 0.0);
    for (int p = 0; p < d; ++p)
      for (int k = 0; k < d; ++k) (*Vwarm)[(size_t)k * d + p] = Vt[(size_t)p * d + k];
  }
  // rows of At are the columns of U * S: normalise (a null column keeps the matching column of V: any
  // orthonormal completion is a valid polar factor there)
  for (int j = 0; j < d; ++j) {
    double *aj = &At[(size_t)j * d];
    const double *vj = &Vt[(size_t)j * d];
    double nrm = 0;
    for (int i = 0; i < d; ++i) nrm += aj[i] * aj[i];
    nrm = sqrt(nrm);
    for (int i = 0; i < d; ++i) aj[i] = nrm > 1e-300 ? aj[i] / nrm : vj[i];
  }
  // out = U V':  out[i][j] = sum_k U[i][k] V[j][k] = sum_k At[k][i] Vt[k][j]
  for (size_t e = 0; e < (size_t)d * d; ++e) out[e] = 0.0;
  for (int k = 0; k < d; ++k) {
    const double *uk = &At[(size_t)k * d], *vk = &Vt[(size_t)k * d];
    for (int i = 0; i < d; ++i) {
      const double u = uk[i];
      double *o = &out[(size_t)i * d];
      for (int j = 0; j < d; ++j) o[j] += u * vk[j];
    }
  }
}

// Phase clock of the training loops (rq_train_profile).  The loop's wall time (slot LOOP) and the iteration count are always
// recorded -- two device synchronisations per call; with tuning TRAIN_PROFILE = 1 every phase is bracketed by
// synchronisations too, so the per-phase figures are exact and the loop total a little longer than an unprofiled run's.
enum { TP_H2D = 0, TP_INIT, TP_QERROR, TP_GRAM, TP_SVD, TP_ROTATE, TP_CENTERS, TP_ENCODE, TP_RECONSTRUCT, TP_CONVERGE,
       TP_D2H, TP_LOOP, TP_ITERS, TP_SWEEPS, TP_NS_STEPS, TP_HOST_POLAR, TP_SLOTS = 16 };
static thread_local double g_train_prof[TP_SLOTS];
struct TrainProf {
  bool fine;
  std::chrono::steady_clock::time_point t0, tl;
  TrainProf() : fine(tuning("TRAIN_PROFILE", 0) != 0) { for (double &x : g_train_prof) x = 0.0; }
  static double since(std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  }
  void start() { if (fine) { (void)hipDeviceSynchronize(); t0 = std::chrono::steady_clock::now(); } }
  void stop(int slot) { if (fine) { (void)hipDeviceSynchronize(); g_train_prof[slot] += since(t0); } }
  void loop_begin() { (void)hipDeviceSynchronize(); tl = std::chrono::steady_clock::now(); }
  void loop_end(int iters) { (void)hipDeviceSynchronize(); g_train_prof[TP_LOOP] = since(tl); g_train_prof[TP_ITERS] = iters; }
};
#define RQ_PH(slot, stmt) do { prof.start(); stmt; prof.stop(slot); } while (0)

struct DevMem {
  void *p = nullptr;
  ~DevMem() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { RQ_HIP(hipMalloc(&p, bytes ? bytes : 16)); return RQ_OK; }
  template <class T> T *as() { return reinterpret_cast<T *>(p); }
};

// the device's persistent non-blocking compute stream (the caller holds the DeviceLock) with the two events that order it
// against the default stream
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;
  int create() {
    hipStream_t xs = nullptr;
    RQ_TRY(aux_streams(&s, &xs));
    RQ_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    RQ_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    return RQ_OK;
  }
  ~SideStream() {
    if (s) (void)hipStreamSynchronize(s);
    if (ready) (void)hipEventDestroy(ready);
    if (done) (void)hipEventDestroy(done);
  }
};

static void offsets(int *off, int d, int m) {
  const int per = d / m, extra = d % m;
  int pos = 0;
  for (int i = 0; i < m; ++i) { off[i] = pos; pos += per + (i < extra ? 1 : 0); }
  off[m] = pos;
}

// initial centres: C_i = h sampled rows of (rotated) X restricted to subspace i
static int init_centers(float *dC, const float *dX, int64_t n, int d, int m, int h, const int *off, Rng &rng) {
  std::vector<int64_t> idx;
  for (int i = 0; i < m; ++i) {
    sample_distinct(rng, n, h, idx);
    const int sub = off[i + 1] - off[i];
    for (int k = 0; k < h; ++k)
      RQ_HIP(hipMemcpy(dC + (size_t)h * off[i] + (size_t)k * sub, dX + idx[k] * d + off[i], sizeof(float) * sub,
                       hipMemcpyDeviceToDevice));
  }
  return RQ_OK;
}

// initial centres by kmeans++ (src/PQ.jl:86 / src/RVQ.jl:104: kmeans(..., init=:kmpp)); seeds_out (host, may be NULL)
// receives the m x h chosen rows.  TRAIN_KMPP=0 falls back to uniformly sampled rows.
static int seed_centers(float *dC, const float *dX, int64_t n, int d, int m, int h, const int *off, Rng &rng,
                        long long *seeds_out = nullptr) {
  if (!tuning("TRAIN_KMPP", 1) && !seeds_out) return init_centers(dC, dX, n, d, m, h, off, rng);
  DevMem dseeds, dmin, dpart, du;
  RQ_TRY(dseeds.alloc((size_t)m * h * 8)); RQ_TRY(dmin.alloc((size_t)m * n * 4));
  RQ_TRY(dpart.alloc((size_t)m * 1024 * 8)); RQ_TRY(du.alloc((size_t)m * h * 8));
  std::vector<double> u((size_t)m * h);
  for (auto &x : u) x = rng.uniform();
  RQ_HIP(hipMemcpy(du.p, u.data(), u.size() * 8, hipMemcpyHostToDevice));
  RQ_TRY(kmpp_init_launch(dC, dseeds.as<long long>(), dmin.as<float>(), dpart.as<double>(), du.as<double>(), dX, n, d, m,
                          h, nullptr));
  RQ_HIP(hipDeviceSynchronize());
  if (seeds_out) RQ_HIP(hipMemcpy(seeds_out, dseeds.p, (size_t)m * h * 8, hipMemcpyDeviceToHost));
  return RQ_OK;
}

static int check_train(int64_t n, int d, int m, int h, int niter) {
  if (n < 1 || d < 1 || m < 1 || m > 32 || d < m || h < 1 || h > 256 || niter < 0)
    return fail(RQ_EINVAL, "train: n=%lld d=%d m=%d h=%d niter=%d", (long long)n, d, m, h, niter);
  if (n < h) return fail(RQ_EINVAL, "train: fewer training vectors (%lld) than codebook entries (%d)", (long long)n, h);
  return RQ_OK;
}

// Clustering.repick_unused_centers (Clustering.jl v0.12.2, called by kmeans' loop right after update_centers! -- the k-means of
// src/PQ.jl:86 and src/RVQ.jl:104): every centre that lost all its points is re-drawn from the data with probability
// proportional to the points' CURRENT cost (distance to their assigned centre), kmeans++ style: the drawn point's cost drops to
// zero and all costs are lowered to the distance to the new centre before the next draw.  Empty clusters are rare (none in the
// bench runs), so this runs on the host: the sub-space columns and the stage's codes come down once per affected sub-space.
// dCsub: device [h][sub] block of the sub-codebook; dX: device rows of `d` floats, the sub-space starts at column col0;
// codes: device bytes, the row's code at codes[row * cstride + ci].  The draws use the library's seeded stream (not Julia's).
// dCassigned: the same block as it was when `codes` were assigned, i.e. BEFORE update_centers -- Clustering draws with the
// costs of the last assignment (`costs` of update_assignments!), not with distances to the centres just recomputed.
static int repick_unused(float *dCsub, const float *dCassigned, const float *dX, int64_t n, int d, int col0, int sub,
                         const uint8_t *codes, int cstride, int ci, int h, const std::vector<int> &unused, Rng &rng) {
  std::vector<float> xs((size_t)n * sub), cs((size_t)h * sub);
  std::vector<uint8_t> cb((size_t)n);
  RQ_HIP(hipMemcpy2D(xs.data(), (size_t)sub * 4, dX + col0, (size_t)d * 4, (size_t)sub * 4, (size_t)n, hipMemcpyDeviceToHost));
  RQ_HIP(hipMemcpy2D(cb.data(), 1, codes + ci, (size_t)cstride, 1, (size_t)n, hipMemcpyDeviceToHost));
  RQ_HIP(hipMemcpy(cs.data(), dCassigned, (size_t)h * sub * 4, hipMemcpyDeviceToHost));
  std::vector<double> tc((size_t)n);
  for (int64_t j = 0; j < n; ++j) {
    const float *x = &xs[(size_t)j * sub], *c = &cs[(size_t)cb[j] * sub];
    double a = 0;
    for (int t = 0; t < sub; ++t) { const double e = (double)x[t] - (double)c[t]; a += e * e; }
    tc[j] = a;
  }
  for (int k : unused) {
    double total = 0;
    for (int64_t j = 0; j < n; ++j) total += tc[j];
    int64_t pick = (int64_t)(rng.next() % (uint64_t)n);          // all costs zero (fewer distinct points than centres): uniform
    if (total > 0) {
      const double u = rng.uniform() * total;
      double run = 0;
      pick = n - 1;
      for (int64_t j = 0; j < n; ++j) { run += tc[j]; if (run > u) { pick = j; break; } }
    }
    const float *v = &xs[(size_t)pick * sub];
    RQ_HIP(hipMemcpy(dCsub + (size_t)k * sub, v, (size_t)sub * 4, hipMemcpyHostToDevice));
    tc[pick] = 0;
    for (int64_t j = 0; j < n; ++j) {
      const float *x = &xs[(size_t)j * sub];
      double a = 0;
      for (int t = 0; t < sub; ++t) { const double e = (double)x[t] - (double)v[t]; a += e * e; }
      if (a < tc[j]) tc[j] = a;
    }
  }
  return RQ_OK;
}

}  // namespace rq

using namespace rq;

extern "C" {

int rq_kmpp_seeds(int64_t *seeds, float *C, const float *X, int64_t n, int d, int m, int h, uint64_t seed) {
  RQ_TRY(check_train(n, d, m, h, 0));
  if (!seeds && !C) return fail(RQ_EINVAL, "rq_kmpp_seeds: nothing to return");
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;
  int off[33];
  offsets(off, d, m);
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 1};
  DevMem dX, dC;
  RQ_TRY(dX.alloc((size_t)n * d * 4)); RQ_TRY(dC.alloc((size_t)h * d * 4));
  RQ_HIP(hipMemcpy(dX.p, X, (size_t)n * d * 4, hipMemcpyHostToDevice));
  std::vector<long long> sd((size_t)m * h);
  RQ_TRY(seed_centers(dC.as<float>(), dX.as<float>(), n, d, m, h, off, rng, sd.data()));
  if (seeds) for (size_t i = 0; i < sd.size(); ++i) seeds[i] = (int64_t)sd[i];
  if (C) RQ_HIP(hipMemcpy(C, dC.p, (size_t)h * d * 4, hipMemcpyDeviceToHost));
  return RQ_OK;
}

int rq_train_pq(float *C, int16_t *B1, double *error, const float *X, int64_t n, int d, int m, int h, int niter,
                uint64_t seed) {
  RQ_TRY(check_train(n, d, m, h, niter));
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;
  int off[33];
  offsets(off, d, m);
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 1};
  DevMem dX, dC, dcodes, dprev, dcnt, dCB, dacc, d16;
  RQ_TRY(dX.alloc((size_t)n * d * 4)); RQ_TRY(dC.alloc((size_t)h * d * 4)); RQ_TRY(dcodes.alloc((size_t)n * m));
  RQ_TRY(dprev.alloc((size_t)n * m)); RQ_TRY(dcnt.alloc((size_t)m * h * 4));
  RQ_TRY(dacc.alloc(8)); RQ_TRY(d16.alloc((size_t)n * m * 2));
  TrainProf prof;
  RQ_PH(TP_H2D, RQ_HIP(hipMemcpy(dX.p, X, (size_t)n * d * 4, hipMemcpyHostToDevice)));
  RQ_PH(TP_INIT, RQ_TRY(seed_centers(dC.as<float>(), dX.as<float>(), n, d, m, h, off, rng)));
  std::vector<unsigned int> counts((size_t)m * h);
  // Convergence.  Clustering.kmeans (v0.12.2 `_kmeans!`) stops when |objv - prev_objv| < tol, tol = 1e-6 ABSOLUTE on
  // objv = sum(costs), a Float32 sum: for any data whose objective exceeds ~10 (float32 resolution 1e-6) that is "the rounded
  // objective did not move", which an iteration without a changed assignment produces exactly (same assignments -> same centres
  // -> same costs) and one WITH changed assignments practically never does.  So the loop stops on "no assignment changed" --
  // the same stopping point on the bench shapes (tests/test_gpu_train.py compares the final error with the oracle loop that
  // applies Clustering's own rule) -- and does not pay a pass over X per iteration for the objective.  The codes of two consecutive
  // iterations are compared ON THE DEVICE (a D2H of n*m bytes + a host compare per iteration cost more than the encode)
  // The change counter sits right behind the cluster counts: ONE small read-back per iteration serves the convergence test
  // and the empty-cluster check (the centres are recomputed before the test is known; with unchanged assignments that
  // reproduces them bit for bit).  The two code buffers swap roles instead of being copied.
  DevMem dCold;        // the centres the current codes were assigned with (repick_unused draws with THOSE costs)
  RQ_TRY(dCold.alloc((size_t)h * d * 4));
  DevMem dcc;
  const size_t cnt_bytes = (size_t)m * h * 4;
  RQ_TRY(dcc.alloc(cnt_bytes + 8));
  unsigned int *dcnt_p = dcc.as<unsigned int>();
  unsigned long long *dchg_p = reinterpret_cast<unsigned long long *>(dcc.as<unsigned char>() + cnt_bytes);
  std::vector<unsigned char> back(cnt_bytes + 8);
  uint8_t *cur = dcodes.as<uint8_t>(), *prev = dprev.as<uint8_t>();
  int iters_done = 0;
  prof.loop_begin();
  for (int it = 0; it < niter; ++it) {
    RQ_PH(TP_ENCODE, RQ_TRY(encode_launch(cur, dX.as<float>(), dC.as<float>(), n, d, m, h, di.num_cu, nullptr)));
    prof.start();
    if (it > 0) RQ_TRY(codes_changed_launch(dchg_p, cur, prev, (size_t)n * m, nullptr));
    prof.stop(TP_CONVERGE);
    RQ_HIP(hipMemcpyAsync(dCold.p, dC.p, (size_t)h * d * 4, hipMemcpyDeviceToDevice, nullptr));
    RQ_PH(TP_CENTERS, RQ_TRY(update_centers_launch(dC.as<float>(), dcnt_p, dX.as<float>(), cur, n, d, m, h, di.num_cu, nullptr)));
    prof.start();
    RQ_HIP(hipMemcpy(back.data(), dcc.p, cnt_bytes + 8, hipMemcpyDeviceToHost));
    unsigned long long changed = 1;
    if (it > 0) memcpy(&changed, back.data() + cnt_bytes, 8);
    prof.stop(TP_CONVERGE);
    if (!changed) break;   // assignments stable: Lloyd has converged
    ++iters_done;
    memcpy(counts.data(), back.data(), cnt_bytes);
    for (int i = 0; i < m; ++i) {             // centres that lost all their points: re-drawn like Clustering.kmeans does
      std::vector<int> unused;
      for (int k = 0; k < h; ++k)
        if (counts[(size_t)i * h + k] == 0) unused.push_back(k);
      if (!unused.empty())
        RQ_TRY(repick_unused(dC.as<float>() + (size_t)h * off[i], dCold.as<float>() + (size_t)h * off[i], dX.as<float>(), n, d,
                             off[i], off[i + 1] - off[i], cur, m, i, h, unused, rng));
    }
    std::swap(cur, prev);
  }
  if (cur != dcodes.as<uint8_t>()) std::swap(dcodes.p, dprev.p);      // the final encode below writes (and the download reads) dcodes
  prof.loop_end(iters_done);
  RQ_PH(TP_ENCODE, RQ_TRY(encode_launch(dcodes.as<uint8_t>(), dX.as<float>(), dC.as<float>(), n, d, m, h, di.num_cu, nullptr)));
  if (codes_forms_ok(d, m, h, false)) {
    RQ_PH(TP_QERROR, RQ_TRY(qerror_codes_launch(dacc.as<double>(), dX.as<float>(), dcodes.as<uint8_t>(), dC.as<float>(), n, d, m, h, di.num_cu, nullptr)));
  } else {
    RQ_TRY(dCB.alloc((size_t)n * d * 4));
    RQ_PH(TP_RECONSTRUCT, RQ_TRY(reconstruct_launch(dCB.as<float>(), dcodes.as<uint8_t>(), dC.as<float>(), n, d, m, h, nullptr)));
    RQ_PH(TP_QERROR, RQ_TRY(qerror_launch(dacc.as<double>(), dX.as<float>(), dCB.as<float>(), n, d, di.num_cu, nullptr)));
  }
  RQ_TRY(widen_codes_launch(d16.as<int16_t>(), dcodes.as<uint8_t>(), n * m, nullptr));
  double acc = 0;
  RQ_HIP(hipMemcpy(&acc, dacc.p, 8, hipMemcpyDeviceToHost));
  if (error) *error = acc / (double)n;
  prof.start();
  RQ_HIP(hipMemcpy(C, dC.p, (size_t)h * d * 4, hipMemcpyDeviceToHost));
  RQ_HIP(hipMemcpy(B1, d16.p, (size_t)n * m * 2, hipMemcpyDeviceToHost));
  prof.stop(TP_D2H);
  return RQ_OK;
}

// train_rvq (src/RVQ.jl:86-127): one k-means per stage on the running residual (Clustering.kmeans with
// kmeans++ seeding and Julia's RNG there; h sampled residual rows from the library's seeded stream here),
// then Xr .-= C[i][:, B[:, i]] (:112).  C [m][h][d]; B1 [n][m] Int16 one-based; error = qerror(X, B, C) (:124).
int rq_train_rvq(float *C, int16_t *B1, double *error, const float *X, int64_t n, int d, int m, int h, int niter,
                 uint64_t seed) {
  if (n < 1 || d < 1 || m < 1 || m > 64 || h < 1 || h > 256 || niter < 0)
    return fail(RQ_EINVAL, "train_rvq: n=%lld d=%d m=%d h=%d niter=%d", (long long)n, d, m, h, niter);
  if (n < h) return fail(RQ_EINVAL, "train_rvq: fewer training vectors (%lld) than codebook entries (%d)", (long long)n, h);
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;
  int off1[2] = {0, d};
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 3};
  DevMem dXr, dCi, dCold, dstage, dcnt, dcodes, dacc, d16;
  RQ_TRY(dXr.alloc((size_t)n * d * 4)); RQ_TRY(dCi.alloc((size_t)h * d * 4)); RQ_TRY(dCold.alloc((size_t)h * d * 4));
  RQ_TRY(dstage.alloc((size_t)n));
  RQ_TRY(dcnt.alloc((size_t)h * 4)); RQ_TRY(dcodes.alloc((size_t)n * m)); RQ_TRY(dacc.alloc(8));
  RQ_TRY(d16.alloc((size_t)n * m * 2));
  RQ_HIP(hipMemcpy(dXr.p, X, (size_t)n * d * 4, hipMemcpyHostToDevice));
  std::vector<unsigned int> counts((size_t)h);
  std::vector<uint8_t> cur((size_t)n), prev;
  for (int i = 0; i < m; ++i) {
    RQ_TRY(seed_centers(dCi.as<float>(), dXr.as<float>(), n, d, 1, h, off1, rng));
    prev.clear();
    for (int it = 0; it < niter; ++it) {
      RQ_TRY(encode_launch(dstage.as<uint8_t>(), dXr.as<float>(), dCi.as<float>(), n, d, 1, h, di.num_cu, nullptr));
      RQ_HIP(hipMemcpy(cur.data(), dstage.p, (size_t)n, hipMemcpyDeviceToHost));
      if (!prev.empty() && prev == cur) break;   // assignments stable: Lloyd has converged
      prev = cur;
      RQ_HIP(hipMemcpyAsync(dCold.p, dCi.p, (size_t)h * d * 4, hipMemcpyDeviceToDevice, nullptr));
      RQ_TRY(update_centers_launch(dCi.as<float>(), dcnt.as<unsigned int>(), dXr.as<float>(), dstage.as<uint8_t>(), n,
                                   d, 1, h, di.num_cu, nullptr));
      RQ_HIP(hipMemcpy(counts.data(), dcnt.p, (size_t)h * 4, hipMemcpyDeviceToHost));
      std::vector<int> unused;                   // centres that lost all their points: re-drawn like Clustering.kmeans does
      for (int k = 0; k < h; ++k)
        if (counts[k] == 0) unused.push_back(k);
      if (!unused.empty())
        RQ_TRY(repick_unused(dCi.as<float>(), dCold.as<float>(), dXr.as<float>(), n, d, 0, d, dstage.as<uint8_t>(), 1, 0, h, unused, rng));
    }
    // the assignments of the final centres (what quantize_rvq would return for this stage), then the residual
    RQ_TRY(encode_launch(dstage.as<uint8_t>(), dXr.as<float>(), dCi.as<float>(), n, d, 1, h, di.num_cu, nullptr));
    RQ_TRY(rvq_residual_launch(dXr.as<float>(), dCi.as<float>(), dstage.as<uint8_t>(), dcodes.as<uint8_t>(), nullptr, n,
                               d, m, i, nullptr));
    RQ_HIP(hipMemcpy(C + (size_t)i * h * d, dCi.p, (size_t)h * d * 4, hipMemcpyDeviceToHost));
  }
  RQ_TRY(qerror_launch(dacc.as<double>(), dXr.as<float>(), nullptr, n, d, di.num_cu, nullptr));
  RQ_TRY(widen_codes_launch(d16.as<int16_t>(), dcodes.as<uint8_t>(), n * m, nullptr));
  double acc = 0;
  RQ_HIP(hipMemcpy(&acc, dacc.p, 8, hipMemcpyDeviceToHost));
  if (error) *error = acc / (double)n;
  RQ_HIP(hipMemcpy(B1, d16.p, (size_t)n * m * 2, hipMemcpyDeviceToHost));
  return RQ_OK;
}

int rq_train_opq(float *C, int16_t *B1, float *R, float *obj, const float *X, int64_t n, int d, int m, int h,
                 int niter, int init, uint64_t seed, const float *R0, const float *C0) {
  RQ_TRY(check_train(n, d, m, h, niter));
  if (init != 0 && init != 1) return fail(RQ_EINVAL, "train_opq: init must be 0 (natural) or 1 (random)");
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DeviceLock call_lock;
  int off[33];
  offsets(off, d, m);
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 2};
  std::vector<float> Rh((size_t)d * d, 0.0f);   // memory image of Julia's R: Rh[i*d+k] = R[k, i]
  std::vector<double> G((size_t)d * d), P((size_t)d * d);
  if (R0) {
    memcpy(Rh.data(), R0, sizeof(float) * d * d);
  } else if (init == 0) {
    for (int i = 0; i < d; ++i) Rh[(size_t)i * d + i] = 1.0f;
  } else {  // R, _, _ = svd(randn(d, d))  (src/OPQ.jl:72): an orthonormal basis of a Gaussian matrix
    for (auto &g : G) g = rng.normal();
    polar_factor(G.data(), P.data(), d);
    for (int i = 0; i < d; ++i)
      for (int k = 0; k < d; ++k) Rh[(size_t)i * d + k] = (float)P[(size_t)k * d + i];
  }
  DevMem dX, dRX, dR, dC, dcodes, dcnt, dCB, dacc, dG, d16;
  RQ_TRY(dX.alloc((size_t)n * d * 4)); RQ_TRY(dRX.alloc((size_t)n * d * 4)); RQ_TRY(dR.alloc((size_t)d * d * 4));
  RQ_TRY(dC.alloc((size_t)h * d * 4)); RQ_TRY(dcodes.alloc((size_t)n * m)); RQ_TRY(dcnt.alloc((size_t)m * h * 4));
  // CB = reconstruct(codes, C) is only materialised for shapes the (codes, C) forms of gram / qerror do not cover
  const bool fused_cb = codes_forms_ok(d, m, h, true);
  if (!fused_cb) RQ_TRY(dCB.alloc((size_t)n * d * 4));
  RQ_TRY(dacc.alloc(8)); RQ_TRY(dG.alloc((size_t)d * d * 4));
  RQ_TRY(d16.alloc((size_t)n * m * 2));
  TrainProf prof;
  RQ_PH(TP_H2D, RQ_HIP(hipMemcpy(dX.p, X, (size_t)n * d * 4, hipMemcpyHostToDevice)));
  prof.start();
  RQ_HIP(hipMemcpy(dR.p, Rh.data(), (size_t)d * d * 4, hipMemcpyHostToDevice));
  RQ_TRY(rotate_launch(dRX.as<float>(), dR.as<float>(), dX.as<float>(), d, n, di.num_cu, nullptr));
  if (C0) RQ_HIP(hipMemcpy(dC.p, C0, (size_t)h * d * 4, hipMemcpyHostToDevice));
  else RQ_TRY(init_centers(dC.as<float>(), dRX.as<float>(), n, d, m, h, off, rng));
  RQ_TRY(encode_launch(dcodes.as<uint8_t>(), dRX.as<float>(), dC.as<float>(), n, d, m, h, di.num_cu, nullptr));
  if (!fused_cb) RQ_TRY(reconstruct_launch(dCB.as<float>(), dcodes.as<uint8_t>(), dC.as<float>(), n, d, m, h, nullptr));
  prof.stop(TP_INIT);
  std::vector<float> Gf((size_t)d * d);
  std::vector<double> Vwarm;   // empty on the first iteration: cold start
  // the d x d polar factor on the device (rq_train.hip): scaled Newton-Schulz (any d <= 1024), then -- should it not converge:
  // a singular G -- the one-sided Jacobi SVD (even d <= 128), then the host Jacobi, which completes the basis.
  // TRAIN_GPU_POLAR: 0 = host only, 1 = the chain above, 2 = skip Newton-Schulz
  const int polar_mode = tuning("TRAIN_GPU_POLAR", 1);
  const bool dev_ns = polar_mode == 1 && d <= 1024;
  const bool dev_polar = polar_mode != 0 && d >= 2 && d <= 128 && (d & 1) == 0;
  DevMem dVw, dscr, dstat, dns;
  RQ_TRY(dstat.alloc(8));
  if (dev_polar) { RQ_TRY(dVw.alloc((size_t)d * d * 8)); RQ_TRY(dscr.alloc(((size_t)d * d + d) * 8)); }
  if (dev_ns) RQ_TRY(dns.alloc(polar_ns_scratch_bytes(d, di.num_cu)));
  bool dev_warm = false;
  prof.loop_begin();
  // The objective of every iteration is computed on a SIDE STREAM while the main stream goes on with X'CB and the polar factor
  // (the objective reads RX, the codes and C; the rotation that overwrites RX waits for it), and the niter + 1 values are
  // read back once, after the loop -- not one blocking 8-byte copy per iteration.
  SideStream side;
  RQ_TRY(side.create());
  DevMem dobj;
  RQ_TRY(dobj.alloc((size_t)(niter + 1) * 8));
  for (int it = 0; it <= niter; ++it) {
    // objective |R CB - X|^2 / n == |CB - R'X|^2 / n (src/OPQ.jl:108)
    prof.start();
    RQ_HIP(hipEventRecord(side.ready, nullptr));
    RQ_HIP(hipStreamWaitEvent(side.s, side.ready, 0));
    if (fused_cb) RQ_TRY(qerror_codes_launch(dobj.as<double>() + it, dRX.as<float>(), dcodes.as<uint8_t>(), dC.as<float>(), n, d, m, h, di.num_cu, side.s));
    else RQ_TRY(qerror_launch(dobj.as<double>() + it, dRX.as<float>(), dCB.as<float>(), n, d, di.num_cu, side.s));
    RQ_HIP(hipEventRecord(side.done, side.s));
    prof.stop(TP_QERROR);
    // update R (src/OPQ.jl:112-113): G = X CB' and its polar factor U V' on the device
    prof.start();
    if (fused_cb) RQ_TRY(gram_codes_launch(dG.as<float>(), dX.as<float>(), dcodes.as<uint8_t>(), dC.as<float>(), n, d, m, h, di.num_cu, nullptr));
    else RQ_TRY(gram_launch(dG.as<float>(), dX.as<float>(), dCB.as<float>(), n, d, di.num_cu, nullptr));
    prof.stop(TP_GRAM);
    prof.start();
    int pstat = 1;
    if (dev_ns) {
      // The Newton-Schulz kernel synchronises its <= num_cu workgroups with a grid barrier and needs them all resident.  The
      // objective kernel on the side stream (num_cu * 8 workgroups) would compete for the CUs, and a barrier that times out
      // falls back to the Jacobi / host factor -- other bits, i.e. the iteration's result would depend on timing.  With
      // TRAIN_DETERMINISTIC = 1 (default) the main stream waits for the objective first (measured: +0.03 ms per iteration
      // of 1.5); 0 keeps the overlap and reports the fallbacks in rq_train_profile()[15].
      if (tuning("TRAIN_DETERMINISTIC", 1)) RQ_HIP(hipStreamWaitEvent(nullptr, side.done, 0));
      RQ_HIP(hipMemsetAsync(dstat.p, 0, 8, nullptr));           // status[1] (steps) is only written on completed paths
      RQ_TRY(polar_ns_launch(dR.as<float>(), dG.as<float>(), d, dstat.as<int>(), dns.p, di.num_cu, nullptr));
      int st2[2] = {1, 0};
      RQ_HIP(hipMemcpy(st2, dstat.p, 8, hipMemcpyDeviceToHost));
      pstat = st2[0];
      g_train_prof[TP_NS_STEPS] += st2[1];
    }
    if (pstat != 0 && dev_polar) {
      RQ_TRY(polar_factor_launch(dR.as<float>(), dG.as<float>(), dVw.as<double>(), dev_warm ? 1 : 0, d, dstat.as<int>(),
                                 dscr.as<double>(), nullptr));
      int st2[2] = {1, 0};
      RQ_HIP(hipMemcpy(st2, dstat.p, 8, hipMemcpyDeviceToHost));
      pstat = st2[0];
      g_train_prof[TP_SWEEPS] += st2[1];
      dev_warm = pstat == 0;
    }
    if (pstat != 0) {
      RQ_HIP(hipMemcpy(Gf.data(), dG.p, (size_t)d * d * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < Gf.size(); ++i) G[i] = Gf[i];
      polar_factor(G.data(), P.data(), d, &Vwarm);       // P = U V' = Julia's R
      for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) Rh[(size_t)i * d + k] = (float)P[(size_t)k * d + i];
      RQ_HIP(hipMemcpy(dR.p, Rh.data(), (size_t)d * d * 4, hipMemcpyHostToDevice));
      g_train_prof[TP_HOST_POLAR] += 1;
    }
    prof.stop(TP_SVD);
    RQ_HIP(hipStreamWaitEvent(nullptr, side.done, 0));      // the objective has read RX, C (and CB)
    RQ_PH(TP_ROTATE, RQ_TRY(rotate_launch(dRX.as<float>(), dR.as<float>(), dX.as<float>(), d, n, di.num_cu, nullptr)));
    RQ_PH(TP_CENTERS, RQ_TRY(update_centers_launch(dC.as<float>(), dcnt.as<unsigned int>(), dRX.as<float>(), dcodes.as<uint8_t>(), n, d,
                                 m, h, di.num_cu, nullptr)));
    RQ_PH(TP_ENCODE, RQ_TRY(encode_launch(dcodes.as<uint8_t>(), dRX.as<float>(), dC.as<float>(), n, d, m, h, di.num_cu, nullptr)));
    if (!fused_cb) RQ_PH(TP_RECONSTRUCT, RQ_TRY(reconstruct_launch(dCB.as<float>(), dcodes.as<uint8_t>(), dC.as<float>(), n, d, m, h, nullptr)));
  }
  prof.loop_end(niter + 1);
  {
    std::vector<double> accs((size_t)niter + 1);
    RQ_HIP(hipStreamSynchronize(side.s));
    RQ_HIP(hipMemcpy(accs.data(), dobj.p, accs.size() * 8, hipMemcpyDeviceToHost));
    if (obj) for (int it = 0; it <= niter; ++it) obj[it] = (float)(accs[it] / (double)n);
  }
  RQ_TRY(widen_codes_launch(d16.as<int16_t>(), dcodes.as<uint8_t>(), n * m, nullptr));
  RQ_HIP(hipDeviceSynchronize());
  prof.start();
  RQ_HIP(hipMemcpy(C, dC.p, (size_t)h * d * 4, hipMemcpyDeviceToHost));
  RQ_HIP(hipMemcpy(B1, d16.p, (size_t)n * m * 2, hipMemcpyDeviceToHost));
  RQ_HIP(hipMemcpy(R, dR.p, sizeof(float) * d * d, hipMemcpyDeviceToHost));     // (the device copy is the current R on both paths)
  prof.stop(TP_D2H);
  return RQ_OK;
}

// Rimg[i * d + k] = R[k][i], R = U V' of the d x d matrix G (row-major) -- the rotation update of src/OPQ.jl:112-113 on the
// device.  method 0: scaled Newton-Schulz (d <= 1024), 1: one-sided Jacobi SVD (even d <= 128).  status[0] = 0 when Rimg was
// written, 1 when the method gave up (rank-deficient G: rq_train_opq then falls back); status[1] = steps / sweeps.  Synchronous.
int rq_dev_polar_factor(float *Rimg, const float *G, int d, int method, int *status_host) {
  if (!Rimg || !G || !status_host) return fail(RQ_EINVAL, "rq_dev_polar_factor: null pointer");
  DeviceInfo di;
  RQ_TRY(device_info(&di));
  DevMem dstat, dscr, dVw;
  RQ_TRY(dstat.alloc(8));
  RQ_HIP(hipMemset(dstat.p, 0, 8));
  if (method == 0) {
    RQ_TRY(dscr.alloc(polar_ns_scratch_bytes(d, di.num_cu)));
    RQ_TRY(polar_ns_launch(Rimg, G, d, dstat.as<int>(), dscr.p, di.num_cu, nullptr));
  } else if (method == 1) {
    if (d < 2 || d > 128 || (d & 1)) return fail(RQ_EUNSUPPORTED, "rq_dev_polar_factor: Jacobi needs an even d <= 128");
    RQ_TRY(dscr.alloc(((size_t)d * d + d) * 8));
    RQ_TRY(dVw.alloc((size_t)d * d * 8));
    RQ_TRY(polar_factor_launch(Rimg, G, dVw.as<double>(), 0, d, dstat.as<int>(), dscr.as<double>(), nullptr));
  } else {
    return fail(RQ_EINVAL, "rq_dev_polar_factor: method %d", method);
  }
  RQ_HIP(hipMemcpy(status_host, dstat.p, 8, hipMemcpyDeviceToHost));
  return RQ_OK;
}

// Phase clock of the calling thread's last rq_train_pq / rq_train_opq call, milliseconds:
//   [0] X upload  [1] initialisation (seeding / first rotation + encode)  [2] qerror  [3] gram X'CB  [4] host SVD incl. its
//   two small copies  [5] rotation  [6] update_centers  [7] encode  [8] reconstruct  [9] convergence check  [10] results D2H
//   [11] wall time of the iteration loop  [12] iterations run  [13] Jacobi sweeps  [14] Newton-Schulz steps  [15] polar factors
//   that fell back to the host.  [2]-[9] are filled with tuning TRAIN_PROFILE = 1 only.
int rq_train_profile(double *out, int cap) {
  if (!out || cap < 1) return fail(RQ_EINVAL, "rq_train_profile: bad arguments");
  for (int i = 0; i < cap && i < TP_SLOTS; ++i) out[i] = g_train_prof[i];
  return RQ_OK;
}

}  // extern "C"
