// rq_train.hip -- device pieces of the PQ / OPQ training loops.
//
// train_opq (src/OPQ.jl:49-139) iterates   obj -> SVD(X CB') -> R'X -> update_centers! ->
// pairwise + update_assignments! -> CB;  train_pq (src/PQ.jl:68-99) is Lloyd's k-means per subspace.
// The assignment step IS the encode kernel (rq_encode.hip) and R'X the rotation kernel; this file
// adds the three reductions around them and the reconstruction:
//   update_centers   C_i[k] = mean of the sub-vectors assigned to k     (Clustering.update_centers!,
//                    call sites src/OPQ.jl:121)                          segment sum, LDS accumulators
//   reconstruct      CB[j][off_i + s] = C_i[b_ji][s]                     (src/OPQ.jl:101,128)
//   qerror           sum_j |RX_j - CB_j|^2                               (src/OPQ.jl:108, src/qerrors.jl:77-90;
//                    same value as |R CB - X|^2 because R is orthonormal)
//   gram             G = X' CB  (d x d, reduced over the n rows)         (src/OPQ.jl:112, input of the host SVD)
//                    f32 MFMA, one 32x32 output tile pair per wavefront, split over row slices
// Summation orders differ from the sequential CPU loops of the reference (which are themselves not
// pinned by any reference test): parity for these is a float tolerance, stated in tests/.
#include "rq_internal.h"

namespace rq {

using f32x16 = float __attribute__((ext_vector_type(16)));

struct TrainParams {
  const float *X;        // [n][d]  (already rotated for OPQ)
  const uint8_t *codes;  // [n][m]
  float *C;              // concat of [h][sub_i]
  float *partial;        // [grid][h*d] sums  then  [grid][m*h] counts (as float)
  float *CB;             // [n][d]
  double *acc;           // scalar accumulator
  unsigned int *counts;  // [m][h] out
  int64_t n;
  int d, m, h;
  int off[33];
};

// ---- update_centers, pass 1: per-workgroup partial sums in LDS, DETERMINISTIC ----------------------------
// Float addition is not associative, so bit-reproducible centres need a fixed summation order.  Every
// accumulator (code, dimension) has exactly ONE owner thread and that thread adds the rows of the workgroup's
// slice in ascending row order: thread t = (g, lane) with g = t / 128 owns the codes [32 g, 32 g + 32) of
// dimension dc0 + lane -- all 1024 threads walk the slice, each accumulates the rows whose code falls into
// its range (1/8 of them on average), no atomics on floats.  The slices are combined in fixed order by pass 2.
// Dimensions are processed in chunks of <= 128 (p.dc0 .. p.dc0 + p.dcw), so h * 128 * 4 B of LDS serve any d.
// LDS: sums [h][dcw] + counts [m][h] (integer atomics: order-free; written out by the chunk that holds the
// sub-quantizer's first dimension).
__global__ __launch_bounds__(1024) void centers_partial_kernel(TrainParams p, int dc0, int dcw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *sums = reinterpret_cast<float *>(smem);                                 // h * dcw
  unsigned int *cnts = reinterpret_cast<unsigned int *>(sums + (size_t)p.h * dcw);   // m * h
  const int tid = threadIdx.x;
  const int hw = p.h * dcw, mh = p.m * p.h;
  for (int i = tid; i < hw + mh; i += 1024) sums[i] = 0.0f;      // 0.0f and 0u share the bit pattern
  __syncthreads();
  const int64_t rows_per = (p.n + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = min(p.n, r0 + rows_per);
  const int ldim = tid & 127, g = tid >> 7, dim = dc0 + ldim;
  if (ldim < dcw) {
    int q = 0;
    while (q + 1 < p.m && dim >= p.off[q + 1]) ++q;
    const bool counts_here = (dim == p.off[q]);       // one dimension per sub-quantizer also counts the rows
    const uint8_t *cq = p.codes + q;
    const float *xd = p.X + dim;
    constexpr int U = 8;
    int64_t r = r0;
    for (; r + U <= r1; r += U) {
      int code[U];
      float x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) code[u] = cq[(r + u) * p.m];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ((code[u] >> 5) == g) ? xd[(r + u) * p.d] : 0.0f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((code[u] >> 5) == g) {
          sums[code[u] * dcw + ldim] += x[u];
          if (counts_here) atomicAdd(&cnts[q * p.h + code[u]], 1u);
        }
      }
    }
    for (; r < r1; ++r) {
      const int code = cq[r * p.m];
      if ((code >> 5) == g) {
        sums[code * dcw + ldim] += xd[r * p.d];
        if (counts_here) atomicAdd(&cnts[q * p.h + code], 1u);
      }
    }
  }
  __syncthreads();
  // partial layout per workgroup: [h][d] sums, then [m][h] counts (u32 bit patterns)
  float *out = p.partial + (size_t)blockIdx.x * ((size_t)p.h * p.d + (size_t)p.m * p.h);
  for (int i = tid; i < hw; i += 1024) out[(size_t)(i / dcw) * p.d + dc0 + (i % dcw)] = sums[i];
  // counts of the sub-quantizers whose first dimension lies in this chunk
  unsigned int *oc = reinterpret_cast<unsigned int *>(out + (size_t)p.h * p.d);
  for (int i = tid; i < mh; i += 1024) {
    const int q0 = p.off[i / p.h];
    if (q0 >= dc0 && q0 < dc0 + dcw) oc[i] = cnts[i];
  }
}

// pass 2: fixed-order sum over the workgroup partials, mean, write C (empty clusters keep their value)
__global__ void centers_finish_kernel(TrainParams p, int nparts) {
  const int hd = p.h * p.d, mh = p.m * p.h;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hd) return;
  const int code = i / p.d, dim = i % p.d;
  int q = 0;
  while (q + 1 < p.m && dim >= p.off[q + 1]) ++q;
  float s = 0.0f;
  unsigned int c = 0;
  for (int w = 0; w < nparts; ++w) {
    const float *part = p.partial + (size_t)w * (hd + mh);
    s += part[i];
    c += reinterpret_cast<const unsigned int *>(part + hd)[q * p.h + code];
  }
  const int sub = p.off[q + 1] - p.off[q];
  if (c > 0) p.C[(size_t)p.h * p.off[q] + (size_t)code * sub + (dim - p.off[q])] = s * (1.0f / (float)c);
  if (dim == p.off[q]) p.counts[q * p.h + code] = c;
}

// ---- reconstruction ----------------------------------------------------------------------------------
__global__ void reconstruct_kernel(TrainParams p) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.n * p.d) return;
  const int64_t r = e / p.d;
  const int dim = (int)(e - r * p.d);
  int q = 0;
  while (q + 1 < p.m && dim >= p.off[q + 1]) ++q;
  const int sub = p.off[q + 1] - p.off[q];
  const int code = p.codes[r * p.m + q];
  p.CB[e] = p.C[(size_t)p.h * p.off[q] + (size_t)code * sub + (dim - p.off[q])];
}

// ---- quantisation error: sum (X - CB)^2 in double, fixed reduction tree (bit-reproducible) ---------------------
__global__ __launch_bounds__(256) void qerror_kernel(TrainParams p, double *partial) {
  __shared__ double red[256];
  const int64_t total = p.n * p.d;
  double s = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const double df = (double)p.X[e] - (p.CB ? (double)p.CB[e] : 0.0);   // CB == NULL: sum of squares of X
    s += df * df;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void qerror_finish_kernel(double *acc, const double *partial, int nparts) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) *acc = red[0];
}

// ---- G = X' CB  (d x d), reduced over rows ---------------------------------------------------------------
// Output tile (a, b) = 32 x 32 block G[32a.., 32b..].  A workgroup owns a row slice; its wavefronts split
// the (d/32)^2 output tiles; per 2 rows one 32x32x2 MFMA per tile: A[i][k] = X[row0+k][32a+i],
// B[k][j] = CB[row0+k][32b+j].  Partial tiles go to `partial`, a second kernel sums the slices.
struct GramParams {
  const float *X, *CB;
  float *partial;   // [grid][d*d]
  float *G;         // [d][d]
  int64_t n;
  int d, NT;
};

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void gram_partial_kernel(GramParams p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int d = p.d, NT = p.NT, ntile = NT * NT;
  const int64_t rows_per = ((p.n + gridDim.x - 1) / gridDim.x + 1) & ~(int64_t)1;  // even
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = min(p.n, r0 + rows_per);
  float *out = p.partial + (size_t)blockIdx.x * d * d;
  for (int t = wave; t < ntile; t += NWAVES) {
    const int a = t / NT, b = t % NT;
    const int ia = 32 * a + j, ib = 32 * b + j;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // 16 row pairs per trip: the 32 loads are issued together, then the 16 chained MFMAs
    const bool ina = ia < d, inb = ib < d;
    const float *xa = p.X + (ina ? ia : 0), *cb = p.CB + (inb ? ib : 0);
    int64_t r = r0;
    for (; r + 32 <= r1; r += 32) {
      float av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t row = r + 2 * u + hi;
        av[u] = xa[row * d];
        bv[u] = cb[row * d];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ina ? av[u] : 0.0f, inb ? bv[u] : 0.0f, acc, 0, 0, 0);
    }
    for (; r < r1; r += 2) {
      const int64_t row = r + hi;
      float av = 0.0f, bv = 0.0f;
      if (row < r1) {
        if (ina) av = xa[row * d];
        if (inb) bv = cb[row * d];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * hi, jj = 32 * b + j;
      if (i < d && jj < d) out[(size_t)i * d + jj] = acc[r];
    }
  }
}

__global__ void gram_finish_kernel(GramParams p, int nparts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.d * p.d) return;
  float s = 0.0f;
  for (int w = 0; w < nparts; ++w) s += p.partial[(size_t)w * p.d * p.d + i];
  p.G[i] = s;
}

// ---- kmeans++ seeding (D^2 sampling) ------------------------------------------------------------------------
// Clustering.jl's `init=:kmpp` as used by train_pq / train_rvq (src/PQ.jl:86, src/RVQ.jl:104): the first seed is a
// uniformly drawn point; every further seed is drawn with probability proportional to mincost[j] = the squared
// distance of point j to its nearest seed so far (the chosen point's own cost is 0, so seeds are distinct while
// any cost is left).  All m sub-spaces advance together (blockIdx.y); the h uniform numbers per sub-space come from
// the library's seeded stream, uploaded once -- no host round trip inside the h steps.  Sums are taken in double
// with fixed trees, so the seeds are bit-reproducible.
//   kmpp_update : mincost <- min(mincost, |x - x_seed|^2) over the sub-space, per-block cost sums -> partial
//   kmpp_select : (one workgroup per sub-space) block with the threshold u * total, then the row inside it;
//                 writes seeds[i][step] and the seed's sub-vector into C_i[step]
struct KmppParams {
  const float *X;       // [n][d]
  float *C;             // concat of [h][sub_i]
  float *mincost;       // [m][n]
  double *partial;      // [m][nblk]
  long long *seeds;     // [m][h]
  const double *u;      // [m][h] uniforms in [0, 1)
  int64_t n, rows_per_blk;
  int d, m, h, nblk;
  int off[33];
};

constexpr int KMPP_THREADS = 1024;

__device__ __forceinline__ double block_sum_1024(double v, double *red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int w = KMPP_THREADS / 2; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(KMPP_THREADS) void kmpp_update_kernel(KmppParams p, int step) {
  __shared__ double red[KMPP_THREADS];
  const int i = blockIdx.y, o = p.off[i], sub = p.off[i + 1] - o;
  const long long seed = p.seeds[(size_t)i * p.h + step - 1];       // the seed chosen in the previous step
  const float *xs = p.X + (size_t)seed * p.d + o;
  float *mc = p.mincost + (size_t)i * p.n;
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_blk, r1 = min(p.n, r0 + p.rows_per_blk);
  double acc = 0.0;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += KMPP_THREADS) {
    const float *x = p.X + (size_t)r * p.d + o;
    float dist = 0.0f;
    for (int s = 0; s < sub; ++s) {
      const float df = x[s] - xs[s];
      dist = __builtin_fmaf(df, df, dist);
    }
    float c = (step == 1) ? dist : fminf(mc[r], dist);
    if (r == seed) c = 0.0f;
    mc[r] = c;
    acc += (double)c;
  }
  const double tot = block_sum_1024(acc, red);
  if (threadIdx.x == 0) p.partial[(size_t)i * p.nblk + blockIdx.x] = tot;
}

// first seed: row floor(u * n)
__global__ void kmpp_first_kernel(KmppParams p) {
  const int i = blockIdx.x, o = p.off[i], sub = p.off[i + 1] - o;
  long long row = (long long)(p.u[(size_t)i * p.h] * (double)p.n);
  row = min(max(row, 0ll), (long long)p.n - 1);
  if (threadIdx.x == 0) p.seeds[(size_t)i * p.h] = row;
  for (int s = threadIdx.x; s < sub; s += blockDim.x)
    p.C[(size_t)p.h * o + s] = p.X[(size_t)row * p.d + o + s];
}

__global__ __launch_bounds__(KMPP_THREADS) void kmpp_select_kernel(KmppParams p, int step) {
  __shared__ double red[KMPP_THREADS];
  __shared__ double scan[KMPP_THREADS];
  __shared__ long long pick[2];
  __shared__ double resid;
  const int tid = threadIdx.x, i = blockIdx.x, o = p.off[i], sub = p.off[i + 1] - o;
  const double *part = p.partial + (size_t)i * p.nblk;
  const float *mc = p.mincost + (size_t)i * p.n;
  // inclusive scan of the block sums (nblk <= 1024: one per thread), Hillis-Steele in double
  scan[tid] = tid < p.nblk ? part[tid] : 0.0;
  __syncthreads();
  for (int w = 1; w < KMPP_THREADS; w <<= 1) {
    const double v = tid >= w ? scan[tid - w] : 0.0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  const double total = scan[KMPP_THREADS - 1];
  const double u = p.u[(size_t)i * p.h + step];
  if (tid == 0) { pick[0] = -1; pick[1] = -1; }
  __syncthreads();
  long long row = -1;
  if (!(total > 0.0)) {
    // every point coincides with a seed: any point is as good as another (Clustering's wsample has no answer here)
    row = min((long long)(u * (double)p.n), (long long)p.n - 1);
  } else {
    const double thr = u * total;
    // first block whose inclusive sum exceeds the threshold (the last non-empty one if rounding leaves none)
    const double before = tid ? scan[tid - 1] : 0.0;
    if (tid < p.nblk && scan[tid] > thr && !(before > thr)) { pick[0] = tid; resid = thr - before; }
    __syncthreads();
    if (pick[0] < 0) {
      if (tid == 0) {
        int b = p.nblk - 1;
        while (b > 0 && !(part[b] > 0.0)) --b;
        pick[0] = b;
        resid = part[b];       // past the end: the last row with a cost in that block
      }
      __syncthreads();
    }
    const int64_t r0 = pick[0] * p.rows_per_blk, r1 = min(p.n, r0 + p.rows_per_blk);
    // rows of the block in 1024 contiguous chunks: chunk sums, scan over the chunks, then a walk inside one chunk
    const int64_t chunk = (r1 - r0 + KMPP_THREADS - 1) / KMPP_THREADS;
    const int64_t c0 = min(r1, r0 + (int64_t)tid * chunk), c1 = min(r1, c0 + chunk);
    double cs = 0.0;
    for (int64_t r = c0; r < c1; ++r) cs += (double)mc[r];
    red[tid] = cs;
    scan[tid] = cs;
    __syncthreads();
    for (int w = 1; w < KMPP_THREADS; w <<= 1) {
      const double v = tid >= w ? scan[tid - w] : 0.0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const double rthr = resid;
    const double cbefore = tid ? scan[tid - 1] : 0.0;
    if (scan[tid] > rthr && !(cbefore > rthr) && c1 > c0) {
      double run = cbefore;
      long long rr = -1;
      for (int64_t r = c0; r < c1; ++r) {
        run += (double)mc[r];
        if (run > rthr && mc[r] > 0.0f) { rr = r; break; }
      }
      if (rr < 0)
        for (int64_t r = c1 - 1; r >= c0; --r) if (mc[r] > 0.0f) { rr = r; break; }
      pick[1] = rr;
    }
    __syncthreads();
    if (pick[1] < 0 && tid == 0) {      // threshold beyond the block's re-summed total: last row with a cost
      for (int64_t r = r1 - 1; r >= r0; --r) if (mc[r] > 0.0f) { pick[1] = r; break; }
      if (pick[1] < 0) pick[1] = r0;
    }
    __syncthreads();
    row = pick[1];
  }
  if (tid == 0) p.seeds[(size_t)i * p.h + step] = row;
  for (int s = tid; s < sub; s += KMPP_THREADS)
    p.C[(size_t)p.h * o + (size_t)step * sub + s] = p.X[(size_t)row * p.d + o + s];
}

int kmpp_init_launch(float *C, long long *seeds, float *mincost, double *partial, const double *u, const float *X,
                     int64_t n, int d, int m, int h, hipStream_t stream) {
  if (n < 1 || m < 1 || m > 32 || d < m || h < 1) return fail(RQ_EINVAL, "kmeans++: n=%lld d=%d m=%d h=%d", (long long)n, d, m, h);
  KmppParams p{};
  p.X = X; p.C = C; p.mincost = mincost; p.partial = partial; p.seeds = seeds; p.u = u;
  p.n = n; p.d = d; p.m = m; p.h = h;
  p.nblk = (int)std::min<int64_t>(KMPP_THREADS, (n + KMPP_THREADS - 1) / KMPP_THREADS);
  p.rows_per_blk = (n + p.nblk - 1) / p.nblk;
  p.nblk = (int)((n + p.rows_per_blk - 1) / p.rows_per_blk);
  {
    const int per = d / m, extra = d % m;
    int pos = 0;
    for (int i = 0; i < m; ++i) { p.off[i] = pos; pos += per + (i < extra ? 1 : 0); }
    p.off[m] = pos;
  }
  hipLaunchKernelGGL(kmpp_first_kernel, dim3(m), dim3(64), 0, stream, p);
  RQ_HIP(hipGetLastError());
  for (int step = 1; step < h; ++step) {
    hipLaunchKernelGGL(kmpp_update_kernel, dim3(p.nblk, m), dim3(KMPP_THREADS), 0, stream, p, step);
    hipLaunchKernelGGL(kmpp_select_kernel, dim3(m), dim3(KMPP_THREADS), 0, stream, p, step);
  }
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}
int kmpp_partial_count(int64_t n) { return KMPP_THREADS; }

// ------------------------------------------------------------------------------------------------------
static void fill_offsets(int *off, int d, int m) {
  const int per = d / m, extra = d % m;
  int pos = 0;
  for (int i = 0; i < m; ++i) { off[i] = pos; pos += per + (i < extra ? 1 : 0); }
  off[m] = pos;
}

int update_centers_launch(float *C, unsigned int *counts, const float *X, const uint8_t *codes, int64_t n, int d,
                          int m, int h, int num_cu, hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  if (m < 1 || m > 32 || d < m || h < 1 || h > 256)
    return fail(RQ_EUNSUPPORTED, "update_centers covers m <= 32, h <= 256 (got m=%d d=%d h=%d)", m, d, h);
  TrainParams p{};
  p.X = X; p.codes = codes; p.C = C; p.counts = counts; p.n = n; p.d = d; p.m = m; p.h = h;
  fill_offsets(p.off, d, m);
  const int grid = (int)std::min<int64_t>(num_cu, (n + 1023) / 1024);
  void *part = nullptr;
  RQ_TRY(workspace(WS_TMP, (size_t)grid * ((size_t)h * d + (size_t)m * h) * sizeof(float), &part, stream));
  p.partial = (float *)part;
  // dimension chunks of <= 128: h * 128 * 4 B of sums (+ m * h counters in the first chunk) always fit the LDS
  for (int dc0 = 0; dc0 < d; dc0 += 128) {
    const int dcw = std::min(128, d - dc0);
    const size_t lds = ((size_t)h * dcw + (size_t)m * h) * sizeof(float);
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(centers_partial_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(centers_partial_kernel, dim3(grid), dim3(1024), lds, stream, p, dc0, dcw);
    RQ_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(centers_finish_kernel, dim3((h * d + 255) / 256), dim3(256), 0, stream, p, grid);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int reconstruct_launch(float *CB, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h,
                       hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  if (m < 1 || m > 32 || d < m) return fail(RQ_EINVAL, "reconstruct: m=%d d=%d", m, d);
  TrainParams p{};
  p.codes = codes; p.C = const_cast<float *>(C); p.CB = CB; p.n = n; p.d = d; p.m = m; p.h = h;
  fill_offsets(p.off, d, m);
  const int64_t total = n * d;
  hipLaunchKernelGGL(reconstruct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int qerror_launch(double *acc_dev, const float *X, const float *CB, int64_t n, int d, int num_cu,
                  hipStream_t stream) {
  if (n <= 0) {
    RQ_HIP(hipMemsetAsync(acc_dev, 0, sizeof(double), stream));
    return RQ_OK;
  }
  TrainParams p{};
  p.X = X; p.CB = const_cast<float *>(CB); p.acc = acc_dev; p.n = n; p.d = d;
  const int grid = num_cu * 8;
  void *part = nullptr;
  RQ_TRY(workspace(WS_MERGE, (size_t)grid * sizeof(double), &part, stream));
  hipLaunchKernelGGL(qerror_kernel, dim3(grid), dim3(256), 0, stream, p, (double *)part);
  RQ_HIP(hipGetLastError());
  hipLaunchKernelGGL(qerror_finish_kernel, dim3(1), dim3(256), 0, stream, acc_dev, (const double *)part, grid);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int gram_launch(float *G, const float *X, const float *CB, int64_t n, int d, int num_cu, hipStream_t stream) {
  if (d < 1 || d > 1024) return fail(RQ_EUNSUPPORTED, "gram: d=%d", d);
  GramParams p;
  p.X = X; p.CB = CB; p.G = G; p.n = n; p.d = d; p.NT = (d + 31) / 32;
  constexpr int NW = 16;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(num_cu, (n + 255) / 256));
  void *part = nullptr;
  RQ_TRY(workspace(WS_TMP, (size_t)grid * d * d * sizeof(float), &part, stream));
  p.partial = (float *)part;
  hipLaunchKernelGGL(gram_partial_kernel<NW>, dim3(grid), dim3(NW * 64), 0, stream, p);
  RQ_HIP(hipGetLastError());
  hipLaunchKernelGGL(gram_finish_kernel, dim3((d * d + 255) / 256), dim3(256), 0, stream, p, grid);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// ------------------------------------------------------------------------------------------
// Polar factor  R = U V'  of the d x d matrix G = X CB'  (src/OPQ.jl:112-113: U, S, VV = svd(X * CB'); R = U * VV') on the
// device: one-sided Jacobi SVD in double (d <= 128, even).  Round 3 ran the same algorithm on ONE host core: 6.6 ms of a
// 10.1 ms OPQ iteration at SIFT1M shape (bench.py --workload train_opq).  Four kernels:
//   (1) polar_prep    A0 = G V0 (V0 = the previous call's right singular vectors: successive G of an OPQ run differ little;
//                     identity on the first call), column-major doubles, grid-wide
//   (2) polar_jacobi  ONE workgroup, the matrix in LDS (columns padded by 8 doubles: 139 KiB at d = 128): sweeps of d - 1
//                     round-robin ("circle") rounds, the d/2 disjoint column pairs of a round rotating in parallel, 8 lanes per
//                     pair, until max |a_p . a_q| / (|a_p| |a_q|) < 1e-9 (G is an f32 accumulation, R is stored as f32);
//                     s_p = |a_p|, U = A / s
//   (3) polar_vt      V' = S^-1 U' G  (A = G V  =>  U' G = S V'), kept for the next warm start -- V never rotates along
//   (4) polar_r       R = U V', written in the memory image the rotation kernel reads
// status[0]: 0 ok, 1 = a vanishing singular value (the polar factor is not unique there: the caller falls back to the host
// path, which completes the basis); status[1]: sweeps.  Measured: 1.7-1.9 ms per call (5.7 sweeps on average over an OPQ run,
// each 127 latency-bound rounds), of which the three products are 30 us.
// ------------------------------------------------------------------------------------------
// (1) A0 = G V0, column-major doubles: A0[p * d + i] = sum_k G[i][k] V0[k][p]   (Vw[p * d + k] = V0[k][p]); warm == 0: A0 = G
__global__ void polar_prep_kernel(double *__restrict__ A0, const float *__restrict__ G, const double *__restrict__ Vw, int warm, int d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d * d) return;
  const int p = e / d, i = e - p * d;
  double acc;
  if (warm) {
    acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < d; ++k) acc = __builtin_fma((double)G[(size_t)i * d + k], Vw[(size_t)p * d + k], acc);
  } else {
    acc = (double)G[(size_t)i * d + p];
  }
  A0[e] = acc;
}

// fast double reciprocal / reciprocal square root: the hardware estimates (v_rcp_f64 / v_rsq_f64, ~2^-27) + two Newton steps.
// The IEEE division and sqrt sequences (~50 instructions each, executed by every lane of every wavefront) were most of a
// Jacobi round: 2.5 ms per polar factor.  A rotation only has to be orthogonal to ~1e-15, which c = rsqrt(1 + t^2),
// s = c t is by construction.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}
__device__ __forceinline__ double group8_sum(double v) {
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) v += __shfl_xor(v, off, 8);
  return v;
}

// (2) the sweeps, one workgroup of 512 threads (8 lanes per column pair, 16 elements per lane at d = 128): A (LDS) <- A0;
//     on exit U = A / s (column-major, back into A0's storage) and s
constexpr int POLAR_THREADS = 512, POLAR_LANES = 8, POLAR_EPL = 16;
__global__ __launch_bounds__(POLAR_THREADS) void polar_jacobi_kernel(double *__restrict__ AU, double *__restrict__ sv, int d, int *status) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // column p at A + p * ld, ld = d + 8: a lane group reads 8 consecutive doubles of its column, the four pairs of a 32-lane
  // LDS pass hold CONSECUTIVE columns (circle ordering below), so their bank groups 8 p mod 32 are distinct.  (With ld = d
  // every column starts on bank 0: 8-way conflicts, 3.3 us per round instead of ~0.5.)
  double *A = reinterpret_cast<double *>(smem_raw);
  const int ld = d + 8;
  __shared__ unsigned long long s_off;
  __shared__ int s_bad;
  __shared__ double s_sv[128];
  const int tid = threadIdx.x, slot = tid / POLAR_LANES, sub = tid % POLAR_LANES;
  const int npairs = d / 2;
  for (int e = tid; e < d * d; e += POLAR_THREADS) A[(e / d) * ld + (e % d)] = AU[e];
  if (tid == 0) s_bad = 0;
  __syncthreads();
  int nsweeps = 0;
  for (int sweep = 0; sweep < 40; ++sweep) {
    ++nsweeps;
    if (tid == 0) s_off = 0ull;
    __syncthreads();
    for (int round = 0; round < d - 1; ++round) {
      if (slot < npairs) {
        // circle method: position d-1 is fixed, the others rotate (round + slot and round - slot + d - 1 are below 2 (d - 1))
        int p, q;
        if (slot == 0) { p = d - 1; q = round; }
        else {
          p = round + slot; if (p >= d - 1) p -= d - 1;
          q = round - slot + (d - 1); if (q >= d - 1) q -= d - 1;
        }
        double *ap = A + (size_t)p * ld, *aq = A + (size_t)q * ld;
        double x[POLAR_EPL], y[POLAR_EPL];
        double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
        for (int i = 0; i < POLAR_EPL; ++i) {
          const int e = sub + POLAR_LANES * i;
          x[i] = e < d ? ap[e] : 0.0;
          y[i] = e < d ? aq[e] : 0.0;
          al = __builtin_fma(x[i], x[i], al);
          be = __builtin_fma(y[i], y[i], be);
          ga = __builtin_fma(x[i], y[i], ga);
        }
        al = group8_sum(al); be = group8_sum(be); ga = group8_sum(ga);
        const double ab = al * be, g2 = ga * ga;
        if (ab > 0.0) {
          // lim^2 = (a_p . a_q)^2 / (|a_p|^2 |a_q|^2); rotate while lim >= 1e-15
          if (sub == 0) atomicMax(&s_off, (unsigned long long)__double_as_longlong(g2 * fast_rcp(ab)));   // >= 0: bit order = value order
          if (g2 >= 1e-30 * ab) {
            const double zeta = (be - al) * 0.5 * fast_rcp(ga);
            const double az = fabs(zeta);
            // t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2));  sqrt(w) = w rsqrt(w)
            const double w = __builtin_fma(zeta, zeta, 1.0);
            double t = fast_rcp(az + w * fast_rsqrt(w));
            if (!(az < 1e150)) t = 0.5 * fast_rcp(az);          // zeta^2 overflows: t -> 1 / (2 zeta)
            t = zeta >= 0.0 ? t : -t;
            const double c = fast_rsqrt(__builtin_fma(t, t, 1.0)), sn = c * t;
#pragma unroll
            for (int i = 0; i < POLAR_EPL; ++i) {
              const int e = sub + POLAR_LANES * i;
              if (e < d) {
                ap[e] = c * x[i] - sn * y[i];
                aq[e] = sn * x[i] + c * y[i];
              }
            }
          }
        }
      }
      __syncthreads();
    }
    const double off2 = __longlong_as_double((long long)s_off);
    __syncthreads();
    // max |cos angle| between two columns below 1e-9: G itself is an f32 accumulation (relative error ~1e-6) and R is stored
    // as f32 (2^-24 = 6e-8), so the 1e-14 of the host version bought nothing -- and every sweep is d - 1 latency-bound rounds
    if (off2 < 1e-18) break;
  }
  for (int p = slot; p < d; p += POLAR_THREADS / POLAR_LANES) {
    double nn = 0.0;
    for (int i = 0; i < POLAR_EPL; ++i) { const int e = sub + POLAR_LANES * i; if (e < d) nn = __builtin_fma(A[(size_t)p * ld + e], A[(size_t)p * ld + e], nn); }
    nn = sqrt(group8_sum(nn));
    if (sub == 0) s_sv[p] = nn;
  }
  __syncthreads();
  {
    double mx = 0.0;
    for (int p = 0; p < d; ++p) mx = fmax(mx, s_sv[p]);
    for (int p = tid; p < d; p += POLAR_THREADS) if (!(s_sv[p] > mx * 1e-12) || !(mx > 0.0) || !(mx < 1e300)) s_bad = 1;
  }
  __syncthreads();
  if (tid == 0) { status[0] = s_bad; status[1] = nsweeps; }
  if (s_bad) return;
  for (int e = tid; e < d * d; e += POLAR_THREADS) AU[e] = A[(e / d) * ld + (e % d)] / s_sv[e / d];
  for (int p = tid; p < d; p += POLAR_THREADS) sv[p] = s_sv[p];
}

// (3) V'[p][b] = (U' G)[p][b] / s_p   (A = G V = U S  =>  U' G = S V');  kept for the next warm start: Vw[p * d + b] = V[b][p]
__global__ void polar_vt_kernel(double *__restrict__ Vw, const double *__restrict__ U, const float *__restrict__ G,
                                const double *__restrict__ sv, int d, const int *status) {
  if (*status != 0) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d * d) return;
  const int p = e / d, b = e - p * d;
  double acc = 0.0;
#pragma unroll 8
  for (int a = 0; a < d; ++a) acc = __builtin_fma(U[(size_t)p * d + a], (double)G[(size_t)a * d + b], acc);
  Vw[e] = acc / sv[p];
}

// (4) R = U V' ; memory image of Julia's R: Rimg[i * d + k] = R[k][i] = sum_p U[k][p] V[i][p]
__global__ void polar_r_kernel(float *__restrict__ Rimg, const double *__restrict__ U, const double *__restrict__ Vw, int d,
                               const int *status) {
  if (*status != 0) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= d * d) return;
  const int i = e / d, k = e - i * d;
  double acc = 0.0;
#pragma unroll 8
  for (int p = 0; p < d; ++p) acc = __builtin_fma(U[(size_t)p * d + k], Vw[(size_t)p * d + i], acc);
  Rimg[e] = (float)acc;
}

// G [d][d] f32 (row-major, G[a][b] = (X CB')[a, b]) -> Rimg [d][d] f32; Vw [d][d] doubles persists between calls (warm != 0:
// it holds the previous V); scratch (d*d + d) doubles; status one int (read it after the stream has drained: 0 = R written).
// d even, 2 <= d <= 128.
int polar_factor_launch(float *Rimg, const float *G, double *Vw, int warm, int d, int *status, double *scratch, hipStream_t stream) {
  if (d < 2 || d > 128 || (d & 1)) return fail(RQ_EUNSUPPORTED, "device polar factor: even d <= 128; got %d", d);
  const size_t lds = (size_t)d * (d + 8) * sizeof(double);
  double *AU = scratch, *sv = scratch + (size_t)d * d;
  const int nb = (d * d + 255) / 256;
  hipLaunchKernelGGL(polar_prep_kernel, dim3(nb), dim3(256), 0, stream, AU, G, (const double *)Vw, warm, d);
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(polar_jacobi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(polar_jacobi_kernel, dim3(1), dim3(POLAR_THREADS), lds, stream, AU, sv, d, status);
  hipLaunchKernelGGL(polar_vt_kernel, dim3(nb), dim3(256), 0, stream, Vw, (const double *)AU, G, (const double *)sv, d, (const int *)status);
  hipLaunchKernelGGL(polar_r_kernel, dim3(nb), dim3(256), 0, stream, Rimg, (const double *)AU, (const double *)Vw, d, (const int *)status);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// number of positions where two code arrays differ (train_pq's convergence test: "no assignment changed")
__global__ void codes_changed_kernel(unsigned long long *out, const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, size_t nbytes) {
  unsigned long long c = 0;
  const size_t nw = nbytes / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nw; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t x = reinterpret_cast<const uint64_t *>(a)[i] ^ reinterpret_cast<const uint64_t *>(b)[i];
    // bytes of x that are non-zero
    const uint64_t nz = ((x | ((x | 0x8080808080808080ull) - 0x0101010101010101ull)) & 0x8080808080808080ull);
    c += (unsigned long long)__popcll(nz);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t i = nw * 8; i < nbytes; ++i) c += a[i] != b[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

int codes_changed_launch(unsigned long long *out, const uint8_t *a, const uint8_t *b, size_t nbytes, hipStream_t stream) {
  RQ_HIP(hipMemsetAsync(out, 0, 8, stream));
  const uint32_t grid = (uint32_t)std::min<size_t>(1024, (nbytes / 8 + 255) / 256 + 1);
  hipLaunchKernelGGL(codes_changed_kernel, dim3(grid), dim3(256), 0, stream, out, a, b, nbytes);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

}  // namespace rq
