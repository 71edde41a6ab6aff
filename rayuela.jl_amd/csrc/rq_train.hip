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

// ---- update_centers, pass 1 on the MATRIX CORES (round 4, the default) ---------------------------------------------------
// The owner-thread kernel above takes 0.81 ms at SIFT1M shape (8 % of the HBM roof).  A second scatter design was measured
// and dropped: one wavefront per (slice, 64 dimensions), every row one `ds_add_f32` per lane in row order, 16-byte loads, 32
// KiB in flight per wavefront -- 0.78 ms; without its adds 0.34 ms, without its loads 0.78: LDS float atomics retire ~1 lane
// per 3.4 cycles and CU, the read-modify-write IS the bound of any scatter.  A segment sum is also a product with a one-hot matrix,
//      sums_q[code][s] = sum_rows [b(row, q) == code] x[row][off_q + s],
// and one-hot entries and products are EXACT in bf16 arithmetic if x is split into three bf16 pieces (8 + 8 + 8 mantissa
// bits: x = p1 + p2 + p3 exactly), so v_mfma_f32_16x16x32_bf16 adds exactly the f32 values the scatter would add -- in the
// matrix core's fixed order: deterministic, no atomics, no LDS.  Per (32 rows, sub-quantizer, 16-code tile): one A operand
// (16 codes x 32 rows of 0 / 2: a row keeps the 16-bit mask 1 << (code >> 4) if its code's low nibble is the lane's, and
// tile t's entry is that mask's bit t moved to bit 14 -- the bf16 number 2.0 -- by one shift and one AND per register: 8
// VALU instructions per tile; compare + select + pack took 24 and bound the kernel; the factor 2 is taken out exactly at
// the end) and four MFMAs (the three pieces of the 32 x 16 block of x, and a column of ones that counts the rows).  A wavefront owns one
// (sub-quantizer, 16-dimension block) for all codes: 16 + 16 accumulator tiles = 128 VGPRs; a workgroup = 8 such units over
// one row slice, blockIdx.y = further units.  Matrix time at SIFT1M shape: 64 MFMAs x 16 cycles per 32 rows and unit = 0.10 ms.
// Non-finite or > 3.3e38 inputs turn into NaN in EVERY centre of their sub-space (0 x inf), not only in their own.
typedef __bf16 cm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cm_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short cm_u16x2 __attribute__((ext_vector_type(2)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2t = float __attribute__((ext_vector_type(2)));

template <int NT>
__global__ __launch_bounds__(512) void centers_mfma_kernel(TrainParams p, int nunits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.y * 8 + wave;
  if (unit >= nunits) return;                 // (no barriers in this kernel)
  int q = 0, cb = 0;
  {
    int before = 0;
    for (q = 0; q < p.m; ++q) {
      const int nb = (p.off[q + 1] - p.off[q] + 15) >> 4;
      if (unit < before + nb) { cb = unit - before; break; }
      before += nb;
    }
  }
  const int sub = p.off[q + 1] - p.off[q], col0 = p.off[q] + 16 * cb, width = min(16, sub - 16 * cb);
  const int i = lane & 15, kg = lane >> 4;
  const int nslice = gridDim.x;
  const int64_t rows_per = (p.n + nslice - 1) / nslice;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = min(p.n, r0 + rows_per);
  f32x4 acc[NT], cnt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; cnt[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const bool colok = i < width;
  // Buffer loads: a slice-wide resource (base = the slice's first row, range = its bytes: rows past r1 read as 0), a 32-bit lane
  // offset (the lane's 8 rows of a 32-row step start at row 8 kg; its column is col0 + i -- an idle lane reads column col0 and
  // multiplies it by zero) and a wave-uniform row offset in an SGPR: no 64-bit address arithmetic per load (it was 280 of the
  // 408 VALU instructions of a step, and the kernel is bound by VALU + MFMA issue)
  const int64_t nrows = r1 > r0 ? r1 - r0 : 0;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.X + r0 * p.d), 0, (int)(nrows * p.d * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.codes + r0 * p.m), 0, (int)(nrows * p.m), 0x00020000);
  const int xoff = ((8 * kg) * p.d + col0 + (colok ? i : 0)) * 4;
  const int coff = (8 * kg) * p.m + q;
  const uint32_t ones = i == 0 ? 0x3F803F80u : 0u;
  const cm_bf16x8 Bc = __builtin_bit_cast(cm_bf16x8, make_uint4(ones, ones, ones, ones));
  // raw operands of ONE pair of 32-row steps; a step re-loads its half for the next pair as soon as it has turned the values
  // into its A mask and B pieces, so the loads fly during the two tile loops that follow (two buffers cost 32 VGPRs more
  // and spilled)
  uint32_t cr[2][8];
  float xr[2][8];
  auto load = [&](int s, int64_t r) {              // rows r + 32 s .. + 31
    const int L0 = (int)(r - r0) + 32 * s;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int L = L0 + u;                                         // wave-uniform
      cr[s][u] = __builtin_amdgcn_raw_buffer_load_b8(rC, coff, L * p.m, 0);
      xr[s][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, xoff, L * p.d * 4, 0));
    }
  };
  // TAIL: the 64 rows from r on may pass r1 (their loads return 0; their one-hot entries are masked here)
  auto step = [&](int s, int64_t r, bool more, auto tail) {     // more: pre-load this half for the pair at r + 64
    constexpr bool TAIL = decltype(tail)::value;
    // A side: mask of row u = 1 << (code >> 4) if the code's low nibble is this lane's (and the row exists), else 0
    uint32_t K[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      uint32_t k2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int u = 2 * pp + e;
        const uint32_t c = cr[s][u];
        bool hit = (c & 15u) == (uint32_t)i;
        if constexpr (TAIL) hit = hit && r + 32 * s + 8 * kg + u < r1;
        k2[e] = hit ? (1u << (c >> 4)) : 0u;
      }
      K[pp] = k2[0] | (k2[1] << 16);
    }
    // B side: the lane's 8 values of x in three bf16 pieces (exact: 8 + 8 + 8 mantissa bits)
    uint32_t P1[4], P2[4], P3[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      f32x2t v = {colok ? xr[s][2 * pp] : 0.0f, colok ? xr[s][2 * pp + 1] : 0.0f};
      const cm_bf16x2 h1 = __builtin_convertvector(v, cm_bf16x2);
      v = v - __builtin_convertvector(h1, f32x2t);
      const cm_bf16x2 h2 = __builtin_convertvector(v, cm_bf16x2);
      v = v - __builtin_convertvector(h2, f32x2t);
      const cm_bf16x2 h3 = __builtin_convertvector(v, cm_bf16x2);
      P1[pp] = __builtin_bit_cast(uint32_t, h1);
      P2[pp] = __builtin_bit_cast(uint32_t, h2);
      P3[pp] = __builtin_bit_cast(uint32_t, h3);
    }
    const cm_bf16x8 B1 = __builtin_bit_cast(cm_bf16x8, make_uint4(P1[0], P1[1], P1[2], P1[3]));
    const cm_bf16x8 B2 = __builtin_bit_cast(cm_bf16x8, make_uint4(P2[0], P2[1], P2[2], P2[3]));
    const cm_bf16x8 B3 = __builtin_bit_cast(cm_bf16x8, make_uint4(P3[0], P3[1], P3[2], P3[3]));
    if (more) load(s, r + 64);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      // bit t of each 16-bit mask -> bit 14 of its half = bf16 2.0.  (A 32-bit shift serves both halves: a left shift by
      // 14 - t <= 14 moves no bit of the low half up to bit 30.)  (Two tiles at a time with alternating MFMAs: no faster.)
      uint32_t a[4];
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) a[pp] = (t <= 14 ? K[pp] << (14 - t) : K[pp] >> 1) & 0x40004000u;
      const cm_bf16x8 A = __builtin_bit_cast(cm_bf16x8, make_uint4(a[0], a[1], a[2], a[3]));
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B1, acc[t], 0, 0, 0);
      cnt[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bc, cnt[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B2, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B3, acc[t], 0, 0, 0);
    }
  };
  {
    const std::false_type F{};
    const std::true_type T{};
    const int64_t nfull = nrows / 64;          // pairs of 32-row steps without a row test
    const bool tail = nfull * 64 < nrows;      // (its loads past r1 return 0)
    if (nrows > 0) { load(0, r0); load(1, r0); }
    for (int64_t k = 0; k < nfull; ++k) {
      const int64_t ra = r0 + 64 * k;
      const bool more = k + 1 < nfull || tail;
      step(0, ra, more, F);
      step(1, ra, more, F);
    }
    if (tail) {
      const int64_t rt = r0 + 64 * nfull;
      step(0, rt, false, T);
      if (rt + 32 < r1) step(1, rt, false, T);
    }
  }
  // D tile t: lane (i, kg) holds codes 16 t + 4 kg + r (r = 0..3) of column i
  float *out = p.partial + (size_t)blockIdx.x * ((size_t)p.h * p.d + (size_t)p.m * p.h);
  unsigned int *oc = reinterpret_cast<unsigned int *>(out + (size_t)p.h * p.d);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int code = 16 * t + 4 * kg + r;
      if (code < p.h) {                      // (the one-hot entries were 2.0: halve, exactly)
        if (colok) out[(size_t)code * p.d + col0 + i] = 0.5f * acc[t][r];
        if (cb == 0 && i == 0) oc[(size_t)q * p.h + code] = (unsigned int)(0.5f * cnt[t][r] + 0.5f);
      }
    }
  }
}

// (Tried for narrow sub-spaces -- Deep1M's sub = 6 uses 6 of the 16 B columns: the three pieces and the counter PACKED side by
// side into two B operands, half the MFMAs, pieces added across lanes at the end.  Correct, and no faster: 0.45 against 0.41 ms
// at Deep1M shape -- with 16 sub-quantizers the kernel is bound by the VALU work per (step, sub-quantizer), the mask build and
// the 8 instructions per tile, not by the matrix pipe.  The cost of this formulation is proportional to m, not to d.)
// dst[i] = sum over the nparts slices of src[w * stride + i] in a FIXED order (16 interleaved groups of slices, then the 16
// group sums in ascending order): the serial loop over 256-512 slices per element of round 3 took 70-140 us per call
template <class T>
__global__ __launch_bounds__(1024) void partials_reduce_kernel(T *__restrict__ dst, const T *__restrict__ src, size_t stride, int nparts, int count) {
  __shared__ T red[16][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  T s = 0;
  if (i < count)
    for (int w = g; w < nparts; w += 16) s += src[(size_t)w * stride + i];
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < count) {
    T t = red[0][e];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][e];
    dst[i] = t;
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
// a per-dimension table (sub-quantizer | width, gather base) in LDS replaces the walk over the offsets per element; a
// persistent grid walks the elements with a 32-bit (row, dimension) pair instead of a 64-bit division each
__global__ __launch_bounds__(256) void reconstruct_kernel(TrainParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];
  int *gbase = reinterpret_cast<int *>(rc_smem);                              // [d]  h * off[q] - off[q] + dim
  unsigned short *gq = reinterpret_cast<unsigned short *>(gbase + p.d);       // [d]  q (m <= 32)
  unsigned short *gsub = gq + p.d;                                            // [d]  width of the sub-space
  for (int dim = threadIdx.x; dim < p.d; dim += 256) {
    int q = 0;
    while (q + 1 < p.m && dim >= p.off[q + 1]) ++q;
    gbase[dim] = p.h * p.off[q] + (dim - p.off[q]);
    gq[dim] = (unsigned short)q;
    gsub[dim] = (unsigned short)(p.off[q + 1] - p.off[q]);
  }
  __syncthreads();
  const int64_t total = p.n * p.d, stride = (int64_t)gridDim.x * 256;
  const int64_t srow = stride / p.d;
  const int sdim = (int)(stride - srow * p.d);
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t r = e / p.d;
  int dim = (int)(e - r * p.d);
  for (; e < total; e += stride) {
    const int code = p.codes[r * p.m + gq[dim]];
    p.CB[e] = p.C[gbase[dim] + code * (int)gsub[dim]];
    r += srow; dim += sdim;
    if (dim >= p.d) { dim -= p.d; ++r; }
  }
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


// ---- round 4: G = X' CB and the objective WITHOUT the n x d reconstruction ---------------------------------------------
// CB is a gather: CB[row][off_q + s] = C_q[b(row, q)][s].  Writing it (0.5 ms at SIFT1M shape) only to read it back twice
// (gram, qerror) is HBM traffic the loop does not need: the two kernels below take (codes, C) instead.
//   gram_codes   per stage of 32 rows the workgroup puts the X rows and the gathered CB rows into LDS (16-byte loads; the CB
//                gather hits the L2-resident codebooks), then every wavefront runs its 32 x 32 output tiles over the stage:
//                two conflict-free ds_read_b32 per v_mfma_f32_32x32x2_f32.  The round-3 kernel fetched both MFMA operands
//                with strided global loads (2 vector-memory instructions per 64-cycle MFMA, every X column block read NT
//                times): 0.99 ms = 21 % of the f32 matrix rate; the matrix pipe's floor for 2 d^2 n flop is 0.21 ms.
//                The loads of stage s+1 (and the code bytes of stage s+2) are in flight while stage s is multiplied.
//   qerror_codes sum (X - C[code])^2 in double, 16 bytes of X and one gather per thread and step
// Both need d % 4 == 0 and sub-spaces that start and end on multiples of 4 (16-byte gathers); gram_codes needs d <= 256.
// Other shapes keep reconstruct + the kernels above.
struct CodesParams {
  const float *X;        // [n][d]
  const uint8_t *codes;  // [n][m]
  const float *C;        // concat of [h][sub_q]
  float *partial;        // gram: [grid][d*d]
  double *dpartial;      // qerror: [grid]
  int64_t n;
  int d, m, h, NT;
  int off[33];
};

__device__ __forceinline__ int codes_subq(const CodesParams &p, int dim) {
  int q = 0;
  while (q + 1 < p.m && dim >= p.off[q + 1]) ++q;
  return q;
}

constexpr int GRAMC_ROWS = 32;

template <int W> struct VecW;
template <> struct VecW<4> { using T = float4; static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); } };
template <> struct VecW<2> { using T = float2; static __device__ __forceinline__ T zero() { return make_float2(0.f, 0.f); } };

// NTT = d / 32 rounded up: NTT wavefronts, wavefront w owns the output row block a = w with ALL NTT column blocks b (NTT
// accumulator tiles): its A operand (the X column block) is read once per K step for NTT MFMAs -- (1 + NTT) / NTT LDS reads
// per MFMA instead of 2 -- and no wavefront idles whatever NTT is (the first layout gave 8 wavefronts 2 slots each: 16 slots
// for Deep1M's 9 tiles).
template <int NTT, int W>
__global__ __launch_bounds__(NTT * 64) void gram_codes_kernel(CodesParams p) {
  using V = typename VecW<W>::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTHREADS = NTT * 64;
  constexpr int UPT = 16 / W;                             // >= 32 rows * (32 NTT / W) units / (64 NTT) threads
  constexpr int KS = NTT <= 4 ? 8 : 4;                    // K steps whose operands are fetched ahead of their MFMAs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int d = p.d, ld = NTT * 32, d4 = ld / W;          // d4: W-wide units per row of the staged column blocks
  // d > 32 NTT (d > 256 at NTT = 8): blockIdx.y = (A, B) picks the output block rows 32 NTT A .. and columns 32 NTT B ..;
  // the stage then holds X's column block A and CB's column block B
  const int nblk = (d + ld - 1) / ld;
  const int acol0 = (int)(blockIdx.y / nblk) * ld, bcol0 = (int)(blockIdx.y % nblk) * ld;
  float *Xs = reinterpret_cast<float *>(smem);          // [32][ld]
  float *Hs = Xs + GRAMC_ROWS * ld;                     // [32][ld]
  for (int i = tid; i < 2 * GRAMC_ROWS * ld; i += NTHREADS) Xs[i] = 0.0f;     // the padding columns stay zero
  const int64_t rows_per = (((p.n + gridDim.x - 1) / gridDim.x) + GRAMC_ROWS - 1) / GRAMC_ROWS * GRAMC_ROWS;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = min(p.n, r0 + rows_per);
  const int nstage = r1 > r0 ? (int)((r1 - r0 + GRAMC_ROWS - 1) / GRAMC_ROWS) : 0;
  const int units = GRAMC_ROWS * d4;
  // this thread's units of a stage: (row, W dimensions); their sub-quantizer and gather base never change
  int urow[UPT], ucol[UPT], uq[UPT], ubase[UPT], usub[UPT];
  bool xin[UPT], hin[UPT];
#pragma unroll
  for (int u = 0; u < UPT; ++u) {
    const int e = tid + u * NTHREADS;
    urow[u] = e < units ? e / d4 : -1;
    ucol[u] = e < units ? W * (e % d4) : 0;                       // column inside the staged blocks
    xin[u] = urow[u] >= 0 && acol0 + ucol[u] < d;
    hin[u] = urow[u] >= 0 && bcol0 + ucol[u] < d;
    const int hcol = hin[u] ? bcol0 + ucol[u] : 0;
    uq[u] = codes_subq(p, hcol);
    ubase[u] = p.h * p.off[uq[u]] + (hcol - p.off[uq[u]]);         // + code * sub
    usub[u] = p.off[uq[u] + 1] - p.off[uq[u]];
  }
  int cd[UPT];
  V xv[UPT], hv[UPT];
  auto load_codes = [&](int s) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int64_t row = r0 + (int64_t)s * GRAMC_ROWS + urow[u];
      cd[u] = (hin[u] && row < r1) ? (int)p.codes[row * p.m + uq[u]] : 0;
    }
  };
  auto load_data = [&](int s) {
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      const int64_t row = r0 + (int64_t)s * GRAMC_ROWS + urow[u];
      const bool in = row < r1;
      xv[u] = (in && xin[u]) ? *reinterpret_cast<const V *>(p.X + row * d + acol0 + ucol[u]) : VecW<W>::zero();
      hv[u] = (in && hin[u]) ? *reinterpret_cast<const V *>(p.C + ubase[u] + cd[u] * usub[u]) : VecW<W>::zero();
    }
  };
  f32x16 acc[NTT];
#pragma unroll
  for (int i = 0; i < NTT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  if (nstage > 0) {
    load_codes(0);
    load_data(0);
    if (nstage > 1) load_codes(1);
  }
  for (int s = 0; s < nstage; ++s) {
    __syncthreads();                       // the previous stage has been multiplied (first trip: LDS zeroed)
#pragma unroll
    for (int u = 0; u < UPT; ++u) {
      if (urow[u] >= 0) {
        *reinterpret_cast<V *>(Xs + urow[u] * ld + ucol[u]) = xv[u];
        *reinterpret_cast<V *>(Hs + urow[u] * ld + ucol[u]) = hv[u];
      }
    }
    __syncthreads();
    if (s + 1 < nstage) {
      load_data(s + 1);                    // cd holds the codes of stage s + 1, loaded one trip ago
      if (s + 2 < nstage) load_codes(s + 2);
    }
    // KS K steps at a time: all operands first, then the MFMAs back to back
#pragma unroll
    for (int part = 0; part < GRAMC_ROWS / 2 / KS; ++part) {
      float av[KS], bv[NTT][KS];
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const int ro = (2 * (KS * part + u) + hi) * ld + j;
        av[u] = Xs[ro + 32 * wave];
#pragma unroll
        for (int i = 0; i < NTT; ++i) bv[i][u] = Hs[ro + 32 * i];
      }
#pragma unroll
      for (int u = 0; u < KS; ++u)
#pragma unroll
        for (int i = 0; i < NTT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[i][u], acc[i], 0, 0, 0);
    }
  }
  float *out = p.partial + (size_t)blockIdx.x * d * d;
#pragma unroll
  for (int i = 0; i < NTT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ii = acol0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hi, jj = bcol0 + 32 * i + j;
      if (ii < d && jj < d) out[(size_t)ii * d + jj] = acc[i][r];
    }
  }
}

// Per W-wide unit (row, block): sub-quantizer and gather base of the block come from a table built once per workgroup (walking
// the offsets per unit -- up to m - 1 compares -- made this kernel VALU-bound: 0.45 ms at Deep1M shape, m = 16), and the
// (row, block) pair advances by the grid stride without a division.
template <int W>
__global__ __launch_bounds__(256) void qerror_codes_kernel(CodesParams p) {
  using V = typename VecW<W>::T;
  __shared__ double red[256];
  __shared__ int gbase[1024 / W];            // block -> h * off[q] - off[q] + column  (the gather address minus code * sub)
  __shared__ unsigned short gq[1024 / W];    // block -> q | sub << 5
  const int d4 = p.d / W;
  for (int c = threadIdx.x; c < d4; c += 256) {
    const int col = W * c, q = codes_subq(p, col);
    gbase[c] = p.h * p.off[q] + (col - p.off[q]);
    gq[c] = (unsigned short)(q | ((p.off[q + 1] - p.off[q]) << 5));
  }
  __syncthreads();
  const int64_t total = p.n * d4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t srow = stride / d4;
  const int scol = (int)(stride - srow * d4);
  int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t row = e / d4;
  int cb = (int)(e - row * d4);
  double s = 0.0;
  for (; e < total; e += stride) {
    const int qs = gq[cb], q = qs & 31, sub = qs >> 5;
    const int code = p.codes[row * p.m + q];
    const V x = *reinterpret_cast<const V *>(p.X + row * p.d + W * cb);
    const V c = *reinterpret_cast<const V *>(p.C + gbase[cb] + code * sub);
    const double a0 = (double)x.x - (double)c.x, a1 = (double)x.y - (double)c.y;
    s += a0 * a0; s += a1 * a1;
    if constexpr (W == 4) {
      const double a2 = (double)x.z - (double)c.z, a3 = (double)x.w - (double)c.w;
      s += a2 * a2; s += a3 * a3;
    }
    row += srow; cb += scol;
    if (cb >= d4) { cb -= d4; ++row; }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) p.dpartial[blockIdx.x] = red[0];
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
  float *mincost;       // [n][m]
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

// One workgroup per block of rows, ALL sub-spaces at once: thread t = (row slot t / m, sub-space t % m), so a wavefront reads
// whole rows (consecutive lanes = consecutive sub-vectors of a row) and X crosses the memory system once per step.  (Round 3
// ran a 1024-thread workgroup per (block, sub-space), one row per thread and a 10-level tree per 977 rows: 1.16 ms per step,
// 0.3 s for h = 256 seeds at SIFT1M shape -- 13 x the 25 Lloyd iterations that follow.)  mincost is [n][m].
__global__ __launch_bounds__(256) void kmpp_update_kernel(KmppParams p, int step) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kmpp_smem[];
  float *xs = reinterpret_cast<float *>(kmpp_smem);                     // [d] the sub-vectors of the seeds of the previous step
  double *red = reinterpret_cast<double *>(xs + ((p.d + 3) & ~3));         // [T]
  const int T = blockDim.x, tid = threadIdx.x, m = p.m;
  for (int e = tid; e < p.d; e += T) {
    int q = 0;
    while (q + 1 < m && e >= p.off[q + 1]) ++q;
    xs[e] = p.X[(size_t)p.seeds[(size_t)q * p.h + step - 1] * p.d + e];
  }
  __syncthreads();
  const int q = tid % m, slot = tid / m, rpi = T / m;
  const int o = p.off[q], sub = p.off[q + 1] - o;
  const long long seed = p.seeds[(size_t)q * p.h + step - 1];
  const int64_t r0 = (int64_t)blockIdx.x * p.rows_per_blk, r1 = min(p.n, r0 + p.rows_per_blk);
  const bool vec = (sub & 3) == 0 && (o & 3) == 0 && (p.d & 3) == 0;
  double acc = 0.0;
  for (int64_t r = r0 + slot; r < r1; r += rpi) {
    const float *x = p.X + (size_t)r * p.d + o;
    float dist = 0.0f;
    if (vec) {
      for (int s4 = 0; s4 < sub; s4 += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(x + s4);
        const float4 w = *reinterpret_cast<const float4 *>(xs + o + s4);
        float df = v.x - w.x; dist = __builtin_fmaf(df, df, dist);
        df = v.y - w.y; dist = __builtin_fmaf(df, df, dist);
        df = v.z - w.z; dist = __builtin_fmaf(df, df, dist);
        df = v.w - w.w; dist = __builtin_fmaf(df, df, dist);
      }
    } else {
      for (int s = 0; s < sub; ++s) {
        const float df = x[s] - xs[o + s];
        dist = __builtin_fmaf(df, df, dist);
      }
    }
    float *mc = p.mincost + (size_t)r * m + q;
    float c = (step == 1) ? dist : fminf(*mc, dist);
    if (r == seed) c = 0.0f;
    *mc = c;
    acc += (double)c;
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < m) {                       // the rpi row slots of sub-space tid, in slot order
    double tot = 0.0;
    for (int k = 0; k < rpi; ++k) tot += red[tid + k * m];
    p.partial[(size_t)tid * p.nblk + blockIdx.x] = tot;
  }
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
  const float *mcq = p.mincost + i;              // [n][m]: the cost of row r in this sub-space is mcq[r * m]
  const int64_t ms = p.m;
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
    for (int64_t r = c0; r < c1; ++r) cs += (double)mcq[r * ms];
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
        run += (double)mcq[r * ms];
        if (run > rthr && mcq[r * ms] > 0.0f) { rr = r; break; }
      }
      if (rr < 0)
        for (int64_t r = c1 - 1; r >= c0; --r) if (mcq[r * ms] > 0.0f) { rr = r; break; }
      pick[1] = rr;
    }
    __syncthreads();
    if (pick[1] < 0 && tid == 0) {      // threshold beyond the block's re-summed total: last row with a cost
      for (int64_t r = r1 - 1; r >= r0; --r) if (mcq[r * ms] > 0.0f) { pick[1] = r; break; }
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
  const int upd_threads = m * (256 / m);           // a multiple of m: every thread keeps one sub-space
  const size_t upd_lds = (size_t)((d + 3) & ~3) * sizeof(float) + (size_t)upd_threads * sizeof(double);
  hipLaunchKernelGGL(kmpp_first_kernel, dim3(m), dim3(64), 0, stream, p);
  RQ_HIP(hipGetLastError());
  for (int step = 1; step < h; ++step) {
    hipLaunchKernelGGL(kmpp_update_kernel, dim3(p.nblk), dim3(upd_threads), upd_lds, stream, p, step);
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

template <class T>
static int partials_reduce(T *dst, const T *src, size_t stride, int nparts, int count, hipStream_t stream) {
  hipLaunchKernelGGL(partials_reduce_kernel<T>, dim3((count + 63) / 64), dim3(1024), 0, stream, dst, src, stride, nparts, count);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// slices [grid][h*d + m*h] -> one reduced slice behind them -> means
static int centers_finish(TrainParams p, int grid, hipStream_t stream) {
  const size_t hd = (size_t)p.h * p.d, mh = (size_t)p.m * p.h, stride = hd + mh;
  float *red = p.partial + (size_t)grid * stride;
  RQ_TRY(partials_reduce<float>(red, p.partial, stride, grid, (int)hd, stream));
  RQ_TRY(partials_reduce<unsigned int>(reinterpret_cast<unsigned int *>(red + hd), reinterpret_cast<const unsigned int *>(p.partial + hd), stride,
                                       grid, (int)mh, stream));
  p.partial = red;
  hipLaunchKernelGGL(centers_finish_kernel, dim3((p.h * p.d + 255) / 256), dim3(256), 0, stream, p, 1);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
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
  RQ_TRY(workspace(WS_TMP, (size_t)(grid + 1) * ((size_t)h * d + (size_t)m * h) * sizeof(float), &part, stream));
  p.partial = (float *)part;
  if (tuning("TRAIN_CENTERS_MFMA", 1) && n / std::max(1, grid) < (1 << 23) && ((n + grid - 1) / grid) * (int64_t)d * 4 < (1ll << 31)) {
    // one-hot products on the bf16 matrix cores; a wavefront per (sub-quantizer, 16-dimension block), 8 of them per workgroup.
    // (counts ride in f32 accumulators: exact below 2^24 rows per slice)
    int nunits = 0;
    for (int q = 0; q < m; ++q) nunits += (p.off[q + 1] - p.off[q] + 15) / 16;
    const int nslice = grid;
    const dim3 g3(nslice, (nunits + 7) / 8);
    if (h <= 64) hipLaunchKernelGGL(centers_mfma_kernel<4>, g3, dim3(512), 0, stream, p, nunits);
    else if (h <= 128) hipLaunchKernelGGL(centers_mfma_kernel<8>, g3, dim3(512), 0, stream, p, nunits);
    else hipLaunchKernelGGL(centers_mfma_kernel<16>, g3, dim3(512), 0, stream, p, nunits);
    RQ_HIP(hipGetLastError());
    return centers_finish(p, grid, stream);
  }
  // dimension chunks of <= 128: h * 128 * 4 B of sums (+ m * h counters in the first chunk) always fit the LDS
  for (int dc0 = 0; dc0 < d; dc0 += 128) {
    const int dcw = std::min(128, d - dc0);
    const size_t lds = ((size_t)h * dcw + (size_t)m * h) * sizeof(float);
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(centers_partial_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(centers_partial_kernel, dim3(grid), dim3(1024), lds, stream, p, dc0, dcw);
    RQ_HIP(hipGetLastError());
  }
  return centers_finish(p, grid, stream);
}

int reconstruct_launch(float *CB, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h,
                       hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  if (m < 1 || m > 32 || d < m) return fail(RQ_EINVAL, "reconstruct: m=%d d=%d", m, d);
  TrainParams p{};
  p.codes = codes; p.C = const_cast<float *>(C); p.CB = CB; p.n = n; p.d = d; p.m = m; p.h = h;
  fill_offsets(p.off, d, m);
  const int64_t total = n * d;
  if (d > 8192) return fail(RQ_EUNSUPPORTED, "reconstruct: d=%d", d);
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(reconstruct_kernel, dim3(grid), dim3(256), (size_t)d * 8, stream, p);
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

// can the (codes, C) forms of gram / qerror serve this shape?  Returns the gather width in floats: 4 (every sub-space starts on
// a multiple of 4: 16-byte gathers), 2 (multiples of 2, e.g. Deep1M's d = 96, m = 16: 8-byte gathers) or 0 (no)
int codes_forms_width(int d, int m, int h, bool for_gram) {
  if (!tuning("TRAIN_FUSED_CB", 1)) return 0;
  if (m < 1 || m > 32 || d < m || (d & 1) || h < 1 || h > 256) return 0;
  if (d > 1024) return 0;
  int off[33];
  fill_offsets(off, d, m);
  int w = 4;
  for (int q = 0; q <= m; ++q) {
    if (off[q] & 1) return 0;
    if (off[q] & 3) w = 2;
  }
  return w;
}
bool codes_forms_ok(int d, int m, int h, bool for_gram) { return codes_forms_width(d, m, h, for_gram) != 0; }

static void fill_codes_params(CodesParams &p, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h) {
  p.X = X; p.codes = codes; p.C = C; p.partial = nullptr; p.dpartial = nullptr; p.n = n; p.d = d; p.m = m; p.h = h; p.NT = (d + 31) / 32;
  fill_offsets(p.off, d, m);
}

template <int NTT, int W>
static int gram_codes_run(CodesParams &p, int grid, size_t lds, hipStream_t stream) {
  auto kern = gram_codes_kernel<NTT, W>;
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nblk = (p.d + 32 * NTT - 1) / (32 * NTT);
  hipLaunchKernelGGL(kern, dim3(grid, nblk * nblk), dim3(NTT * 64), lds, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

template <int W>
static int gram_codes_pick(CodesParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.NT > 8) return gram_codes_run<8, W>(p, grid, lds, stream);        // d > 256: 256-wide output blocks, blockIdx.y
  switch (p.NT) {
    case 1: return gram_codes_run<1, W>(p, grid, lds, stream);
    case 2: return gram_codes_run<2, W>(p, grid, lds, stream);
    case 3: return gram_codes_run<3, W>(p, grid, lds, stream);
    case 4: return gram_codes_run<4, W>(p, grid, lds, stream);
    case 5: return gram_codes_run<5, W>(p, grid, lds, stream);
    case 6: return gram_codes_run<6, W>(p, grid, lds, stream);
    case 7: return gram_codes_run<7, W>(p, grid, lds, stream);
    case 8: return gram_codes_run<8, W>(p, grid, lds, stream);
  }
  return fail(RQ_EUNSUPPORTED, "gram_codes: d=%d", p.d);
}

// G = X' CB with CB given as (codes, C); codes_forms_ok(d, m, h, true) must hold
int gram_codes_launch(float *G, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h, int num_cu,
                      hipStream_t stream) {
  const int w = codes_forms_width(d, m, h, true);
  if (!w) return fail(RQ_EUNSUPPORTED, "gram_codes: d=%d m=%d h=%d", d, m, h);
  CodesParams p;
  fill_codes_params(p, X, codes, C, n, d, m, h);
  // workgroups of NT wavefronts, 8 wavefronts per CU (measured 4 ... 32: 8 is best at d = 128 and d = 96 -- more workgroups mean more
  // partial matrices to write and reduce, fewer leave the matrix pipe idle at the stage barriers); d > 256: nblk^2 output
  // blocks of 256 x 256 per row slice
  const int ntt = std::min(p.NT, 8);
  const int nblk = (p.NT + ntt - 1) / ntt;
  const int per_cu = std::max(1, tuning("GRAM_WAVES_PER_CU", 8) / ntt);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(1, (int64_t)per_cu * num_cu / (nblk * nblk)), (n + 255) / 256));
  void *part = nullptr;
  RQ_TRY(workspace(WS_TMP, (size_t)grid * d * d * sizeof(float), &part, stream));
  p.partial = (float *)part;
  const size_t lds = (size_t)2 * GRAMC_ROWS * ntt * 32 * sizeof(float);
  if (w == 4) RQ_TRY(gram_codes_pick<4>(p, grid, lds, stream));
  else RQ_TRY(gram_codes_pick<2>(p, grid, lds, stream));
  return partials_reduce<float>(G, p.partial, (size_t)d * d, grid, d * d, stream);
}

// acc = sum |X - CB|^2 with CB given as (codes, C); codes_forms_ok(d, m, h, false) must hold
int qerror_codes_launch(double *acc_dev, const float *X, const uint8_t *codes, const float *C, int64_t n, int d, int m, int h,
                        int num_cu, hipStream_t stream) {
  const int w = codes_forms_width(d, m, h, false);
  if (!w) return fail(RQ_EUNSUPPORTED, "qerror_codes: d=%d m=%d h=%d", d, m, h);
  if (n <= 0) {
    RQ_HIP(hipMemsetAsync(acc_dev, 0, sizeof(double), stream));
    return RQ_OK;
  }
  CodesParams p;
  fill_codes_params(p, X, codes, C, n, d, m, h);
  const int grid = num_cu * 8;
  void *part = nullptr;
  RQ_TRY(workspace(WS_MERGE, (size_t)grid * sizeof(double), &part, stream));
  p.dpartial = (double *)part;
  if (w == 4) hipLaunchKernelGGL(qerror_codes_kernel<4>, dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(qerror_codes_kernel<2>, dim3(grid), dim3(256), 0, stream, p);
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
  return partials_reduce<float>(G, p.partial, (size_t)d * d, grid, d * d, stream);
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

// ------------------------------------------------------------------------------------------
// The same polar factor by a SCALED NEWTON-SCHULZ iteration (round 4, the default; the Jacobi SVD above is its fallback).
// The Jacobi sweeps are 127 latency-bound rounds each inside ONE workgroup: 1.74 ms of a 4.96 ms OPQ iteration.  The
// orthogonal polar factor of G (= U V' of src/OPQ.jl:112-113 whenever G has full rank) needs no SVD:
//      X_0 = G / |G|_F          (singular values in (0, 1])
//      X_{k+1} = X_k (a_k I + b_k X_k' X_k),     a_k = 1.5 rho_k,  b_k = -0.5 rho_k^3,  rho_k^2 = 3 / (1 + l_k + l_k^2)
// where l_k is a lower bound of the singular values of X_k (l_{k+1} = rho l (1.5 - 0.5 rho^2 l^2); Chen & Chow's scaling of
// the cubic Newton-Schulz map g(y) = 1.5 y - 0.5 y^3: rho stretches [l, 1] so that both ends land on the same value).  Every
// singular value in [0, 1] stays in [0, 1] for ANY l (g <= 1 on [0, sqrt 3]), so a wrong guess of l_0 only costs iterations;
// small values grow by up to 2.6x per step instead of 1.5x, and from l ~ 1 on the map is the plain one with its quadratic
// convergence.  Two d x d x d products in double per step, all matrix work spread over (d/16)^2 tiles: ONE persistent
// launch of min(tiles, CUs) workgroups with a grid barrier (a monotone counter in global memory, agent-scope release /
// acquire) between products -- every workgroup is resident (grid <= number of CUs), so the barrier cannot starve.
// Convergence: |X'X - I|_F < 1e-9 (the same 1e-9 as the Jacobi's angle test; R is stored as f32), measured on the product
// the step computes anyway; the per-workgroup parts are summed in a fixed order by everyone, so the iteration count -- and
// with it every bit of R -- is reproducible.  status[0] = 1 when the iteration does not get there in `maxit` steps (a
// singular G: the factor is not unique) or |G|_F is not a positive finite number: the caller falls back.
// ------------------------------------------------------------------------------------------
constexpr int NS_T = 16, NS_KC = 128, NS_THREADS = 256, NS_LDA = 17;

struct NsParams {
  const float *G;        // [d][d] row-major
  double *X0, *X1, *Y;   // [d][d] each
  double *part;          // [maxit + 1][nwg] per-workgroup parts of |X'X - I|_F^2
  unsigned int *bar;     // [0] barrier counter, [1] abandon flag; zeroed before the launch
  float *Rimg;           // Rimg[i * d + k] = R[k][i]
  int *status;           // [0] 0 = R written, 1 = fall back; [1] steps taken
  int d, maxit;
  double l0, tol2;
};

// Returns false when the barrier was abandoned: a workgroup that waits longer than ~2 s (a grid that is not fully resident
// for some reason this code did not foresee) raises bar[1], everybody else sees it at its next poll, the kernel reports
// status 1 and the caller falls back -- a wrong assumption must cost a fallback, not a hung device.
constexpr unsigned int NS_SPIN_LIMIT = 1u << 21;
__device__ __forceinline__ bool ns_grid_barrier(unsigned int *bar, unsigned int &target, unsigned int nwg, int *s_ok) {
  // __syncthreads waits for every thread's stores to be acknowledged by the L2; thread 0 alone then releases (writes this XCD's
  // L2 back: the XCDs have their own) and, after the wait, acquires (drops this CU's L1 and the stale L2 lines) for everybody
  // -- the L1 is the CU's, so one invalidate serves all wavefronts of the workgroup.  (Fences by all 256 threads of all
  // workgroups cost ~15 us per barrier.)
  __syncthreads();
  target += nwg;
  if (threadIdx.x == 0) {
    int ok = 1;
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u) {
        if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
        if (spins > NS_SPIN_LIMIT) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_ok = ok;
  }
  __syncthreads();
  return *s_ok != 0;
}

// sum of v over the workgroup, the same fixed tree everywhere
__device__ __forceinline__ double ns_block_sum(double v, double *red) {
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (int w = NS_THREADS / 2; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  return red[0];
}

// one 16 x 16 output tile: out(ti, tj) = sum_k A(k, ti) B(k, tj), k over the d rows in chunks of NS_KC through LDS.
//   MODE 0 (Y = X'X):          A(k, c) = X[k][16 I + c]         B(k, c) = X[k][16 J + c]
//   MODE 1 (X (a I + b Y)):    A(k, c) = X[16 I + c][k]         B(k, c) = b Y[k][16 J + c] + a [k == 16 J + c]   (Y is symmetric)
template <int MODE>
__device__ __forceinline__ double ns_tile(const double *__restrict__ X, const double *__restrict__ Y, int d, int I, int J, double a,
                                          double b, double *As, double *Bs) {
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  for (int k0 = 0; k0 < d; k0 += NS_KC) {
    __syncthreads();
    if (MODE == 0) {
      const int c = tid & 15;
      for (int k = tid >> 4; k < NS_KC; k += NS_THREADS / 16) {
        const int kk = k0 + k;
        As[k * NS_LDA + c] = (kk < d && 16 * I + c < d) ? X[(size_t)kk * d + 16 * I + c] : 0.0;
        Bs[k * NS_T + c] = (kk < d && 16 * J + c < d) ? X[(size_t)kk * d + 16 * J + c] : 0.0;
      }
    } else {
      const int k = tid & (NS_KC - 1);
      for (int c = tid >> 7; c < NS_T; c += NS_THREADS / NS_KC)
        As[k * NS_LDA + c] = (k0 + k < d && 16 * I + c < d) ? X[(size_t)(16 * I + c) * d + k0 + k] : 0.0;
      const int c = tid & 15;
      for (int kq = tid >> 4; kq < NS_KC; kq += NS_THREADS / 16) {
        const int kk = k0 + kq, col = 16 * J + c;
        double v = 0.0;
        if (kk < d && col < d) v = __builtin_fma(b, Y[(size_t)kk * d + col], kk == col ? a : 0.0);
        Bs[kq * NS_T + c] = v;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < NS_KC; k += 4) {
      acc0 = __builtin_fma(As[(k + 0) * NS_LDA + ti], Bs[(k + 0) * NS_T + tj], acc0);
      acc1 = __builtin_fma(As[(k + 1) * NS_LDA + ti], Bs[(k + 1) * NS_T + tj], acc1);
      acc2 = __builtin_fma(As[(k + 2) * NS_LDA + ti], Bs[(k + 2) * NS_T + tj], acc2);
      acc3 = __builtin_fma(As[(k + 3) * NS_LDA + ti], Bs[(k + 3) * NS_T + tj], acc3);
    }
  }
  return (acc0 + acc1) + (acc2 + acc3);
}

// The same product on 64 x 64 output tiles, 4 x 4 outputs per thread (d >= 384: at d = 960 the 16 x 16 tiles above -- two LDS
// reads per FMA, 3600 tiles over 256 workgroups -- ran at 3 TF: 39 ms per polar factor): per k step a thread reads 4 + 4
// doubles (four ds_read_b128) for 16 FMAs.  K in chunks of 32 through LDS (the same 34 KiB).
constexpr int NSB_T = 64, NSB_KC = 32, NSB_LDA = 66;
template <int MODE>
__device__ __forceinline__ void ns_tile_big(const double *__restrict__ X, const double *__restrict__ Y, int d, int I, int J, double a,
                                            double b, double *As, double *Bs, double (&acc)[4][4]) {
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  constexpr int PER = NSB_KC * NSB_T / NS_THREADS;        // 8 elements of each operand per thread and chunk
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  // the chunk after the one being multiplied is already on its way from memory (registers), so that a chunk costs
  // max(load latency, its 32 k steps) and not their sum
  double pa[PER], pb[PER];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * NS_THREADS;
      if (MODE == 0) {
        const int k = e >> 6, c = e & 63, kk = k0 + k;
        pa[u] = (kk < d && NSB_T * I + c < d) ? X[(size_t)kk * d + NSB_T * I + c] : 0.0;
        pb[u] = (kk < d && NSB_T * J + c < d) ? X[(size_t)kk * d + NSB_T * J + c] : 0.0;
      } else {
        const int ka = e & (NSB_KC - 1), ca = e >> 5;                    // k fastest: rows of X are read along k
        pa[u] = (k0 + ka < d && NSB_T * I + ca < d) ? X[(size_t)(NSB_T * I + ca) * d + k0 + ka] : 0.0;
        const int k = e >> 6, c = e & 63, kk = k0 + k, col = NSB_T * J + c;
        pb[u] = (kk < d && col < d) ? __builtin_fma(b, Y[(size_t)kk * d + col], kk == col ? a : 0.0) : 0.0;
      }
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * NS_THREADS;
      if (MODE == 0) {
        As[(e >> 6) * NSB_LDA + (e & 63)] = pa[u];
      } else {
        As[(e & (NSB_KC - 1)) * NSB_LDA + (e >> 5)] = pa[u];
      }
      Bs[(e >> 6) * NSB_T + (e & 63)] = pb[u];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < d; k0 += NSB_KC) {
    __syncthreads();
    stash();
    __syncthreads();
    if (k0 + NSB_KC < d) fetch(k0 + NSB_KC);
#pragma unroll 4
    for (int k = 0; k < NSB_KC; ++k) {
      double av[4], bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) av[r] = As[k * NSB_LDA + 4 * ti + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = Bs[k * NSB_T + 4 * tj + c];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = __builtin_fma(av[r], bv[c], acc[r][c]);
    }
  }
}

// TS = 16: one output per thread (ns_tile); TS = 64: 4 x 4 outputs per thread (ns_tile_big)
template <int TS>
__global__ __launch_bounds__(NS_THREADS) void polar_ns_kernel(NsParams p) {
  constexpr int R = TS / 16;
  constexpr int AS_SZ = TS == 16 ? NS_KC * NS_LDA : NSB_KC * NSB_LDA, BS_SZ = TS == 16 ? NS_KC * NS_T : NSB_KC * NSB_T;
  __shared__ double As[AS_SZ];
  __shared__ double Bs[BS_SZ];
  __shared__ double red[NS_THREADS];
  __shared__ int s_ok;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const int d = p.d, T = (d + TS - 1) / TS, ntile = T * T;
  const unsigned int nwg = gridDim.x, wg = blockIdx.x;
  unsigned int target = 0;
  // |G|_F^2, by every workgroup in the same order
  double f = 0.0;
  for (int e = tid; e < d * d; e += NS_THREADS) { const double g = (double)p.G[e]; f = __builtin_fma(g, g, f); }
  const double fro2 = ns_block_sum(f, red);
  if (!(fro2 > 0.0) || !(fro2 < 1.0e300)) {
    if (wg == 0 && tid == 0) { p.status[0] = 1; p.status[1] = 0; }
    return;
  }
  const double inv = 1.0 / sqrt(fro2);
  for (int e = (int)wg * NS_THREADS + tid; e < d * d; e += (int)nwg * NS_THREADS) p.X0[e] = (double)p.G[e] * inv;
  if (!ns_grid_barrier(p.bar, target, nwg, &s_ok)) { if (tid == 0) p.status[0] = 1; return; }
  double *cur = p.X0, *nxt = p.X1;
  double l = p.l0;
  int it = 0, ok = 0;
  // the thread's outputs of tile (I, J): rows TS I + R ti + r, columns TS J + R tj + c
  auto tile = [&](int mode, const double *Xc, int I, int J, double a, double b, double (&o)[R][R]) {
    if constexpr (TS == 16) {
      o[0][0] = mode == 0 ? ns_tile<0>(Xc, nullptr, d, I, J, 0.0, 0.0, As, Bs) : ns_tile<1>(Xc, p.Y, d, I, J, a, b, As, Bs);
    } else {
      if (mode == 0) ns_tile_big<0>(Xc, nullptr, d, I, J, 0.0, 0.0, As, Bs, o);
      else ns_tile_big<1>(Xc, p.Y, d, I, J, a, b, As, Bs, o);
    }
  };
  for (;; ++it) {
    double perr = 0.0;
    for (int t = wg; t < ntile; t += nwg) {
      const int I = t / T, J = t % T;
      double o[R][R];
      tile(0, cur, I, J, 0.0, 0.0, o);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < R; ++c) {
          const int i = TS * I + R * ti + r, j = TS * J + R * tj + c;
          if (i < d && j < d) {
            p.Y[(size_t)i * d + j] = o[r][c];
            const double dv = o[r][c] - (i == j ? 1.0 : 0.0);
            perr = __builtin_fma(dv, dv, perr);
          }
        }
    }
    const double mine = ns_block_sum(perr, red);
    if (tid == 0) p.part[(size_t)it * nwg + wg] = mine;
    if (!ns_grid_barrier(p.bar, target, nwg, &s_ok)) { if (tid == 0) p.status[0] = 1; return; }
    double e = 0.0;
    for (unsigned int w = tid; w < nwg; w += NS_THREADS) e += p.part[(size_t)it * nwg + w];
    e = ns_block_sum(e, red);
    if (e < p.tol2) { ok = 1; break; }
    if (it >= p.maxit || !(e < 1.0e300)) break;
    double a = 1.5, b = -0.5;
    if (l < 1.0 - 1e-9) {
      const double rho2 = 3.0 / (1.0 + l + l * l), rho = sqrt(rho2);
      a = 1.5 * rho; b = -0.5 * rho * rho2;
      l = rho * l * (1.5 - 0.5 * rho2 * l * l);
      if (!(l < 1.0)) l = 1.0;
    }
    for (int t = wg; t < ntile; t += nwg) {
      const int I = t / T, J = t % T;
      double o[R][R];
      tile(1, cur, I, J, a, b, o);
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < R; ++c) {
          const int i = TS * I + R * ti + r, j = TS * J + R * tj + c;
          if (i < d && j < d) nxt[(size_t)i * d + j] = o[r][c];
        }
    }
    if (!ns_grid_barrier(p.bar, target, nwg, &s_ok)) { if (tid == 0) p.status[0] = 1; return; }
    double *tmp = cur; cur = nxt; nxt = tmp;
  }
  if (wg == 0 && tid == 0) { p.status[0] = ok ? 0 : 1; p.status[1] = it; }
  if (!ok) return;
  // X[i][j] = R[i][j]  ->  Rimg[j * d + i]
  for (int e = (int)wg * NS_THREADS + tid; e < d * d; e += (int)nwg * NS_THREADS) {
    const int i = e / d, j = e - i * d;
    p.Rimg[(size_t)j * d + i] = (float)cur[e];
  }
}

// scratch: polar_ns_scratch_bytes(d, num_cu) bytes; status two ints (read after the stream has drained)
size_t polar_ns_scratch_bytes(int d, int num_cu) {
  const int maxit = 96;
  return ((size_t)3 * d * d + (size_t)(maxit + 2) * (size_t)std::max(1, num_cu)) * sizeof(double) + 64;
}

int polar_ns_launch(float *Rimg, const float *G, int d, int *status, void *scratch, int num_cu, hipStream_t stream) {
  if (d < 1 || d > 1024) return fail(RQ_EUNSUPPORTED, "device polar factor: d <= 1024; got %d", d);
  NsParams p;
  // 16 x 16 output tiles, one per thread-block of 256 outputs; from d = 384 on 64 x 64 tiles with 4 x 4 outputs per thread
  const bool big = d >= tuning("TRAIN_NS_BIG_D", 384);
  const int T = big ? (d + NSB_T - 1) / NSB_T : (d + NS_T - 1) / NS_T;
  const int nwg = std::max(1, std::min(T * T, num_cu));
  p.G = G; p.Rimg = Rimg; p.status = status; p.d = d;
  p.maxit = 96;
  p.l0 = (double)tuning("TRAIN_NS_L0_MICRO", 1000) * 1e-6;
  if (!(p.l0 > 0.0) || p.l0 > 1.0) p.l0 = 1e-3;
  p.tol2 = 1e-18;     // (1e-12 saves no step at kappa ~ 1e3: the last step goes from ~1e-4 straight below 1e-9)
  double *s = reinterpret_cast<double *>(scratch);
  p.X0 = s; p.X1 = s + (size_t)d * d; p.Y = s + (size_t)2 * d * d;
  p.part = s + (size_t)3 * d * d;
  p.bar = reinterpret_cast<unsigned int *>(p.part + (size_t)(p.maxit + 2) * nwg);
  RQ_HIP(hipMemsetAsync(p.bar, 0, 8, stream));
  if (big) hipLaunchKernelGGL(polar_ns_kernel<64>, dim3(nwg), dim3(NS_THREADS), 0, stream, p);
  else hipLaunchKernelGGL(polar_ns_kernel<16>, dim3(nwg), dim3(NS_THREADS), 0, stream, p);
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
