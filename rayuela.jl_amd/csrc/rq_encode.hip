// rq_encode.hip -- PQ encode (quantize_pq) and the OPQ rotation R'X on gfx950 f32 MFMA.
//
// Replaces the per-subspace  Distances.pairwise(SqEuclidean(), C[i], X[subdims[i],:])  +
// Clustering.update_assignments!  pair of src/PQ.jl:37-43 (1 GB h x n distance matrix written
// and re-read per subspace on the CPU) and the  R' * X  sgemm of src/OPQ.jl:26.
//
// Canonical arithmetic (oracle/rq_oracle.c, DESIGN.md "numerics"):
//   g_k = <c_k, x>, sa_k = |c_k|^2, sb = |x|^2 : k-ordered fmaf chains from +0
//   v_k = max( fl( fl(sa_k + sb) - 2 g_k ), 0 ) ;  code = first index of the minimum (strict '<')
// v_mfma_f32_32x32x2_f32 IS a k-ordered fmaf chain, bit for bit, so the 256 x sub inner
// products of a subspace come straight off the matrix cores with CPU-reproducible bits.
//
// Encode kernel layout (one 256- or 512-thread workgroup per CU, persistent over row tiles):
//   LDS   cbA [m][NT][KS][64]  the sub-codebooks pre-swizzled into MFMA A-fragment order
//                              (lane l of k-step kk holds C_i[t*32 + (l&31)][2kk + (l>>5)]),
//                              128 KiB at SIFT (m=8,h=256,sub=16), 96 KiB at Deep (m=16,sub=6)
//         saL [m][NT][2][16]   |c_k|^2 in C/D-fragment order (+inf for padded centroids)
//         xs  [wave][2KS][33]  the wave's 32 x sub slice of X, transposed, bank-padded
//   wave  owns tiles of 32 vectors; per subspace: stage slice -> B fragments in VGPRs ->
//         NT x KS MFMAs (32 centroids x 32 vectors x 2 dims each) -> fused epilogue
//         (add, sub, clamp, compare/select) on the 16 accumulator registers -> one cross-half
//         shuffle -> packed code bytes, written once per tile (m contiguous bytes per vector).
// Nothing of size h x n ever exists; X is read once, codes written once.
#include "rq_internal.h"
#include "rq_encode_split.h"

namespace rq {


__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }


constexpr int XS_STRIDE = 33;

// NT = number of 32-centroid tiles (compile time so the whole subspace is straight-line code and
// the scheduler can slide tile t's epilogue under tile t+1's MFMA chain); KS = k-steps.
template <int KS, int NT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void encode_pq_kernel(EncParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int m = p.m, h = p.h, d = p.d;
  const int i0 = p.i0, mg = p.i1 - p.i0;                        // this launch's group of sub-quantizers
  float *cbA = reinterpret_cast<float *>(smem);                 // mg*NT*KS*64
  float *saL = cbA + (size_t)mg * NT * KS * 64;                 // mg*NT*32
  float *xs_all = saL + (size_t)mg * NT * 32;                   // NWAVES * 2KS*33
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  float *xs = xs_all + (size_t)wave * (2 * KS * XS_STRIDE);

  // ---- prologue: codebooks -> A-fragment order, norms -> C/D-fragment order -------------------
  for (int idx = tid; idx < mg * NT * KS * 64; idx += NWAVES * 64) {
    const int l = idx & 63;
    int rest = idx >> 6;
    const int kk = rest % KS; rest /= KS;
    const int t = rest % NT;
    const int i = i0 + rest / NT;
    const int sub = p.off[i + 1] - p.off[i];
    const int cen = t * 32 + (l & 31);
    const int s = 2 * kk + (l >> 5);
    float v = 0.0f;
    if (cen < h && s < sub) v = p.C[(size_t)h * p.off[i] + (size_t)cen * sub + s];
    cbA[idx] = v;
  }
  for (int idx = tid; idx < mg * NT * 32; idx += NWAVES * 64) {
    const int c32 = idx & 31;
    const int t = (idx >> 5) % NT;
    const int il = (idx >> 5) / NT;
    const int i = i0 + il;
    const int sub = p.off[i + 1] - p.off[i];
    const int cen = t * 32 + c32;
    float sa = __uint_as_float(0x7f800000u);
    if (cen < h) {
      const float *c = p.C + (size_t)h * p.off[i] + (size_t)cen * sub;
      sa = 0.0f;
      for (int s = 0; s < sub; ++s) sa = __builtin_fmaf(c[s], c[s], sa);
    }
    const int hh = (c32 >> 2) & 1;
    const int r = (c32 & 3) + 4 * (c32 >> 3);
    saL[((size_t)(il * NT + t) * 2 + hh) * 16 + r] = sa;
  }
  __syncthreads();

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t total_waves = (int64_t)gridDim.x * NWAVES;
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave;

  // X slice of (tile, sub-quantizer) in flight in registers: element e = lane + 64u of the
  // 32 x sub slice, row-major -> coalesced runs of sub floats per row.  The (row, s) split of e
  // needs an integer division by sub; it is hoisted: recomputed only when sub changes (uneven
  // splits, src/utils.jl:179-203), i.e. never in the d % m == 0 case.
  float xr[KS];
  int xrow[KS], xs_off[KS];   // row of element u; its LDS offset s*33+row (or -1 past the slice)
  int cur_sub = -1;
  auto set_sub = [&](int sub) {
    if (sub == cur_sub) return;
    cur_sub = sub;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int e = lane + 64 * u;
      const int row = e / sub, sx = e - row * sub;
      xrow[u] = row * d + sx;                       // offset inside the tile, before + off_i
      xs_off[u] = (e < 32 * sub) ? sx * XS_STRIDE + row : -1;
    }
  };
  auto gload = [&](int64_t tile, int il) {
    const int i = i0 + il;
    const int o = p.off[i], sub = p.off[i + 1] - o;
    set_sub(sub);
    const int64_t row0 = tile * 32;
    const bool full = row0 + 32 <= p.n;
    const float *base = p.X + row0 * d + o;
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      float v = 0.0f;
      if (xs_off[u] >= 0) {
        if (full) v = base[xrow[u]];
        else {  // ragged last tile: clamp the row
          const int e = lane + 64 * u;
          int64_t gr = row0 + e / sub;
          if (gr >= p.n) gr = p.n - 1;
          v = p.X[gr * d + o + (e % sub)];
        }
      }
      xr[u] = v;
    }
  };
  if (tile0 < ntiles) gload(tile0, 0);

  for (int64_t tile = tile0; tile < ntiles; tile += total_waves) {
    const int64_t row0 = tile * 32;
    uint64_t cw[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int il = 0; il < mg; ++il) {
      const int i = i0 + il;
      const int sub = p.off[i + 1] - p.off[i];
      // registers -> LDS, transposed: xs[s][row]   (xs_off still describes THIS slice: gload for
      // the next one is issued below, after the write)
      wave_lds_sync();
#pragma unroll
      for (int u = 0; u < KS; ++u)
        if (xs_off[u] >= 0) xs[xs_off[u]] = xr[u];
      wave_lds_sync();
      // next slice's global loads fly while this one is computed
      if (il + 1 < mg) gload(tile, il + 1);
      else if (tile + total_waves < ntiles) gload(tile + total_waves, 0);
      float b[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const int sx = 2 * kk + hi;
        b[kk] = sx < sub ? xs[sx * XS_STRIDE + j] : 0.0f;
      }
      // |x|^2 over the subspace, chain s = 0..sub-1 (padded steps add fma(0,0,sb) == sb)
      float xv[2 * KS];
#pragma unroll
      for (int sx = 0; sx < 2 * KS; ++sx) xv[sx] = sx < sub ? xs[sx * XS_STRIDE + j] : 0.0f;
      float sb = 0.0f;
#pragma unroll
      for (int sx = 0; sx < 2 * KS; ++sx) sb = __builtin_fmaf(xv[sx], xv[sx], sb);
      ArgminState st;
      st.best_v = __uint_as_float(0x7f800000u);
      st.best_t = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) st.ub[r] = f32x2{0.0f, 0.0f};
      const float *cb_i = cbA + (size_t)il * NT * KS * 64 + lane;
      const float4 *sa_i = reinterpret_cast<const float4 *>(saL + ((size_t)il * NT * 2 + hi) * 16);
      // two named accumulators: tile t+1's MFMA chain is in flight while tile t is reduced
      f32x16 accA = tile_dots<KS>(cb_i, b), accB;
#pragma unroll
      for (int t = 0; t < NT; t += 2) {
        if (t + 1 < NT) accB = tile_dots<KS>(cb_i + (size_t)(t + 1) * KS * 64, b);
        tile_argmin(accA, sa_i + (size_t)t * 8, sb, t, st);
        if (t + 2 < NT) accA = tile_dots<KS>(cb_i + (size_t)(t + 2) * KS * 64, b);
        if (t + 1 < NT) tile_argmin(accB, sa_i + (size_t)(t + 1) * 8, sb, t + 1, st);
      }
      float best_v = st.best_v;
      int best_i = argmin_finish(st, hi);
      // the two half-waves hold disjoint centroid subsets of the same vector
      const float ov = __shfl_xor(best_v, 32);
      const int oi = __shfl_xor(best_i, 32);
      if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if ((i >> 3) == w) cw[w] |= (uint64_t)(uint32_t)best_i << (8 * (i & 7));
    }
    if (hi == 0 && row0 + j < p.n) {
      uint8_t *o = p.codes + (size_t)(row0 + j) * m;
      if ((m & 7) == 0 && mg == m) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
          if (w * 8 < m) reinterpret_cast<uint64_t *>(o)[w] = cw[w];
      } else {
        for (int i = i0; i < p.i1; ++i) o[i] = (uint8_t)(cw[i >> 3] >> (8 * (i & 7)));
      }
    }
  }
}

// Fast path: every sub-quantizer spans exactly sub = 2*KS dimensions (d % m == 0, sub even).  The
// 32 x sub slice of X never touches LDS: both half-lanes of vector j load its whole sub-vector straight
// into registers (sub*4 contiguous bytes; the two lanes of a pair hit the same addresses, which the
// memory pipe merges), take their B fragments by a per-lane select (even / odd dimension of each
// k-step) and run the |x|^2 chain locally.  LDS then holds only the codebooks and their norms, no
// intra-wave LDS hand-off (no s_waitcnt lgkmcnt drains) sits between two sub-quantizers, and the next
// sub-vector is in flight during the MFMAs.
template <int KS, int NT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void encode_pq_direct_kernel(EncParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SUB = 2 * KS;
  const int m = p.m, h = p.h, d = p.d;
  const int i0 = p.i0, mg = p.i1 - p.i0;
  float *cbA = reinterpret_cast<float *>(smem);                 // mg*NT*KS*64
  float *saL = cbA + (size_t)mg * NT * KS * 64;                 // mg*NT*32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  for (int idx = tid; idx < mg * NT * KS * 64; idx += NWAVES * 64) {
    const int l = idx & 63;
    int rest = idx >> 6;
    const int kk = rest % KS; rest /= KS;
    const int t = rest % NT;
    const int i = i0 + rest / NT;
    const int cen = t * 32 + (l & 31);
    const int sx = 2 * kk + (l >> 5);
    cbA[idx] = cen < h ? p.C[(size_t)h * SUB * i + (size_t)cen * SUB + sx] : 0.0f;
  }
  for (int idx = tid; idx < mg * NT * 32; idx += NWAVES * 64) {
    const int c32 = idx & 31;
    const int t = (idx >> 5) % NT;
    const int il = (idx >> 5) / NT;
    const int cen = t * 32 + c32;
    float sa = __uint_as_float(0x7f800000u);
    if (cen < h) {
      const float *c = p.C + (size_t)h * SUB * (i0 + il) + (size_t)cen * SUB;
      sa = 0.0f;
#pragma unroll
      for (int sx = 0; sx < SUB; ++sx) sa = __builtin_fmaf(c[sx], c[sx], sa);
    }
    const int hh = (c32 >> 2) & 1;
    const int r = (c32 & 3) + 4 * (c32 >> 3);
    saL[((size_t)(il * NT + t) * 2 + hh) * 16 + r] = sa;
  }
  __syncthreads();

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t total_waves = (int64_t)gridDim.x * NWAVES;
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave;
  f32x2 xn[KS];   // the sub-vector in flight (next (tile, sub-quantizer))
  auto gload = [&](int64_t tile, int il) {
    int64_t gr = tile * 32 + j;
    if (gr >= p.n) gr = p.n - 1;
    const f32x2 *src = reinterpret_cast<const f32x2 *>(p.X + gr * d + (size_t)(i0 + il) * SUB);
#pragma unroll
    for (int u = 0; u < KS; ++u) xn[u] = src[u];
  };
  // the next sub-vector is prefetched during the MFMAs, except for full-dimensional sub-spaces
  // (KS = 64: the prefetch alone would be 128 registers; two wavefronts per SIMD cover the load)
  constexpr bool PF = KS < 64;
  if (PF && tile0 < ntiles) gload(tile0, 0);

  for (int64_t tile = tile0; tile < ntiles; tile += total_waves) {
    const int64_t row0 = tile * 32;
    uint64_t cw[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int il = 0; il < mg; ++il) {
      const int i = i0 + il;
      if constexpr (!PF) gload(tile, il);
      float b[KS];
      float sb = 0.0f;
#pragma unroll
      for (int u = 0; u < KS; ++u) {
        const f32x2 v = xn[u];
        b[u] = hi ? v.y : v.x;                 // k-step u: lanes 0-31 the even, 32-63 the odd dimension
        sb = __builtin_fmaf(v.x, v.x, sb);     // chain s = 0..sub-1 in order
        sb = __builtin_fmaf(v.y, v.y, sb);
      }
      if constexpr (PF) {
        if (il + 1 < mg) gload(tile, il + 1);
        else if (tile + total_waves < ntiles) gload(tile + total_waves, 0);
      }
      ArgminState st;
      st.best_v = __uint_as_float(0x7f800000u);
      st.best_t = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) st.ub[r] = f32x2{0.0f, 0.0f};
      const float *cb_i = cbA + (size_t)il * NT * KS * 64 + lane;
      const float4 *sa_i = reinterpret_cast<const float4 *>(saL + ((size_t)il * NT * 2 + hi) * 16);
      f32x16 accA = tile_dots<KS>(cb_i, b), accB;
#pragma unroll
      for (int t = 0; t < NT; t += 2) {
        if (t + 1 < NT) accB = tile_dots<KS>(cb_i + (size_t)(t + 1) * KS * 64, b);
        tile_argmin(accA, sa_i + (size_t)t * 8, sb, t, st);
        if (t + 2 < NT) accA = tile_dots<KS>(cb_i + (size_t)(t + 2) * KS * 64, b);
        if (t + 1 < NT) tile_argmin(accB, sa_i + (size_t)(t + 1) * 8, sb, t + 1, st);
      }
      float best_v = st.best_v;
      int best_i = argmin_finish(st, hi);
      const float ov = __shfl_xor(best_v, 32);
      const int oi = __shfl_xor(best_i, 32);
      if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if ((i >> 3) == w) cw[w] |= (uint64_t)(uint32_t)best_i << (8 * (i & 7));
    }
    if (hi == 0 && row0 + j < p.n) {
      uint8_t *o = p.codes + (size_t)(row0 + j) * m;
      if ((m & 7) == 0 && mg == m) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
          if (w * 8 < m) reinterpret_cast<uint64_t *>(o)[w] = cw[w];
      } else {
        for (int i = i0; i < p.i1; ++i) o[i] = (uint8_t)(cw[i >> 3] >> (8 * (i & 7)));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// encode_pq_split_kernel -- the same codes, bit for bit, at several times the f32-MFMA rate.
//
// The direct kernel above is bound by the f32 matrix rate (64 cycles per 32x32x2 step) plus a VALU epilogue that cannot
// overlap it.  But the ARGMIN does not need every distance in canonical arithmetic -- only the winner (and whatever
// could tie with it) does.  So, per (32 vectors, sub-quantizer):
//
//  1. FILTER on the bf16 matrix cores (16x the f32 rate).  x and -2c are split into bf16 pieces, x = xh + xl (+ <= 2^-16|x|),
//     and   W_k = |c_k|^2 - 2 <c_k, x>  ~  sa_k + sum_s (-2c)h xh + (-2c)l xh + (-2c)h xl
//     comes out of 3 (sub <= 8: 2) v_mfma_f32_32x32x16_bf16 per 32-centroid tile, the accumulator pre-loaded with the
//     f32 norms sa_k (so the whole "sa - 2g" is MFMA work; |x|^2 is constant per vector and plays no part in the order).
//     Error analysis (sub <= 16; S = sum_s |x_s c_ks| <= (sa_k + sb)/2, sb = |x|^2):
//        dropped terms  2 (xl cl + xh ec + xl ec + ex c)   <= 6.2 * 2^-16 S        <= 2^-14.4 (sa_k + sb)
//        f32 accumulation of <= 49 terms inside the MFMA    <= 49 * 2^-23 (sa + 2S) <= 2^-16.3 (sa_k + sb)
//        canonical u_k (fmaf chains, two roundings) vs the real value               <= 2^-18.7 (sa_k + sb)
//     so |(W_k + sb) - u_k| <= e_k := 1.02 * 2^-14 (sa_k + sb), clamp included (a clamped u_k has real value <= its own
//     error).  (The three terms add up to 1.015 * 2^-14; the accumulation term assumes <= 1 ulp per add inside the MFMA.
//     MEASURED on the kernel's own W values, tests/test_gpu_encode_margin.py: <= 0.46 * 2^-14 on hostile inputs, 0.05 / 0.36
//     on the 2e9 / 4e9 values of the SIFT / Deep bench shapes.)
//  2. If k* is the canonical argmin then W_k* <= W_k + e_k* + e_k for every k, so k* -- and every k that ties with it --
//     satisfies  W_k <= min_k W + DELTA,  DELTA = 3 * 2^-14 (max_k sa_k + sb)   (needed: 2 e_k = 2.04 * 2^-14; the rest covers
//     the f32 roundings of the threshold itself.  Walking DELTA down -- tuning ENC_SPLIT_DELTA_MILLI -- codes first change
//     at 0.023 / 0.047 * 2^-14 on the bench shapes: 130x / 64x below the shipped value).  Per lane the kernel keeps the running minimum b1, a copy of the 16 W values of the tile
//     that holds it, and a bit mask of the tiles that came within DELTA of the running minimum at the time (a superset of
//     the tiles within DELTA of the final one).
//  3. REFINE: the candidates {W_k <= b1 + DELTA} -- 1.02 per vector on SIFT-like, 1.005 on Deep-like data -- get the
//     canonical evaluation (oracle/rq_oracle.c: fmaf chains s = 0..sub-1 from +0, v = max(fl(fl(sa + sb) - 2g), 0)) on the
//     VALU, and the smallest (v, index) pair wins: exactly the first-index argmin of the reference.  Tiles flagged in the
//     mask are re-run through the matrix cores at the end (their W values are not kept), lanes whose bound is not usable
//     (non-finite or vanishing norms) evaluate all their centroids exactly.
// Nothing approximate reaches the output: the filter only decides WHICH centroids get the exact evaluation.
// ------------------------------------------------------------------------------------------
// DBG: the instantiation that stores the filter's W values (rq_dev_encode_pq_filter_w, tests only); the product kernel carries
// none of that code (16 lane masks and a 64-bit address per tile)
template <int SUB, int NT, int NWAVES, bool DBG = false>
__global__ __launch_bounds__(NWAVES * 64) void encode_pq_split_kernel(EncParams p) {
  using Shape = SplitShape<SUB>;
  constexpr bool PACK = Shape::PACK;
  constexpr int NPIECE = Shape::NPIECE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int m = p.m, h = p.h, d = p.d;
  const int i0 = p.i0, mg = p.i1 - p.i0;
  uint4 *cbA = reinterpret_cast<uint4 *>(smem);                                   // [mg][NT][NPIECE][64] bf16x8 A fragments of -2c
  float *saL = reinterpret_cast<float *>(cbA + (size_t)mg * NT * NPIECE * 64);    // [mg][NT][2][16] |c|^2, C/D-fragment order
  float *saMax = saL + (size_t)mg * NT * 32;                                      // [mg] max_k |c_k|^2 (finite entries)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;

  // ---- prologue: -2c split into bf16 pieces, in the A-fragment order of v_mfma_f32_32x32x16_bf16 (lane l: centroid
  // l & 31, K elements 8 (l >> 5) .. + 7).  PACK: K elements 0-7 = hi pieces, 8-15 = lo pieces of dimensions 0-7. ----
  for (int idx = tid; idx < mg * NT * NPIECE * 64; idx += NWAVES * 64) {
    const int l = idx & 63;
    int rest = idx >> 6;
    const int piece = rest % NPIECE; rest /= NPIECE;
    const int t = rest % NT;
    const int il = rest / NT;
    const int cen = t * 32 + (l & 31);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sx = PACK ? e : 8 * (l >> 5) + e;            // dimension of this K element
      const bool lo = PACK ? (l >> 5) != 0 : piece != 0;      // which bf16 piece
      uint32_t bits = 0;
      if (cen < h && sx < SUB) {
        const float v = -2.0f * p.C[((size_t)(i0 + il) * h + cen) * SUB + sx];
        const uint32_t hb = bf16_bits(v);
        bits = lo ? bf16_bits(v - bf16_val(hb)) : hb;
      }
      w[e >> 1] |= bits << (16 * (e & 1));
    }
    cbA[idx] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  for (int idx = tid; idx < mg * NT * 32; idx += NWAVES * 64) {
    const int c32 = idx & 31;
    const int t = (idx >> 5) % NT;
    const int il = (idx >> 5) / NT;
    const int cen = t * 32 + c32;
    float sa = __uint_as_float(0x7f800000u);
    if (cen < h) {
      const float *c = p.C + ((size_t)(i0 + il) * h + cen) * SUB;
      sa = 0.0f;
#pragma unroll
      for (int sx = 0; sx < SUB; ++sx) sa = __builtin_fmaf(c[sx], c[sx], sa);
    }
    const int hh = (c32 >> 2) & 1;
    const int r = (c32 & 3) + 4 * (c32 >> 3);
    saL[((size_t)(il * NT + t) * 2 + hh) * 16 + r] = sa;
  }
  __syncthreads();
  for (int il = wave; il < mg; il += NWAVES) {       // max of the real centroids' norms, one wavefront per sub-quantizer
    float mx = 0.0f;
    for (int e = lane; e < NT * 32; e += 64) {
      const float v = saL[(size_t)il * NT * 32 + e];
      const int t = e >> 5, hh = (e >> 4) & 1, r = e & 15;
      const int cen = t * 32 + 4 * hh + 8 * (r >> 2) + (r & 3);
      if (cen < h) mx = __builtin_fmaxf(mx, v) + (v != v ? v : 0.0f);    // a NaN norm poisons the bound -> exact path
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_xor(mx, off);
      mx = (o != o || mx != mx) ? __uint_as_float(0x7fc00000u) : __builtin_fmaxf(mx, o);
    }
    if (lane == 0) saMax[il] = mx;
  }
  __syncthreads();

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t total_waves = (int64_t)gridDim.x * NWAVES;
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave;
  // The lane's 8 K elements of the sub-vector in flight (next (tile, sub-quantizer)): dimensions 8 hi .. 8 hi + 7 (PACK:
  // dimensions 0 .. 7 in both halves) -- 32 contiguous bytes per lane, two 16-byte loads where the rows allow it.  (Loading
  // the whole sub-vector 8 bytes at a time, as the f32 kernel does, costs 8 wave-wide loads of 32 cache lines each per
  // sub-quantizer; at this kernel's pace the texture-address unit became the bound: 0.47 ms for loads + MFMAs alone.)
  constexpr int NPAIR = PACK ? SUB / 2 : 4;         // f32x2 slots per lane
  const int my_pairs = PACK ? SUB / 2 : (hi ? (SUB - 8) / 2 : 4);
  const bool vec4 = (d % 4 == 0) && (((uintptr_t)p.X & 15) == 0) && (SUB % 4 == 0);
  f32x2 xn[NPAIR];
  auto gload = [&](int64_t tile, int il) {
    int64_t gr = tile * 32 + j;
    if (gr >= p.n) gr = p.n - 1;
    const float *src = p.X + gr * d + (size_t)(i0 + il) * SUB + (PACK ? 0 : 8 * hi);
    if (vec4) {
#pragma unroll
      for (int u = 0; u < NPAIR; u += 2) {
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (u < my_pairs) v = *reinterpret_cast<const float4 *>(src + 2 * u);
        xn[u] = f32x2{v.x, v.y};
        if (u + 1 < NPAIR) xn[u + 1] = f32x2{v.z, v.w};
      }
    } else {
#pragma unroll
      for (int u = 0; u < NPAIR; ++u) xn[u] = u < my_pairs ? *reinterpret_cast<const f32x2 *>(src + 2 * u) : f32x2{0.0f, 0.0f};
    }
  };
  if (tile0 < ntiles) gload(tile0, 0);

  for (int64_t tile = tile0; tile < ntiles; tile += total_waves) {
    const int64_t row0 = tile * 32;
    uint64_t cw[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int il = 0; il < mg; ++il) {
      const int i = i0 + il;
      // |x|^2 (any summation order will do here: it only scales the filter's margin) and the B fragments: this lane's
      // 8 K elements of x, split into bf16 pieces (v_cvt_pk_bf16_f32 rounds to nearest even)
      f32x2 sel[4];
      f32x2 sq = {0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < NPAIR; ++u) sq = __builtin_elementwise_fma(xn[u], xn[u], sq);
#pragma unroll
      for (int u = 0; u < 4; ++u) sel[u] = u < NPAIR ? xn[u < NPAIR ? u : 0] : f32x2{0.0f, 0.0f};
      uint32_t bh[4], bl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#ifndef RQ_SPLIT_SWCVT
        const bf16x2_t hb = __builtin_convertvector(sel[u], bf16x2_t);
        const f32x2 rest = sel[u] - __builtin_convertvector(hb, f32x2);
        bh[u] = __builtin_bit_cast(uint32_t, hb);
        bl[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, bf16x2_t));
#else
        const uint32_t h0 = bf16_bits(sel[u].x), h1 = bf16_bits(sel[u].y);
        bh[u] = h0 | (h1 << 16);
        bl[u] = bf16_bits(sel[u].x - bf16_val(h0)) | (bf16_bits(sel[u].y - bf16_val(h1)) << 16);
#endif
      }
      if (PACK && hi) { bl[0] = bl[1] = bl[2] = bl[3] = 0; }                // second MFMA: [xl | 0] against [ch | cl]
      const bf16x8_t Bh = __builtin_bit_cast(bf16x8_t, make_uint4(bh[0], bh[1], bh[2], bh[3]));
      const bf16x8_t Bl = __builtin_bit_cast(bf16x8_t, make_uint4(bl[0], bl[1], bl[2], bl[3]));
      float sb = sq.x + sq.y;
      if constexpr (!PACK) sb += __shfl_xor(sb, 32);       // the other half-wave holds dimensions 8..15
      f32x2 xc[NPAIR];                                     // this lane's share of the sub-vector, kept for the refine step
#pragma unroll
      for (int u = 0; u < NPAIR; ++u) xc[u] = xn[u];
      // the next sub-vector travels while this one is filtered
      if (il + 1 < mg) gload(tile, il + 1);
      else if (tile + total_waves < ntiles) gload(tile + total_waves, 0);
      const float smax = saMax[il];
      const float ssum = smax + sb;
      const float delta = p.delta_rel * ssum;
      // the bound needs finite, non-vanishing magnitudes; otherwise this lane evaluates all its centroids exactly
      const bool slow = !(delta < __uint_as_float(0x7f800000u)) || !(ssum >= SplitCfg::TINY);

      float b1 = __uint_as_float(0x7f800000u);   // running minimum of W over this lane's centroids
      uint32_t t1bit = 1u;                       // 1 << (tile that holds it)
      uint32_t extra = 0u;                       // tiles that came within delta of the running minimum
      f32x16 ub;                                 // the 16 W values of the best tile
#pragma unroll
      for (int r = 0; r < 16; ++r) ub[r] = 0.0f;
      const uint4 *cb_i = cbA + (size_t)il * NT * NPIECE * 64 + lane;
      const float4 *sa_i = reinterpret_cast<const float4 *>(saL + ((size_t)il * NT * 2 + hi) * 16);
      auto tile_w = [&](int t) -> f32x16 {
        const float4 *s4 = sa_i + (size_t)t * 8;
        f32x16 acc;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 v = s4[g4];
          acc[g4 * 4 + 0] = v.x; acc[g4 * 4 + 1] = v.y; acc[g4 * 4 + 2] = v.z; acc[g4 * 4 + 3] = v.w;
        }
#if !defined(RQ_SPLIT_ABL) || RQ_SPLIT_ABL != 3
        const bf16x8_t A0 = __builtin_bit_cast(bf16x8_t, cb_i[(size_t)t * NPIECE * 64]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bl, acc, 0, 0, 0);
        if constexpr (!PACK) {
          const bf16x8_t A1 = __builtin_bit_cast(bf16x8_t, cb_i[(size_t)t * NPIECE * 64 + 64]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bh, acc, 0, 0, 0);
        }
#else
        acc[0] += __uint_as_float(bh[0] ^ bl[1]);
#endif
        return acc;
      };
      auto tile_min = [&](const f32x16 &a) -> float {
        float m1 = __builtin_fminf(__builtin_fminf(a[0], a[1]), a[2]);
        float m2 = __builtin_fminf(__builtin_fminf(a[3], a[4]), a[5]);
        float m3 = __builtin_fminf(__builtin_fminf(a[6], a[7]), a[8]);
        float m4 = __builtin_fminf(__builtin_fminf(a[9], a[10]), a[11]);
        float m5 = __builtin_fminf(__builtin_fminf(a[12], a[13]), a[14]);
        float mm = __builtin_fminf(__builtin_fminf(m1, m2), m3);
        mm = __builtin_fminf(__builtin_fminf(mm, m4), m5);
        return __builtin_fminf(mm, a[15]);
      };
      auto tile_filter = [&](const f32x16 &a, int t) {
        if constexpr (DBG) {    // tests/test_gpu_encode_margin.py: the very values the filter decides on
          if (row0 + j < p.n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int k = t * 32 + 4 * hi + 8 * (r >> 2) + (r & 3);
              if (k < h) p.dbg_w[((size_t)(row0 + j) * m + i) * h + k] = a[r];
            }
          }
        }
#if defined(RQ_SPLIT_ABL) && RQ_SPLIT_ABL == 4
        b1 = __builtin_fminf(b1, a[t & 15]); return;
#endif
        const float mm = tile_min(a);
        const bool imp = mm < b1;
        const bool near = __builtin_fabsf(mm - b1) <= delta;       // (+inf - x: false; NaN: false)
        extra |= near ? (imp ? t1bit : (1u << t)) : 0u;
        if (imp) {
          b1 = mm;
          t1bit = 1u << t;
          // eight 64-bit moves under the exec mask.  (Written as `ub = a` the compiler if-converts the copy into 16-18
          // v_cndmask per tile -- a third of the loop's VALU time.  The asm reads accumulator registers that tile_min's
          // compiler-scheduled reads have already waited for, so no MFMA hazard is left open here.)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            f32x2 src = {a[r], a[r + 1]}, dst;
            asm volatile("v_mov_b64 %0, %1" : "=v"(dst) : "v"(src));
            ub[r] = dst.x; ub[r + 1] = dst.y;
          }
        }
      };
      if constexpr (NWAVES >= 16 && RQ_SPLIT_SINGLE_ACC) {
        // four wavefronts per SIMD cover each other's MFMA latency: one accumulator set (128-register budget)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x16 acc = tile_w(t);
          tile_filter(acc, t);
        }
      } else {
        f32x16 accA = tile_w(0), accB;
#pragma unroll
        for (int t = 0; t < NT; t += 2) {
          if (t + 1 < NT) accB = tile_w(t + 1);
          tile_filter(accA, t);
          if (t + 2 < NT) accA = tile_w(t + 2);
          if (t + 1 < NT) tile_filter(accB, t + 1);
        }
      }

      // ---- refine: canonical evaluation of the candidates --------------------------------------------------------
      const float *cb = p.C + (size_t)i * h * SUB;
      float bv = __uint_as_float(0x7f800000u);
      int bk = 0;
      float bo_a = b1, bo_b = b1;                                  // the other half of this vector's centroids
      swap32(bo_a, bo_b);
      const float bo = hi ? bo_a : bo_b;
      const float thr = __builtin_fminf(b1, bo) + delta;
      const bool contend = b1 <= thr;
      const int cbase = 4 * hi;                                    // centroid of (tile t, register r): 32 t + 4 hi + 8 (r >> 2) + (r & 3)
#if defined(RQ_SPLIT_ABL) && RQ_SPLIT_ABL >= 2
      uint32_t cm = 0;
      bk = 31 - __builtin_clz(t1bit); bv = b1;
#else
      uint32_t cm = mask_leq16(ub, thr);
#endif
      if (!contend || slow) cm = 0;
      const int t1 = 31 - __builtin_clz(t1bit);
#if defined(RQ_SPLIT_ABL) && RQ_SPLIT_ABL >= 1
      if (cm) { bk = t1 * 32 + __builtin_ctz(cm); bv = b1; }
      cm = 0; extra = 0;
#endif
      // A vector with exactly ONE candidate (and no flagged tile, no unusable bound) in its two half-waves together is
      // done: the candidate set provably contains the canonical argmin, so its only element IS the argmin -- no exact
      // evaluation needed (98 % of the vectors at SIFT shape; half of the wavefronts skip the evaluation code entirely).
      uint32_t todo = (contend && !slow) ? (extra & ~t1bit) : 0u;
      const uint32_t mine_w = (uint32_t)__builtin_popcount(cm) | (todo != 0u ? 0x100u : 0u) | (slow ? 0x200u : 0u);
      float ow_a = __uint_as_float(mine_w), ow_b = __uint_as_float(mine_w);
      swap32(ow_a, ow_b);
      const uint32_t both_w = mine_w + __float_as_uint(hi ? ow_a : ow_b);       // candidate counts add; a flag in either half shows
      const bool single = both_w == 1u;
      if (single) {
        if (cm != 0u) { bk = t1 * 32 + cbase + 8 * (__builtin_ctz(cm) >> 2) + (__builtin_ctz(cm) & 3); bv = 0.0f; }
        cm = 0u;                // (the half without the candidate keeps bv = +inf and loses the merge below)
      }
      if (__ballot(!single)) {
      // the whole sub-vector in every lane: the other half-wave's 8 dimensions come over by v_permlane32_swap; then
      // the canonical |x|^2 (chain s = 0..sub-1 from +0)
      float x[SUB];
      if constexpr (PACK) {
#pragma unroll
        for (int u = 0; u < SUB / 2; ++u) { x[2 * u] = xc[u].x; x[2 * u + 1] = xc[u].y; }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float lo0 = xc[u].x, hi0 = xc[u].x, lo1 = xc[u].y, hi1 = xc[u].y;
          swap32(lo0, hi0);       // lo0: dimension 2u of the vector in all lanes; hi0: dimension 8 + 2u
          swap32(lo1, hi1);
          x[2 * u] = lo0; x[2 * u + 1] = lo1;
          if (8 + 2 * u < SUB) { x[8 + 2 * u] = hi0; x[8 + 2 * u + 1] = hi1; }
        }
      }
      float sbx = 0.0f;
#pragma unroll
      for (int sx = 0; sx < SUB; ++sx) sbx = __builtin_fmaf(x[sx], x[sx], sbx);
      // |c_k|^2 of centroid k from the C/D-ordered norm table of the prologue
      const float *sa_il = saL + (size_t)il * NT * 32;
      auto sa_of = [&](int k) -> float {
        const int c32 = k & 31;
        return sa_il[((k >> 5) * 2 + ((c32 >> 2) & 1)) * 16 + (c32 & 3) + 4 * (c32 >> 3)];
      };
      while (__ballot(cm != 0u)) {
        if (cm != 0u) {
          const int r = __builtin_ctz(cm);
          cm &= cm - 1u;
          const int k = t1 * 32 + cbase + 8 * (r >> 2) + (r & 3);
          if (k < h) split_exact<SUB>(cb, k, x, sbx, sa_of(k), bv, bk);
        }
      }
      // tiles that came within delta of the running minimum: their W values were not kept -- once more through the
      // matrix cores, wave-uniform tile by tile (the best tile itself is covered by the copy above)
      if (__ballot(todo != 0u)) {                                  // (no lane of the wavefront in ~60 % of the cases)
#pragma unroll 1
      for (int T = 0; T < NT; ++T) {
        if (!__ballot((todo >> T) & 1u)) continue;
        const f32x16 acc = tile_w(T);
        // (plain C here: the hazard recogniser does not look inside inline asm, and an asm compare issued straight after
        // the MFMA read stale accumulator registers now and then -- 2 to 12 wrong codes per 1e7, different ones every run)
        uint32_t cm2 = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) cm2 |= (acc[r] <= thr) ? (1u << r) : 0u;
        if (!((todo >> T) & 1u)) cm2 = 0;
        todo &= ~(1u << T);
        while (__ballot(cm2 != 0u)) {
          if (cm2 != 0u) {
            const int r = __builtin_ctz(cm2);
            cm2 &= cm2 - 1u;
            const int k = T * 32 + cbase + 8 * (r >> 2) + (r & 3);
            if (k < h) split_exact<SUB>(cb, k, x, sbx, sa_of(k), bv, bk);
          }
        }
      }
      }
      // lanes without a usable bound: every centroid of this lane, exactly
      if (__ballot(slow)) {
        if (slow) {
#pragma unroll 1
          for (int kk = 0; kk < NT * 16; ++kk) {
            const int k = (kk >> 4) * 32 + cbase + 8 * ((kk & 15) >> 2) + (kk & 3);
            if (k < h) split_exact<SUB>(cb, k, x, sbx, sa_of(k), bv, bk);
          }
        }
      }
      }
      // the two half-waves hold disjoint centroid subsets of the same vector
      float ov_a = bv, ov_b = bv, ok_a = __int_as_float(bk), ok_b = __int_as_float(bk);
      swap32(ov_a, ov_b);
      swap32(ok_a, ok_b);
      const float ov = hi ? ov_a : ov_b;
      const int ok = __float_as_int(hi ? ok_a : ok_b);
      if (ov < bv || (ov == bv && ok < bk)) { bv = ov; bk = ok; }
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if ((i >> 3) == w) cw[w] |= (uint64_t)(uint32_t)bk << (8 * (i & 7));
    }
    if (hi == 0 && row0 + j < p.n) {
      uint8_t *o = p.codes + (size_t)(row0 + j) * m;
      if ((m & 7) == 0 && mg == m) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
          if (w * 8 < m) reinterpret_cast<uint64_t *>(o)[w] = cw[w];
      } else {
        for (int i = i0; i < p.i1; ++i) o[i] = (uint8_t)(cw[i >> 3] >> (8 * (i & 7)));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Wide sub-spaces (sub > 64 that are not exactly 96 / 128 wide, or whose codebook does not fit LDS:
// PQ on GIST-960 / MNIST-784, RVQ / k-means assignment at any d).  The work is one flat sequence of
// chunks (tile group, sub-quantizer, KC k-steps of the sub-space).  While the MFMAs of a chunk run,
// the NEXT chunk of the sequence -- the next slice of the same sub-space, the first slice of the next
// sub-quantizer, or of the next tile group -- is in flight: its codebook slice (NT x KC x 64 floats,
// A-fragment order) goes to the other half of a double-buffered LDS area, its X slice (coalesced
// loads) to the wavefront's own staging tile.  Every wavefront keeps the NT accumulator tiles of its
// 32 vectors across the chunks of a sub-space, so the MFMA chain of a (centroid, vector) pair is still
// s = 0..sub-1 in order, i.e. the oracle's fmaf chain (padded dimensions multiply 0 by 0).
// ------------------------------------------------------------------------------------------
template <int NT>
struct WideCfg {
  static constexpr int KC = NT >= 8 ? 8 : NT >= 4 ? 16 : 32;   // k-steps per chunk: 128 accumulator registers
  static constexpr int NW = 8;                                  // at NT = 8 leave room for 8-step staging only
};

template <int NT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void encode_wide_kernel(EncParams p) {
  constexpr int KC = WideCfg<NT>::KC;
  constexpr int CHUNK = NT * KC * 64;                  // floats per staged codebook chunk
  constexpr int PER = CHUNK / (NWAVES * 64);           // ... per thread
  constexpr int DIMS = 2 * KC;                         // dimensions per chunk
  constexpr int RPI = 64 / DIMS;                       // rows one wave-wide X load covers
  constexpr int XL = 32 / RPI;                         // X loads per lane per chunk
  constexpr int XSTR = DIMS + 1;                       // padded row stride of the staged X tile
  static_assert(CHUNK % (NWAVES * 64) == 0 && 64 % DIMS == 0, "chunk split");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *cb0 = reinterpret_cast<float *>(smem);       // [2][CHUNK]
  float *xs_all = cb0 + 2 * CHUNK;                     // [NWAVES][32][XSTR]: each wave's X chunk, row-major
  float *saL = xs_all + NWAVES * 32 * XSTR;            // [m][NT*32], C/D-fragment order
  const int m = p.m, h = p.h, d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  float *xs = xs_all + wave * 32 * XSTR;

  for (int idx = tid; idx < m * NT * 32; idx += NWAVES * 64) {
    const int c32 = idx & 31;
    const int t = (idx >> 5) % NT;
    const int i = (idx >> 5) / NT;
    const int sub = p.off[i + 1] - p.off[i];
    const int cen = t * 32 + c32;
    float sa = __uint_as_float(0x7f800000u);
    if (cen < h) {
      const float *c = p.C + (size_t)h * p.off[i] + (size_t)cen * sub;
      sa = 0.0f;
      for (int s = 0; s < sub; ++s) sa = __builtin_fmaf(c[s], c[s], sa);
    }
    const int hh = (c32 >> 2) & 1;
    const int r = (c32 & 3) + 4 * (c32 >> 3);
    saL[((size_t)(i * NT + t) * 2 + hh) * 16 + r] = sa;
  }

  // element `e` of a staged chunk -> (tile t, k-step kk, lane l): centroid t*32 + (l & 31), dimension 2kk + (l >> 5)
  auto cb_load = [&](int i, int c, int e) -> float {
    const int l = e & 63;
    const int kk = (e >> 6) % KC;
    const int t = (e >> 6) / KC;
    const int sub = p.off[i + 1] - p.off[i];
    const int cen = t * 32 + (l & 31);
    const int s = c * DIMS + 2 * kk + (l >> 5);
    return (cen < h && s < sub) ? p.C[(size_t)h * p.off[i] + (size_t)cen * sub + s] : 0.0f;
  };
  // X slice of (tile group tg, sub-quantizer i, chunk c): lane -> (row u*RPI + lane/DIMS, dimension lane%DIMS);
  // rows past the end repeat the last one (their codes are never written)
  const int xdim = lane % DIMS, xrow_in = lane / DIMS;
  auto x_load = [&](int64_t tg, int i, int c, int u) -> float {
    const int sub = p.off[i + 1] - p.off[i];
    int64_t gr = (tg * NWAVES + wave) * 32 + u * RPI + xrow_in;
    if (gr >= p.n) gr = p.n - 1;
    const int sdim = c * DIMS + xdim;
    return sdim < sub ? p.X[gr * d + p.off[i] + sdim] : 0.0f;
  };

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t ngroups = (ntiles + NWAVES - 1) / NWAVES;
  int64_t tg = blockIdx.x;
  int i = 0, c = 0;
  if (tg >= ngroups) return;
#pragma unroll
  for (int u = 0; u < PER; ++u) cb0[tid + u * NWAVES * 64] = cb_load(0, 0, tid + u * NWAVES * 64);
#pragma unroll
  for (int u = 0; u < XL; ++u) xs[(u * RPI + xrow_in) * XSTR + xdim] = x_load(tg, 0, 0, u);
  __syncthreads();

  f32x16 acc[NT];
  float sb = 0.0f;
  int buf = 0;
#pragma unroll 1
  for (;;) {
    const int sub = p.off[i + 1] - p.off[i];
    const int nchunks = (sub + DIMS - 1) / DIMS;
    // the position after this one in the flat sequence
    int64_t ntg = tg;
    int ni = i, nc = c + 1;
    if (nc == nchunks) { nc = 0; ++ni; if (ni == m) { ni = 0; ntg += gridDim.x; } }
    const bool more = ntg < ngroups;
    if (c == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
      sb = 0.0f;
    }
    float nxt[PER], xn[XL];
    if (more) {
#pragma unroll
      for (int u = 0; u < PER; ++u) nxt[u] = cb_load(ni, nc, tid + u * NWAVES * 64);
#pragma unroll
      for (int u = 0; u < XL; ++u) xn[u] = x_load(ntg, ni, nc, u);
    }
    const float *cb = cb0 + (size_t)buf * CHUNK + lane;
    float b[KC];
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) {
      const float x0 = xs[j * XSTR + 2 * kk], x1 = xs[j * XSTR + 2 * kk + 1];
      b[kk] = hi ? x1 : x0;
      sb = __builtin_fmaf(x0, x0, sb);
      sb = __builtin_fmaf(x1, x1, sb);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int kk = 0; kk < KC; ++kk)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cb[(t * KC + kk) * 64], b[kk], acc[t], 0, 0, 0);
    }
    if (c + 1 == nchunks) {   // the sub-space is complete: argmin over the NT tiles
      ArgminState st;
      st.best_v = __uint_as_float(0x7f800000u);
      st.best_t = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) st.ub[r] = f32x2{0.0f, 0.0f};
      const float4 *sa_i = reinterpret_cast<const float4 *>(saL + ((size_t)i * NT * 2 + hi) * 16);
#pragma unroll
      for (int t = 0; t < NT; ++t) tile_argmin(acc[t], sa_i + (size_t)t * 8, sb, t, st);
      float best_v = st.best_v;
      int best_i = argmin_finish(st, hi);
      const float ov = __shfl_xor(best_v, 32);
      const int oi = __shfl_xor(best_i, 32);
      if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
      const int64_t row = (tg * NWAVES + wave) * 32 + j;
      if (hi == 0 && row < p.n) p.codes[(size_t)row * m + i] = (uint8_t)best_i;
    }
    if (!more) break;
    {
      float *dstb = cb0 + (size_t)(buf ^ 1) * CHUNK;
#pragma unroll
      for (int u = 0; u < PER; ++u) dstb[tid + u * NWAVES * 64] = nxt[u];
#pragma unroll
      for (int u = 0; u < XL; ++u) xs[(u * RPI + xrow_in) * XSTR + xdim] = xn[u];   // this wave's own tile
    }
    __syncthreads();
    buf ^= 1;
    tg = ntg; i = ni; c = nc;
  }
}

template <int NT>
static int launch_encode_wide(EncParams p, int num_cu, hipStream_t stream) {
  constexpr int NW = WideCfg<NT>::NW;
  constexpr int KC = WideCfg<NT>::KC;
  p.NT = NT;
  const size_t lds = (size_t)2 * NT * KC * 64 * sizeof(float) + (size_t)NW * 32 * (2 * KC + 1) * sizeof(float) +
                     (size_t)p.m * NT * 32 * sizeof(float);
  if (lds > 160 * 1024)
    return fail(RQ_EUNSUPPORTED, "wide encode: m=%d sub-quantizers of h=%d need %zu B of LDS", p.m, p.h, lds);
  auto kern = encode_wide_kernel<NT, NW>;
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds));
  const int64_t ngroups = ((p.n + 31) / 32 + NW - 1) / NW;
  const int grid = (int)std::min<int64_t>(num_cu, ngroups);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// ------------------------------------------------------------------------------------------
// Rotation  RX[j][i] = sum_k Rc[i][k] X[j][k]   (src/OPQ.jl:26, src/Linscan.jl:102)
// R sits in LDS in A-fragment order; each wave stages a 32-vector tile of X transposed in LDS
// and runs NT x KK 32x32x2 MFMAs; the k loop is the fmaf chain k = 0..d-1 of the oracle.
// ------------------------------------------------------------------------------------------
struct RotParams {
  const float *R;  // [d][d]  Rc[i][k]
  const float *X;  // [n][d]
  float *RX;       // [n][d]
  int64_t n;
  int d, NT, KK;
};

template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void rotate_kernel(RotParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int d = p.d, NT = p.NT, KK = p.KK;
  float *RA = reinterpret_cast<float *>(smem);            // NT*KK*64
  float *xs_all = RA + (size_t)NT * KK * 64;              // NWAVES * (2KK)*33
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  float *xs = xs_all + (size_t)wave * (2 * KK * XS_STRIDE);
  for (int idx = tid; idx < NT * KK * 64; idx += NWAVES * 64) {
    const int l = idx & 63;
    const int kk = (idx >> 6) % KK, t = (idx >> 6) / KK;
    const int i = t * 32 + (l & 31), k = 2 * kk + (l >> 5);
    RA[idx] = (i < d && k < d) ? p.R[(size_t)i * d + k] : 0.0f;
  }
  __syncthreads();
  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t total_waves = (int64_t)gridDim.x * NWAVES;
  for (int64_t tile = (int64_t)blockIdx.x * NWAVES + wave; tile < ntiles; tile += total_waves) {
    const int64_t row0 = tile * 32;
    wave_lds_sync();
    for (int e = lane; e < 32 * d; e += 64) {
      const int row = e / d, c = e - row * d;
      int64_t gr = row0 + row;
      if (gr >= p.n) gr = p.n - 1;
      xs[c * XS_STRIDE + row] = p.X[gr * d + c];
    }
    if (hi == 1 && (d & 1)) xs[d * XS_STRIDE + j] = 0.0f;  // odd d: zero the padded k column
    wave_lds_sync();
    const float *xb = xs + hi * XS_STRIDE + j;
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const float *ra = RA + (size_t)t * KK * 64 + lane;
#pragma unroll 4
      for (int kk = 0; kk < KK; ++kk) {
        const float a = ra[kk * 64];
        const float b = xb[2 * kk * XS_STRIDE];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      if (row0 + j < p.n) {
        float *o = p.RX + (size_t)(row0 + j) * d;
        const int ibase = t * 32 + 4 * hi;
        if ((d & 3) == 0) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int i0 = ibase + 8 * g4;
            if (i0 < d)
              *reinterpret_cast<float4 *>(o + i0) =
                  make_float4(acc[g4 * 4 + 0], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = ibase + 8 * (r >> 2) + (r & 3);
            if (i < d) o[i] = acc[r];
          }
        }
      }
    }
  }
}

// Rotation for any d (GIST-960, MNIST-784, ...): neither R nor a 32 x d tile of X fits the LDS, so both are
// chunked.  A wavefront owns 32 vectors and keeps TG = 8 output tiles (256 outputs, 128 accumulator registers)
// alive across the k chunks; per chunk of KC = 32 dimensions the workgroup stages the 256 x 32 panel of R in
// A-fragment order (32 KiB) and every wavefront its 32 x 32 slice of X (transposed, stride 33).  The chain of an
// output is still k = 0..d-1 in order (zero padding adds fma(0, x, acc) = acc), i.e. the oracle's fmaf chain.
// X is re-read ceil(d / 256) times; this path is about coverage, not the roofline (the BASELINE shapes take
// the register-resident kernels above).
template <int NWAVES, int TG, int KC>
__global__ __launch_bounds__(NWAVES * 64) void rotate_wide_kernel(RotParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KS = KC / 2;
  float *RA = reinterpret_cast<float *>(smem);                     // TG * KS * 64
  float *xs_all = RA + (size_t)TG * KS * 64;                        // NWAVES * KC * XS_STRIDE
  const int d = p.d, NT = p.NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  float *xs = xs_all + (size_t)wave * (KC * XS_STRIDE);
  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nbatches = (ntiles + NWAVES - 1) / NWAVES;
  for (int64_t batch = blockIdx.x; batch < nbatches; batch += gridDim.x) {
    const int64_t tile = batch * NWAVES + wave;
    const int64_t row0 = tile * 32;
    for (int g0 = 0; g0 < NT; g0 += TG) {
      f32x16 acc[TG];
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
      for (int k0 = 0; k0 < d; k0 += KC) {
        __syncthreads();
        for (int idx = tid; idx < TG * KS * 64; idx += NWAVES * 64) {
          const int l = idx & 63, kk = (idx >> 6) % KS, t = (idx >> 6) / KS;
          const int i = (g0 + t) * 32 + (l & 31), k = k0 + 2 * kk + (l >> 5);
          RA[idx] = (i < d && k < d) ? p.R[(size_t)i * d + k] : 0.0f;
        }
        for (int e = lane; e < 32 * KC; e += 64) {
          const int row = e / KC, c = e - row * KC;
          int64_t gr = row0 + row;
          if (gr >= p.n) gr = p.n - 1;
          xs[c * XS_STRIDE + row] = (k0 + c < d) ? p.X[gr * d + k0 + c] : 0.0f;
        }
        __syncthreads();
        const float *xb = xs + hi * XS_STRIDE + j;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          const float *ra = RA + (size_t)t * KS * 64 + lane;
#pragma unroll
          for (int kk = 0; kk < KS; ++kk)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk * 64], xb[2 * kk * XS_STRIDE], acc[t], 0, 0, 0);
        }
      }
      if (row0 + j < p.n) {
        float *o = p.RX + (size_t)(row0 + j) * d;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = (g0 + t) * 32 + 4 * hi + 8 * (r >> 2) + (r & 3);
            if (i < d) o[i] = acc[t][r];
          }
        }
      }
    }
  }
}

// The same chunked rotation with COALESCED, PREFETCHED staging (d % 4 == 0, 16-byte aligned X and R; round 4).  The kernel
// above fills its LDS panels with 48 scalar loads per thread and chunk (R read across its rows: one cache line per lane) and
// multiplies only after they have landed: 36 TF at d = 960.  Here a thread fetches the next chunk as twelve 16-byte loads
// along k (8 of the R panel, 4 of its wavefront's X slice) into registers BEFORE the 128 MFMAs of the current chunk and
// writes them into the A-fragment / transposed layouts afterwards; the panel rows are padded to 65 floats so that those
// writes spread over the banks.  Same k-ordered chain per output, same bits.
constexpr int RW_RA_STRIDE = 65;
template <int NWAVES, int TG>
__global__ __launch_bounds__(NWAVES * 64) void rotate_wide2_kernel(RotParams p) {
  static_assert(NWAVES == 4 && TG == 8, "thread mapping of the staging loads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KC = 32, KS = KC / 2;
  float *RA = reinterpret_cast<float *>(smem);                       // [TG][KS] rows of RW_RA_STRIDE (64 used)
  float *xs_all = RA + (size_t)TG * KS * RW_RA_STRIDE;               // NWAVES * KC * XS_STRIDE
  const int d = p.d, NT = p.NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  float *xs = xs_all + (size_t)wave * (KC * XS_STRIDE);
  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t nbatches = (ntiles + NWAVES - 1) / NWAVES;
  // staging roles: R panel -- thread -> (panel row tid / 8 of every tile t, k quad tid % 8); X slice -- lane -> (row lane / 8 + 8 v, k quad lane % 8)
  const int rrow = tid >> 3, rq4 = tid & 7;
  const int xrow = lane >> 3, xq4 = lane & 7;
  for (int64_t batch = blockIdx.x; batch < nbatches; batch += gridDim.x) {
    const int64_t tile = batch * NWAVES + wave;
    const int64_t row0 = tile * 32;
    for (int g0 = 0; g0 < NT; g0 += TG) {
      f32x16 acc[TG];
#pragma unroll
      for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
      float4 ra4[TG], x4[4];
      auto fetch = [&](int k0) {
        const int kr = k0 + 4 * rq4;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          const int i = (g0 + t) * 32 + rrow;
          ra4[t] = (i < d && kr < d) ? *reinterpret_cast<const float4 *>(p.R + (size_t)i * d + kr) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int kx = k0 + 4 * xq4;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          int64_t gr = row0 + xrow + 8 * v;
          if (gr >= p.n) gr = p.n - 1;
          x4[v] = kx < d ? *reinterpret_cast<const float4 *>(p.X + gr * d + kx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto stash = [&]() {
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          // k = 4 rq4 + e  ->  k step 2 rq4 + e / 2, half e & 1
          float *dst = RA + (size_t)(t * KS + 2 * rq4) * RW_RA_STRIDE + rrow;
          dst[0] = ra4[t].x; dst[32] = ra4[t].y; dst[RW_RA_STRIDE] = ra4[t].z; dst[RW_RA_STRIDE + 32] = ra4[t].w;
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          float *dst = xs + (4 * xq4) * XS_STRIDE + xrow + 8 * v;
          dst[0] = x4[v].x; dst[XS_STRIDE] = x4[v].y; dst[2 * XS_STRIDE] = x4[v].z; dst[3 * XS_STRIDE] = x4[v].w;
        }
      };
      fetch(0);
      for (int k0 = 0; k0 < d; k0 += KC) {
        __syncthreads();
        stash();
        __syncthreads();
        if (k0 + KC < d) fetch(k0 + KC);
        const float *xb = xs + hi * XS_STRIDE + j;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
          const float *ra = RA + (size_t)t * KS * RW_RA_STRIDE + lane;
#pragma unroll
          for (int kk = 0; kk < KS; ++kk)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk * RW_RA_STRIDE], xb[2 * kk * XS_STRIDE], acc[t], 0, 0, 0);
        }
      }
      if (row0 + j < p.n) {
        float *o = p.RX + (size_t)(row0 + j) * d;
#pragma unroll
        for (int t = 0; t < TG; ++t) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int i0 = (g0 + t) * 32 + 4 * hi + 8 * g4;
            if (i0 < d)
              *reinterpret_cast<float4 *>(o + i0) = make_float4(acc[t][g4 * 4 + 0], acc[t][g4 * 4 + 1], acc[t][g4 * 4 + 2], acc[t][g4 * 4 + 3]);
          }
        }
      }
    }
  }
}

// Rotation, fast path for d % 8 == 0 (KK = d/2 compile time): the 32 x d tile of X never touches
// LDS.  Lane (j, hi) loads the 16-byte pieces X[j][8q + 4hi .. +3]; two v_permlane32_swap per piece
// pair turn them into the B fragments of k-steps 4q..4q+3 (lanes 0-31 the even, 32-63 the odd
// dimension of each step).  R stays in LDS in A-fragment order, so LDS holds only d*d*4 bytes and a
// workgroup can run 8-16 wavefronts; loads of the next tile are issued before the MFMA chains.
template <int KK, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void rotate_kernel_v2(RotParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = 2 * KK, NT = (D + 31) / 32, NP = D / 8;   // NP 16-byte pieces per lane
  float *RA = reinterpret_cast<float *>(smem);                 // NT*KK*64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  for (int idx = tid; idx < NT * KK * 64; idx += NWAVES * 64) {
    const int l = idx & 63;
    const int kk = (idx >> 6) % KK, t = (idx >> 6) / KK;
    const int i = t * 32 + (l & 31), k = 2 * kk + (l >> 5);
    RA[idx] = (i < D) ? p.R[(size_t)i * D + k] : 0.0f;
  }
  __syncthreads();
  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t total_waves = (int64_t)gridDim.x * NWAVES;
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave;
  float4 nx[NP];
  auto gload = [&](int64_t tile) {
    int64_t gr = tile * 32 + j;
    if (gr >= p.n) gr = p.n - 1;
    const float4 *src = reinterpret_cast<const float4 *>(p.X + gr * D + 4 * hi);
#pragma unroll
    for (int q = 0; q < NP; ++q) nx[q] = src[2 * q];
  };
  if (tile0 < ntiles) gload(tile0);
  for (int64_t tile = tile0; tile < ntiles; tile += total_waves) {
    const int64_t row0 = tile * 32;
    float b[KK];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      float x = nx[q].x, y = nx[q].y, z = nx[q].z, w = nx[q].w;
      swap32(x, y);   // x: dims (8q, 8q+1) = k-step 4q      y: dims (8q+4, 8q+5) = k-step 4q+2
      swap32(z, w);   // z: dims (8q+2, 8q+3) = k-step 4q+1  w: dims (8q+6, 8q+7) = k-step 4q+3
      b[4 * q + 0] = x; b[4 * q + 1] = z; b[4 * q + 2] = y; b[4 * q + 3] = w;
    }
#pragma unroll 1      // (unrolled, so that a tile's stores overlap the next chain: spills, 0.31 -> 0.61 ms)
    for (int t = 0; t < NT; ++t) {
      // next tile's loads go out under the last chain of this one (keeps nx's live range short)
      if (t == NT - 1 && tile + total_waves < ntiles) gload(tile + total_waves);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      const float *ra = RA + (size_t)t * KK * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[kk * 64], b[kk], acc, 0, 0, 0);
      if (row0 + j < p.n) {
        float *o = p.RX + (size_t)(row0 + j) * D;
        const int ibase = t * 32 + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int i0 = ibase + 8 * g4;
          if (i0 < D)
            *reinterpret_cast<float4 *>(o + i0) =
                make_float4(acc[g4 * 4 + 0], acc[g4 * 4 + 1], acc[g4 * 4 + 2], acc[g4 * 4 + 3]);
        }
      }
    }
  }
}

#ifndef RQ_ROT_NW
#define RQ_ROT_NW 8
#endif
template <int KK>
static int launch_rotate_v2(const RotParams &p, int num_cu, hipStream_t stream) {
  constexpr int NW = RQ_ROT_NW;
  constexpr int NT = (2 * KK + 31) / 32;
  const size_t lds = (size_t)NT * KK * 64 * sizeof(float);
  auto kern = rotate_kernel_v2<KK, NW>;
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ntiles = (p.n + 31) / 32;
  const int grid = (int)std::min<int64_t>(num_cu, (ntiles + NW - 1) / NW);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

__global__ void widen_codes_kernel(int16_t *out1, const uint8_t *codes, size_t nelem) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nelem) out1[i] = (int16_t)((int)codes[i] + 1);  // src/PQ.jl:45-47: Int16, one-based
}

// ------------------------------------------------------------------------------------------
template <int KS, int NT, int NWAVES, bool DIRECT>
static int launch_encode(EncParams p, int num_cu, hipStream_t stream) {
  p.NT = NT;  // centroids are padded to NT*32 (+inf norms, zero rows)
  // All m sub-codebooks in LDS when they fit (SIFT: 128 KiB, Deep: 96 KiB); otherwise the
  // sub-quantizers are encoded in groups, one launch per group (X is re-read per group).
  const size_t per_sub = ((size_t)p.NT * KS * 64 + (size_t)p.NT * 32) * sizeof(float);
  const size_t fixed_staged = (size_t)NWAVES * 2 * KS * XS_STRIDE * sizeof(float);
  const size_t budget = 160 * 1024;
  // every sub-quantizer exactly 2*KS wide (and 8-byte aligned rows): the LDS-free-X fast path
  constexpr bool direct = DIRECT;
  const size_t fixed = direct ? 0 : fixed_staged;
  if (per_sub + fixed > budget)
    return fail(RQ_EUNSUPPORTED, "one sub-codebook needs %zu B of LDS (> 160 KiB): h=%d ksteps=%d",
                per_sub + fixed, p.h, KS);
  const int gmax = (int)std::min<size_t>((budget - fixed) / per_sub, (size_t)p.m);
  void (*kern)(EncParams);
  if constexpr (DIRECT) kern = encode_pq_direct_kernel<KS, NT, NWAVES>;
  else kern = encode_pq_kernel<KS, NT, NWAVES>;
  const int64_t ntiles = (p.n + 31) / 32;
  const int grid = (int)std::min<int64_t>(num_cu, (ntiles + NWAVES - 1) / NWAVES);
  for (int i0 = 0; i0 < p.m; i0 += gmax) {
    p.i0 = i0;
    p.i1 = std::min(p.m, i0 + gmax);
    const size_t lds = per_sub * (size_t)(p.i1 - p.i0) + fixed;
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, p);
    RQ_HIP(hipGetLastError());
  }
  return RQ_OK;
}

template <int SUB, int NT, int NWAVES>
static int launch_encode_split(EncParams p, int num_cu, hipStream_t stream) {
  p.NT = NT;
  constexpr int NPIECE = SplitShape<SUB>::NPIECE;
  const size_t per_sub = (size_t)NT * NPIECE * 64 * 16 + (size_t)NT * 32 * sizeof(float) + sizeof(float);
  const size_t budget = 160 * 1024 - 64;
  const int gmax = (int)std::min<size_t>(budget / per_sub, (size_t)p.m);
  if (gmax < 1) return fail(RQ_EUNSUPPORTED, "split encode: one sub-codebook needs %zu B of LDS", per_sub);
  auto kern = p.dbg_w ? encode_pq_split_kernel<SUB, NT, NWAVES, true> : encode_pq_split_kernel<SUB, NT, NWAVES, false>;
  const int64_t ntiles = (p.n + 31) / 32;
  const int grid = (int)std::min<int64_t>(num_cu, (ntiles + NWAVES - 1) / NWAVES);
  for (int i0 = 0; i0 < p.m; i0 += gmax) {
    p.i0 = i0;
    p.i1 = std::min(p.m, i0 + gmax);
    const size_t lds = per_sub * (size_t)(p.i1 - p.i0) + 64;
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, p);
    RQ_HIP(hipGetLastError());
  }
  return RQ_OK;
}

static thread_local int g_last_encode_kernel = 0;      // 1 split, 2 f32-MFMA direct (X in registers), 3 f32-MFMA LDS-staged, 4 wide
const char *last_encode_kernel_name() {
  switch (g_last_encode_kernel) {
    case 1: return "encode_pq_split_kernel";
    case 2: return "encode_pq_direct_kernel";
    case 3: return "encode_pq_kernel";
    case 4: return "encode_wide_kernel";
    case 5: return "encode_pq_filter_kernel";
    default: return "";
  }
}

int encode_launch(uint8_t *codes, const float *X, const float *C, int64_t n, int d, int m, int h,
                  int num_cu, hipStream_t stream, float *dbg_w) {
  if (n <= 0) return RQ_OK;
  if (m < 1 || m > 32) return fail(RQ_EUNSUPPORTED, "encode covers 1 <= m <= 32; got m=%d", m);
  if (h < 1 || h > 256) return fail(RQ_EUNSUPPORTED, "encode emits uint8 codes: 1 <= h <= 256; got h=%d", h);
  if (d < m) return fail(RQ_EINVAL, "d=%d < m=%d", d, m);
  EncParams p;
  p.X = X; p.C = C; p.codes = codes; p.n = n; p.d = d; p.m = m; p.h = h;
  p.NT = (h + 31) / 32;
  // ENC_SPLIT_DELTA_MILLI: the margin's numerator in thousandths (3000 = the shipped 3 * 2^-14); tests walk it down to
  // find where codes first change -- the measured safety factor of the filter (tests/test_gpu_encode_margin.py)
  p.delta_rel = (float)tuning("ENC_SPLIT_DELTA_MILLI", 3000) * 1e-3f * 6.103515625e-05f;
  p.dbg_w = dbg_w;
  const int per = d / m, extra = d % m;
  int pos = 0, maxsub = 0;
  for (int i = 0; i < m; ++i) {
    p.off[i] = pos;
    const int s = per + (i < extra ? 1 : 0);
    pos += s;
    maxsub = std::max(maxsub, s);
  }
  p.off[m] = pos;
  const int ks = (maxsub + 1) / 2;
  const int nw = tuning("ENC_WAVES", 16);
  const int nt = (h + 31) / 32;
  // even sub-space widths up to 16 (BASELINE's 16 and 6): bf16 matrix-core filter + exact re-evaluation of the candidates
  if (tuning("ENC_SPLIT", 1) && (d % m == 0) && (d / m) % 2 == 0 && d / m <= 16 && (((uintptr_t)X & 7) == 0)) {
    const int sw = tuning("ENC_SPLIT_WAVES", 0);
    // ENC_SPLIT=1 (shipped): filter launch + exact pass (rq_encode_filter.hip); 2: the one-pass kernel below
    if (tuning("ENC_SPLIT", 1) == 1) {
      g_last_encode_kernel = 5;
      return encode_filter_launch(p, d / m, nt, sw, num_cu, stream);
    }
#define RQ_SPLIT_NT(SUBV, NW)                                                        \
  do {                                                                               \
    g_last_encode_kernel = 1;                                                        \
    if (nt <= 1) return launch_encode_split<SUBV, 1, NW>(p, num_cu, stream);           \
    if (nt <= 2) return launch_encode_split<SUBV, 2, NW>(p, num_cu, stream);           \
    if (nt <= 4) return launch_encode_split<SUBV, 4, NW>(p, num_cu, stream);           \
    return launch_encode_split<SUBV, 8, NW>(p, num_cu, stream);                        \
  } while (0)
#define RQ_SPLIT_CASE(SUBV)                                                          \
  if (d / m == SUBV) {                                                               \
    if (sw == 8) RQ_SPLIT_NT(SUBV, 8);                                               \
    if (sw == 12 || (sw == 0 && SUBV > 8)) RQ_SPLIT_NT(SUBV, 12);                    \
    RQ_SPLIT_NT(SUBV, 16);                                                           \
  }
    RQ_SPLIT_CASE(2) RQ_SPLIT_CASE(4) RQ_SPLIT_CASE(6) RQ_SPLIT_CASE(8)
    RQ_SPLIT_CASE(10) RQ_SPLIT_CASE(12) RQ_SPLIT_CASE(14) RQ_SPLIT_CASE(16)
#undef RQ_SPLIT_CASE
#undef RQ_SPLIT_NT
  }
  if (dbg_w) return fail(RQ_EUNSUPPORTED, "the filter's W values exist for the split kernel only (even sub-space widths <= 16)");
#define RQ_ENC_NT(KSV, NW, DIR)                                             \
  do {                                                                     \
    g_last_encode_kernel = DIR ? 2 : 3;                                    \
    if (nt <= 1) return launch_encode<KSV, 1, NW, DIR>(p, num_cu, stream);  \
    if (nt <= 2) return launch_encode<KSV, 2, NW, DIR>(p, num_cu, stream);  \
    if (nt <= 4) return launch_encode<KSV, 4, NW, DIR>(p, num_cu, stream);  \
    return launch_encode<KSV, 8, NW, DIR>(p, num_cu, stream);               \
  } while (0)
  // fast path: every sub-quantizer exactly 2*KS wide and rows 8-byte aligned -> X straight to registers,
  // 8 or 16 wavefronts per workgroup; otherwise the LDS-staged kernel (8 wavefronts)
#define RQ_ENC_CASE(KSV)                                                   \
  if (ks <= KSV) {                                                         \
    const bool direct = tuning("ENC_DIRECT", 1) && (d % m == 0) && (d / m == 2 * KSV) && \
                        (((uintptr_t)X & 7) == 0);                         \
    if (direct && nw == 16) RQ_ENC_NT(KSV, 16, true);                       \
    if (direct) RQ_ENC_NT(KSV, 8, true);                                    \
    RQ_ENC_NT(KSV, 8, false);                                               \
  }
  RQ_ENC_CASE(1)
  RQ_ENC_CASE(2)
  RQ_ENC_CASE(3)
  RQ_ENC_CASE(4)
  RQ_ENC_CASE(6)
  RQ_ENC_CASE(8)
  RQ_ENC_CASE(16)
  RQ_ENC_CASE(32)
  // wide sub-spaces (RVQ stages are full-dimensional: sub = d = 96 / 128): 8 wavefronts, X in registers;
  // the LDS-staged fallback runs 4 wavefronts and only fits while codebook + staging <= 160 KiB
#define RQ_ENC_CASE_WIDE(KSV)                                              \
  if (ks == KSV && tuning("ENC_DIRECT", 1) && (d % m == 0) && (d / m == 2 * KSV) && \
      (((uintptr_t)X & 7) == 0) && (size_t)nt * KSV * 64 * 4 + (size_t)nt * 128 <= 160 * 1024)   \
    RQ_ENC_NT(KSV, 8, true);
  RQ_ENC_CASE_WIDE(48)
  RQ_ENC_CASE_WIDE(64)
#undef RQ_ENC_CASE_WIDE
#undef RQ_ENC_NT
#undef RQ_ENC_CASE
  // any other width: chunked kernel, codebook streamed through LDS
  g_last_encode_kernel = 4;
  if (nt <= 1) return launch_encode_wide<1>(p, num_cu, stream);
  if (nt <= 2) return launch_encode_wide<2>(p, num_cu, stream);
  if (nt <= 4) return launch_encode_wide<4>(p, num_cu, stream);
  return launch_encode_wide<8>(p, num_cu, stream);
}

int rotate_launch(float *RX, const float *R, const float *X, int d, int64_t n, int num_cu,
                  hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  RotParams p;
  p.R = R; p.X = X; p.RX = RX; p.n = n; p.d = d;
  p.NT = (d + 31) / 32;
  p.KK = (d + 1) / 2;
  if (tuning("ROT_V2", 1) && ((uintptr_t)X & 15) == 0) {
    switch (d) {
      case 32: return launch_rotate_v2<16>(p, num_cu, stream);
      case 64: return launch_rotate_v2<32>(p, num_cu, stream);
      case 96: return launch_rotate_v2<48>(p, num_cu, stream);
      case 128: return launch_rotate_v2<64>(p, num_cu, stream);
      default: break;   // other d: generic LDS-staged kernel below
    }
  }
  constexpr int NW = 4;
  const size_t lds = ((size_t)p.NT * p.KK * 64 + (size_t)NW * 2 * p.KK * XS_STRIDE) * sizeof(float);
  if (lds > 160 * 1024) {
    // R does not fit the LDS: chunked kernel (any d)
    constexpr int TG = 8, KC = 32;
    if (tuning("ROT_WIDE2", 1) && (d & 3) == 0 && (((uintptr_t)X | (uintptr_t)R | (uintptr_t)RX) & 15) == 0) {
      const size_t lds2 = ((size_t)TG * (KC / 2) * RW_RA_STRIDE + (size_t)NW * KC * XS_STRIDE) * sizeof(float);
      auto wk2 = rotate_wide2_kernel<NW, TG>;
      RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wk2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      const int64_t nb2 = ((n + 31) / 32 + NW - 1) / NW;
      hipLaunchKernelGGL(wk2, dim3((int)std::min<int64_t>(2 * (int64_t)num_cu, nb2)), dim3(NW * 64), lds2, stream, p);
      RQ_HIP(hipGetLastError());
      return RQ_OK;
    }
    const size_t wlds = ((size_t)TG * (KC / 2) * 64 + (size_t)NW * KC * XS_STRIDE) * sizeof(float);
    auto wk = rotate_wide_kernel<NW, TG, KC>;
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds));
    const int64_t nb = ((n + 31) / 32 + NW - 1) / NW;
    hipLaunchKernelGGL(wk, dim3((int)std::min<int64_t>(2 * (int64_t)num_cu, nb)), dim3(NW * 64), wlds, stream, p);
    RQ_HIP(hipGetLastError());
    return RQ_OK;
  }
  auto kern = rotate_kernel<NW>;
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t ntiles = (n + 31) / 32;
  const int grid = (int)std::min<int64_t>(num_cu, (ntiles + NW - 1) / NW);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// RVQ stage epilogue (src/RVQ.jl:56  Xr .-= C[i][:, B[i]]): Xr[j][:] -= C_i[code_j][:], the stage's
// codes go to column `stage` of the [n][m] code matrix, and the per-centre counts of the stage
// (update_assignments!'s `counts`, src/RVQ.jl:43-47) are accumulated.  One thread per float4 of Xr.
__global__ __launch_bounds__(256) void rvq_residual_kernel(float *Xr, const float *Ci, const uint8_t *stage_codes,
                                                           uint8_t *codes, unsigned int *counts, int64_t n, int d,
                                                           int m, int stage) {
  const int d4 = d >> 2;   // d % 4 == 0 on this path
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * d4) return;
  const int64_t j = e / d4;
  const int c4 = (int)(e - j * d4);
  const int code = stage_codes[j];
  float4 x = reinterpret_cast<float4 *>(Xr)[e];
  const float4 c = reinterpret_cast<const float4 *>(Ci)[(size_t)code * d4 + c4];
  x.x = x.x - c.x; x.y = x.y - c.y; x.z = x.z - c.z; x.w = x.w - c.w;
  reinterpret_cast<float4 *>(Xr)[e] = x;
  if (c4 == 0) {
    codes[j * m + stage] = (uint8_t)code;
    if (counts) atomicAdd(&counts[code], 1u);
  }
}

__global__ __launch_bounds__(256) void rvq_residual_scalar_kernel(float *Xr, const float *Ci, const uint8_t *stage_codes,
                                                                  uint8_t *codes, unsigned int *counts, int64_t n,
                                                                  int d, int m, int stage) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * d) return;
  const int64_t j = e / d;
  const int c = (int)(e - j * d);
  const int code = stage_codes[j];
  Xr[e] = Xr[e] - Ci[(size_t)code * d + c];
  if (c == 0) {
    codes[j * m + stage] = (uint8_t)code;
    if (counts) atomicAdd(&counts[code], 1u);
  }
}

// quantize_rvq (src/RVQ.jl:18-66) on resident data: m full-dimensional stages, each the encode kernel
// with one sub-quantizer of width d on the running residual.  Xr [n][d] is overwritten (in: X or a copy
// of it, out: the final residual); stage_codes is n bytes of scratch; counts is [m][h] or NULL.
// one stage's epilogue: Xr -= Ci[stage_codes], codes[:, stage] = stage_codes, cnt[code] += 1 (cnt may be NULL)
int rvq_residual_launch(float *Xr, const float *Ci, const uint8_t *stage_codes, uint8_t *codes, unsigned int *cnt,
                        int64_t n, int d, int m, int stage, hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  if ((d & 3) == 0 && (((uintptr_t)Xr | (uintptr_t)Ci) & 15) == 0) {
    const int64_t total = n * (d >> 2);
    hipLaunchKernelGGL(rvq_residual_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, Xr, Ci,
                       stage_codes, codes, cnt, n, d, m, stage);
  } else {
    const int64_t total = n * d;
    hipLaunchKernelGGL(rvq_residual_scalar_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, Xr,
                       Ci, stage_codes, codes, cnt, n, d, m, stage);
  }
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int rvq_encode_launch(uint8_t *codes, float *Xr, uint8_t *stage_codes, unsigned int *counts, const float *C,
                      int64_t n, int d, int m, int h, int num_cu, hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  if (counts) RQ_HIP(hipMemsetAsync(counts, 0, (size_t)m * h * sizeof(unsigned int), stream));
  for (int i = 0; i < m; ++i) {
    const float *Ci = C + (size_t)i * h * d;
    RQ_TRY(encode_launch(stage_codes, Xr, Ci, n, d, 1, h, num_cu, stream));
    RQ_TRY(rvq_residual_launch(Xr, Ci, stage_codes, codes, counts ? counts + (size_t)i * h : nullptr, n, d, m, i, stream));
  }
  return RQ_OK;
}

int widen_codes_launch(int16_t *out1, const uint8_t *codes, int64_t nelem, hipStream_t stream) {
  if (nelem <= 0) return RQ_OK;
  hipLaunchKernelGGL(widen_codes_kernel, dim3((uint32_t)((nelem + 255) / 256)), dim3(256), 0, stream, out1,
                     codes, (size_t)nelem);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

}  // namespace rq
