// rq_encode_filter.hip -- PQ encode (quantize_pq, src/PQ.jl:18-48) as two launches: a bf16 matrix-core FILTER that settles
// every (vector, sub-quantizer) pair whose nearest centroid is beyond doubt, and an EXACT pass over the pairs it leaves.
//
// The arithmetic that defines the answer is the canonical one of oracle/rq_oracle.c:264-328 (g, sa, sb = k-ordered fmaf
// chains from +0, v = max(fl(fl(sa + sb) - 2g), 0), first index of the minimum); see rq_encode.hip for the derivation of
// the filter's margin.  What the filter proves, per (vector, sub-quantizer):
//     W_k = |c_k|^2 - 2<c_k, x>  (bf16 hi/lo pieces on v_mfma_f32_32x32x16_bf16, accumulator pre-loaded with |c_k|^2)
//     |(W_k + |x|^2) - u_k| <= e = 1.02 * 2^-14 (|c_k|^2 + |x|^2)          (measured: tests/test_gpu_encode_margin.py)
//  => the canonical argmin, and everything that ties with it, lies in { k : W_k <= min W + DELTA },
//     DELTA = 3 * 2^-14 (max_k |c_k|^2 + |x|^2).
// If that set has ONE element, it is the argmin and nothing else needs computing (98 % of the pairs on SIFT-like data,
// 99.5 % on Deep-like).  Otherwise the pair's bit is set in flags[row]; encode_pq_fix_kernel then evaluates ALL h
// centroids of such a pair canonically on the VALU and overwrites the code.  Nothing approximate reaches the output.
//
// Why two launches (round 5): with the exact evaluation inside the filter kernel (rq_encode.hip, encode_pq_split_kernel)
// half of the wavefronts walked a divergent candidate loop with L2 latencies in it at 3 wavefronts per SIMD (0.08 ms of
// the 0.41 per 1e6 SIFT vectors), and the `if (improved) copy` of the tile loop cut the loop into basic blocks, so no LDS
// read was ever in flight across an MFMA chain (loads + MFMAs alone: 0.25 ms against floors of 0.08).  Here the tile loop
// is one straight-line block (the copy of the winning tile runs under an exec mask set inside the asm statement), the
// fragments of tile t + 1 are requested before the MFMAs of tile t issue, and the exact pass runs at full occupancy.
#include "rq_encode_split.h"

#include <type_traits>

namespace rq {

// ---- filter --------------------------------------------------------------------------------------------------------
// ub <- a in the lanes of `msk` (a lane mask under the current exec): eight 64-bit moves with exec narrowed INSIDE the
// statement -- no branch, so the tile loop stays one basic block.  `a` holds MFMA results: the statement depends on `msk`,
// which the caller derives from ordinary VALU reads of the same accumulator (tile_min), so the compiler's hazard wait for
// that MFMA precedes it (tests/test_isa.py walks the generated code for exactly this).
__device__ __forceinline__ void copy_lanes(f32x16 &ub, const f32x16 &a, uint64_t msk) {
  f32x2 d0 = {ub[0], ub[1]}, d1 = {ub[2], ub[3]}, d2 = {ub[4], ub[5]}, d3 = {ub[6], ub[7]};
  f32x2 d4 = {ub[8], ub[9]}, d5 = {ub[10], ub[11]}, d6 = {ub[12], ub[13]}, d7 = {ub[14], ub[15]};
  const f32x2 s0 = {a[0], a[1]}, s1 = {a[2], a[3]}, s2 = {a[4], a[5]}, s3 = {a[6], a[7]};
  const f32x2 s4 = {a[8], a[9]}, s5 = {a[10], a[11]}, s6 = {a[12], a[13]}, s7 = {a[14], a[15]};
  uint64_t sv;
  asm("s_mov_b64 %[sv], exec\n\t"
      "s_mov_b64 exec, %[m]\n\t"
      "v_mov_b64 %[d0], %[s0]\n\t"
      "v_mov_b64 %[d1], %[s1]\n\t"
      "v_mov_b64 %[d2], %[s2]\n\t"
      "v_mov_b64 %[d3], %[s3]\n\t"
      "v_mov_b64 %[d4], %[s4]\n\t"
      "v_mov_b64 %[d5], %[s5]\n\t"
      "v_mov_b64 %[d6], %[s6]\n\t"
      "v_mov_b64 %[d7], %[s7]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [d4] "+v"(d4), [d5] "+v"(d5), [d6] "+v"(d6),
        [d7] "+v"(d7), [sv] "=&s"(sv)
      : [m] "s"(msk), [s0] "v"(s0), [s1] "v"(s1), [s2] "v"(s2), [s3] "v"(s3), [s4] "v"(s4), [s5] "v"(s5), [s6] "v"(s6),
        [s7] "v"(s7));
  ub[0] = d0.x; ub[1] = d0.y; ub[2] = d1.x; ub[3] = d1.y; ub[4] = d2.x; ub[5] = d2.y; ub[6] = d3.x; ub[7] = d3.y;
  ub[8] = d4.x; ub[9] = d4.y; ub[10] = d5.x; ub[11] = d5.y; ub[12] = d6.x; ub[13] = d6.y; ub[14] = d7.x; ub[15] = d7.y;
}

// 16 values <= thr, as a bit mask: four independent v_cmp / v_addc chains (one chain of 32 dependent instructions cost
// 0.04 ms per 1e6 SIFT vectors at 3 wavefronts per SIMD).  Only for values ordinary VALU instructions produced (see copy_lanes).
__device__ __forceinline__ uint32_t mask_leq16_4(const f32x16 &v, float thr) {
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
  for (int r = 3; r >= 0; --r) {
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(c0) : "v"(v[r]), "v"(thr) : "vcc");
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(c1) : "v"(v[4 + r]), "v"(thr) : "vcc");
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(c2) : "v"(v[8 + r]), "v"(thr) : "vcc");
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(c3) : "v"(v[12 + r]), "v"(thr) : "vcc");
  }
  return c0 | (c1 << 4) | (c2 << 8) | (c3 << 12);
}

__device__ __forceinline__ float min16(const f32x16 &a) {
  const float m1 = __builtin_fminf(__builtin_fminf(a[0], a[1]), a[2]);
  const float m2 = __builtin_fminf(__builtin_fminf(a[3], a[4]), a[5]);
  const float m3 = __builtin_fminf(__builtin_fminf(a[6], a[7]), a[8]);
  const float m4 = __builtin_fminf(__builtin_fminf(a[9], a[10]), a[11]);
  const float m5 = __builtin_fminf(__builtin_fminf(a[12], a[13]), a[14]);
  float mm = __builtin_fminf(__builtin_fminf(m1, m2), m3);
  mm = __builtin_fminf(__builtin_fminf(mm, m4), m5);
  return __builtin_fminf(mm, a[15]);
}

// LDS image of the sub-codebooks of one launch (same layout as encode_pq_split_kernel's), built ONCE per launch in global
// scratch by encode_tables_kernel and copied by every workgroup of the filter with straight 16-byte loads.  (Built inside
// each workgroup -- 8 strided scalar loads per fragment, a dozen dependent round trips to L2 -- the prologue took 50 us of
// the 280 a 1e6-row SIFT-shape launch lasts, and nearly all of a short one.)
//   cbA [mg][NT][NPIECE][64] uint4  bf16 pieces of -2c in the A-fragment order of v_mfma_f32_32x32x16_bf16 (lane l: centroid
//                                   l & 31, K elements 8 (l >> 5) .. + 7; PACK: K 0-7 = hi pieces, 8-15 = lo pieces)
//   saL [mg][NT][2][16] float       |c_k|^2 (canonical chain s = 0..sub-1 from +0) in C/D-fragment order, +inf for k >= h
//   saMax [mg] float                max_k |c_k|^2 over the real centroids (NaN if any norm is NaN), padded to 16 bytes
template <int SUB>
__host__ __device__ constexpr size_t filter_image_bytes(int mg, int NT) {
  return (size_t)mg * NT * SplitShape<SUB>::NPIECE * 64 * 16 + (size_t)mg * NT * 32 * 4 + (((size_t)mg * 4 + 15) & ~(size_t)15);
}

// grid = mg workgroups (one per sub-quantizer of the launch) x 256 threads
template <int SUB, int NT>
__global__ __launch_bounds__(256) void encode_tables_kernel(EncParams p) {
  constexpr bool PACK = SplitShape<SUB>::PACK;
  constexpr int NPIECE = SplitShape<SUB>::NPIECE;
  const int h = p.h, i0 = p.i0, mg = p.i1 - p.i0, il = blockIdx.x;
  uint4 *cbA = reinterpret_cast<uint4 *>(p.image);
  float *saL = reinterpret_cast<float *>(cbA + (size_t)mg * NT * NPIECE * 64);
  float *saMax = saL + (size_t)mg * NT * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float wmax[4];
  for (int idx = tid; idx < NT * NPIECE * 64; idx += 256) {
    const int l = idx & 63;
    const int piece = (idx >> 6) % NPIECE;
    const int t = (idx >> 6) / NPIECE;
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
    cbA[(size_t)il * NT * NPIECE * 64 + idx] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  float mx = 0.0f;
  for (int idx = tid; idx < NT * 32; idx += 256) {
    const int c32 = idx & 31, t = idx >> 5;
    const int cen = t * 32 + c32;
    float sa = __uint_as_float(0x7f800000u);
    if (cen < h) {
      const float *c = p.C + ((size_t)(i0 + il) * h + cen) * SUB;
      sa = 0.0f;
#pragma unroll
      for (int sx = 0; sx < SUB; ++sx) sa = __builtin_fmaf(c[sx], c[sx], sa);
      mx = __builtin_fmaxf(mx, sa) + (sa != sa ? sa : 0.0f);    // a NaN norm poisons the bound -> exact pass
    }
    saL[((size_t)(il * NT + t) * 2 + ((c32 >> 2) & 1)) * 16 + (c32 & 3) + 4 * (c32 >> 3)] = sa;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(mx, off);
    mx = (o != o || mx != mx) ? __uint_as_float(0x7fc00000u) : __builtin_fmaxf(mx, o);
  }
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  if (tid == 0) {
    float r = wmax[0];
    for (int w = 1; w < 4; ++w) r = (r != r || wmax[w] != wmax[w]) ? __uint_as_float(0x7fc00000u) : __builtin_fmaxf(r, wmax[w]);
    saMax[il] = r;
  }
}

template <int SUB, int NT, int NWAVES, bool DBG = false>
__global__ __launch_bounds__(NWAVES * 64) void encode_pq_filter_kernel(EncParams p) {
  using Shape = SplitShape<SUB>;
  constexpr bool PACK = Shape::PACK;
  constexpr int NPIECE = Shape::NPIECE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int m = p.m, h = p.h, d = p.d;
  const int i0 = p.i0, mg = p.i1 - p.i0;
  uint4 *cbA = reinterpret_cast<uint4 *>(smem);
  float *saL = reinterpret_cast<float *>(cbA + (size_t)mg * NT * NPIECE * 64);
  float *saMax = saL + (size_t)mg * NT * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  {   // the launch's table image, as encode_tables_kernel left it
    const uint4 *src = reinterpret_cast<const uint4 *>(p.image);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    const int n16 = (int)(filter_image_bytes<SUB>(mg, NT) / 16);
    for (int idx = tid; idx < n16; idx += NWAVES * 64) dst[idx] = src[idx];
    __syncthreads();
  }

  const int64_t ntiles = (p.n + 31) / 32;
  const int64_t total_waves = (int64_t)gridDim.x * NWAVES;
  const int64_t tile0 = (int64_t)blockIdx.x * NWAVES + wave;
  // the lane's 8 K elements of the sub-vector in flight: dimensions 8 hi .. 8 hi + 7 (PACK: 0 .. 7 in both halves)
  constexpr int NPAIR = PACK ? SUB / 2 : 4;
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
  const int cbase = 4 * hi;          // centroid of (tile t, register r): 32 t + 4 hi + 8 (r >> 2) + (r & 3)

  for (int64_t tile = tile0; tile < ntiles; tile += total_waves) {
    const int64_t row0 = tile * 32;
    uint64_t cw[4] = {0, 0, 0, 0};
    uint32_t fl = 0;                 // bit il: (this row, sub-quantizer i0 + il) goes to the exact pass
    // the tile loop's result for one sub-quantizer, consumed by `settle` (one candidate, or the exact pass)
    struct Pend { f32x16 ub; float b1, delta; int t1, il; uint64_t amb; bool slow; };
    auto settle_v = [&](const f32x16 &q_ub, float q_b1, float q_delta, int q_t1, int q_il, uint64_t q_amb, bool q_slow) {
      const int i = i0 + q_il;
      float bo_a = q_b1, bo_b = q_b1;                              // the other half of this vector's centroids
      swap32(bo_a, bo_b);
      const float bo = hi ? bo_a : bo_b;
      const float thr = __builtin_fminf(q_b1, bo) + q_delta;
      const bool contend = (q_b1 <= thr) && !q_slow;
#if defined(RQ_FILT_ABL) && RQ_FILT_ABL == 5
      uint32_t cm = __float_as_uint(q_ub[3]) & 0xffffu;
#else
      uint32_t cm = mask_leq16_4(q_ub, thr);
#endif
      if (!contend) cm = 0;
      const bool todo = contend && __builtin_amdgcn_inverse_ballot_w64(q_amb);
      const int r1 = __builtin_ctz(cm | 0x10000u);
      const uint32_t kmine = cm != 0u ? (uint32_t)(q_t1 * 32 + cbase + 8 * (r1 >> 2) + (r1 & 3)) : 0xffffu;
      // candidate counts of the two halves add; a flagged tile or an unusable bound in either half shows
      const uint32_t mine_w = ((uint32_t)__builtin_popcount(cm) | (todo ? 0x100u : 0u) | (q_slow ? 0x200u : 0u)) << 16 | kmine;
      float ow_a = __uint_as_float(mine_w), ow_b = __uint_as_float(mine_w);
      swap32(ow_a, ow_b);
      const uint32_t other_w = __float_as_uint(hi ? ow_a : ow_b);
      const bool single = (mine_w >> 16) + (other_w >> 16) == 1u;
      const uint32_t kk = min(mine_w & 0xffffu, other_w & 0xffffu);       // the candidate (single) or any stand-in
      fl |= single ? 0u : (1u << q_il);
      const uint64_t bk = (uint64_t)(kk & 0xffu) << (8 * (i & 7));
      switch (i >> 3) {          // (uniform: one 64-bit shift and OR instead of four selected ones)
        case 0: cw[0] |= bk; break;
        case 1: cw[1] |= bk; break;
        case 2: cw[2] |= bk; break;
        default: cw[3] |= bk; break;
      }
    };
    auto settle = [&](const Pend &q) { settle_v(q.ub, q.b1, q.delta, q.t1, q.il, q.amb, q.slow); };
    Pend pend;
    // One sub-quantizer: B fragments, tile loop.  EPI: the PREVIOUS sub-quantizer's `settle` -- ~80 VALU instructions in
    // dependent chains with two half-wave exchanges, no matrix work of its own -- is placed inside this one's straight-line
    // tile loop, where the scheduler can slide it under the MFMAs (build knob RQ_FILT_PIPE = 1; default: it runs right after its
    // own loop).  MEASURED, round 5: the carried state (16 W values + 6 scalars) costs more than the overlap returns -- 252
    // registers at 8 wavefronts per CU: 0.356 ms per 1e6 SIFT-shape vectors against 0.342 unpipelined at 8 and 0.334 at 12
    // wavefronts (where the pipelined build spills 88 registers: 0.390); Deep shape 0.572 / 0.581 / 0.501.  Not shipped.
    auto unit = [&](int il, auto epi_tag) {
      constexpr bool EPI = decltype(epi_tag)::value;
      const int i = i0 + il;
      // |x|^2 (any order: it only scales the margin) and the B fragments: this lane's 8 K elements of x as bf16 pieces
      f32x2 sel[4];
      f32x2 sq = {0.0f, 0.0f};
#pragma unroll
      for (int u = 0; u < NPAIR; ++u) sq = __builtin_elementwise_fma(xn[u], xn[u], sq);
#pragma unroll
      for (int u = 0; u < 4; ++u) sel[u] = u < NPAIR ? xn[u < NPAIR ? u : 0] : f32x2{0.0f, 0.0f};
      uint32_t bh[4], bl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16x2_t hb = __builtin_convertvector(sel[u], bf16x2_t);
        const f32x2 rest = sel[u] - __builtin_convertvector(hb, f32x2);
        bh[u] = __builtin_bit_cast(uint32_t, hb);
        bl[u] = __builtin_bit_cast(uint32_t, __builtin_convertvector(rest, bf16x2_t));
      }
      if (PACK && hi) { bl[0] = bl[1] = bl[2] = bl[3] = 0; }                // second MFMA: [xl | 0] against [ch | cl]
      const bf16x8_t Bh = __builtin_bit_cast(bf16x8_t, make_uint4(bh[0], bh[1], bh[2], bh[3]));
      const bf16x8_t Bl = __builtin_bit_cast(bf16x8_t, make_uint4(bl[0], bl[1], bl[2], bl[3]));
      float sb = sq.x + sq.y;
      if constexpr (!PACK) {                               // the other half-wave holds dimensions 8..15
        float sa_ = sb, sb_ = sb;                          // (v_permlane32_swap: no trip through the LDS crossbar)
        swap32(sa_, sb_);
        sb += hi ? sa_ : sb_;
      }
      // the next sub-vector travels while this one is filtered
      if (il + 1 < mg) gload(tile, il + 1);
      else if (tile + total_waves < ntiles) gload(tile + total_waves, 0);
      const float smax = saMax[il];
      const float ssum = smax + sb;
      const float delta = p.delta_rel * ssum;
      // the bound needs finite, non-vanishing magnitudes; otherwise the pair goes to the exact pass
      const bool slow = !(delta < __uint_as_float(0x7f800000u)) || !(ssum >= SplitCfg::TINY);

      float b1 = 0.0f;                           // running minimum of W over this lane's centroids
      int t1 = 0;                                // the tile that holds it (the first one, on ties)
      uint64_t amb = 0;                          // LANE MASK (scalar registers): another tile came within delta of the running
                                                 // minimum and has not been left behind by more than delta since
      f32x16 ub;                                 // the 16 W values of tile t1
      const uint4 *cb_i = cbA + (size_t)il * NT * NPIECE * 64 + lane;
      const float4 *sa_i = reinterpret_cast<const float4 *>(saL + ((size_t)il * NT * 2 + hi) * 16);
      // fragments of one tile: |c|^2 straight into the accumulator registers, the A pieces beside them
      struct Frag { f32x16 acc; uint4 a0, a1; };
      auto fetch = [&](int t) -> Frag {
        Frag f;
        const float4 *s4 = sa_i + (size_t)t * 8;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float4 v = s4[g4];
          f.acc[g4 * 4 + 0] = v.x; f.acc[g4 * 4 + 1] = v.y; f.acc[g4 * 4 + 2] = v.z; f.acc[g4 * 4 + 3] = v.w;
        }
        f.a0 = cb_i[(size_t)t * NPIECE * 64];
        f.a1 = PACK ? f.a0 : cb_i[(size_t)t * NPIECE * 64 + 64];
        return f;
      };
      auto products = [&](const Frag &f) -> f32x16 {
#if defined(RQ_FILT_ABL) && RQ_FILT_ABL == 1
        { f32x16 a = f.acc; a[0] += __uint_as_float(f.a0.x ^ f.a1.y ^ bh[0] ^ bl[1]); return a; }
#endif
        const bf16x8_t A0 = __builtin_bit_cast(bf16x8_t, f.a0);
        f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bh, f.acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, Bl, acc, 0, 0, 0);
        if constexpr (!PACK) {
          const bf16x8_t A1 = __builtin_bit_cast(bf16x8_t, f.a1);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, Bh, acc, 0, 0, 0);
        }
        return acc;
      };
      auto filter = [&](const f32x16 &a, int t) {
        if constexpr (DBG) {    // tests/test_gpu_encode_margin.py: the very values the filter decides on
          if (row0 + j < p.n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int k = t * 32 + 4 * hi + 8 * (r >> 2) + (r & 3);
              if (k < h) p.dbg_w[((size_t)(row0 + j) * m + i) * h + k] = a[r];
            }
          }
        }
#if defined(RQ_FILT_ABL) && RQ_FILT_ABL == 2
        b1 = __builtin_fminf(b1, a[t & 15]); return;
#endif
        const float mm = min16(a);
        if (t == 0) {                   // (compile-time: the loop is unrolled)
          b1 = mm;
          ub = a;
          return;
        }
        // Invariant after tile t: b1 = the smallest tile minimum so far, t1 = the first tile that reached it, and -- where
        // `amb` is clear -- every other tile seen so far lies more than delta above b1.  A tile within delta of the running
        // minimum sets the flag, whichever of the two is smaller; a minimum that improves by MORE than delta leaves all
        // earlier tiles out of reach and clears it.  The flag lives in a scalar register pair: two SALU instructions per
        // tile instead of a handful of per-lane selects.
        const bool imp = mm < b1;
        const bool near = __builtin_fabsf(mm - b1) <= delta;       // (NaN: false)
        const uint64_t impm = __builtin_amdgcn_ballot_w64(imp), nearm = __builtin_amdgcn_ballot_w64(near);
        amb = nearm | (amb & ~impm);
#if !defined(RQ_FILT_ABL) || RQ_FILT_ABL != 3
        copy_lanes(ub, a, impm);
#else
        ub[t & 15] += a[t & 15];
#endif
        b1 = __builtin_fminf(b1, mm);
        t1 = imp ? t : t1;
      };
      // straight-line tile loop: fragments of tile t + 1 are requested before the MFMAs of tile t issue, the filter of
      // tile t - 1 runs on the VALU under them
      {
        Frag cur = fetch(0), nxt;
        f32x16 done;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#if defined(RQ_FILT_ABL) && RQ_FILT_ABL == 4
          if (t + 1 < NT) { nxt = cur; nxt.a0.x += t; }
#else
          if (t + 1 < NT) nxt = fetch(t + 1);
#endif
          const f32x16 acc = products(cur);
          if (t > 0) filter(done, t - 1);
          if constexpr (EPI) { if (t == (NT > 2 ? 2 : NT - 1)) settle(pend); }
          done = acc;
          if (t + 1 < NT) cur = nxt;
        }
        filter(done, NT - 1);
      }
#if defined(RQ_FILT_PIPE) && RQ_FILT_PIPE
      pend.ub = ub; pend.b1 = b1; pend.delta = delta; pend.t1 = t1; pend.il = il; pend.amb = amb; pend.slow = slow;
#else
      settle_v(ub, b1, delta, t1, il, amb, slow);       // (directly on the live registers: no copy of the 16 kept values)
#endif
    };

#if defined(RQ_FILT_PIPE) && RQ_FILT_PIPE
    unit(0, std::false_type{});
#pragma unroll 1
    for (int il = 1; il < mg; ++il) unit(il, std::true_type{});
    settle(pend);
#else
#pragma unroll 1
    for (int il = 0; il < mg; ++il) unit(il, std::false_type{});
    (void)settle;
    (void)pend;
#endif
    if (hi == 0 && row0 + j < p.n) {
      uint8_t *o = p.codes + (size_t)(row0 + j) * m;
      if ((m & 7) == 0 && mg == m) {
#pragma unroll
        for (int w = 0; w < 4; ++w)
          if (w * 8 < m) reinterpret_cast<uint64_t *>(o)[w] = cw[w];
      } else {
        for (int i = i0; i < p.i1; ++i) o[i] = (uint8_t)(cw[i >> 3] >> (8 * (i & 7)));
      }
      p.flags[row0 + j] = fl;
    }
  }
}

// The canonical evaluation of one work item -- sub-quantizer i, the rows the 32 lane pairs hold (lane (j, hi): row of lane j) --
// against ALL h centroids on v_mfma_f32_32x32x2_f32; returns the first index of the minimum for the lane's row.  sa_i: this
// sub-quantizer's |c|^2 table in C/D-fragment order, already offset by the lane's half (LDS).
template <int SUB, int NT>
__device__ __forceinline__ int exact_item(const EncParams &p, int i, int64_t row, const float4 *sa_i) {
  constexpr int KS = SUB / 2;
  const int lane = threadIdx.x & 63, j = lane & 31, hi = lane >> 5;
  const int h = p.h, d = p.d;
  // the row's sub-vector: |x|^2 (canonical chain) and the B fragments (lane: k = 2 kk + hi of vector j)
  float x[SUB];
  const float *xs = p.X + (size_t)row * d + (size_t)i * SUB;
  if ((SUB % 4 == 0) && (d % 4 == 0) && (((uintptr_t)p.X & 15) == 0)) {
#pragma unroll
    for (int s4 = 0; s4 < SUB / 4; ++s4) {
      const float4 v = reinterpret_cast<const float4 *>(xs)[s4];
      x[4 * s4] = v.x; x[4 * s4 + 1] = v.y; x[4 * s4 + 2] = v.z; x[4 * s4 + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int s2 = 0; s2 < SUB / 2; ++s2) {
      const f32x2 v = reinterpret_cast<const f32x2 *>(xs)[s2];
      x[2 * s2] = v.x; x[2 * s2 + 1] = v.y;
    }
  }
  float sb = 0.0f;
#pragma unroll
  for (int s = 0; s < SUB; ++s) sb = __builtin_fmaf(x[s], x[s], sb);
  float b[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) b[kk] = hi ? x[2 * kk + 1] : x[2 * kk];
  ArgminState st;
  st.best_v = __uint_as_float(0x7f800000u);
  st.best_t = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) st.ub[r] = f32x2{0.0f, 0.0f};
  // A fragments: lane (j, hi) wants C_i[32 t + j][2 kk + hi].  For 8-wide halves (sub = 16) the lane loads floats
  // 8 hi .. 8 hi + 7 of its centroid (two 16-byte loads) and one v_permlane32_swap per register PAIR turns
  // (c[2q] | c[8 + 2q]), (c[2q + 1] | c[9 + 2q]) into the fragments of k-steps q and 4 + q; other widths load the whole
  // row and select.  Four tiles are requested at a time, so an item waits for L2 twice, not once per tile.
  auto cload = [&](int t, float (&a)[KS]) {
    const int cen = t * 32 + j;
    if constexpr (SUB == 16) {
      float4 v0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), v1 = v0;
      if (cen < h) {
        const float4 *src = reinterpret_cast<const float4 *>(p.C + ((size_t)i * h + cen) * SUB + 8 * hi);
        v0 = src[0]; v1 = src[1];
      }
      float r[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        swap32(r[2 * q], r[2 * q + 1]);
        a[q] = r[2 * q]; a[4 + q] = r[2 * q + 1];
      }
    } else {
      float c[SUB];
#pragma unroll
      for (int s = 0; s < SUB; ++s) c[s] = 0.0f;
      if (cen < h) {
        const float *src = p.C + ((size_t)i * h + cen) * SUB;
        if constexpr (SUB % 4 == 0) {
#pragma unroll
          for (int s4 = 0; s4 < SUB / 4; ++s4) {
            const float4 v = reinterpret_cast<const float4 *>(src)[s4];
            c[4 * s4] = v.x; c[4 * s4 + 1] = v.y; c[4 * s4 + 2] = v.z; c[4 * s4 + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < SUB / 2; ++s2) {
            const f32x2 v = reinterpret_cast<const f32x2 *>(src)[s2];
            c[2 * s2] = v.x; c[2 * s2 + 1] = v.y;
          }
        }
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) a[kk] = hi ? c[2 * kk + 1] : c[2 * kk];
    }
  };
  constexpr int TB = NT < 4 ? NT : (KS >= 8 ? 2 : 4);        // tiles per batch (registers: 128 per lane at 4 wavefronts per SIMD)
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += TB) {
    float a[TB][KS];
#pragma unroll
    for (int u = 0; u < TB; ++u) cload(t0 + u, a[u]);
#pragma unroll
    for (int u = 0; u < TB; ++u) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][kk], b[kk], acc, 0, 0, 0);
      tile_argmin(acc, sa_i + (size_t)(t0 + u) * 8, sb, t0 + u, st);
    }
  }
  float best_v = st.best_v;
  int best_i = argmin_finish(st, hi);
  const float ov = __shfl_xor(best_v, 32);
  const int oi = __shfl_xor(best_i, 32);
  if (ov < best_v || (ov == best_v && oi < best_i)) { best_v = ov; best_i = oi; }
  return best_i;
}

// ---- exact pass ----------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per chunk of p.fix_rows rows (4096 where the lists fit LDS; fewer for many sub-quantizers).  The flagged rows of the chunk are collected per sub-quantizer in LDS;
// a work item is (sub-quantizer, 32 flagged rows), taken by the wavefronts in turn.  The canonical evaluation is the one of
// the direct kernels (rq_encode.hip): the h x 32 inner products come off v_mfma_f32_32x32x2_f32 -- bit for bit the k-ordered
// fmaf chain of oracle/rq_oracle.c:264-328 -- with the centroids (A fragments) read straight from global memory (the
// sub-codebooks are L2-resident: 16 KiB per item), then tile_argmin / argmin_finish on all h centroids: the first index of
// the minimum of v = max(fl(fl(sa + sb) - 2g), 0), whatever the filter thought of the pair.  (Round 5's first version
// evaluated the centroids on the VALU, four per lane and one row per wavefront at a time: 0.34 ns per pair, 0.1 ms for the
// 2.4 % of the pairs of 1e6 SIFT-like vectors; on the matrix cores it is ~0.1 ns.)
// (Round 5, measured and not kept: the same pass at the END of the filter kernel -- per-workgroup lists of the open pairs, the
// |c|^2 table already in LDS, no flags array, one launch less.  Same speed at 1e6 rows (0.348 / 0.350 vs 0.345 / 0.344 ms at SIFT
// shape, 0.513 / 0.520 vs 0.522 / 0.532 at Deep shape; 41 vs 46 us at 8192 rows): a workgroup's 16 items over 12 wavefronts are two
// rounds of the same dependent chain -- row -> sub-vector from HBM -> centroid rows from L2 -> 64 MFMAs -- behind a barrier all
// its wavefronts reach at different times, and it needs 4 m bytes of list scratch per row instead of 4.)
// Chunk size, measured (rocprofv3, 1e6 rows; flagged pairs 1.5 % / 0.5 %): 1024 rows x 256 threads 46 / 49 us (SIFT / Deep shape),
// 2048 x 512 40 / 37, 4096 x 512 38 / 23, 4096 x 1024 38 / 21.5 -- sparse flags want big chunks (fuller 32-row items).
#ifndef RQ_FIX_ROWS
#define RQ_FIX_ROWS 4096
#endif
#ifndef RQ_FIX_THREADS
#define RQ_FIX_THREADS 1024
#endif
constexpr int FIX_ROWS_MAX = RQ_FIX_ROWS;       // rows per workgroup when the per-sub-quantizer lists (2 bytes per row) fit LDS
constexpr int FIX_THREADS = RQ_FIX_THREADS;

template <int SUB, int NT>
__global__ __launch_bounds__(FIX_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void encode_pq_fix_kernel(EncParams p) {
  constexpr int KS = SUB / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int mg = p.i1 - p.i0, h = p.h, d = p.d, m = p.m;
  float *saL = reinterpret_cast<float *>(smem);                                   // [mg][NT][2][16] |c|^2, C/D-fragment order
  uint32_t *cnt = reinterpret_cast<uint32_t *>(saL + (size_t)mg * NT * 32);       // [mg]
  uint16_t *list = reinterpret_cast<uint16_t *>(cnt + 32);                        // [mg][fix_rows] flagged rows of the chunk
  const int FIX_ROWS = p.fix_rows;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int64_t row_base = (int64_t)blockIdx.x * FIX_ROWS;
  if (tid < 32) cnt[tid] = 0;
  __syncthreads();
  uint32_t any = 0;
  for (int r = tid; r < FIX_ROWS && row_base + r < p.n; r += FIX_THREADS) {
    uint32_t fl = p.flags[row_base + r];
    any |= fl;
    while (fl) {
      const int il = __builtin_ctz(fl);
      fl &= fl - 1u;
      list[(size_t)il * FIX_ROWS + atomicAdd(&cnt[il], 1u)] = (uint16_t)r;
    }
  }
  if (!__syncthreads_or(any != 0u)) return;
  if (p.stat && tid < 32 && tid < mg && cnt[tid]) atomicAdd(p.stat, (unsigned long long)cnt[tid]);
  // the norm table (canonical chain s = 0..sub-1 from +0; +inf for centroids >= h) of the launch's table image
  const float *sa_img = reinterpret_cast<const float *>(reinterpret_cast<const uint4 *>(p.image) + (size_t)mg * NT * SplitShape<SUB>::NPIECE * 64);
  for (int idx = tid; idx < mg * NT * 32; idx += FIX_THREADS) saL[idx] = sa_img[idx];
  __syncthreads();

  int item = 0;       // items are numbered sub-quantizer by sub-quantizer; wavefront w takes items w, w + 8, ...
  for (int il = 0; il < mg; ++il) {
    const int ne = (int)cnt[il];
    const int i = p.i0 + il;
    for (int e0 = 0; e0 < ne; e0 += 32, ++item) {
      if ((item & (FIX_THREADS / 64 - 1)) != wave) continue;
      const int e = min(e0 + j, ne - 1);                  // (lanes past the end repeat the last row; nothing is stored for them)
      const int64_t row = row_base + list[(size_t)il * FIX_ROWS + e];
      const int best_i = exact_item<SUB, NT>(p, i, row, reinterpret_cast<const float4 *>(saL + ((size_t)il * NT * 2 + hi) * 16));
      if (hi == 0 && e0 + j < ne) p.codes[(size_t)row * m + i] = (uint8_t)best_i;
    }
  }
}

static thread_local unsigned long long g_enc_stats[2] = {0, 0};
void last_encode_stats(unsigned long long out[2]) { out[0] = g_enc_stats[0]; out[1] = g_enc_stats[1]; }

template <int SUB, int NT, int NWAVES>
static int launch_encode_filter(EncParams p, int num_cu, hipStream_t stream) {
  p.NT = NT;
  DeviceLock launch_lock;      // flags scratch (per device and stream) is written by the filter and read by the exact pass
  constexpr int NPIECE = SplitShape<SUB>::NPIECE;
  const size_t per_sub = (size_t)NT * NPIECE * 64 * 16 + (size_t)NT * 32 * sizeof(float) + sizeof(float);
  const size_t budget = 160 * 1024 - 64;
  const int gmax = (int)std::min<size_t>(std::min<size_t>(budget / per_sub, (size_t)p.m), 32);
  if (gmax < 1) return fail(RQ_EUNSUPPORTED, "split encode: one sub-codebook needs %zu B of LDS", per_sub);
  // Rows go through in pieces of ENC_CHUNK_ROWS (4 Mi rows: 16 MiB of flags), so the library's scratch for this path is bounded
  // whatever n is (an encode of 1e8 rows would otherwise keep 400 MB of flags per device and stream); a 1e6-row call is one piece.
  const int64_t rows_all = p.n;
  // rows per workgroup of the exact pass: as many as the LDS lists allow (2 bytes per row and sub-quantizer of the launch group)
  {
    const size_t fixed = (size_t)gmax * NT * 32 * sizeof(float) + 32 * sizeof(uint32_t);
    int64_t fr = ((int64_t)(160 * 1024 - 256) - (int64_t)fixed) / ((int64_t)gmax * 2);
    fr = std::min<int64_t>(FIX_ROWS_MAX, fr / 32 * 32);
    if (p.n < 500000) fr = std::min<int64_t>(fr, FIX_ROWS_MAX / 4);      // short inputs: more, smaller workgroups (latency)
    if (fr < 32) return fail(RQ_EUNSUPPORTED, "split encode: the exact pass's lists do not fit LDS (m=%d)", p.m);
    p.fix_rows = (int)fr;
  }
  const int FIX_ROWS = p.fix_rows;
  const int64_t piece = std::max<int64_t>(FIX_ROWS, (int64_t)tuning("ENC_CHUNK_ROWS", 1 << 22));
  void *fl = nullptr;
  const size_t img_bytes = (filter_image_bytes<SUB>(gmax, NT) + 255) & ~(size_t)255;
  RQ_TRY(workspace(WS_ENCFLAG, img_bytes + (size_t)std::min(rows_all, piece) * sizeof(uint32_t), &fl, stream));
  p.image = static_cast<unsigned char *>(fl);
  p.flags = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(fl) + img_bytes);
  // ENC_STATS = 1 (tests, DESIGN's figures): count the pairs that take the exact pass; costs a synchronous read-back
  p.stat = nullptr;
  const bool want_stats = tuning("ENC_STATS", 0) != 0;
  void *stat_dev = nullptr;
  if (want_stats) {
    RQ_TRY(workspace(WS_COUNTER, WS_COUNTER_BYTES, &stat_dev, stream));
    p.stat = reinterpret_cast<unsigned long long *>(static_cast<unsigned char *>(stat_dev) + 128);
    RQ_HIP(hipMemsetAsync(p.stat, 0, 8, stream));
  }
  auto kern = p.dbg_w ? encode_pq_filter_kernel<SUB, NT, NWAVES, true> : encode_pq_filter_kernel<SUB, NT, NWAVES, false>;
  const float *X_all = p.X;
  uint8_t *codes_all = p.codes;
  float *dbg_all = p.dbg_w;
  for (int i0 = 0; i0 < p.m; i0 += gmax) {
    p.i0 = i0;
    p.i1 = std::min(p.m, i0 + gmax);
    const size_t lds = filter_image_bytes<SUB>(p.i1 - p.i0, NT);
    hipLaunchKernelGGL((encode_tables_kernel<SUB, NT>), dim3(p.i1 - p.i0), dim3(256), 0, stream, p);
    RQ_HIP(hipGetLastError());
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const size_t fix_lds = (size_t)(p.i1 - p.i0) * NT * 32 * sizeof(float) + 32 * sizeof(uint32_t) + (size_t)(p.i1 - p.i0) * FIX_ROWS * sizeof(uint16_t);
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(encode_pq_fix_kernel<SUB, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fix_lds));
    for (int64_t r0 = 0; r0 < rows_all; r0 += piece) {
      p.n = std::min(piece, rows_all - r0);
      p.X = X_all + (size_t)r0 * p.d;
      p.codes = codes_all + (size_t)r0 * p.m;
      p.dbg_w = dbg_all ? dbg_all + (size_t)r0 * p.m * p.h : nullptr;
      const int64_t ntiles = (p.n + 31) / 32;
      const int grid = (int)std::min<int64_t>(num_cu, (ntiles + NWAVES - 1) / NWAVES);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, stream, p);
      RQ_HIP(hipGetLastError());
      hipLaunchKernelGGL((encode_pq_fix_kernel<SUB, NT>), dim3((unsigned)((p.n + FIX_ROWS - 1) / FIX_ROWS)), dim3(FIX_THREADS), fix_lds, stream, p);
      RQ_HIP(hipGetLastError());
    }
  }
  p.n = rows_all;
  if (want_stats) {
    unsigned long long fl_pairs = 0;
    RQ_HIP(hipMemcpyAsync(&fl_pairs, p.stat, 8, hipMemcpyDeviceToHost, stream));
    RQ_HIP(hipStreamSynchronize(stream));
    g_enc_stats[0] = (unsigned long long)p.n * (unsigned long long)p.m;
    g_enc_stats[1] = fl_pairs;
  }
  return RQ_OK;
}

int encode_filter_launch(const EncParams &p, int sub, int nt, int waves, int num_cu, hipStream_t stream) {
#define RQ_FILTER_NT(SUBV, NW)                                                        \
  do {                                                                                \
    if (nt <= 1) return launch_encode_filter<SUBV, 1, NW>(p, num_cu, stream);           \
    if (nt <= 2) return launch_encode_filter<SUBV, 2, NW>(p, num_cu, stream);           \
    if (nt <= 4) return launch_encode_filter<SUBV, 4, NW>(p, num_cu, stream);           \
    return launch_encode_filter<SUBV, 8, NW>(p, num_cu, stream);                        \
  } while (0)
#define RQ_FILTER_CASE(SUBV)                                                          \
  if (sub == SUBV) {                                                                  \
    if (waves == 8) RQ_FILTER_NT(SUBV, 8);                                            \
    if (waves == 12 || (waves == 0 && SUBV > 8)) RQ_FILTER_NT(SUBV, 12);              \
    RQ_FILTER_NT(SUBV, 16);                                                           \
  }
  RQ_FILTER_CASE(2) RQ_FILTER_CASE(4) RQ_FILTER_CASE(6) RQ_FILTER_CASE(8)
  RQ_FILTER_CASE(10) RQ_FILTER_CASE(12) RQ_FILTER_CASE(14) RQ_FILTER_CASE(16)
#undef RQ_FILTER_CASE
#undef RQ_FILTER_NT
  return fail(RQ_EUNSUPPORTED, "filter encode covers even sub-space widths up to 16; got %d", sub);
}

}  // namespace rq
