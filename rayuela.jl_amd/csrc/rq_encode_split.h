// rq_encode_split.h -- pieces shared by the bf16 matrix-core filter kernels of the PQ encode (rq_encode.hip: the one-pass
// kernel with the exact re-evaluation inside; rq_encode_filter.hip: filter + separate exact pass).
#pragma once
#include "rq_internal.h"

namespace rq {

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x2 = float __attribute__((ext_vector_type(2)));

struct EncParams {
  const float *X;   // [n][d]
  const float *C;   // concat of [h][sub_i]
  uint8_t *codes;   // [n][m]
  int64_t n;
  int d, m, h, NT;
  int i0, i1;       // sub-quantizers handled by this launch (codebooks of [i0,i1) sit in LDS)
  int off[33];      // splitarray offsets (src/utils.jl:179-203)
  float delta_rel;  // split kernel: candidate margin relative to max|c|^2 + |x|^2 (SplitCfg::DELTA_REL unless tuned, tests)
  float *dbg_w;     // split kernel, tests only: [n][m][h] receives the filter's W values (nullptr in every product call)
  uint32_t *flags;  // filter + exact pass: [n] words, bit (i - i0) set = (row, sub-quantizer i) goes to the exact pass
  unsigned char *image;  // filter + exact pass: the launch's LDS table image (encode_tables_kernel, rq_encode_filter.hip)
  int fix_rows;          // exact pass: rows per workgroup (its per-sub-quantizer LDS lists hold 2 bytes per row)
  unsigned long long *stat;  // tuning ENC_STATS only: [0] += flagged (row, sub-quantizer) pairs (exact pass); nullptr otherwise
};

// lanes 32-63 of a  <->  lanes 0-31 of b   (v_permlane32_swap_b32)
__device__ __forceinline__ void swap32(float &a, float &b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// bf16 round-to-nearest-even of a finite f32 (inf stays inf); returns the 16 payload bits
__device__ __forceinline__ uint32_t bf16_bits(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_val(uint32_t b) { return __uint_as_float(b << 16); }

#ifndef RQ_SPLIT_SINGLE_ACC
#define RQ_SPLIT_SINGLE_ACC 0
#endif
struct SplitCfg {
  static constexpr float DELTA_REL = 3.0f * 6.103515625e-05f;     // 3 * 2^-14
  static constexpr float TINY = 8.673617379884035e-19f;            // 2^-60: below it bf16 flush-to-zero could matter
};

template <int SUB>
struct SplitShape {
  static_assert(SUB >= 2 && SUB <= 16 && SUB % 2 == 0, "split encode: even sub-space widths up to 16");
  static constexpr bool PACK = SUB <= 8;          // hi and lo pieces of -2c share one K = 16 fragment
  static constexpr int NPIECE = PACK ? 1 : 2;     // 16-byte A fragments per (tile, lane)
};

// canonical evaluation of centroid k of sub-quantizer `cb` (oracle/rq_oracle.c:264-328): g = fmaf chain s = 0..sub-1 from
// +0, sa = |c_k|^2 (the same chain, computed once in the prologue: sa_k), v = max(fl(fl(sa + sb) - 2g), 0); then the
// lexicographic (v, index) update.
template <int SUB>
__device__ __forceinline__ void split_exact(const float *__restrict__ cb, int k, const float (&x)[SUB], float sb, float sa,
                                            float &bv, int &bk) {
  float c[SUB];
  if constexpr (SUB % 4 == 0) {
    const float4 *c4 = reinterpret_cast<const float4 *>(cb + (size_t)k * SUB);
#pragma unroll
    for (int s4 = 0; s4 < SUB / 4; ++s4) { const float4 v = c4[s4]; c[4 * s4] = v.x; c[4 * s4 + 1] = v.y; c[4 * s4 + 2] = v.z; c[4 * s4 + 3] = v.w; }
  } else {
    const f32x2 *c2 = reinterpret_cast<const f32x2 *>(cb + (size_t)k * SUB);
#pragma unroll
    for (int s2 = 0; s2 < SUB / 2; ++s2) { const f32x2 v = c2[s2]; c[2 * s2] = v.x; c[2 * s2 + 1] = v.y; }
  }
  float g = 0.0f;
#pragma unroll
  for (int sx = 0; sx < SUB; ++sx) g = __builtin_fmaf(c[sx], x[sx], g);
  const float t = sa + sb;
  const float u = __builtin_fmaf(-2.0f, g, t);
  const float v = __builtin_fmaxf(u, 0.0f);
  if (v < bv || (v == bv && k < bk)) { bv = v; bk = k; }
}

// bit r of the result: v[r] <= thr  (v_cmp + v_addc per value: the carry shifts into the mask).  Only for values that
// were produced by ordinary VALU instructions (see the note at the tile re-run below).
__device__ __forceinline__ uint32_t mask_leq16(const f32x16 &v, float thr) {
  uint32_t cm = 0;
#pragma unroll
  for (int r = 15; r >= 0; --r)
    asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(cm) : "v"(v[r]), "v"(thr) : "vcc");
  return cm;
}


// ---- canonical evaluation on the f32 matrix cores (direct kernels of rq_encode.hip, exact pass of rq_encode_filter.hip) ----
// One 32-centroid x 32-vector tile of inner products: KS chained 32x32x2 MFMAs (= the fmaf chain
// s = 0..2KS-1 of the oracle; padded k-steps multiply 0 by 0 and leave the chain untouched).
template <int KS>
__device__ __forceinline__ f32x16 tile_dots(const float *cb_tile, const float (&b)[KS]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cb_tile[kk * 64], b[kk], acc, 0, 0, 0);
  return acc;
}

// Epilogue of one tile on the lane's 16 accumulator registers (16 centroids of ONE vector):
//   u_r = fl(fl(sa_r + sb) - 2 g_r)           (fma(-2, g, t) has the same bits: 2g is exact)
//   the tile's clamped minimum is cm = max(min_r u_r, 0).
// f32 MFMA and f32 VALU share the SIMD's FP32 lanes on gfx950 (measured: busy cycles add up, they
// do not overlap), so every VALU instruction here is paid in full.  Per tile we therefore only
// keep the running best value and, for the lanes that improved (strict '<': the earliest tile wins
// ties), a copy of the tile's 16 u values (one shared mask, 16 v_cndmask).  The first-index search
// runs ONCE per sub-quantizer on that copy (argmin_finish) instead of once per tile.

struct ArgminState {
  float best_v;   // clamped minimum so far
  int best_t;     // tile that holds it
  f32x2 ub[8];    // that tile's 16 u values (register pairs, as the packed ops leave them)
};

__device__ __forceinline__ void tile_argmin(const f32x16 &acc, const float4 *sa4, float sb, int t,
                                            ArgminState &st) {
  // packed f32 arithmetic (v_pk_add_f32 / v_pk_fma_f32): two elements per VALU issue
  f32x2 u[8];
  sa4 = reinterpret_cast<const float4 *>(__builtin_assume_aligned(sa4, 16));
  const f32x2 sb2 = {sb, sb};
  const f32x2 m2 = {-2.0f, -2.0f};
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 sav = sa4[g4];
    const f32x2 s01 = {sav.x, sav.y}, s23 = {sav.z, sav.w};
    const f32x2 g01 = {acc[g4 * 4 + 0], acc[g4 * 4 + 1]}, g23 = {acc[g4 * 4 + 2], acc[g4 * 4 + 3]};
    u[g4 * 2 + 0] = __builtin_elementwise_fma(m2, g01, s01 + sb2);
    u[g4 * 2 + 1] = __builtin_elementwise_fma(m2, g23, s23 + sb2);
  }
  float m01 = __builtin_fminf(__builtin_fminf(u[0].x, u[0].y), u[1].x);
  float m02 = __builtin_fminf(__builtin_fminf(u[1].y, u[2].x), u[2].y);
  float m03 = __builtin_fminf(__builtin_fminf(u[3].x, u[3].y), u[4].x);
  float m04 = __builtin_fminf(__builtin_fminf(u[4].y, u[5].x), u[5].y);
  float m05 = __builtin_fminf(__builtin_fminf(u[6].x, u[6].y), u[7].x);
  float mm = __builtin_fminf(__builtin_fminf(m01, m02), m03);
  mm = __builtin_fminf(__builtin_fminf(mm, m04), m05);
  mm = __builtin_fminf(mm, u[7].y);
  const float cm = __builtin_fmaxf(mm, 0.0f);
  // a real (exec-masked) branch: f32 MFMA and VALU do not overlap on this chip, so there is nothing to
  // interleave the copy with, and under the mask it is 8 64-bit moves instead of 16 selects
  if (cm < st.best_v) {
    st.best_v = cm;
    st.best_t = t;
#pragma unroll
    for (int r = 0; r < 8; ++r) st.ub[r] = u[r];
  }
}

// First index of the clamped minimum inside the winning tile: the first r with u_r <= cm (for
// cm > 0 that is the first minimum; for cm == 0 the first value the reference's max(.,0) clamps).
__device__ __forceinline__ int argmin_finish(const ArgminState &st, int hi) {
  int rf = 15;
#pragma unroll
  for (int r = 14; r >= 0; --r) rf = (((r & 1) ? st.ub[r >> 1].y : st.ub[r >> 1].x) <= st.best_v) ? r : rf;
  return st.best_t * 32 + 4 * hi + 8 * (rf >> 2) + (rf & 3);
}

// rq_encode_filter.hip: filter launch + exact pass per group of sub-quantizers that fits LDS (even widths <= 16)
int encode_filter_launch(const EncParams &p, int sub, int nt, int waves, int num_cu, hipStream_t stream);

}  // namespace rq
