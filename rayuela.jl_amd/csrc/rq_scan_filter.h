// rq_scan_filter.h -- the integer pre-filter: byte tables (build_qtab), the alive test, exact evaluation per (row, query) pair
// Part of the ADC scan (rq_scan.hip); device code only, included by that file alone.
#pragma once
#include "rq_scan_tables.h"

namespace rq {

// ------------------------------------------------------------------------------------------
// Integer pre-filter (M = 8 tiles, LUT modes with entries >= 0).
//
// 63 % of the LDS cycles of the exact loop are bank-conflict replays of 16-byte gathers, and only ~0.3 % of
// the (row, query) pairs survive the threshold.  So the hot loop first evaluates a LOWER BOUND of every
// distance from a table of one BYTE per (sub-quantizer, code, query) -- ONE 8-byte gather serves the 8 queries
// of the group, and one v_add_u32 accumulates 4 of them -- and only rows whose bound can still beat tau for
// some query are queued (row id, per wavefront) for the exact f32 evaluation above, 64 queued rows at a time.
//
//   entry  e_q[k][r] = min( floor( (T_q[k][r] - min_r T_q[k][.]) * inv_q ), CLAMP ),   CLAMP * (sub-quantizers per byte sum) <= 255
//   inv_q  a little BELOW  THR / (tau_q (1 + 2^-18) - sum_k min_k (1 - 2^-19))
//   pass   sum_k e_q[k][b_k] <= THR            (no byte sum can wrap)
//
// Soundness (every row with f32 distance d < tau passes): the real sum S of the M <= 16 table entries is within
// 15 * 2^-24 relative of the sequential f32 sum d (all entries >= 0), so S < tau (1 + 0.9e-6); the margins in
// inv_q dominate every rounding of its own computation and of (T - min) * inv (accounting in build_qtab), so the
// computed entry never exceeds the real (T - min) * THR / range with range >= S - sum_k min_k, the real sum of
// those is <= THR, and the integer sum of their floors is <= THR.  Clamping only lowers entries.
// The filter therefore passes a SUPERSET of {d <= tau} (the margins are strict); the exact evaluation decides, so results
// do not change.
// ------------------------------------------------------------------------------------------
// Byte accumulators.  M = 8, FINE kernels (chosen for k >= 8192): TWO sets of 4 sub-quantizers with 6-bit entries (4 * 63 <= 255) and
// THR8 = 191 -- half the quantisation step of one set of 8 with 5-bit entries and THR 95, same relative clamp (1/3 of
// the range): 27 % fewer rows reach the exact evaluation (first block at SIFT1M shape: 7.0 -> 5.1 % of the rows at
// K = 1000, 25.8 -> 19.9 % at K = 10000; K = 10000 7.38 -> 6.79 ms, K <= 1000 within 1 %; THR 159 / 223 / 255 are level
// or worse: beyond 191 the clamp bites).  A + B <= THR is tested on the per-byte AVERAGE, which needs no wider
// fields: floor((A + B) / 2) = (A & B) + (((A ^ B) >> 1) & 0x7f..) <= (THR - 1) / 2 for odd THR.
// M = 16: two sets of 8 (8 * 31 <= 255), compared against THR16 = 159 through their per-byte average (<= 79).
constexpr uint32_t FILT_CLAMP = 31;
#ifndef RQ_FILT_THR8
#define RQ_FILT_THR8 95
#endif
constexpr uint32_t filt_thr8(bool fine) { return fine ? 191u : (uint32_t)RQ_FILT_THR8; }
// Round 5 (m = 8, coarse tables): the entries of sub-quantizer 0 carry an OFFSET of 127 - THR, so that "sum <= THR" IS bit 7
// of the byte sum -- the alive test needs no compare arithmetic (5 of the hot loop's 25 VALU instructions per row) -- at the
// price of a lower per-entry clamp: 8 * clamp + offset <= 255 keeps the byte sums from carrying into their neighbours
// (THR 95: offset 32, clamp 27 instead of 31; clamping only lowers entries, i.e. lets a few more rows through).
#ifndef RQ_FILT_OFFSET
#define RQ_FILT_OFFSET 1
#endif
constexpr uint32_t filt_off8(bool fine) { return (!fine && RQ_FILT_OFFSET && RQ_FILT_THR8 < 127) ? 127u - (uint32_t)RQ_FILT_THR8 : 0u; }
constexpr bool filt_bit7(bool fine) { return !fine && (filt_off8(false) != 0u || RQ_FILT_THR8 == 127); }
// Round 6, the FINE tables (two sets of 4, k >= 8192): the same trick on the per-byte AVERAGE of the two sums -- the first
// sub-quantizer of BOTH sets carries 127 - (THR - 1) / 2 = 32, the offsets add 64 (even) to A + B, so floor((A + B) / 2) moves by
// exactly 32 and "A + B <= 191" is bit 7 of the average being clear: 6 of the 36 VALU instructions per row go, the clamp drops
// from 63 to (255 - 32) / 4 = 55 (first block at k = 10000: 19.9 -> 20.3 % of the rows alive).  Same-box A/B at SIFT1M shape,
// k = 10000, arrival order: 4.795 -> 4.725 ms against the same tree without it, 4.709 -> 4.704 ms against round 5's final build
// (tools/k10000_ab.py, tools/ab_shard.py): the instructions saved and the rows let through cancel -- level, kept for the
// shorter loop.  PQ / CQ scans only; LSQ sets hold 5 entries of <= 51.
#ifndef RQ_FILT_OFFSET_FINE
#define RQ_FILT_OFFSET_FINE 1
#endif
constexpr uint32_t filt_off8_fine() { return RQ_FILT_OFFSET_FINE ? 127u - (filt_thr8(true) - 1u) / 2u : 0u; }
constexpr uint32_t filt_clamp8(bool fine) { return fine ? (filt_off8_fine() ? (255u - filt_off8_fine()) / 4u : 63u) : (filt_off8(false) ? (255u - filt_off8(false)) / 8u : 31u); }
constexpr uint32_t FILT_THR16 = 159;
// m = 16, same trick (build knob RQ_FILT_OFFSET16, OFF): the first sub-quantizer of BOTH sets would carry (127 - (THR16 - 1) / 2)
// = 48, so that the per-byte average of the two sums is <= (THR16 - 1) / 2 exactly when its bit 7 is clear (the offsets add 96,
// an even number, to A + B: the floor of the average moves by exactly 48); clamp (255 - 48) / 8 = 25 instead of 31.  Measured
// at Deep1M shape: k = 100 level (4.16 ms), k = 1000 4.93 vs 4.86 ms -- at m = 16 the lower clamp costs more rows than the
// six instructions per row buy.  m = 8 (shipped): k = 1 1.458 vs 1.504 ms, k = 100 1.553 vs 1.596, k = 1000 2.019 vs 2.026.
#ifndef RQ_FILT_OFFSET16
#define RQ_FILT_OFFSET16 0
#endif
constexpr uint32_t filt_off16() { return RQ_FILT_OFFSET16 ? 127u - (FILT_THR16 - 1u) / 2u : 0u; }

// (byte k of w) << SH in ONE VALU instruction (SDWA operand select; the compiler emits v_bfe_u32 + v_lshl_add_u32)
template <int K, int SH>
__device__ __forceinline__ uint32_t byte_shl(uint32_t w, uint32_t sh_reg) {
  uint32_t r;
  if constexpr (K == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(sh_reg), "v"(w));
  else if constexpr (K == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(sh_reg), "v"(w));
  else if constexpr (K == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(sh_reg), "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(sh_reg), "v"(w));
  return r;
}

// load from an ABSOLUTE LDS byte address (no symbol involved: constant parts fold into the instruction's offset field)
template <class T> __device__ __forceinline__ T lds_abs_load(uint32_t addr);
template <> __device__ __forceinline__ uint2 lds_abs_load<uint2>(uint32_t addr) {
  typedef const unsigned long long __attribute__((address_space(3))) lds_u64_t;
  const unsigned long long v = *(lds_u64_t *)(uintptr_t)addr;
  return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
template <> __device__ __forceinline__ uint4 lds_abs_load<uint4>(uint32_t addr) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef const u32x4 __attribute__((address_space(3))) lds_u128_t;
  const u32x4 v = *(lds_u128_t *)(uintptr_t)addr;
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ uint32_t lds_abs_load<uint32_t>(uint32_t addr) {
  typedef const uint32_t __attribute__((address_space(3))) lds_u32_t;
  return *(lds_u32_t *)(uintptr_t)addr;
}

// dword j of a byte-table entry (4 queries per dword)
__device__ __forceinline__ uint32_t fv_word(const uint2 &v, int j) { return j == 0 ? v.x : v.y; }
__device__ __forceinline__ uint32_t fv_word(const uint4 &v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

template <int M> struct FiltVec;               // table entry: one byte per query of the group
#if RQ_QG16
template <> struct FiltVec<8> { using type = uint4; };    // experiment: 16 queries per ds_read_b128
#else
template <> struct FiltVec<8> { using type = uint2; };    // 8 queries per ds_read_b64
#endif
template <> struct FiltVec<16> { using type = uint2; };
template <> struct FiltVec<4> { using type = uint2; };

// per-entry clamp of the byte tables: (entries per byte sum) * clamp <= 255.  LSQ scans add the row-norm entry to the
// LAST byte sum: 5 entries of <= 51 at m = 8 (two sets of 4 and 4 + 1), 9 of <= 28 at m = 16 (sets of 8 and 8 + 1)
template <int M, bool FINE, bool LSQ>
constexpr uint32_t filt_clamp() {
  return LSQ ? (M <= 8 ? 51u : 28u)
             : (M == 4 ? (filt_off8(false) ? (255u - filt_off8(false)) / 4u : 63u)      // m = 4: ONE set of 4 entries (round 6)
                : M == 8 ? filt_clamp8(FINE) : (M == 16 && filt_off16() ? (255u - filt_off16()) / 8u : FILT_CLAMP));
}

// row norm -> its quantisation cell's LOWER edge, with exactly these two rounded operations (the quantiser checks its
// choice against the same expression, so EDGE(byte of a row) <= the row's norm holds in exact arithmetic)
__device__ __forceinline__ float norm_edge(float nmin, float nstep, uint32_t b) {
  const float t = (float)b * nstep;
  return nmin + t;
}

template <int M, bool FINE, bool LSQ = false>
__device__ __forceinline__ void build_qtab(ScanCtrl<ScanCfg<M>::QG> *ctrl, const float4 *lut4, const float4 *gtab4,
                                           uint32_t *qtab, int tid, const float *norm_info = nullptr,
                                           const float *cnorm = nullptr) {
  using Cfg = ScanCfg<M>;
  constexpr int QG = Cfg::QG, NQUAD = Cfg::NQUAD, KL = Cfg::KL;
  static_assert(Cfg::QPG == 4, "pre-filter tiling: float4 table entries");
  constexpr float THR = (float)(M <= 8 ? filt_thr8(FINE) : FILT_THR16);
  const int wave = tid >> 6, lane = tid & 63;
  auto entry = [&](int kk, int quad, int r) -> float4 {
    float4 v = kk < KL ? lut4[(kk * NQUAD + quad) * 256 + r] : gtab4[((kk - KL) * NQUAD + quad) * 256 + r];
    if constexpr (LSQ) {
      // LSQ: T = -2 <q, c> and the row adds |x_hat|^2.  Bounding the two separately is useless (the centroid with the
      // largest <q, c> also has a large norm: 76 % of the rows stayed alive); so the filter works on
      //   T'_k[r] = T_k[r] + |c_k[r]|^2      and      rho(row) = norm(row) - sum_k |c_k[b_k]|^2   (the cross terms),
      // whose sum is the same distance in exact arithmetic.  rho is what the row byte quantises.
      const float cn = cnorm[kk * 256 + r];
      v.x = v.x + cn; v.y = v.y + cn; v.z = v.z + cn; v.w = v.w + cn;
    }
    return v;
  };
  // 1. minima of the 256 entries of (k, q): one wavefront per sub-quantizer, lane handles r = lane, lane + 64, ...
  for (int k = wave; k < M; k += ScanCfg<M>::THREADS / 64) {
    float mn[QG], mx[LSQ ? QG : 1];
#pragma unroll
    for (int q = 0; q < QG; ++q) mn[q] = __uint_as_float(0x7f800000u);
    if constexpr (LSQ) {
#pragma unroll
      for (int q = 0; q < QG; ++q) mx[q] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int quad = 0; quad < NQUAD; ++quad) {
        const float4 v = entry(k, quad, lane + 64 * i);
        mn[quad * 4 + 0] = fminf(mn[quad * 4 + 0], v.x);
        mn[quad * 4 + 1] = fminf(mn[quad * 4 + 1], v.y);
        mn[quad * 4 + 2] = fminf(mn[quad * 4 + 2], v.z);
        mn[quad * 4 + 3] = fminf(mn[quad * 4 + 3], v.w);
        if constexpr (LSQ) {
          mx[quad * 4 + 0] = fmaxf(mx[quad * 4 + 0], fabsf(v.x));
          mx[quad * 4 + 1] = fmaxf(mx[quad * 4 + 1], fabsf(v.y));
          mx[quad * 4 + 2] = fmaxf(mx[quad * 4 + 2], fabsf(v.z));
          mx[quad * 4 + 3] = fmaxf(mx[quad * 4 + 3], fabsf(v.w));
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int q = 0; q < QG; ++q) mn[q] = fminf(mn[q], __shfl_xor(mn[q], off));
      if constexpr (LSQ) {
#pragma unroll
        for (int q = 0; q < QG; ++q) mx[q] = fmaxf(mx[q], __shfl_xor(mx[q], off));
      }
    }
    float mc = 0.0f;      // LSQ: max_r |c_k[r]|^2 -- the ORIGINAL magnitudes are bounded by |T| <= |T'| + |c|^2, |norm| <= |rho| + sum |c|^2
    if constexpr (LSQ) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mc = fmaxf(mc, cnorm[k * 256 + lane + 64 * i]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mc = fmaxf(mc, __shfl_xor(mc, off));
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < QG; ++q) ctrl->fmin[k][q] = mn[q];
      if constexpr (LSQ) {
#pragma unroll
        for (int q = 0; q < QG; ++q) ctrl->fmax[k][q] = mx[q] + 2.0f * mc;
      }
    }
  }
  __syncthreads();
  float nmin = 0.0f, nstep = 0.0f;
  if constexpr (LSQ) { nmin = norm_info[0]; nstep = norm_info[1]; }
  if constexpr (LSQ) {
    // LSQ tables are signed (-2 <q, c>) and every row adds its norm.  Shifted by their minima the entries are >= 0 again;
    // the norm enters as one more table, indexed by the row's norm BYTE, whose entry is the cell's lower edge -- a lower
    // bound of it.  Rounding is accounted for ABSOLUTELY: with A >= sum_k max|T_k| + max|norm| (original magnitudes,
    // bounded through the folded ones: fmax = max|T'| + 2 max|c|^2), the sequential f32
    // distance of a row (M + 1 <= 17 terms) is within 17u A < 2^-19.9 A of the real sum, so is the f32 sum of the minima,
    // and (tau - base) itself rounds by 2^-24 |tau - base|.  The margin 2^-16 A + 2^-18 |tau - base| covers the three
    // seven times over; A is a few ranges, so it costs < 1e-3 of a filter step.
    if (tid < QG) {
      float base = nmin, A = fabsf(norm_info[2]);
      for (int kk = 0; kk < M; ++kk) { base = base + ctrl->fmin[kk][tid]; A = A + ctrl->fmax[kk][tid]; }
      const float tau = ctrl->tau[tid];
      const float gap = tau - base;
      const float range = gap + (A * 1.52587890625e-5f + fabsf(gap) * 3.814697265625e-6f);
      float inv = 0.0f;     // 0: the filter passes everything for this query
      if (tau < __uint_as_float(0x7f800000u) && range > 0.0f && A < __uint_as_float(0x7f800000u)) {
        const float step = range / THR;
        const float cand = (1.0f / step) * (1.0f - 1.9073486328125e-6f);
        if (step > 0.0f && cand < __uint_as_float(0x7f800000u)) inv = cand;
      }
      ctrl->finv[tid] = inv;
    }
  } else

  if (tid < QG) {
    float base = 0.0f;
    for (int kk = 0; kk < M; ++kk) base = base + ctrl->fmin[kk][tid];
    const float tau = ctrl->tau[tid];
    // Margins: 2^-18 on tau, 2^-19 on the minima and on 1/step.  What they have to cover (u = 2^-24): the sequential
    // f32 sum of M <= 16 non-negative terms is within 15u/(1-15u) < 0.9e-6 of the real sum -- for the row's distance
    // (so S < tau (1 + 0.9e-6) whenever d < tau) and for `base` against the real sum of the minima -- plus one rounding
    // for each of the two products, the subtraction, the division, the reciprocal and its product (< 7u = 0.42e-6 in
    // all).  2^-18 = 3.8e-6 and 2^-19 = 1.9e-6 leave a factor of two everywhere; their cost is nil (one filter step
    // is range / THR, i.e. 1e-2 of the range).
    const float range = tau * (1.0f + 3.814697265625e-6f) - base * (1.0f - 1.9073486328125e-6f);
    float inv = 0.0f;     // 0: every entry quantises to 0, i.e. the filter passes everything for this query
    if (tau < __uint_as_float(0x7f800000u) && range > 0.0f && base >= 0.0f) {
      const float step = range / THR;
      const float cand = (1.0f / step) * (1.0f - 1.9073486328125e-6f);
      if (step > 0.0f && cand < __uint_as_float(0x7f800000u)) inv = cand;
    }
    ctrl->finv[tid] = inv;
  }
  __syncthreads();
  // 2. one byte per (k, r, query): QG bytes per entry
  for (int e = tid; e < M * 256; e += ScanCfg<M>::THREADS) {
    const int kk = e >> 8, r = e & 255;
#pragma unroll
    for (int quad = 0; quad < NQUAD; ++quad) {
      const float4 v = entry(kk, quad, r);
      const float t[4] = {v.x, v.y, v.z, v.w};
      uint32_t w = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float diff = t[c] - ctrl->fmin[kk][quad * 4 + c];
        const float x = diff * ctrl->finv[quad * 4 + c];
        w |= (uint32_t)fminf(fmaxf(x, 0.0f), (float)filt_clamp<M, FINE, LSQ>()) << (8 * c);   // float -> uint truncates = floor (x >= 0)
      }
      if constexpr (M <= 8 && !LSQ && filt_off8(FINE) != 0u) {
        if (kk == 0) w += filt_off8(FINE) * 0x01010101u;         // (clamp + offset <= 255: no carry between the bytes)
      }
      if constexpr (M == 8 && !LSQ && FINE && filt_off8_fine() != 0u) {
        if ((kk & 3) == 0) w += filt_off8_fine() * 0x01010101u;  // first sub-quantizer of each of the two sets of 4
      }
      if constexpr (M == 16 && !LSQ && filt_off16() != 0u) {
        if ((kk & 7) == 0) w += filt_off16() * 0x01010101u;      // first sub-quantizer of each of the two sets
      }
      qtab[e * NQUAD + quad] = w;
    }
  }
  if constexpr (LSQ) {
    // 3. the row-norm table: entry r = the lower edge of cell r above the smallest norm, in the query's steps
    for (int r = tid; r < 256; r += ScanCfg<M>::THREADS) {
      const float diff = norm_edge(nmin, nstep, (uint32_t)r) - nmin;
#pragma unroll
      for (int quad = 0; quad < NQUAD; ++quad) {
        uint32_t w = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x = diff * ctrl->finv[quad * 4 + c];
          w |= (uint32_t)fminf(fmaxf(x, 0.0f), (float)filt_clamp<M, FINE, LSQ>()) << (8 * c);
        }
        qtab[(M * 256 + r) * NQUAD + quad] = w;
      }
    }
  }
}

// "Can this row still beat a threshold?" from the byte sums of the group's queries.
//   M = 8 : a[0], a[1] (and a[2], a[3] for the second set; then s = their per-byte average, T = (THR - 1) / 2) hold 8 byte
//           sums s <= 252.  ((s | 0x80) - (T+1)) has bit 7 set iff (s & 0x7f) > T, and any s >= 0x80 is > T as well; no
//           borrow crosses a byte because (s | 0x80) >= T + 1.
//   M = 16: two sets of 4 byte sums; per query A + B <= THR16  <=>  their per-byte average <= (THR16 - 1) / 2, same trick.
template <int M, bool FINE, bool LSQ = false>
__device__ __forceinline__ bool filt_alive(const uint32_t (&a)[ScanCfg<M>::NACC * ScanCfg<M>::NQUAD]) {
  if constexpr (M <= 8) {
    // NQUAD dwords of 4 byte sums per set; FINE: two sets (k < 4, k >= 4), compared through their per-byte average
    constexpr int NQ = ScanCfg<M>::NQUAD;
    constexpr uint32_t H = 0x80808080u;
    constexpr uint32_t TC = (FINE ? (filt_thr8(true) - 1u) / 2u + 1u : filt_thr8(false) + 1u) * 0x01010101u;
    uint32_t all = H;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      uint32_t v = a[j];
      if constexpr (FINE) v = (a[j] & a[NQ + j]) + (((a[j] ^ a[NQ + j]) >> 1) & 0x7f7f7f7fu);
      // THR = 127 (coarse tables): "sum <= THR" IS bit 7 of the byte sum -- no compare arithmetic at all
      if constexpr ((filt_bit7(FINE) && !LSQ) || (FINE && !LSQ && filt_off8_fine() != 0u)) all &= v;
      else all &= ((v | H) - TC) | v;
    }
    return (all & H) != H;
  } else if constexpr (ScanCfg<M>::NQUAD == 2) {
    // two sets (k < 8, k >= 8) of 8 byte sums, each <= 248: A + B <= THR  <=>  floor((A + B) / 2) <= (THR - 1) / 2 for
    // odd THR, and the per-byte average needs no wider fields: (A & B) + (((A ^ B) >> 1) & 0x7f..)
    static_assert(FILT_THR16 % 2 == 1 && (FILT_THR16 - 1) / 2 < 128, "average trick");
    constexpr uint32_t H = 0x80808080u, TC = ((FILT_THR16 - 1u) / 2u + 1u) * 0x01010101u;
    const uint32_t v0 = (a[0] & a[2]) + (((a[0] ^ a[2]) >> 1) & 0x7f7f7f7fu);
    const uint32_t v1 = (a[1] & a[3]) + (((a[1] ^ a[3]) >> 1) & 0x7f7f7f7fu);
    if constexpr (M == 16 && !LSQ && filt_off16() != 0u) return (v0 & v1 & H) != H;      // offset tables (PQ / CQ scans only)
    const uint32_t g0 = ((v0 | H) - TC) | v0, g1 = ((v1 | H) - TC) | v1;
    return (g0 & g1 & H) != H;
  } else {
    // two sets of 4 byte sums, each <= 248: the per-byte average again (no 16-bit widening: 10 instead of 15 VALU)
    static_assert(FILT_THR16 % 2 == 1 && (FILT_THR16 - 1) / 2 < 128, "average trick");
    constexpr uint32_t H = 0x80808080u, TC = ((FILT_THR16 - 1u) / 2u + 1u) * 0x01010101u;
    const uint32_t v = (a[0] & a[1]) + (((a[0] ^ a[1]) >> 1) & 0x7f7f7f7fu);
    return ((((v | H) - TC) | v) & H) != H;
  }
}

// Exact evaluation of the queued rows, per (row, query) PAIR: of the QG queries of an alive row typically one or two passed the
// byte bound (the bound is per query, so only those can beat their tau).  Lane i recomputes the byte sums of its row
// (M gathers from the byte tables, the filter's own arithmetic), and the wavefront then walks the alive queries in
// rounds: in round j every lane that still has one takes its next alive query q, gathers the M f32 entries
// T_q[k][b_k] (4-byte gathers), sums them in the reference's order and appends the key if it beats tau_q (one LDS
// atomic per survivor).  Against evaluating whole rows (M * QG / 4 16-byte gathers per row, QG compares and ballots) this is
// ~M 4-byte gathers per alive pair.  Soundness: a pair the bound rules out has d >= tau_q (build_qtab), so skipping
// it cannot change the candidate set below tau.
__device__ __forceinline__ uint32_t high_bits4(uint32_t x) {   // bits 7, 15, 23, 31 of x -> bits 0..3
  return (((x >> 7) & 0x01010101u) * 0x00204081u >> 21) & 0xfu;
}

template <int M, bool FINE, bool LSQ = false>
__device__ __forceinline__ uint32_t filt_alive_bits(const uint32_t (&a)[ScanCfg<M>::NACC * ScanCfg<M>::NQUAD]) {
  using Cfg = ScanCfg<M>;
  if constexpr (M <= 8) {
    constexpr int NQ = Cfg::NQUAD;
    constexpr uint32_t H = 0x80808080u;
    constexpr uint32_t TC = (FINE ? (filt_thr8(true) - 1u) / 2u + 1u : filt_thr8(false) + 1u) * 0x01010101u;
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      uint32_t v = a[j];
      if constexpr (FINE) v = (a[j] & a[NQ + j]) + (((a[j] ^ a[NQ + j]) >> 1) & 0x7f7f7f7fu);
      if constexpr ((filt_bit7(FINE) && !LSQ) || (FINE && !LSQ && filt_off8_fine() != 0u)) bits |= high_bits4(~v) << (4 * j);
      else bits |= high_bits4(~(((v | H) - TC) | v)) << (4 * j);
    }
    return bits;
  } else if constexpr (Cfg::NQUAD == 2) {
    constexpr uint32_t H = 0x80808080u, TC = ((FILT_THR16 - 1u) / 2u + 1u) * 0x01010101u;
    const uint32_t v0 = (a[0] & a[2]) + (((a[0] ^ a[2]) >> 1) & 0x7f7f7f7fu);
    const uint32_t v1 = (a[1] & a[3]) + (((a[1] ^ a[3]) >> 1) & 0x7f7f7f7fu);
    if constexpr (M == 16 && !LSQ && filt_off16() != 0u) return (high_bits4(~v0) | (high_bits4(~v1) << 4));
    const uint32_t g0 = ((v0 | H) - TC) | v0, g1 = ((v1 | H) - TC) | v1;
    return (high_bits4(~g0) | (high_bits4(~g1) << 4));
  } else {
    constexpr uint32_t H = 0x80808080u, TC = ((FILT_THR16 - 1u) / 2u + 1u) * 0x01010101u;
    const uint32_t v = (a[0] & a[1]) + (((a[0] ^ a[1]) >> 1) & 0x7f7f7f7fu);
    return high_bits4(~(((v | H) - TC) | v));
  }
}

template <int M, bool BIAS, bool FINE>
__device__ __noinline__ void refine_pairs(ScanCtrl<ScanCfg<M>::QG> *ctrl, uint64_t *cand_wg, const uint8_t *codes,
                                          const float *row_bias, uint32_t id_offset, uint32_t cap, const float4 *lut4,
                                          const float4 *gtab, const uint32_t *qtab, const uint32_t *queue, uint32_t count,
                                          const uint8_t *norm_bytes, const uint32_t *__restrict__ perm) {
  using Cfg = ScanCfg<M>;
  constexpr int QG = Cfg::QG, NQUAD = Cfg::NQUAD, KL = Cfg::KL;
  static_assert(Cfg::QPG == 4, "pair refinement: float4 table entries");
  using FV = typename FiltVec<M>::type;
  const int lane = threadIdx.x & 63;
  const bool valid = (uint32_t)lane < count;
  const uint32_t row = queue[valid ? lane : 0];
  uint32_t w1[(M + 3) / 4];
  load_row<M>(w1, codes, row);
  // byte sums of the row, exactly as the hot loop forms them
  uint32_t a[Cfg::NACC * NQUAD];
#pragma unroll
  for (int i = 0; i < Cfg::NACC * NQUAD; ++i) a[i] = 0;
  const FV *qt = reinterpret_cast<const FV *>(qtab);
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const uint32_t byte = (w1[k >> 2] >> (8 * (k & 3))) & 0xffu;
    const FV e = qt[k * 256 + byte];
#pragma unroll
    for (int j = 0; j < NQUAD; ++j) a[(k / Cfg::kpa(FINE)) * NQUAD + j] += fv_word(e, j);
  }
  if constexpr (BIAS) {        // LSQ: the row-norm entry belongs to the last byte sum, as in the hot loop
    const FV e = qt[M * 256 + norm_bytes[row]];
#pragma unroll
    for (int j = 0; j < NQUAD; ++j) a[(Cfg::NACC - 1) * NQUAD + j] += fv_word(e, j);
  }
  uint32_t alive = valid ? filt_alive_bits<M, FINE, BIAS>(a) : 0u;
  const uint32_t selmask = __builtin_amdgcn_readfirstlane(ctrl->selmask);
  const float bias = BIAS ? row_bias[row] : 0.0f;
  const float *lutf = reinterpret_cast<const float *>(lut4);
  const float *__restrict__ gtf = reinterpret_cast<const float *>(gtab);
  while (__ballot(alive != 0u)) {
    if (alive != 0u) {
      const uint32_t q = (uint32_t)__builtin_ctz(alive);
      alive &= alive - 1u;
      const uint32_t qoff = (q >> 2) * 1024u + (q & 3u);       // float index of (quad, component) inside a k block
      float tg[Cfg::KG > 0 ? Cfg::KG : 1];
#pragma unroll
      for (int k = KL; k < M; ++k) {
        const uint32_t byte = (w1[k >> 2] >> (8 * (k & 3))) & 0xffu;
        tg[k - KL] = gtf[(uint32_t)(k - KL) * NQUAD * 1024u + qoff + byte * 4u];
      }
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < M; ++k) {
        const uint32_t byte = (w1[k >> 2] >> (8 * (k & 3))) & 0xffu;
        const float t = k < KL ? lutf[(uint32_t)k * NQUAD * 1024u + qoff + byte * 4u] : tg[k - KL];
        acc = (k == 0) ? t : acc + t;        // deps/src/linscan_aqd.cpp:85-87, sequential f32
      }
      if (BIAS) acc = acc + bias;
      if (acc <= ctrl->tau[q]) {
        const uint32_t pos = atomicAdd(&ctrl->cnt[q], 1u);
        uint64_t *buf = cand_wg + ((size_t)q * 2 + ((selmask >> q) & 1u)) * cap;
        // ordered bases: the key carries the row's ORIGINAL number -- read here, for true survivors only (a load per queued
        // row was 1.6e8 random 4-byte reads per launch on a 1.25e8-row shard: more HBM traffic than the codes)
        const uint32_t kid = (perm ? perm[row] : row) + id_offset;
        buf[pos] = make_key(acc, kid);
      }
    }
  }
}

// (the call sites sit behind run-time `filt_on` tests; kernels without the pre-filter never instantiate refine_pairs)
template <int M, bool BIAS, bool FILT, bool FINE>
__device__ __forceinline__ void refine_queue(ScanCtrl<ScanCfg<M>::QG> *ctrl, uint64_t *cand_wg, const uint8_t *codes,
                                             const float *row_bias, uint32_t id_offset, uint32_t cap, const float4 *lut4,
                                             const float4 *gtab, const uint32_t *qtab, const uint32_t *queue, uint32_t count,
                                             const uint8_t *norm_bytes, const uint32_t *perm) {
  if constexpr (FILT && ScanCfg<M>::HAS_FILT)
    refine_pairs<M, BIAS, FINE>(ctrl, cand_wg, codes, row_bias, id_offset, cap, lut4, gtab, qtab, queue, count, norm_bytes, perm);
}
#define RQ_REFINE(c, cw, cd, rb, io, cp, l4, gt, qu, n) refine_queue<M, BIAS, FILT, FINE>(c, cw, cd, rb, io, cp, l4, gt, qtab, qu, n, p.norm_bytes, p.perm)


}  // namespace rq
