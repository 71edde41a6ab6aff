// rq_scan_tables.h -- tiling constants (ScanCfg), per-item state, f32 look-up tables and the exact row evaluation
// Part of the ADC scan (rq_scan.hip); device code only, included by that file alone.
#pragma once
#include "rq_internal.h"
#include "rq_topk.h"

#ifndef RQ_SCAN_PACE_BUILD
#define RQ_SCAN_PACE_BUILD 0
#endif
// EXPERIMENT build (EXPERIMENTS.md 9.3): m = 8 scans with 16 queries per group -- byte tables of 16 queries read by ds_read_b128,
// one 1024-thread workgroup per CU, half of the f32 tables through L1.  tools/build_variant.sh qg16 rq_scan.hip "-DRQ_QG16=1"
#ifndef RQ_QG16
#define RQ_QG16 0
#endif

namespace rq {


constexpr int SCAN_THREADS = 512;       // two workgroups per CU (ScanCfg<M>::THREADS is 1024 where one group needs more than half of the LDS)

// v_writelane_b32 (no clang builtin in ROCm 7.2): lane `L` of `old` <- wave-uniform `val`
template <class T>
__device__ __forceinline__ uint32_t writelane_u32(uint32_t old, T val, int L) {
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(L));
  return old;
}

// LUT entry of QPG queries
template <int QPG> struct LutVec;
template <> struct LutVec<4> {
  using type = float4;
  static __device__ __forceinline__ type make(const float *a) { return make_float4(a[0], a[1], a[2], a[3]); }
  static __device__ __forceinline__ float get(const type &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};
template <> struct LutVec<2> {
  using type = float2;
  static __device__ __forceinline__ type make(const float *a) { return make_float2(a[0], a[1]); }
  static __device__ __forceinline__ float get(const type &v, int i) { return i == 0 ? v.x : v.y; }
};

// Experiments that did not ship (history: git log): 16 queries per 16-byte gather at m = 8 (one
// 1024-thread workgroup per CU; correct and slower: k = 1 2.25 vs 2.01 ms, k = 1000 3.28 vs 2.46 -- a 16-byte entry costs
// four adds, the wider accumulators spill), exact re-evaluation of whole rows instead of (row, query) pairs, all tables in
// LDS (no L1-gathered part), other block / vote periods.
constexpr uint32_t FILT_MAX_SHARE_PCT = 60;   // first block: share of rows the pre-filter may let through before it is switched off
template <int M>
struct ScanCfg {
  // queries per LDS gather: a float4 entry (ds_read_b128) up to m = 32; m = 64 only fits the 160 KiB of
  // LDS with float2 entries (ds_read_b64, 2 queries per gather)
  static constexpr int QPG = (M <= 32) ? 4 : 2;
  static constexpr int QG = (RQ_QG16 && M == 8) ? 16 : (M <= 16) ? 8 : (M <= 32) ? 4 : 2;   // queries per group
  static constexpr int NQUAD = QG / QPG;                         // gathers per code byte
  // rows per thread per sub-step: ~32 gathers' worth, and a whole number of 16-byte code loads
  static constexpr int RPT = (32 / (M * NQUAD)) > (M < 16 ? 16 / M : 1) ? 32 / (M * NQUAD) : (M < 16 ? 16 / M : 1);
  // threads per workgroup: two 512-thread workgroups per CU, or ONE of 1024 where a group's tables take more than half of
  // the LDS (m = 16: 96 KiB of f32 tables + 32 KiB of byte tables; m = 32, 64: 112 KiB of f32 tables) -- the same 16
  // wavefronts per CU either way (m = 32 exact scan 27.1 -> 24.6 ms against one 512-thread workgroup per CU)
  static constexpr int THREADS = (M * QG >= 128) ? 1024 : SCAN_THREADS;
  static constexpr int SUB = THREADS * RPT;    // rows per sub-step (one 16/32-byte load per lane)
  static constexpr int U = (M <= 8) ? 8 : (M <= 16) ? 4 : (M <= 32) ? 2 : 1;  // sub-steps per block: loads of a block fly together
  static constexpr int BLK = SUB * U;               // rows per workgroup block
  // blocks between two capacity votes (the only barrier of the streaming loop): one vote per ~32768 rows.  Measured at
  // SIFT1M shape, votes every 1 / 2 / 4 / 8 blocks: k = 1 2.13 / 2.05 / 2.00 / 1.99 ms, k = 1000 2.62 / 2.62 / 2.46 / 2.44;
  // the candidate buffers grow by 2 * VP * BLK keys (a slow wavefront may still be appending the previous period's rows)
  static constexpr int VP = (32768 / BLK) < 1 ? 1 : (32768 / BLK) > 8 ? 8 : (32768 / BLK);
  static constexpr int LUT_BYTES = M * QG * 1024;   // full table; the LDS part is LUT_LDS_BYTES below
  // The LDS gather pipe is the kernel's bound (~11-12 cycles per 64-lane ds_read_b128 with random
  // slots).  The vector-memory path can gather the same 16 bytes from an L1-resident table in ~29
  // cycles per wavefront (tools/micro/gather_l1.hip) and runs beside the LDS, so the LAST KG
  // sub-quantizers (~25 % of the gathers, <= 16 KiB of table) are looked up through L1 instead.
  static constexpr int KG = (M == 8) ? (RQ_QG16 ? 4 : 2) : (M == 16) ? 4 : (M == 32) ? 4 : (M == 4) ? 1 : 0;   // M = 64: all in LDS
  static constexpr int KL = M - KG;                 // sub-quantizers [0, KL) gather from LDS
  static constexpr int GTAB_F4 = KG * NQUAD * 256;  // entries (float4 for QPG = 4) of the global (L1) table
  static constexpr int LUT_LDS_BYTES = KL * QG * 1024;
  // integer pre-filter (see build_qtab): one byte per (sub-quantizer, code, query); tiled for M = 8 and 16
  static constexpr bool HAS_FILT = (M == 4 || M == 8 || M == 16) && SCAN_THREADS == 512;     // (the default build; m = 4 since round 6: PQ / CQ tables only)
  // byte accumulator sets: 8 sub-quantizers each; m = 8 in FINE mode: two sets of 4 with 6-bit entries (half the step)
  static constexpr int NACC = HAS_FILT ? (M == 8 ? 2 : M >= 8 ? M / 8 : 1) : 1;
  static constexpr int kpa(bool fine) { return (M == 8 && fine) ? 4 : 8; }   // sub-quantizers per accumulator set
  static constexpr int QTAB_BYTES = HAS_FILT ? (M + 1) * 256 * QG : 0;   // + 1: the row-norm table of LSQ scans
  // scratch behind the staged queries: the threshold sample's [QG][THREADS] minima, later the filter table
  static constexpr int AUX_BYTES = (QG * THREADS * 4 > QTAB_BYTES) ? QG * THREADS * 4 : QTAB_BYTES;
  static_assert(RPT >= 1, "M too large for this tiling");
};

template <int QG>
struct ScanCtrl {
  float tau[QG];
  uint32_t cnt[QG];
  uint32_t sel[QG];     // which half of the candidate ping-pong buffer is current
  uint32_t item;
  uint32_t selmask;     // bit q == sel[q] (one LDS word the hot loop reads per block)
  uint32_t pad[2];
  // integer pre-filter (FILT kernels): per-(sub-quantizer, query) table minima and the per-query scale
  float fmin[16][QG];
  float fmax[16][QG];   // LSQ scans: per-(sub-quantizer, query) max |entry| (absolute rounding margins)
  float finv[QG];
  uint32_t fpush;       // rows the pre-filter let through in the item's first block (it is switched off if too many)
  SelState<QG> st;      // st.hist doubles as the per-wavefront queues of rows waiting for the exact evaluation
};

struct ScanParams {
  const uint8_t *codes;     // [n][M]
  const float *centers;     // [M][256][sub]
  const float *queries;     // [nq][d]
  uint32_t n, nq;
  int sub, d, K;
  int m_real;               // sub-quantizers that exist; tables k >= m_real are all-zero padding
  int lut_mode;             // 0 PQ sub-space (c-q)^2 | 1 LSQ -2<q,c> full-dim | 2 CQ (q-c)^2 full-dim
  const float *row_bias;    // LSQ: dbnorms[n], added after the table sum; else nullptr
  const uint8_t *norm_bytes;  // LSQ pre-filter: row norms quantised to 256 lower edges, [n]
  const float *norm_info;     // ... {nmin, nstep, max |.|} of the quantised quantity: norm[row] - sum_k |c_k[b_k]|^2
  const float *cnorm;         // ... |c_k[r]|^2, [M][256]: folded into the filter's tables (see build_qtab)
  const uint32_t *perm;     // bank-aware row order (rq_order.hip): perm[position] = original row, read for survivors only; or nullptr
  uint32_t samp_end;        // ... below this position one block in every samp_stride is an arrival-order sample of the base
  uint32_t samp_stride;     //     (order_sample_rows, rq_order.hip); samp_end == 0: none
  uint32_t id_offset;
  int id_base;
  uint32_t nslices, rows_per_slice, ngroups;
  uint32_t whole;           // query groups [0, whole) are ONE item over all rows (answer written directly);
                            // groups [whole, ngroups) are cut into nslices row slices (key lists -> merge)
  uint32_t xcd_mode;        // 1: big base -- row windows are handed out per XCD (work_counter[0..7]), see the item loop
  uint32_t xcd_slack;       // ... an item may start while at most this many items of the XCD's earlier rounds still run
  uint32_t xcd_round;       // ... items per pacing round (a window's items, or the XCD's resident workgroups)
#if RQ_SCAN_PACE_BUILD      // (experiment builds only: even unused kernel arguments moved the streaming loop's register allocation)
  uint32_t *pace;           // chunk pacing inside a round (round 6): [8][SCAN_PACE_SLOTS][2] {chunks done, members}; nullptr = off
  uint32_t pace_lag;        // ... a workgroup runs at most this many chunks ahead of its wave's average
  uint32_t pace_votes;      // ... a chunk = this many capacity votes (VP blocks each)
#endif
  uint32_t cap;             // candidate buffer capacity per query (keys)
  uint32_t trigger;         // compact when cnt > trigger  (cap - 2*VP*BLK >= trigger >= K)
  uint32_t p2;              // next_pow2(K)
  uint32_t scratch_keys;    // LDS sort scratch capacity in keys
  uint32_t sample;          // rows sampled per slice to initialise tau (0 = off)
  uint32_t sample_rt;       // ... for slices that get the second estimate (a looser first tau only rules 1/8 of the rows)
  uint32_t srank_mul;       // target survivors per slice = srank_mul * K (3; 0 forces the fallback, tests)
  int retune_z;             // second threshold estimate after 1/8 of the rows: rank = mean + z sigma (6; 0 = off; < 0: tests)
  int retune_min_k;         // ... only for K >= this
  int retune_div;           // ... after rows / retune_div rows (8)
  uint32_t *work_counter;
  float4 *gtab;               // [gridDim][GTAB_F4] L1-gathered part of the LUT
  unsigned long long *stats;  // optional [8]: cycles in lut, sample, stream, cuts, final cut, sort; #cuts; #fallbacks
  uint64_t *cand;           // [gridDim][QG][2][cap]
  uint16_t *bkt;            // [gridDim][cap] bucket ids of the large-K sample sort
  int filter;               // 1: 8-bit lower-bound pre-filter in front of the exact evaluation (FILT kernels)
  int bigk;                 // K > SCAN_SS_MIN_K: finish with samplesort_topk instead of cut + LDS bitonic (1: map buckets first, 2: sorted splitters only)
  int bfin;                 // K <= SCAN_SS_MIN_K: cut + sort per query by one wavefront through distance buckets (SCAN_BUCKET_FINISH)
  // outputs of whole items: dists/ids [nq][K], or packed keys [nq][K] when keys != nullptr;
  // of sliced items: part [(query - whole*QG)][nslices][K] packed keys
  float *dists;
  uint32_t *ids;
  uint64_t *keys;
  uint64_t *part;
};

// ------------------------------------------------------------------------------------------
// LUT construction, one (k, r) entry at a time for all QG queries of the group; compiled with
// -ffp-contract=off so every product, square and add stays a separately rounded f32 operation
// like the reference's x86-64 builds.
//   mode 0  deps/src/linscan_aqd.cpp:66-74                 T = sum_s (c[s] - q[k*sub+s])^2, s < sub
//   mode 1  deps/src/linscan_aqd_pairwise_byte.cpp:42-49   T = T - (2*q[s])*c[s],            s < d
//   mode 2  deps/src/linscan_aqd_pairwise_byte.cpp:126-133 T = T + (q[s]-c[s])^2,            s < d
// ------------------------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ void build_lut(float *lut, float4 *gtab, const float *qstage,
                                          const float *centers, int sub, int d, int mode, int m_real, int tid) {
  using Cfg = ScanCfg<M>;
  constexpr int QG = Cfg::QG, NQUAD = Cfg::NQUAD;
  const int cdim = mode == 0 ? sub : d;
  for (int e = tid; e < M * 256; e += ScanCfg<M>::THREADS) {
    const int k = e >> 8, r = e & 255;
    const float *c = centers + (size_t)e * cdim;
    const int qoff = mode == 0 ? k * sub : 0;
    float acc[QG];
#pragma unroll
    for (int q = 0; q < QG; ++q) acc[q] = 0.0f;
    if (k >= m_real) {
      // padding sub-quantizer (m rounded up to a supported tile width): T = 0, and x + 0.0f == x
    } else {
      // one table entry per thread, its codebook row streamed 16 bytes at a time (full-dimensional rows
      // are 512 B apart between lanes: scalar loads would touch 64 cache lines per instruction, 4x as often)
      auto step = [&](float cs, int s) {
        if (mode == 1) {
#pragma unroll
          for (int q = 0; q < QG; ++q) {
            const float two_q = 2.0f * qstage[q * d + s];
            const float prod = two_q * cs;
            acc[q] = acc[q] - prod;
          }
        } else {
#pragma unroll
          for (int q = 0; q < QG; ++q) {
            const float diff = cs - qstage[q * d + qoff + s];   // (q-c)^2 has the same bits
            const float sq = diff * diff;
            acc[q] = acc[q] + sq;
          }
        }
      };
      if ((cdim & 3) == 0 && ((uintptr_t)centers & 15) == 0) {
        const float4 *c4 = reinterpret_cast<const float4 *>(c);
#pragma unroll 2
        for (int s4 = 0; s4 < cdim / 4; ++s4) {
          const float4 cv = c4[s4];
          step(cv.x, 4 * s4 + 0);
          step(cv.y, 4 * s4 + 1);
          step(cv.z, 4 * s4 + 2);
          step(cv.w, 4 * s4 + 3);
        }
      } else {
        for (int s = 0; s < cdim; ++s) step(c[s], s);
      }
    }
    using LV = LutVec<Cfg::QPG>;
    using Vec = typename LV::type;
#pragma unroll
    for (int quad = 0; quad < NQUAD; ++quad) {
      const Vec v = LV::make(&acc[quad * Cfg::QPG]);
      if (k < Cfg::KL) reinterpret_cast<Vec *>(lut)[(k * NQUAD + quad) * 256 + r] = v;
      else reinterpret_cast<Vec *>(gtab)[((k - Cfg::KL) * NQUAD + quad) * 256 + r] = v;
    }
  }
}

// ADC distances of row r of the packed byte string w for the QG queries of the group:
// acc_q = ((T_q[0][b0] + T_q[1][b1]) + ...)  -- deps/src/linscan_aqd.cpp:85-87, sequential f32.
template <int M>
__device__ __forceinline__ void row_dists(const uint32_t *w, int r, const float4 *lut4,
                                          const float4 *__restrict__ gtab4, float (&acc)[ScanCfg<M>::QG]) {
  using Cfg = ScanCfg<M>;
  using LV = LutVec<Cfg::QPG>;
  using Vec = typename LV::type;
  constexpr int NQUAD = Cfg::NQUAD, KL = Cfg::KL, QPG = Cfg::QPG;
  const Vec *lutv = reinterpret_cast<const Vec *>(lut4);
  const Vec *__restrict__ gtab = reinterpret_cast<const Vec *>(gtab4);
  // issue the L1 gathers of the last sub-quantizers first: their latency hides under the LDS ones
  Vec tg[(Cfg::KG > 0 ? Cfg::KG : 1) * NQUAD];
  auto load_tg = [&]() {
#pragma unroll
    for (int k = KL; k < M; ++k) {
      const uint32_t byte = (w[(r * M + k) >> 2] >> (8 * ((r * M + k) & 3))) & 0xffu;
#pragma unroll
      for (int quad = 0; quad < NQUAD; ++quad) tg[(k - KL) * NQUAD + quad] = gtab[((k - KL) * NQUAD + quad) * 256 + byte];
    }
  };
  constexpr bool WIDE = M * NQUAD >= 32;     // a row of 32 gathers is summed in two halves (see below)
  if constexpr (!WIDE) load_tg();
#pragma unroll
  for (int k = 0; k < M; ++k) {
    // at most 16 gathers (64 registers of table entries) in flight: a row of 32 (m = 16, 8 queries) is summed in two
    // halves -- the scheduler otherwise hoists all 32 and spills their results (LSQ variant: 160 spilled registers)
    if constexpr (WIDE) {
      if (k * NQUAD == 16) {
        __builtin_amdgcn_sched_barrier(0);
        load_tg();                             // the L1 part belongs to the second half (KL >= M / 2)
      }
    }
    const uint32_t byte = (w[(r * M + k) >> 2] >> (8 * ((r * M + k) & 3))) & 0xffu;
#pragma unroll
    for (int quad = 0; quad < NQUAD; ++quad) {
      const Vec t = k < KL ? lutv[(k * NQUAD + quad) * 256 + byte] : tg[(k - KL) * NQUAD + quad];
#pragma unroll
      for (int c = 0; c < QPG; ++c) {
        if (k == 0) acc[quad * QPG + c] = LV::get(t, c);
        else acc[quad * QPG + c] = acc[quad * QPG + c] + LV::get(t, c);
      }
    }
  }
}

// one row's M code bytes into w[0 .. M/4) (packed like the hot loop's byte string, r = 0)
template <int M>
__device__ __forceinline__ void load_row(uint32_t *w, const uint8_t *codes, uint32_t row) {
  if constexpr (M % 4 == 0) {
#pragma unroll
    for (int i = 0; i < M / 4; ++i) w[i] = reinterpret_cast<const uint32_t *>(codes + (size_t)row * M)[i];
  } else {
#pragma unroll
    for (int i = 0; i < (M + 3) / 4; ++i) w[i] = 0;
#pragma unroll
    for (int k = 0; k < M; ++k) w[k >> 2] |= (uint32_t)codes[(size_t)row * M + k] << (8 * (k & 3));
  }
}


// ------------------------------------------------------------------------------------------
// Survivors of one wave-row-step: lane `lane` holds the exact distances acc[q] of ONE row (key id `kid`) to the
// QG queries; rows with acc[q] <= tau[q] are appended to query q's candidate buffer (INCLUSIVE: tau is a sampled
// distance, and on heavily duplicated codes hundreds of rows share the K-th neighbour's distance exactly -- with a strict
// test they all drop out and the slice has to be redone exactly; 3 % of the groups on 1024-cluster data).  ONE LDS atomic for all QG
// queries: lane q reserves popc(mk[q]) slots of query q.
// ------------------------------------------------------------------------------------------
template <int QG>
__device__ __forceinline__ void emit_survivors(const float (&acc)[QG], const float (&tau)[QG], bool valid, uint32_t row,
                                               const uint32_t *__restrict__ perm, uint32_t id_offset,
                                               uint32_t selmask, ScanCtrl<QG> *ctrl, uint64_t *cand_wg, uint32_t cap,
                                               int lane) {
  uint64_t mk[QG];
  uint64_t any = 0;
#pragma unroll
  for (int q = 0; q < QG; ++q) {
    mk[q] = __ballot(valid && (acc[q] <= tau[q]));
    any |= mk[q];
  }
  if (any) {
    // the key's id: the row's ORIGINAL number (ordered bases: one load per surviving row, never in the common path)
    uint32_t kid = row;
    if (perm && ((any >> lane) & 1ull)) kid = perm[row];
    kid += id_offset;
    uint32_t want = 0;
#pragma unroll
    for (int q = 0; q < QG; ++q)
      want = writelane_u32(want, (uint32_t)__popcll(mk[q]), q);
    uint32_t got = 0;
    if (lane < QG && want) got = atomicAdd(&ctrl->cnt[lane], want);
#pragma unroll
    for (int q = 0; q < QG; ++q) {
      if (mk[q]) {
        const uint32_t basep = __builtin_amdgcn_readlane(got, q);
        if ((mk[q] >> lane) & 1ull) {
          const uint32_t pos = basep + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk[q] >> 32),
                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mk[q], 0u));
          uint64_t *buf = cand_wg + ((size_t)q * 2 + ((selmask >> q) & 1u)) * cap;
          buf[pos] = make_key(acc[q], kid);
        }
      }
    }
  }
}


}  // namespace rq
