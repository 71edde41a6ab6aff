// rq_scan_select.h -- threshold bookkeeping of a work item: capacity cut, second threshold estimate, large-k finish
// Part of the ADC scan (rq_scan.hip); device code only, included by that file alone.
#pragma once
#include "rq_scan_tables.h"

namespace rq {

// Cut the candidate buffers of the flagged queries back to exactly K keys and refresh tau.
template <int M>
__device__ __forceinline__ void compact_group(ScanCtrl<ScanCfg<M>::QG> *ctrl, uint64_t *cand_wg,
                                              const ScanParams &p, bool need, int g, int gi, uint32_t &vseq) {
  constexpr int QG = ScanCfg<M>::QG;
  constexpr int TPG = ScanCfg<M>::THREADS / QG;
  const uint32_t cnt = ctrl->cnt[g];
  const uint32_t sel = ctrl->sel[g];
  const uint64_t *src = cand_wg + ((size_t)g * 2 + sel) * p.cap;
  uint64_t *dst = cand_wg + ((size_t)g * 2 + (sel ^ 1u)) * p.cap;
  radix_select<QG, TPG>(&ctrl->st, src, cnt, (uint32_t)p.K, need, g, gi, vseq);
  const uint64_t tau_key = ctrl->st.prefix[g];
  compact_leq<QG, TPG>(&ctrl->st, src, dst, cnt, tau_key, need, g, gi);
  if (need && gi == 0) {
    ctrl->cnt[g] = ctrl->st.newcnt[g];  // == K (keys are unique)
    ctrl->sel[g] = sel ^ 1u;
    atomicXor(&ctrl->selmask, 1u << g);
    ctrl->tau[g] = key_dist(tau_key);
  }
  __syncthreads();
}

// Second threshold estimate, once per item after the first `f` of the slice's rows: the candidates collected so far
// are an exact sample of that fraction, so the number of them below the true K-th distance is Binomial(K, f); tau
// becomes the distance of the candidate of rank  K f + z sqrt(K f (1 - f)) + 2  (z = 6), which lets ~K + z sqrt(K/f)
// rows through the whole slice instead of the first estimate's 1.5-2.5 K: fewer exact re-evaluations, appends and
// keys to cut at the end.  The candidates ABOVE the new tau are dropped on the spot (one compaction pass over the few
// hundred keys collected so far): they can only matter if fewer than K rows beat the new tau, and in that case the
// end-of-slice check (cnt < K) redoes the slice exactly anyway.  So the invariant of the streaming loop holds before and
// after: the buffer is exactly the set of rows seen so far with dist <= tau -- which is what makes a later capacity cut
// (K smallest keys, tau = the K-th) exact.  (Round 2 kept the stale candidates and a correction count instead; a
// capacity cut after the estimate could then keep stale keys above the new tau and silently lose rows in between --
// ADVICE r2; flagging such cuts for the exact redo instead made 6 % of the items of a 1e9-row scan fall back.)
template <int M>
__device__ __noinline__ uint32_t retune_tau(ScanCtrl<ScanCfg<M>::QG> *ctrl, uint64_t *cand_wg, uint32_t cap, uint32_t r2,
                                            uint32_t vseq) {
  constexpr int QG = ScanCfg<M>::QG;
  constexpr int TPG = ScanCfg<M>::THREADS / QG;
  const int g = threadIdx.x / TPG, gi = threadIdx.x % TPG;
  const uint32_t cnt = ctrl->cnt[g];
  const uint32_t sel = ctrl->sel[g];
  const bool act = cnt > r2 && r2 >= 1;
  const uint64_t *src = cand_wg + ((size_t)g * 2 + sel) * cap;
  uint64_t *dst = cand_wg + ((size_t)g * 2 + (sel ^ 1u)) * cap;
  radix_select<QG, TPG>(&ctrl->st, src, cnt, r2, act, g, gi, vseq);
  const uint32_t td = (uint32_t)(ctrl->st.prefix[g] >> 32);     // ordered bits of the r2-th smallest distance
  const bool tighten = act && ord2f(td) < ctrl->tau[g];          // (tau is rewritten behind compact_leq's barriers)
  // keep every key whose DISTANCE is <= the new tau (all ids): the inclusive rule of emit_survivors
  compact_leq<QG, TPG>(&ctrl->st, src, dst, cnt, ((uint64_t)td << 32) | 0xFFFFFFFFull, tighten, g, gi);
  if (tighten && gi == 0) {
    ctrl->tau[g] = ord2f(td);
    ctrl->cnt[g] = ctrl->st.newcnt[g];
    ctrl->sel[g] = sel ^ 1u;
    atomicXor(&ctrl->selmask, 1u << g);
  }
  __syncthreads();
  return vseq;
}

// Large-K finish of one work item (out of line: keeps the streaming loop's register allocation
// independent of it).  cnt/sel: the item's per-query candidate counts and current buffer halves.
template <int NT>
__device__ __noinline__ void finish_bigk(const uint32_t *cnt_q, const uint32_t *sel_q, uint32_t QG, uint64_t *cand_wg,
                                         uint16_t *bkt, uint32_t cap, uint32_t K, uint32_t q0, uint32_t nq,
                                         uint64_t *keys_base, uint32_t key_stride, float *dists, uint32_t *ids,
                                         uint32_t id_base, unsigned char *lds, unsigned long long *stats, bool use_map) {
  const uint32_t tid = threadIdx.x;
#pragma unroll 1
  for (uint32_t q = 0; q < QG; ++q) {
    const uint32_t qq = q0 + q;
    if (qq >= nq) break;
    const uint32_t cnt = cnt_q[q], sel = sel_q[q];
    const uint64_t *src = cand_wg + ((size_t)q * 2 + sel) * cap;
    uint64_t *dst = cand_wg + ((size_t)q * 2 + (sel ^ 1u)) * cap;
    const uint32_t n_out = min(K, cnt);
    // Fewer candidates than K: a slice shorter than K, or rows whose distance is NaN (they compare false with every threshold
    // and are never candidates: include/rayuela_hip.h "Non-finite inputs") -- the tail is the padding key, (NaN, no row).
    // cnt is workgroup-uniform, so skipping the sort (which has barriers inside) for an empty list is uniform too.
    if (keys_base) {
      uint64_t *o = keys_base + (size_t)qq * key_stride;
      for (uint32_t i = n_out + tid; i < K; i += NT) o[i] = KEY_MAX;
      if (cnt != 0u)
        samplesort_topk<NT>(src, dst, bkt, cnt, n_out, lds, [o](uint32_t r, uint64_t key) { o[r] = key; }, stats, use_map);
    } else {
      float *od = dists + (size_t)qq * K;
      uint32_t *oi = ids + (size_t)qq * K;
      for (uint32_t i = n_out + tid; i < K; i += NT) { od[i] = key_dist(KEY_MAX); oi[i] = key_id(KEY_MAX) + id_base; }
      if (cnt != 0u)
        samplesort_topk<NT>(src, dst, bkt, cnt, n_out, lds, [od, oi, id_base](uint32_t r, uint64_t key) {
          od[r] = key_dist(key);
          oi[r] = key_id(key) + id_base;
        }, stats, use_map);
    }
  }
}


}  // namespace rq
