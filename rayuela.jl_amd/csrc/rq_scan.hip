// rq_scan.hip -- ADC linear scan with exact top-k for gfx950 (MI355X).
//
// Replaces deps/src/linscan_aqd.cpp:37-102 (_linscan_aqd_query).  Same arithmetic:
//   LUT   T[k][r] = sum_s (c[(k*256+r)*sub+s] - q[k*sub+s])^2   sequential f32, mul/add unfused (:66-74)
//   dist  d_j     = ((T[0][b_j0] + T[1][b_j1]) + ...)           sequential f32               (:85-87)
//   top-k         = k smallest (dist, id) pairs, lexicographic, ascending                    (:91-97)
//
// MI355X design:
//   * Two 512-thread workgroups per CU, persistent over work items (query-group x row-slice)
//     handed out by an atomic counter; the second workgroup streams while the first one sits in
//     a barrier, a cut or its final sort.
//   * A query group is QG queries.  Their LUTs live in LDS interleaved 4 queries per entry:
//     lut[k][quad][r] is a float4 = T_q[k][r] for the 4 queries of the quad, so ONE
//     ds_read_b128 gather serves 4 queries; slot = r mod 16 spreads a 16-lane group over
//     all 64 banks (the [k][r][QG] layout would only reach every other 16-byte slot).
//     The last KG sub-quantizers are gathered through L1 from a per-workgroup global table
//     instead (the LDS pipe is the bound; the vector-memory pipe runs beside it).
//   * Codes stream from HBM/L2 coalesced, 16 or 32 bytes per lane per sub-step, U sub-steps per
//     block; every code byte is read once per query GROUP, not per query.
//   * Top-k: per-query threshold tau.  A row survives the hot loop only if dist <= tau (INCLUSIVE); survivors
//     are appended (one LDS atomic per row for all QG queries) to a per-query candidate buffer in
//     global memory.  tau starts from a sampled estimate (per-thread minima of a stratified
//     sample, rank selected in LDS) and is tightened once after 1/8 of the slice (retune_tau); if fewer
//     than k rows beat it the slice is redone from tau = +inf, which cuts the buffer back to exactly k keys
//     by an 8-pass radix select whenever it could overflow.  The inclusive test is exact: a cut keeps the k
//     smallest (dist, id) KEYS and sets tau to the k-th key's distance, so every later row that could still
//     enter the top-k has dist <= tau and is appended (rows that tie tau's distance with a larger id are
//     appended too and lose to the k-th key at the next cut or in the final select -- at most one key per
//     row per query, which the capacity accounting already assumes).  With a strict '<' a whole tie group
//     at a SAMPLED tau would drop out and force the exact redo (emit_survivors).
//   * At the end of a slice the k survivors are bitonic-sorted in LDS (the LUT is dead by then
//     and its space is reused) and written either as final (dist,id) or as packed keys for
//     the slice/GPU merge kernel.
//   * LUT modes 1/2 and the per-row bias serve linscan_lsq / linscan_cq
//     (deps/src/linscan_aqd_pairwise_byte.cpp) with the same kernel.
#include "rq_internal.h"
#include "rq_topk.h"

#include "rq_scan_tables.h"
#include "rq_scan_filter.h"
#include "rq_scan_select.h"


namespace rq {

// phase accounting (diagnostics only; p.stats == nullptr in normal runs)
#define RQ_STAT_T() ((p.stats && threadIdx.x == 0) ? (unsigned long long)clock64() : 0ull)
#define RQ_STAT_ADD(slot, t0) do { if (p.stats && threadIdx.x == 0) atomicAdd(&p.stats[slot], (unsigned long long)clock64() - (t0)); } while (0)
#define RQ_STAT_INC(slot) do { if (p.stats && threadIdx.x == 0) atomicAdd(&p.stats[slot], 1ull); } while (0)

// One work item's bucket finish (K <= 1024): a look at every query's candidates for distance ties, then one wavefront per query
// through bucket_finish_wave.  Returns whether this thread's query gave up or the group skipped the attempt (the caller votes);
// nothing has been written for a query that gave up.  `scratch`: the dead table space, split evenly over the QG queries.  Out of
// line so that the streaming loop's register allocation does not depend on it; LDS pointers are cast back to the LDS address
// space here (through a generic pointer every access would be a flat instruction).
struct BucketFinishArgs {      // by value: taking the address of the kernel's ScanParams would move the whole struct to scratch
  uint32_t cap, scratch_keys, nq, K, id_base;
  int bfin;
  float *dists;
  uint32_t *ids;
  unsigned long long *stats;
};
template <int M>
__device__ __noinline__ bool bucket_finish_item(ScanCtrl<ScanCfg<M>::QG> *ctrl_g, uint64_t *cand_wg, BucketFinishArgs p,
                                                uint64_t *scratch_g, int g, int gi, uint32_t q0, uint64_t *keys_base,
                                                uint32_t key_stride, uint32_t vseq_in, uint32_t *vseq_out) {
  constexpr int QG = ScanCfg<M>::QG;
  uint32_t vseq = vseq_in;
  typedef ScanCtrl<QG> __attribute__((address_space(3))) lds_ctrl_t;
  typedef unsigned char __attribute__((address_space(3))) lds_byte_t;
  // (address-space casts: the same addresses, as LDS pointers)
  ScanCtrl<QG> *ctrl = (ScanCtrl<QG> *)(lds_ctrl_t *)ctrl_g;
  unsigned char *scratch = (unsigned char *)(lds_byte_t *)reinterpret_cast<unsigned char *>(scratch_g);
  bool gave_up = false;
  const uint32_t share = (p.scratch_keys / (uint32_t)QG) * 8u & ~7u;          // bytes of a query's share
  const bool mine_on = gi < 64 && q0 + (uint32_t)g < p.nq;
  // first a look at 64 candidates of every query: a group with a tie-heavy query (rows sharing their codes) goes straight to
  // select + sort -- the bucket ranking is quadratic in a tie group, and giving up half-way would waste the group's work
  {
    bool heavy = false;
    if (mine_on) {
      uint64_t *tab = reinterpret_cast<uint64_t *>(scratch + (size_t)g * share);
      const uint64_t *src = cand_wg + ((size_t)g * 2 + ctrl->sel[g]) * p.cap;
      const uint32_t twins = share >= 8192u ? bf_tie_twins<10>(src, ctrl->cnt[g], tab, (uint32_t)gi)
                                            : bf_tie_twins<8>(src, ctrl->cnt[g], tab, (uint32_t)gi);
      // ... and so does a query with more than max(BF_MAX_CNT, 3 K) candidates (short slices, K a large share of the rows): the map is laid
      // over ALL of them, so the K it keeps would sit in a few crowded buckets (m = 32, 33 K-row slices, k = 1000: 0.37 against 0.30 ms)
      heavy = twins >= BF_TIE_MIN || ctrl->cnt[g] > max(BF_MAX_CNT, 3u * p.K);
    }
    gave_up = block_any(heavy, ctrl->st.vote, vseq);
#ifndef RQ_NO_FINSTATS
    if (p.stats && threadIdx.x == 0) {     // diagnostics: groups that came here / that the look sent to select + sort
      atomicAdd(&p.stats[16], 1ull);
      if (gave_up) atomicAdd(&p.stats[17], 1ull);
    }
#endif
  }
  if (!gave_up && mine_on) {
    const uint32_t cnt = ctrl->cnt[g];
    const uint32_t sel = ctrl->sel[g];
    const uint64_t *src = cand_wg + ((size_t)g * 2 + sel) * p.cap;
    uint64_t *dst = cand_wg + ((size_t)g * 2 + (sel ^ 1u)) * p.cap;
    const uint32_t qq = q0 + (uint32_t)g;
    // the wavefront's share of the (dead) table space: BF_NB bucket words, then room for the kept keys
    unsigned char *mine = scratch + (size_t)g * share;
    uint32_t *nxt = reinterpret_cast<uint32_t *>(mine) + 2;                  // nxt[-1] is part of the share
    uint64_t *kbuf = reinterpret_cast<uint64_t *>(mine + BF_NB * 4u + 8u);
    const uint32_t kcap = p.bfin == 2 ? 0u : (share - BF_NB * 4u - 8u) / 8u;     // (SCAN_BUCKET_FINISH=2: tests, kept keys through global memory)
    uint64_t *ok = keys_base ? keys_base + (size_t)qq * key_stride : nullptr;
    float *od = p.dists + (size_t)qq * p.K;
    uint32_t *oi = p.ids + (size_t)qq * p.K;
    const uint32_t idb = p.id_base;
    auto emit = [&](uint32_t r, uint64_t key) {
      if (ok) ok[r] = key;
      else { od[r] = key_dist(key); oi[r] = key_id(key) + idb; }
    };
    gave_up = !bucket_finish_wave(src, dst, cnt, (uint32_t)p.K, nxt, kbuf, kcap, (uint32_t)gi, emit, p.stats);
    if (!gave_up)      // fewer candidates than K (slices shorter than K): the tail is padding, as the LDS sort leaves it
      for (uint32_t i = cnt + (uint32_t)gi; i < (uint32_t)p.K; i += 64u) emit(i, KEY_MAX);
  }
  *vseq_out = vseq;
  return gave_up;
}

#if RQ_SCAN_PACE_BUILD
// EXPERIMENT, compiled in with -DRQ_SCAN_PACE_BUILD=1 only (tools/build_variant.sh; EXPERIMENTS.md section 9.2: it buys L2 hits with
// idle time -- 11.8x -> 5.6x the code bytes fetched on the 1.25e8-row shard for 15.4 -> 20.6 ms).
// Chunk pacing of a big-base item (xcd_mode; one thread per workgroup, out of line so that the streaming loop's registers do
// not depend on it).  The workgroups of an XCD that run the items of one pacing wave stream the same 32 MB window; they share it
// through the XCD's 4 MiB L2 only while they are within a few hundred KB of each other.  slot[0] sums the chunks the wave's
// workgroups have finished, slot[1] counts the workgroups that joined: a workgroup starts chunk c only when the wave's AVERAGE
// progress is at least c - lag.  The slowest workgroups never wait (no cycle), stale or shared slots only shorten waits, the spin
// is bounded: a speed hint like the item pacing above -- no data depends on it.
__device__ __noinline__ void pace_chunk(uint32_t *slot, uint32_t chunk, uint32_t lag) {
  if (chunk != 0u) atomicAdd(slot, 1u);
  if (chunk <= lag) return;
  const uint32_t want = chunk - lag;
  for (uint32_t spin = 0; spin < (1u << 9); ++spin) {
    const uint32_t done = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t members = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done >= want * members) break;
    __builtin_amdgcn_s_sleep(32);
  }
}
#endif

template <int M, bool BIAS, bool FILT, bool FINE = false>
__global__ __launch_bounds__(ScanCfg<M>::THREADS, 4) void adc_scan_kernel(ScanParams p) {
  using Cfg = ScanCfg<M>;
  constexpr int QG = Cfg::QG, RPT = Cfg::RPT, BLK = Cfg::BLK;
  constexpr int TPG = ScanCfg<M>::THREADS / QG;
  constexpr int CTRL_BYTES = (sizeof(ScanCtrl<QG>) + 15) & ~15;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ScanCtrl<QG> *ctrl = reinterpret_cast<ScanCtrl<QG> *>(smem);
  // LDS: [ctrl][aux: sample minima, later the byte tables][f32 tables][staged queries].  The byte tables sit
  // right behind ctrl so that their addresses are byte * 8 + a COMPILE-TIME offset below 64 KiB: the hot loop's
  // address is then one SDWA shift and the offset rides in the ds_read instruction.
  uint32_t *samp = reinterpret_cast<uint32_t *>(smem + CTRL_BYTES);   // [QG][ScanCfg<M>::THREADS] sample minima / qtab
  float *lut = reinterpret_cast<float *>(smem + CTRL_BYTES + Cfg::AUX_BYTES);
  float *qstage = lut + Cfg::LUT_LDS_BYTES / 4;
  uint64_t *scratch = reinterpret_cast<uint64_t *>(smem + CTRL_BYTES);  // aliases lut (dead by then)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int g = tid / TPG, gi = tid % TPG;  // query-lane view used by select / sort
  uint64_t *cand_wg = p.cand + (size_t)blockIdx.x * QG * 2 * p.cap;
  float4 *gtab = p.gtab + (size_t)blockIdx.x * (Cfg::GTAB_F4 > 0 ? Cfg::GTAB_F4 : 1);
  const uint32_t tail_groups = p.ngroups - p.whole;
  const uint32_t nitems = p.whole + tail_groups * p.nslices;
  uint32_t vseq = 0;                         // block_any() call counter (workgroup-uniform)
  if (tid < 2) ctrl->st.vote[tid] = 0;
  if (tid == 0) ctrl->pad[0] = 0xffffffffu;  // xcd_mode: the XCD whose item this workgroup is working on
  uint32_t xcd = 0;                          // the XCD this workgroup runs on (a speed hint only: L2 affinity)
  if (p.xcd_mode) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd));
    xcd &= 7u;
  }

  for (;;) {
    __syncthreads();
    if (tid == 0) {
      if (!p.xcd_mode) {
        ctrl->item = atomicAdd(p.work_counter, 1u);
      } else {
        // Big bases (plan_for): the rows are cut into windows of a few tens of MB and window s belongs to XCD s mod 8,
        // which runs ALL query groups over it before its next window -- the 64 workgroups of an XCD then stream the same
        // rows at about the same time and share them through that XCD's 4 MiB L2, instead of every workgroup pulling its
        // own copy of the base through the fabric (8 GB base, 128 groups: 79x the code bytes fetched per launch).
        // One work counter per XCD; an XCD that runs dry takes windows of the next one.  Placement only changes speed.
        // Pacing (a speed hint, no data depends on it): with nothing else the workgroups of an XCD finish their items
        // at evenly spread times after a few windows, their positions inside the current window spread uniformly and the
        // L2 sharing is gone again (measured at 1e9 rows: 65x the code bytes fetched instead of 75x).  So the XCD's items
        // go in rounds of `xcd_round` (one per resident workgroup), and an item starts only when all but `xcd_slack`
        // items of the earlier rounds are finished: the pack re-forms every round (numbers: scan_launch).  work_counter[8 + x] counts XCD x's finished items; earlier items never wait for
        // later ones, so the waits cannot form a cycle, and the spin is bounded anyway.
        if (ctrl->pad[0] != 0xffffffffu) atomicAdd(p.work_counter + 8u + ctrl->pad[0], 1u);     // the item just finished
        ctrl->pad[0] = 0xffffffffu;
        uint32_t it = 0xffffffffu;
        for (uint32_t y = 0; y < 8u && it == 0xffffffffu; ++y) {
          const uint32_t x = (xcd + y) & 7u;
          const uint32_t nwin = x < p.nslices ? (p.nslices - x + 7u) / 8u : 0u;       // windows s = x, x + 8, ...
          const uint32_t cnt_x = nwin * p.ngroups;
          if (cnt_x == 0u) continue;
          const uint32_t j = atomicAdd(p.work_counter + x, 1u);
          if (j < cnt_x) {
            it = ((j / p.ngroups) * 8u + x) * p.ngroups + (j % p.ngroups);             // = slice * ngroups + group
            ctrl->pad[0] = x;
            // chunk pacing: the items of a pacing round run in waves of one item per resident workgroup of the XCD; a wave
            // shares one slot (workgroups that took another XCD's window sit behind another L2: they stay out)
#if RQ_SCAN_PACE_BUILD
            ctrl->pad[1] = 0xffffffffu;
            if (p.pace != nullptr && y == 0u) {
              const uint32_t nres = max(1u, gridDim.x / 8u);
              const uint32_t wave = (j / p.xcd_round) * ((p.xcd_round + nres - 1u) / nres) + (j % p.xcd_round) / nres;
              ctrl->pad[1] = (x * (uint32_t)SCAN_PACE_SLOTS + wave % (uint32_t)SCAN_PACE_SLOTS) * 2u;
              atomicAdd(p.pace + ctrl->pad[1] + 1u, 1u);
            }
#endif
            const uint32_t before = (j / p.xcd_round) * p.xcd_round;                   // items of the earlier rounds
            const uint32_t need = before > p.xcd_slack ? before - p.xcd_slack : 0u;
            for (uint32_t spin = 0; spin < (1u << 10); ++spin) {      // bounded: ~2 ms at most, then the item starts anyway
              if (__hip_atomic_load(p.work_counter + 8u + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) break;
              __builtin_amdgcn_s_sleep(64);
            }
          }
        }
        ctrl->item = it;
      }
    }
    __syncthreads();
    const uint32_t item = ctrl->item;
    if (item >= nitems) break;
    // whole items first (the long ones), then the sliced tail, slice-major so that the workgroups
    // running together stream the same rows
    const bool sliced = item >= p.whole && p.nslices > 1;
    const uint32_t t_item = item - min(item, p.whole);
    const uint32_t slice = sliced ? t_item / tail_groups : 0u;
    const uint32_t group = item < p.whole ? item : p.whole + (sliced ? t_item % tail_groups : t_item);
    const uint32_t q0 = group * QG;
    // where this item's sorted keys go (nullptr: dists/ids)
    uint64_t *const keys_base = sliced ? p.part + (size_t)slice * p.K - (size_t)p.whole * QG * p.nslices * p.K
                                       : p.keys;
    const uint32_t key_stride = sliced ? p.nslices * (uint32_t)p.K : (uint32_t)p.K;

    // ---- stage the group's queries, reset the per-query state -------------------------------
    for (int e = tid; e < QG * p.d; e += ScanCfg<M>::THREADS) {
      const int q = e / p.d, c = e - q * p.d;
      const uint32_t qq = min(q0 + (uint32_t)q, p.nq - 1u);  // ragged last group: repeat a query
      qstage[e] = p.queries[(size_t)qq * p.d + c];
    }
    __syncthreads();
    unsigned long long t_ph = RQ_STAT_T();
    build_lut<M>(lut, gtab, qstage, p.centers, p.sub, p.d, p.lut_mode, p.m_real, tid);
    // The L1-gathered part of the table lives in global memory and is written by all wavefronts of the
    // workgroup, and read back only by it: every wavefront RELEASES its stores at workgroup scope before the
    // barrier (s_waitcnt vmcnt(0): the write-through stores are complete; an agent-scope release would also write
    // the whole L2 back, 0.2 ms per launch) and ACQUIRES after it at agent scope (buffer_inv: this CU's L1 may
    // still hold the previous item's lines).  tests/test_isa.py asserts that sequence in the generated code.
    if (Cfg::KG > 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      // LLVM's gfx942/gfx950 memory model needs no wait here (the wavefronts of a workgroup share the CU's L1 and
      // its request order); the explicit drain makes the hand-over independent of that reasoning -- once per item
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (Cfg::KG > 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    RQ_STAT_ADD(0, t_ph);
    const uint32_t r_begin = sliced ? slice * p.rows_per_slice : 0u;
    const uint32_t r_end = sliced ? min(p.n, r_begin + p.rows_per_slice) : p.n;
    const uint32_t rows = r_end - r_begin;
    const float4 *lut4 = reinterpret_cast<const float4 *>(lut);
    // Threshold initialisation.  attempt 0: tau = the `srank`-th smallest of the per-thread minima of a
    // stratified sample of S rows, so about 1.5-2.5 K rows survive the whole slice instead of
    // K*(1+ln(rows/K)) and no cut is needed before the end.  That tau is an estimate: if fewer
    // than K rows beat it (needs a sample ~3x off) attempt 1 redoes the slice with tau = +inf,
    // which is exact by construction.  Either way the answer is exact: whenever >= K rows beat
    // tau, the K best rows are among them.
    // The sample statistic is the per-thread MINIMUM of gsz rows, so the fraction of threads whose
    // minimum beats the q-quantile is 1-(1-q)^gsz (~ q*gsz only while that is small); gsz shrinks
    // for large K so that the selected rank stays in the well-conditioned middle of the 512 minima.
    // slices long enough for the second estimate (retune_tau) start from a smaller sample
    // Ordered bases: sorted rows are no random sample of anything -- the estimate is taken from the slice's SAMPLE blocks
    // (one block in every samp_stride below samp_end holds an arrival-order sample of the base, rq_order.hip), which the
    // block loop visits first.  nsamp: those blocks inside [r_begin, r_end).
    uint32_t nsamp = 0;
    if (p.perm != nullptr && p.samp_end != 0u) {
      const uint32_t b0 = r_begin / (uint32_t)BLK, b1 = min(r_end, p.samp_end) / (uint32_t)BLK;     // (samp_end is a block multiple)
      const uint32_t s0 = (b0 + p.samp_stride - 1u) / p.samp_stride, s1 = (max(b1, b0) + p.samp_stride - 1u) / p.samp_stride;
      nsamp = s1 > s0 ? s1 - s0 : 0u;
    }
    const bool will_retune = p.retune_z != 0 && p.K >= p.retune_min_k && rows >= 16u * (uint32_t)BLK &&
                             (p.perm == nullptr || nsamp > 0u);
    uint32_t S = will_retune ? p.sample_rt : p.sample;
    uint32_t srank = 0;
    bool sampled = false;
    const uint32_t Ks = max((uint32_t)p.K, 8u);   // k < 8 aims at the 8th neighbour: same machinery, still exact
    if (S >= (uint32_t)ScanCfg<M>::THREADS && rows >= 32u * (uint32_t)ScanCfg<M>::THREADS && (uint64_t)rows >= 16ull * (uint64_t)Ks) {
      // The number of thread minima below the true K-th distance is Binomial(512, frac), frac = 1-(1-K/rows)^g.
      // tau = the minimum of rank  mean + z sigma + 2  (z = 3 * srank_mul = 6): fewer than K survivors -- the
      // only cost of a miss is one redone slice -- is a 6-sigma event, and the surplus over K shrinks with K:
      // ~2.5 K survivors at K = 1000 (sigma/mean = 25 %), ~1.5 K at K = 10000 (7 %).
      const float q = (float)Ks / (float)rows;                            // the K-th neighbour's quantile (<= 1/16)
      uint32_t gsz = S / (uint32_t)ScanCfg<M>::THREADS;
      gsz = min(gsz, rows / (8u * (uint32_t)ScanCfg<M>::THREADS));               // short slice: sample at most 1/8 of it
      if (q * (float)gsz > 0.45f) gsz = max(1u, (uint32_t)(0.45f / q));   // keep the rank in the middle of the 512
      S = gsz * (uint32_t)ScanCfg<M>::THREADS;
      const float frac = 1.0f - __expf((float)gsz * __logf(fmaxf(1.0f - q, 1e-6f)));
      const float mean = frac * (float)ScanCfg<M>::THREADS;
      const float z = 3.0f * (float)p.srank_mul;
      srank = (uint32_t)ceilf(mean + z * sqrtf(mean * (1.0f - frac))) + 2u;
      sampled = srank * 4u <= 3u * (uint32_t)ScanCfg<M>::THREADS;
    }
#pragma unroll 1
    for (int attempt = sampled ? 0 : 1; attempt < 2; ++attempt) {
    __syncthreads();
    if (tid < QG) {
      ctrl->tau[tid] = __uint_as_float(0x7f800000u);  // +inf: everything passes until the first cut
      ctrl->cnt[tid] = 0;
      ctrl->sel[tid] = 0;
      if (tid == 0) { ctrl->selmask = 0; ctrl->fpush = 0; }
    }
    __syncthreads();
    if (attempt == 1 && sampled) RQ_STAT_INC(7);
    t_ph = RQ_STAT_T();
    if (attempt == 0) {
      // every thread keeps the minimum of its S/ScanCfg<M>::THREADS sample rows per query; the `srank`-th
      // smallest of those minima (an upper bound of the srank-th smallest sample distance, and equal
      // to it unless two of the srank best rows fell to one thread) is selected in LDS
      const uint32_t step = rows / S;
      float smin[QG];
#pragma unroll
      for (int q = 0; q < QG; ++q) smin[q] = __uint_as_float(0x7f800000u);
#pragma unroll 1
      for (uint32_t i = tid; i < S; i += ScanCfg<M>::THREADS) {
        const uint32_t row = r_begin + i * step + ((i * 2654435761u) >> 8) % step;
        uint32_t w1[(M + 3) / 4];
        load_row<M>(w1, p.codes, row);
        float acc[QG];
        row_dists<M>(w1, 0, lut4, gtab, acc);
        if (BIAS) {
          const float bias = p.row_bias[row];
#pragma unroll
          for (int q = 0; q < QG; ++q) acc[q] = acc[q] + bias;
        }
#pragma unroll
        for (int q = 0; q < QG; ++q) smin[q] = fminf(smin[q], acc[q]);
      }
#pragma unroll
      for (int q = 0; q < QG; ++q) samp[q * ScanCfg<M>::THREADS + tid] = f2ord(smin[q] + 0.0f);
      __syncthreads();
      RQ_STAT_ADD(8, t_ph);
      const uint32_t tk = radix_select_lds<QG, TPG, uint32_t>(&ctrl->st, samp + g * ScanCfg<M>::THREADS, ScanCfg<M>::THREADS,
                                                                srank, true, g, gi);
      if (gi == 0) ctrl->tau[g] = ord2f(tk);
      __syncthreads();
      RQ_STAT_ADD(1, t_ph);
    }
    // ---- pre-filter tables for this threshold (they live where the sample minima were) ---------------------
    // (__builtin_amdgcn_groupstaticsize() == 0: the byte tables are addressed absolutely, see the hot loop)
    bool filt_on = FILT && p.filter && attempt == 0 && (uint64_t)rows >= 64ull * (uint64_t)Ks &&
                   __builtin_amdgcn_groupstaticsize() == 0;
    constexpr uint32_t FILT_QCAP = QG * 256 / (ScanCfg<M>::THREADS / 64);   // queue entries per wavefront: st.hist split over the waves
    static_assert(!FILT || FILT_QCAP >= 128, "a wavefront's queue holds 64 waiting rows + one row-step of pushes");
    const uint32_t *qtab = samp;
    // this wavefront's queue: the offset is wave-uniform, so it lives in an SGPR (a VGPR pointer got spilled to
    // scratch and reloaded on every push: +10 % kernel time)
    uint32_t *const myq = &ctrl->st.hist[0][0] + __builtin_amdgcn_readfirstlane((uint32_t)(tid >> 6) * FILT_QCAP);
    uint32_t qtail = 0;      // wave-uniform
    if constexpr (FILT) {
      if (filt_on) {
        build_qtab<M, FINE, BIAS>(ctrl, lut4, gtab, samp, tid, p.norm_info, p.cnorm);
        __syncthreads();
      }
    }
    // second threshold estimate after ~1/8 of the rows (see retune_tau)
    uint32_t retune_at = 0xffffffffu, retune_rank = 0;
    const bool two_pass = attempt == 0 && will_retune && p.perm != nullptr;      // sample blocks first, then the sorted ones
    if (attempt == 0 && will_retune) {
      const uint32_t nb = two_pass ? nsamp : max(1u, rows / ((uint32_t)p.retune_div * (uint32_t)BLK));
      retune_at = two_pass ? nb : r_begin + nb * (uint32_t)BLK;     // two_pass: a count of processed blocks, else a position
      const float f = (float)(nb * (uint32_t)BLK) / (float)rows, mean = (float)p.K * f;
      const float rk = mean + (float)p.retune_z * sqrtf(mean * (1.0f - f)) + 2.0f;
      retune_rank = rk < 1.0f ? 1u : (uint32_t)ceilf(rk);
    }
    // wave-uniform values: keep them in SGPRs (as VGPRs they pushed the block's code words into scratch)
    retune_at = __builtin_amdgcn_readfirstlane(retune_at);
    retune_rank = __builtin_amdgcn_readfirstlane(retune_rank);
    t_ph = RQ_STAT_T();

    // ---- stream the slice -----------------------------------------------------------------------
    // (all of this loop control is wave-uniform: readfirstlane keeps it in SGPRs -- as VGPRs it spilled the block's code words)
    uint32_t bi = 0;       // blocks processed in this attempt
#if RQ_SCAN_PACE_BUILD
    // chunk pacing (big bases): blocks per chunk, 0 = this item is not paced (pad[1]: written with the item, read behind its barrier)
    const uint32_t pace_every = __builtin_amdgcn_readfirstlane(
        (p.pace != nullptr && attempt == 0 && ctrl->pad[1] != 0xffffffffu) ? p.pace_votes * (uint32_t)Cfg::VP : 0u);
#endif
    const uint32_t npass = __builtin_amdgcn_readfirstlane(two_pass ? 2u : 1u);
    const uint32_t samp_lim = __builtin_amdgcn_readfirstlane(two_pass ? min(r_end, p.samp_end) : 0u);
    const uint32_t samp_step = __builtin_amdgcn_readfirstlane(p.samp_stride * (uint32_t)BLK);
#pragma unroll 1
    for (uint32_t pass = 0; pass < npass; pass = __builtin_amdgcn_readfirstlane(pass + 1u))
#pragma unroll 1
    for (uint32_t base = r_begin; base < r_end; base += BLK) {
      if (npass == 2u) {
        const bool smp = base < samp_lim && base % samp_step == 0u;
        if (smp != (pass == 0u)) continue;
      }
      if (npass == 2u ? (pass == 1u && bi == retune_at) : (base == retune_at)) {
        if (FILT && filt_on) {
          while (qtail) {
            const uint32_t take = min(qtail, 64u);
            RQ_REFINE(ctrl, cand_wg, p.codes, p.row_bias, p.id_offset, p.cap, lut4, gtab, myq + (qtail - take), take);
            qtail -= take;
          }
        }
        __syncthreads();       // every append of the first rows has landed, the queues are empty
        vseq = retune_tau<M>(ctrl, cand_wg, p.cap, retune_rank, vseq);
        if constexpr (FILT) {
          if (filt_on) {
            build_qtab<M, FINE, BIAS>(ctrl, lut4, gtab, samp, tid, p.norm_info, p.cnorm);
            __syncthreads();
          }
        }
      }
      // ONE barrier per VP blocks.  Capacity invariant: cnt[q] + VP * BLK <= cap for every q when a period starts.
      // The pre-barrier read of cnt may miss what slower wavefronts are still appending for the
      // previous period (at most VP * BLK keys), hence cap = trigger + 2 * VP * BLK; behind the barrier cnt is exact.
      const bool vote_now = Cfg::VP == 1 || bi % (uint32_t)Cfg::VP == 0u;
#if RQ_SCAN_PACE_BUILD
      if (pace_every != 0u && vote_now && bi % pace_every == 0u && tid == 0)
        pace_chunk(p.pace + ctrl->pad[1], bi / pace_every, p.pace_lag);       // (the vote's barrier holds the other threads)
#endif
      const bool maybe = ctrl->cnt[g] > p.trigger;
      if (vote_now && block_any(maybe, ctrl->st.vote, vseq)) {
        const unsigned long long t_c = RQ_STAT_T();
        if (FILT && filt_on) {
          // the cut uses st.hist: every wavefront first runs its queued rows through the exact evaluation
          while (qtail) {
            const uint32_t take = min(qtail, 64u);
            RQ_REFINE(ctrl, cand_wg, p.codes, p.row_bias, p.id_offset, p.cap, lut4, gtab, myq + (qtail - take), take);
            qtail -= take;
          }
          __syncthreads();
        }
        const bool need = ctrl->cnt[g] > p.trigger;
        compact_group<M>(ctrl, cand_wg, p, need, g, gi, vseq);
        RQ_STAT_ADD(3, t_c);
        RQ_STAT_INC(6);
      }
      if constexpr (FILT) {
        // The filter pays off while few rows pass it (0.5 - 5 % on clustered data).  Tables without contrast
        // (e.g. random codes against random codebooks at m = 16) let a large share through; the first block's
        // count decides for the rest of the item, and the exact loop takes over after the queues are drained.
        if (filt_on && bi == (uint32_t)Cfg::VP) {
          // (60 % since the exact evaluation is per pair: at Deep1M shape, k = 10000, 29 % of the first block's rows are alive
          // and the filter still wins -- 10.2 ms against 14.9 with the old 12 / 30 % limits)
          constexpr uint32_t MAX_SHARE_PCT = FILT_MAX_SHARE_PCT;
          if (__builtin_amdgcn_readfirstlane(ctrl->fpush) * 100u > (uint32_t)BLK * MAX_SHARE_PCT) {
            while (qtail) {
              const uint32_t take = min(qtail, 64u);
              RQ_REFINE(ctrl, cand_wg, p.codes, p.row_bias, p.id_offset, p.cap, lut4, gtab, myq + (qtail - take), take);
              qtail -= take;
            }
            filt_on = false;
          }
        }
      }
      float tau[QG];
#pragma unroll
      for (int q = 0; q < QG; ++q) tau[q] = ctrl->tau[q];
      const uint32_t selmask = __builtin_amdgcn_readfirstlane(ctrl->selmask);
      const bool full_blk = __builtin_amdgcn_readfirstlane((uint32_t)(base + (uint32_t)BLK <= r_end)) != 0u;

      // (norm-adding kernels with 32 gathers per row: one sub-step at a time -- with all U code words live next to a row's
      // 64 + 32 registers of table entries the allocator spilled the code words themselves, 160 registers in all)
      constexpr int UCH = ((BIAS && M * Cfg::NQUAD >= 32 && 4 < Cfg::U) || (FILT && Cfg::U * RPT > 31)) ? 4 : Cfg::U;      // (m = 4: 4 rows per sub-step, one alive bit per row of a chunk)
#pragma unroll 1
      for (int uc = 0; uc < Cfg::U; uc += UCH) {
      // the thread's rows of this block: U sub-steps of RPT rows, each one packed little-endian
      // byte string (byte (r*M + k) of w[u]); all U loads are issued before the first gather
      static_assert((RPT * M) % 16 == 0, "a thread's rows are a whole number of 16-byte loads");
      uint32_t wu[UCH][RPT * M / 4];
      // LSQ, rows of 32 gathers: the rows' norms travel with their code words (a load per row inside the gather sequence
      // made every row wait for all memory traffic in flight: m = 16 43 -> 12.6 ms; at m = 8 the 16 extra registers
      // spill and the per-row load stays: 7.3 against 22.5 ms)
      constexpr bool PRE_BIAS = BIAS && M * Cfg::NQUAD >= 32;
      float bu[PRE_BIAS ? UCH : 1][RPT];
      uint32_t nbw[(FILT && BIAS) ? UCH : 1];   // LSQ pre-filter: RPT norm bytes per sub-step (RPT <= 4)
#pragma unroll
      for (int u = 0; u < UCH; ++u) {
        uint32_t *w = wu[u];
        const uint32_t row0 = base + (uint32_t)(uc + u) * Cfg::SUB + (uint32_t)tid * RPT;
#pragma unroll
        for (int r = 0; r < RPT; ++r)
          if constexpr (PRE_BIAS) bu[u][r] = (row0 + r < r_end && !(FILT && filt_on)) ? p.row_bias[row0 + r] : 0.0f;
        if constexpr (FILT && BIAS) {        // the rows' norm bytes: the "codes" of the row-norm table
          nbw[u] = 0;
          if (filt_on) {
            if (RPT == 2 && row0 + 2 <= r_end) nbw[u] = *reinterpret_cast<const uint16_t *>(p.norm_bytes + row0);
            else {
#pragma unroll
              for (int r = 0; r < RPT; ++r)
                if (row0 + r < r_end) nbw[u] |= (uint32_t)p.norm_bytes[row0 + r] << (8 * r);
            }
          }
        }
        if (row0 + RPT <= r_end) {
          const uint4 *src = reinterpret_cast<const uint4 *>(p.codes + (size_t)row0 * M);
#pragma unroll
          for (int i = 0; i < RPT * M / 16; ++i) {
            const uint4 v = src[i];
            w[4 * i + 0] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
          }
        } else {
          // ragged end of the slice: row by row, never touching bytes past row r_end-1
#pragma unroll
          for (int i = 0; i < RPT * M / 4; ++i) w[i] = 0;
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            if (row0 + r < r_end) {
              if constexpr (M % 4 == 0) {
#pragma unroll
                for (int i = 0; i < M / 4; ++i)
                  w[r * (M / 4) + i] = reinterpret_cast<const uint32_t *>(p.codes + (size_t)(row0 + r) * M)[i];
              } else {
#pragma unroll
                for (int k = 0; k < M; ++k)
                  w[(r * M + k) >> 2] |= (uint32_t)p.codes[(size_t)(row0 + r) * M + k] << (8 * ((r * M + k) & 3));
              }
            }
          }
        }
      }

      bool filtered = false;
      if constexpr (FILT) {
      if (filt_on) {
        filtered = true;
        uint32_t npush = 0;      // wave-uniform: rows this wavefront queued in this block
        uint32_t amask = 0;      // per lane: bit (u * RPT + r) = row r of sub-step u may still beat a threshold
        static_assert(UCH * RPT <= 31, "one alive bit per row of a chunk");
        asm volatile("; RQ_FILTER_LOOP_BEGIN" ::: "memory");     // markers for tests/test_isa.py (no instructions)
        // ---- pre-filter: byte lower bounds for the 8 queries, 8 bytes per gather; rows that may still beat a
        // threshold are queued for the exact evaluation, which runs 64 queued rows at a time
#pragma unroll
        for (int u = 0; u < UCH; ++u) {
          const uint32_t *w = wu[u];
          // all RPT * M gathers of the sub-step are issued before the first sum (16 x ds_read_b64 / 32 x ds_read_b32)
          using FV = typename FiltVec<M>::type;
          static_assert(sizeof(FV) == (size_t)QG, "byte tables: one byte per query of the group (ds_read_b64; b128 in the QG = 16 experiment)");
          const uint32_t shreg = sizeof(FV) == 16 ? 4u : 3u;
          // gathers in flight together: 16 (M = 8: both rows of the sub-step; M = 16: one row -- 32 of them with
          // their 32 addresses spill registers in this loop)
          constexpr int RB = (RPT * M * (int)sizeof(FV) > 128) ? 1 : RPT;      // rows per gather batch: <= 32 registers of entries
          FV e[RPT][M];
          FV en[BIAS ? RPT : 1];
#pragma unroll
          for (int rb = 0; rb < RPT; rb += RB) {
#pragma unroll
          for (int r = rb; r < rb + RB; ++r) {
#pragma unroll
            for (int k = 0; k < M; ++k) {
              // address = byte * sizeof(FV) (one SDWA shift) + compile-time offset of table k (in the instruction)
              constexpr int SH = sizeof(FV) == 16 ? 4 : 3;
              const uint32_t w32 = w[(r * M + k) >> 2];
              uint32_t boff;
              switch ((r * M + k) & 3) {
                case 0: boff = byte_shl<0, SH>(w32, shreg); break;
                case 1: boff = byte_shl<1, SH>(w32, shreg); break;
                case 2: boff = byte_shl<2, SH>(w32, shreg); break;
                default: boff = byte_shl<3, SH>(w32, shreg); break;
              }
              // absolute LDS address = CTRL_BYTES + k * table + byte * entry: the kernel has no static LDS (its dynamic
              // segment starts at 0, checked once per launch through filt_on), so the constant part folds into the
              // ds_read offset field and the whole address costs the one SDWA shift above
              e[r][k] = lds_abs_load<FV>(boff + (uint32_t)(CTRL_BYTES + k * 256 * sizeof(FV)));
            }
            if constexpr (BIAS) {      // the row-norm table, indexed by the row's norm byte
              constexpr int SHN = sizeof(FV) == 16 ? 4 : 3;
              uint32_t boff;
              switch (r & 3) {
                case 0: boff = byte_shl<0, SHN>(nbw[u], shreg); break;
                case 1: boff = byte_shl<1, SHN>(nbw[u], shreg); break;
                case 2: boff = byte_shl<2, SHN>(nbw[u], shreg); break;
                default: boff = byte_shl<3, SHN>(nbw[u], shreg); break;
              }
              en[r] = lds_abs_load<FV>(boff + (uint32_t)(CTRL_BYTES + M * 256 * sizeof(FV)));
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = rb; r < rb + RB; ++r) {
            uint32_t a[Cfg::NACC * Cfg::NQUAD];
#pragma unroll
            for (int i = 0; i < Cfg::NACC * Cfg::NQUAD; ++i) a[i] = 0;
#pragma unroll
            for (int k = 0; k < M; ++k) {
#pragma unroll
              for (int j = 0; j < Cfg::NQUAD; ++j) a[(k / Cfg::kpa(FINE)) * Cfg::NQUAD + j] += fv_word(e[r][k], j);
            }
            if constexpr (BIAS) {
#pragma unroll
              for (int j = 0; j < Cfg::NQUAD; ++j) a[(Cfg::NACC - 1) * Cfg::NQUAD + j] += fv_word(en[r], j);
            }
            // one bit per row of the chunk in a per-lane mask: the rows are queued once per chunk below (a ballot + prefix
            // count + LDS store per ROW cost ~11 VALU instructions, a fifth of the loop's VALU time -- and with 5 % of the
            // rows alive almost every ballot of 64 rows finds somebody)
            amask |= filt_alive<M, FINE, BIAS>(a) ? (1u << (u * RPT + r)) : 0u;
          }
          __builtin_amdgcn_sched_barrier(0);
          }  // gather batches
        }
        asm volatile("; RQ_FILTER_GATHERS_END" ::: "memory");    // end of the streaming part (gathers + byte sums + alive bits)
        if (!full_blk) {          // rows behind the slice's end (only a slice's last block is ragged)
#pragma unroll
          for (int u = 0; u < UCH; ++u) {
            const uint32_t row0 = base + (uint32_t)(uc + u) * Cfg::SUB + (uint32_t)tid * RPT;
#pragma unroll
            for (int r = 0; r < RPT; ++r)
              if (!(row0 + r < r_end)) amask &= ~(1u << (u * RPT + r));
          }
        }
        // queue the alive rows: every lane hands over one row per round (at most 64 per round, so the queue -- which is
        // drained whenever 64 rows wait -- never holds more than 127 <= FILT_QCAP); 1-4 rounds per chunk of 16 rows
        {
          const uint32_t rowt = base + (uint32_t)uc * Cfg::SUB + (uint32_t)tid * RPT;
          for (;;) {
            const bool has = amask != 0u;
            const uint64_t mq = __builtin_amdgcn_ballot_w64(has);
            if (!mq) break;
            const uint32_t b = (uint32_t)__builtin_ctz(amask | 0x80000000u);
            amask &= amask - 1u;
            if (has) myq[qtail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u))] =
                rowt + (b / (uint32_t)RPT) * (uint32_t)Cfg::SUB + (b % (uint32_t)RPT);
            qtail += (uint32_t)__popcll(mq);
            npush += (uint32_t)__popcll(mq);
            if (qtail >= 64u) {
              // (two 64-row batches per call would overlap the call's dependent latencies -- ~2 us for queue -> code
              // bytes -> table gathers -- but the larger callee makes every call save more registers: 3.64 -> 4.16 ms)
              RQ_REFINE(ctrl, cand_wg, p.codes, p.row_bias, p.id_offset, p.cap, lut4, gtab, myq + (qtail - 64u), 64u);
              qtail -= 64u;
            }
          }
        }
        asm volatile("; RQ_FILTER_LOOP_END" ::: "memory");
        if (bi == 0u && lane == 0) atomicAdd(&ctrl->fpush, npush);
      }
      }
      if (!filtered) {
#pragma unroll
      for (int u = 0; u < UCH; ++u) {
      const uint32_t *w = wu[u];
      const uint32_t row0 = base + (uint32_t)(uc + u) * Cfg::SUB + (uint32_t)tid * RPT;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        float acc[QG];
        row_dists<M>(w, r, lut4, gtab, acc);
        if (BIAS) {   // deps/src/linscan_aqd_pairwise_byte.cpp:74  pairs[j].first += dbnorms[normidx]
          float bias;
          if constexpr (PRE_BIAS) bias = bu[u][r];
          else bias = (row0 + r < r_end) ? p.row_bias[row0 + r] : 0.0f;
#pragma unroll
          for (int q = 0; q < QG; ++q) acc[q] = acc[q] + bias;
        }
        // ---- survivors: rows whose distance beats the query's threshold ------------------------
        emit_survivors<QG>(acc, tau, row0 + r < r_end, row0 + r, p.perm, p.id_offset, selmask, ctrl, cand_wg, p.cap, lane);
        // one row at a time when a row's gathers alone fill half of the register budget (m = 16, 8 queries: 32 x 16 bytes):
        // interleaving two rows spilled 160 registers in the norm-adding (LSQ) variant
        if constexpr (M * Cfg::NQUAD >= 32) __builtin_amdgcn_sched_barrier(0);
      }
      }  // sub-steps
      }
      }  // sub-step chunks
      bi = __builtin_amdgcn_readfirstlane(bi + 1u);
    }
    if (FILT && filt_on) {
      while (qtail) {          // slice end: the rest of the queue
        const uint32_t take = min(qtail, 64u);
        RQ_REFINE(ctrl, cand_wg, p.codes, p.row_bias, p.id_offset, p.cap, lut4, gtab, myq + (qtail - take), take);
        qtail -= take;
      }
    }
    __syncthreads();   // every append of the slice has landed

    RQ_STAT_ADD(2, t_ph);
    if (p.stats && tid == 0) {       // filter bookkeeping: items, items that kept the filter, first-block pushes, rows
      atomicAdd(&p.stats[12], 1ull);
      if (FILT && filt_on) atomicAdd(&p.stats[13], 1ull);
      atomicAdd(&p.stats[14], (unsigned long long)ctrl->fpush);
      atomicAdd(&p.stats[15], (unsigned long long)min(rows, (uint32_t)BLK));
    }
    t_ph = RQ_STAT_T();
    // ---- finish the item: cut to K, sort, write ----------------------------------------------
    if (attempt == 0) {
      // the sampled tau must have let at least min(K, rows) rows through for EVERY query
      // (the buffer holds exactly the rows with dist <= tau, also after a second estimate: retune_tau)
      const bool shortfall = ctrl->cnt[g] < min((uint32_t)p.K, rows);
      if (block_any(shortfall, ctrl->st.vote, vseq)) continue;  // exact fallback: redo the slice from tau = +inf
    }
    if (!p.bigk && !p.bfin) {
      bool need = ctrl->cnt[g] > (uint32_t)p.K;
      if (block_any(need, ctrl->st.vote, vseq)) compact_group<M>(ctrl, cand_wg, p, need, g, gi, vseq);
    }
    break;
    }  // attempts
    RQ_STAT_ADD(4, t_ph);
    t_ph = RQ_STAT_T();
    if (p.bigk) {
      // large K: select + sort in one sample-sort pass per query, keys stay in global memory
      finish_bigk<Cfg::THREADS>(ctrl->cnt, ctrl->sel, QG, cand_wg, p.bkt + (size_t)blockIdx.x * p.cap, p.cap, (uint32_t)p.K,
                  q0, p.nq, keys_base, key_stride, p.dists, p.ids,
                  (uint32_t)p.id_base, smem + CTRL_BYTES, p.stats, p.bigk != 2);
      RQ_STAT_ADD(5, t_ph);
      continue;
    }
    if (p.bfin) {
      // Round 5: one wavefront per query cuts and sorts its candidates through distance buckets (bucket_finish_item above,
      // bucket_finish_wave in rq_topk.h) -- no barrier inside, a fraction of the select + sorting-network work.  Out of line: with
      // the attempt inlined the streaming loop's register allocation changed (k = 1 +2 %, the 1.25e8-row shard +8 %).
      BucketFinishArgs fa;
      fa.cap = p.cap; fa.scratch_keys = p.scratch_keys; fa.nq = p.nq; fa.K = (uint32_t)p.K; fa.id_base = (uint32_t)p.id_base;
      fa.bfin = p.bfin; fa.dists = p.dists; fa.ids = p.ids; fa.stats = p.stats;
      uint32_t vseq_new = vseq;
      const bool gave_up = bucket_finish_item<M>(ctrl, cand_wg, fa, scratch, g, gi, q0, keys_base, key_stride, vseq, &vseq_new);
      vseq = vseq_new;
      // (a group that skipped the attempt votes "gave up" with every thread: uniform either way)
      if (!block_any(gave_up, ctrl->st.vote, vseq)) {
        RQ_STAT_ADD(5, t_ph);
        continue;
      }
#ifndef RQ_NO_FINSTATS
      RQ_STAT_INC(18);     // groups that went on to select + sort (the look's skips included)
#endif
      bool need = ctrl->cnt[g] > (uint32_t)p.K;
      if (block_any(need, ctrl->st.vote, vseq)) compact_group<M>(ctrl, cand_wg, p, need, g, gi, vseq);
    }
    // sort `nconc` queries at a time in LDS (scratch aliases the LUT, which is dead now)
    uint32_t nconc = p.scratch_keys / p.p2;
    if (nconc > (uint32_t)QG) nconc = QG;
    // round down to a power of two so the thread split is even
    while (nconc & (nconc - 1)) nconc &= nconc - 1;
    const uint32_t tps = ScanCfg<M>::THREADS / nconc;  // threads per concurrent sort
    const uint32_t sg = tid / tps, sgi = tid % tps;
    for (uint32_t qb = 0; qb < (uint32_t)QG; qb += nconc) {
      const uint32_t q = qb + sg;
      const bool act = q < (uint32_t)QG;
      uint64_t *a = scratch + (size_t)sg * p.p2;
      const uint32_t cnt = act ? ctrl->cnt[q] : 0;
      const uint64_t *src = cand_wg + ((size_t)(act ? q : 0) * 2 + (act ? ctrl->sel[q] : 0)) * p.cap;
      __syncthreads();
      unsigned long long t_s = RQ_STAT_T();
      if (act)
        for (uint32_t i = sgi; i < p.p2; i += tps) a[i] = i < cnt ? src[i] : KEY_MAX;
      __syncthreads();
      RQ_STAT_ADD(9, t_s);
      t_s = RQ_STAT_T();
      bitonic_sort_tiled(a, p.p2, tps / 64, sgi / 64, sgi & 63, act);
      RQ_STAT_ADD(10, t_s);
      t_s = RQ_STAT_T();
      const uint32_t qq = q0 + q;
      if (act && qq < p.nq) {
        if (keys_base) {
          uint64_t *o = keys_base + (size_t)qq * key_stride;
          for (uint32_t i = sgi; i < (uint32_t)p.K; i += tps) o[i] = a[i];
        } else {
          float *od = p.dists + (size_t)qq * p.K;
          uint32_t *oi = p.ids + (size_t)qq * p.K;
          for (uint32_t i = sgi; i < (uint32_t)p.K; i += tps) {
            const uint64_t key = a[i];
            od[i] = key_dist(key);
            oi[i] = key_id(key) + (uint32_t)p.id_base;
          }
        }
      }
      __syncthreads();
      RQ_STAT_ADD(11, t_s);
    }
    RQ_STAT_ADD(5, t_ph);
  }
}

// ------------------------------------------------------------------------------------------
// Merge P sorted key lists per query (slices of one GPU and/or shards of several GPUs).
// One 256-thread workgroup per query: radix-select the K-th key over the P*K keys in global
// memory, pull the K survivors into LDS, sort, write.
// ------------------------------------------------------------------------------------------
constexpr int MERGE_THREADS = 256;

struct MergeParams {
  const uint64_t *keys_in;  // [nq][P][K]
  uint32_t nq, P;
  int K, id_base;
  int use_map;              // large K: buckets from the distance map first (SCAN_SS_MAP)
  uint32_t p2;
  float *dists;
  uint32_t *ids;
  uint64_t *keys_out;
};

struct MergeCtrl {
  SelState<1> st;
};

__global__ __launch_bounds__(MERGE_THREADS) void merge_topk_kernel(MergeParams p) {
  constexpr int CTRL_BYTES = (sizeof(MergeCtrl) + 15) & ~15;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MergeCtrl *ctrl = reinterpret_cast<MergeCtrl *>(smem);
  uint64_t *a = reinterpret_cast<uint64_t *>(smem + CTRL_BYTES);
  const int tid = threadIdx.x;
  const uint32_t q = blockIdx.x;
  const uint64_t *src = p.keys_in + (size_t)q * p.P * p.K;
  const uint32_t cnt = p.P * (uint32_t)p.K;
  for (uint32_t i = tid; i < p.p2; i += MERGE_THREADS) a[i] = KEY_MAX;
  uint32_t vseq = 0;
  if (tid < 2) ctrl->st.vote[tid] = 0;
  __syncthreads();
  radix_select<1, MERGE_THREADS>(&ctrl->st, src, cnt, (uint32_t)p.K, true, 0, tid, vseq);
  const uint64_t tau = ctrl->st.prefix[0];
  // survivors straight into LDS (padding keys equal to KEY_MAX never survive unless tau is one)
  if (tid == 0) ctrl->st.newcnt[0] = 0;
  __syncthreads();
  {
    const int lane = tid & 63;
    const uint32_t cnt_up = (cnt + 63u) & ~63u;
    for (uint32_t idx = tid; idx < cnt_up; idx += MERGE_THREADS) {
      const uint64_t key = idx < cnt ? src[idx] : KEY_MAX;
      const bool take = idx < cnt && key <= tau && key != KEY_MAX;
      const uint64_t mask = __ballot(take);
      if (mask) {
        const int leader = __ffsll((unsigned long long)mask) - 1;
        uint32_t basep = 0;
        if (lane == leader) basep = atomicAdd(&ctrl->st.newcnt[0], (uint32_t)__popcll(mask));
        basep = __builtin_amdgcn_readlane(basep, leader);
        if (take) {
          const uint32_t pos = basep + __popcll(mask & ((1ull << lane) - 1ull));
          if (pos < p.p2) a[pos] = key;
        }
      }
    }
  }
  __syncthreads();
  bitonic_sort_tiled(a, p.p2, MERGE_THREADS / 64, tid / 64, tid & 63, true);
  for (uint32_t i = tid; i < (uint32_t)p.K; i += MERGE_THREADS) {
    const uint64_t key = a[i];
    if (p.keys_out) p.keys_out[(size_t)q * p.K + i] = key;
    if (p.dists) p.dists[(size_t)q * p.K + i] = key_dist(key);
    if (p.ids) p.ids[(size_t)q * p.K + i] = key_id(key) + (uint32_t)p.id_base;
  }
}

// Large K: the same merge as a sample sort over the P*K keys (no K-sized LDS array).  Persistent
// 512-thread workgroups, each with P*K keys + P*K u16 of global scratch.
__global__ __launch_bounds__(SCAN_THREADS) void merge_topk_big_kernel(MergeParams p, uint64_t *scratch, uint16_t *bkt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t cnt = p.P * (uint32_t)p.K, K = (uint32_t)p.K, tid = threadIdx.x;
  uint64_t *dst = scratch + (size_t)blockIdx.x * cnt;
  uint16_t *b = bkt + (size_t)blockIdx.x * cnt;
  for (uint32_t q = blockIdx.x; q < p.nq; q += gridDim.x) {
    const uint64_t *src = p.keys_in + (size_t)q * cnt;
    uint64_t *ok = p.keys_out ? p.keys_out + (size_t)q * K : nullptr;
    float *od = p.dists ? p.dists + (size_t)q * K : nullptr;
    uint32_t *oi = p.ids ? p.ids + (size_t)q * K : nullptr;
    const uint32_t id_base = (uint32_t)p.id_base;
    auto emit = [ok, od, oi, id_base](uint32_t r, uint64_t key) {
      if (ok) ok[r] = key;
      if (od) od[r] = key_dist(key);
      if (oi) oi[r] = key_id(key) + id_base;
    };
    const uint32_t got = samplesort_topk<SCAN_THREADS>(src, dst, b, cnt, K, smem, emit, nullptr, p.use_map != 0);
    for (uint32_t i = got + tid; i < K; i += SCAN_THREADS) emit(i, KEY_MAX);   // fewer than K real keys
  }
}

// P == 1: the list is already the answer, only unpack it
__global__ void unpack_keys_kernel(MergeParams p) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)p.nq * p.K) return;
  const uint64_t key = p.keys_in[i];
  if (p.keys_out) p.keys_out[i] = key;
  if (p.dists) p.dists[i] = key_dist(key);
  if (p.ids) p.ids[i] = key_id(key) + (uint32_t)p.id_base;
}

// Stand-alone LUT kernel (test aid, rq_dev_adc_lut): lut[q][k][r], one thread per (q,k,r).
__global__ void adc_lut_kernel(float *lut, const float *centers, const float *queries, uint32_t nq,
                               int m, int sub) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per_q = (size_t)m * 256;
  if (e >= per_q * nq) return;
  const size_t q = e / per_q;
  const int t = (int)(e - q * per_q);
  const int k = t >> 8;
  const float *c = centers + (size_t)t * sub;
  const float *qv = queries + q * (size_t)(m * sub) + (size_t)k * sub;
  float acc = 0.0f;
  for (int s = 0; s < sub; ++s) {
    const float diff = c[s] - qv[s];
    const float sq = diff * diff;
    acc = acc + sq;
  }
  lut[e] = acc;
}

// codes [n][m] -> [n][mp] with zero padding bytes (m not one of the tiled widths)
__global__ void pad_codes_kernel(uint8_t *dst, const uint8_t *src, size_t n, int m, int mp) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * (size_t)mp) return;
  const size_t r = e / mp;
  const int k = (int)(e - r * mp);
  dst[e] = k < m ? src[r * m + k] : (uint8_t)0;
}

__global__ void synth_codes_kernel(uint8_t *codes, size_t nbytes, uint64_t seed, uint64_t e0) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= nbytes) return;
  uint64_t out = 0;
  const int lim = (int)((nbytes - i) < 8 ? (nbytes - i) : 8);
  for (int b = 0; b < lim; ++b) {
    uint64_t z = ((e0 + i + b) ^ seed) + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    out |= (z >> 56) << (8 * b);
  }
  if (lim == 8) *reinterpret_cast<uint64_t *>(codes + i) = out;
  else for (int b = 0; b < lim; ++b) codes[i + b] = (uint8_t)(out >> (8 * b));
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local char g_last_scan_kernel[64] = "";      // the calling thread's last scan launch (template arguments as rocprofv3 prints them)
const char *last_scan_kernel_name() { return g_last_scan_kernel; }

template <int M>
static int launch_scan(ScanParams &p, const ScanPlan &plan, hipStream_t stream) {
  using Cfg = ScanCfg<M>;
  p.samp_end = 0; p.samp_stride = 16;
  if (p.perm) {       // the ordered base's sample blocks (the same arithmetic as order_rows_launch)
    uint32_t groups = 0;
    (void)order_sample_rows((int64_t)p.n, Cfg::BLK, &groups);
    p.samp_stride = (uint32_t)std::max(2, order_sample_stride());
    p.samp_end = groups * p.samp_stride * (uint32_t)Cfg::BLK;
  }
  constexpr int CTRL_BYTES = (sizeof(ScanCtrl<Cfg::QG>) + 15) & ~15;
  size_t lds = CTRL_BYTES + (size_t)std::max<size_t>(Cfg::LUT_LDS_BYTES + (size_t)Cfg::QG * p.d * 4 + Cfg::AUX_BYTES,
                                                     (size_t)p.scratch_keys * 8);
  if (p.bigk) lds = std::max<size_t>(lds, CTRL_BYTES + SS_LDS_BYTES);
  // SCAN_SPREAD: asking for more than half of the LDS forces one workgroup per CU
  if (plan.spread) lds = std::max<size_t>(lds, 84 * 1024);
  // the pre-filter is tiled for M = 8 and 16 and needs non-negative table entries (PQ and CQ tables are sums of squares)
  void (*kern)(ScanParams) = p.row_bias ? adc_scan_kernel<M, true, false> : adc_scan_kernel<M, false, false>;
  bool t_bias = p.row_bias != nullptr, t_filt = false, t_fine = false;       // the instantiation, for rq_last_scan_kernel()
  if constexpr (Cfg::HAS_FILT) {
    if (p.filter) { kern = adc_scan_kernel<M, false, true>; t_filt = true; }
    // LSQ (signed tables + row norms): the two-set byte sums (m = 8: 4 and 4 + 1 entries, m = 16: 8 and 8 + 1)
    if constexpr (M >= 8) {      // (m = 4: PQ / CQ tables only -- the LSQ preparation is written for the 8- and 16-byte tilings)
      if (p.filter && p.row_bias) { kern = adc_scan_kernel<M, true, true, (M == 8)>; t_fine = (M == 8); }
    }
    // m = 8, large k: the finer byte tables (6-bit entries, two sum sets) -- more VALU work per row, fewer rows for
    // the exact evaluation.  Measured at SIFT1M shape, coarse vs fine: k = 1 2.23 / 2.34 ms, k = 100 2.36 / 2.39,
    // k = 1000 2.91 / 2.91, k = 10000 6.62 / 6.26.
    if constexpr (M == 8) {
      int fine_k = tuning("SCAN_FINE_MIN_K", 0);
      if (fine_k <= 0) fine_k = 8192;            // crossover measured between k = 4096 (level) and 10000
      if (p.filter && !p.row_bias && p.K >= fine_k) { kern = adc_scan_kernel<M, false, true, true>; t_fine = true; }
    }
  } else {
    p.filter = 0;
  }
  snprintf(g_last_scan_kernel, sizeof(g_last_scan_kernel), "adc_scan_kernel<%d, %s, %s, %s>", M, t_bias ? "true" : "false",
           t_filt ? "true" : "false", t_fine ? "true" : "false");
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(plan.grid), dim3(Cfg::THREADS), lds, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

template <int M>
static void plan_for(ScanPlan &pl, int64_t n, int64_t nq, int d, int K, int num_cu, int force_slices) {
  using Cfg = ScanCfg<M>;
  constexpr int CTRL_BYTES = (sizeof(ScanCtrl<Cfg::QG>) + 15) & ~15;
  pl.qg = Cfg::QG;
  pl.blk = Cfg::BLK;
  pl.ngroups = (uint32_t)((nq + Cfg::QG - 1) / Cfg::QG);
  // slack between cuts: with the sampled tau about 2-3K rows survive a slice, so a slack of 4K means
  // "no cut before the end"; the exact fallback (tau from +inf) cuts every `slack` survivors.
  int slack_i = tuning("SCAN_SLACK", 0);
  if (slack_i <= 0) slack_i = std::max(std::min(std::max(4 * K, 2048), 16384), K + K / 2);
  const uint32_t slack = (uint32_t)slack_i;
  pl.trigger = (uint32_t)K + slack;
  pl.sample = (uint32_t)tuning("SCAN_SAMPLE", 16384);
  // + 2048: rows waiting in the pre-filter's queues (< 128 per wavefront) are appended outside their block
  pl.cap = pl.trigger + 2 * Cfg::VP * Cfg::BLK + 2048;
  pl.p2 = next_pow2((uint32_t)K);
  // large K finishes with the global-memory sample sort: its LDS need does not grow with K
  pl.bigk = K > tuning("SCAN_SS_MIN_K", 1024);
  // LDS scratch for the final sort: at least one query, at most QG, within 160 KiB total
  const size_t lds_max = 160 * 1024 - CTRL_BYTES;
  size_t want_keys = (size_t)pl.p2 * Cfg::QG;
  size_t base_keys = ((size_t)Cfg::LUT_LDS_BYTES + (size_t)Cfg::QG * d * 4 + Cfg::AUX_BYTES) / 8;
  size_t keys = std::max(base_keys, std::min(want_keys, (size_t)64 * 1024 / 8 * 2));
  keys = std::max(keys, (size_t)pl.p2);
  if (pl.bigk) keys = std::max(base_keys, (size_t)(SS_LDS_BYTES + 7) / 8);
  keys = std::min(keys, lds_max / 8);
  pl.scratch_keys = (uint32_t)keys;
  // workgroups per CU: as many as the LDS request admits (2 x 512 threads when it is <= 80 KiB)
  const size_t lds_req = CTRL_BYTES + std::max<size_t>(base_keys * 8, (size_t)pl.scratch_keys * 8);
  const int wgs = std::max(1, std::min<int>(1024 / Cfg::THREADS, (int)(160 * 1024 / lds_req)));
  // Work items = (8-query group, row range), handed out by an atomic counter to the resident workgroups
  // (2 per CU).  Per item there is a fixed cost (LUT, threshold sample, select + sort, and a merge pass for
  // sliced groups), so whole-base items are best whenever they fill the chip:
  //  * fewer groups than resident workgroups: every group is cut into slots/groups row slices
  //    (1000 queries, K = 1000: 4 slices 1.11 ms; 1, 2, 6, 9 slices 1.85, 1.18, 1.37, 1.64 ms);
  //  * otherwise whole items, except a small last partial round: when the groups beyond a multiple of
  //    num_cu are at most num_cu/2 they are cut in two, so that they occupy twice as many CUs for half
  //    as long (5000 queries: 3.71 -> 3.05 ms, 7000: 4.76 -> 4.10 ms; with a larger remainder, or finer
  //    cuts, the per-item cost wins: 10000 queries 5.80 -> 5.91 ms, so those stay whole).
  int64_t min_rows = std::max<int64_t>(tuning("SCAN_MIN_ROWS", 16384), 32LL * K);
  min_rows = (min_rows + Cfg::BLK - 1) / Cfg::BLK * Cfg::BLK;
  const int64_t max_slices = std::max<int64_t>(1, n / min_rows);
  const int64_t U = num_cu, slots = (int64_t)num_cu * wgs;
  int64_t whole = 0, tail = pl.ngroups, ns = 1;
  if ((int64_t)pl.ngroups < slots) {
    ns = std::min<int64_t>(std::max<int64_t>(1, slots / pl.ngroups), max_slices);
  } else {
    whole = (pl.ngroups / U) * U;
    tail = pl.ngroups - whole;
    if (tail > 0 && 2 * tail <= U && max_slices >= 2) ns = 2;
  }
  const int tail_ns = tuning("SCAN_TAIL_SLICES", 0);           // experiments: slices of the tail groups only
  if (tail_ns > 0 && tail > 0) ns = std::min<int64_t>(tail_ns, max_slices);
  if (force_slices > 0) { ns = force_slices; whole = 0; }   // tests: every group sliced
  // Big bases (code array >= SCAN_XCD_MIN_MB, far beyond the 8 x 4 MiB of L2): short row windows, handed out per XCD
  // (adc_scan_kernel: xcd_mode).  A window is SCAN_WINDOW_MB of codes: small enough that the workgroups of an XCD
  // that started it together are still within L2 reach of each other at its end (their speeds differ by a few percent),
  // large enough that the per-item work (tables, threshold sample, final select + sort) stays around one percent.  Bounds: the per-slice
  // key lists (nq * ns * K * 8 bytes) stay below 1 GiB and a query's merge below 2^20 keys.
  pl.xcd = false;
  // the window hand-out and its pacing are written for the 8 XCDs x 32 CUs of an MI355X in SPX mode (xcd & 7, grid / 8); a
  // partitioned device (CPX / DPX: fewer XCDs behind one agent) takes the plain planner -- results are the same either way
  const bool eight_xcds = num_cu == 256 || tuning("SCAN_XCD", 1) > 1;
  if (force_slices <= 0 && tail_ns <= 0 && tuning("SCAN_XCD", 1) && eight_xcds &&
      (int64_t)n * M >= ((int64_t)(tuning("SCAN_XCD_MIN_MB", 0) > 0 ? tuning("SCAN_XCD_MIN_MB", 0) : 128) << 20)) {
    int64_t wrows = ((int64_t)(tuning("SCAN_WINDOW_MB", 0) > 0 ? tuning("SCAN_WINDOW_MB", 0) : 32) << 20) / M;
    wrows = std::max<int64_t>(wrows, min_rows);
    int64_t nw = (n + wrows - 1) / wrows;
    // per-window key lists: at most 256 MB of scratch per (device, stream) (1 GiB x 8 streams was ~10 GiB hidden)
    const int64_t cap_part = ((int64_t)256 << 20) / std::max<int64_t>(1, nq * (int64_t)K * 8);
    const int64_t cap_merge = ((int64_t)1 << 20) / std::max(1, K);
    nw = std::min(nw, std::min(cap_part, cap_merge));
    if (nw >= 16) { ns = nw; whole = 0; pl.xcd = true; }
  }
  if (ns == 1) whole = pl.ngroups;
  int64_t rps = (n + ns - 1) / ns;
  rps = (rps + Cfg::BLK - 1) / Cfg::BLK * Cfg::BLK;
  ns = (n + rps - 1) / rps;
  if (ns == 1) whole = pl.ngroups;
  if (ns < 16) pl.xcd = false;
  pl.nslices = (uint32_t)ns;
  pl.rows_per_slice = (uint32_t)rps;
  pl.whole = (uint32_t)whole;
  const uint64_t items = (uint64_t)pl.whole + (uint64_t)(pl.ngroups - pl.whole) * pl.nslices;
  // experiment knob: one workgroup per CU (made no difference for small grids -- the dispatcher already
  // spreads them -- and costs 24 % at 4096 queries, so it is off)
  pl.spread = tuning("SCAN_SPREAD", 0) > 0;
  pl.grid = (uint32_t)std::min<uint64_t>(items, (uint64_t)num_cu * (pl.spread ? 1 : wgs));
  pl.cand_bytes = (size_t)pl.grid * Cfg::QG * 2 * pl.cap * sizeof(uint64_t);
  pl.gtab_off = pl.cand_bytes;   // the L1-gathered LUT parts live behind the candidate buffers
  pl.cand_bytes += (size_t)pl.grid * (Cfg::GTAB_F4 > 0 ? Cfg::GTAB_F4 : 1) * sizeof(float4);
  pl.bkt_off = pl.cand_bytes;
  if (pl.bigk) pl.cand_bytes += ((size_t)pl.grid * pl.cap * sizeof(uint16_t) + 15) & ~(size_t)15;
  pl.lds_ok = (pl.bigk || (size_t)pl.p2 * 8 <= lds_max) &&
              ((size_t)Cfg::LUT_LDS_BYTES + (size_t)Cfg::QG * d * 4 + Cfg::AUX_BYTES <= lds_max);
}

int scan_padded_m(int m) {
  for (int mp : {2, 4, 8, 16, 32, 64})
    if (m <= mp) return mp;
  return -1;
}

int scan_plan(ScanPlan &pl, int64_t n, int64_t nq, int m, int d, int K, int num_cu, int force_slices) {
  m = scan_padded_m(m);
  switch (m) {
    case 2: plan_for<2>(pl, n, nq, d, K, num_cu, force_slices); break;
    case 4: plan_for<4>(pl, n, nq, d, K, num_cu, force_slices); break;
    case 8: plan_for<8>(pl, n, nq, d, K, num_cu, force_slices); break;
    case 16: plan_for<16>(pl, n, nq, d, K, num_cu, force_slices); break;
    case 32: plan_for<32>(pl, n, nq, d, K, num_cu, force_slices); break;
    case 64: plan_for<64>(pl, n, nq, d, K, num_cu, force_slices); break;
    default:
      return fail(RQ_EUNSUPPORTED, "the ADC scan kernels cover 1 <= m <= 64 sub-quantizers");
  }
  if (!pl.lds_ok) return fail(RQ_EUNSUPPORTED, "k=%d / d=%d does not fit the 160 KiB LDS plan", K, d);
  return RQ_OK;
}

// ---- LSQ pre-filter: the rows' norms as one byte each (the "code" of the row-norm table, see build_qtab) -----------
// info (uint32 view): [0] ordered(min), [1] ordered(max); (float view) [4] nmin, [5] nstep, [6] max |norm|
// |c_k[r]|^2 of the full-dimensional codebooks [m * 256][d] (any rounding will do: the residual below uses these values)
__global__ void cnorm_kernel(const float *__restrict__ cb, int entries, int d, float *cn) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= entries) return;
  float s = 0.0f;
  for (int i = 0; i < d; ++i) s = __builtin_fmaf(cb[(size_t)e * d + i], cb[(size_t)e * d + i], s);
  cn[e] = s;
}

// rho(row) = norm(row) - sum_k cn[k][b_k], in float64 (exact for f32 inputs up to m = 16) and rounded DOWN to f32
__global__ void norm_residual_kernel(const float *__restrict__ nrm, const uint8_t *__restrict__ codes, uint32_t n, int mrow,
                                     int mreal, const float *__restrict__ cn, float *rho) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double s = (double)nrm[i];
    for (int k = 0; k < mreal; ++k) s -= (double)cn[k * 256 + codes[(size_t)i * mrow + k]];
    rho[i] = __double2float_rd(s);
  }
}

__global__ void norm_minmax_kernel(const float *__restrict__ nrm, uint32_t n, uint32_t *info) {
  uint32_t lo = 0xffffffffu, hi = 0u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t o = f2ord(nrm[i] + 0.0f);
    lo = min(lo, o);
    hi = max(hi, o);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, off));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, off));
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&info[0], lo); atomicMax(&info[1], hi); }
}

__global__ void norm_info_kernel(uint32_t *info) {
  float *f = reinterpret_cast<float *>(info);
  const float nmin = ord2f(info[0]), nmax = ord2f(info[1]);
  float nstep = (nmax - nmin) / 255.0f;
  if (!(nstep > 0.0f) || !(nstep < __uint_as_float(0x7f800000u))) nstep = 0.0f;   // constant norms (or not finite): one cell
  f[4] = nmin;
  f[5] = nstep;
  f[6] = fmaxf(fabsf(nmin), fabsf(nmax));
}

__global__ void norm_quant_kernel(const float *__restrict__ nrm, uint32_t n, const uint32_t *info, uint8_t *nb) {
  const float *f = reinterpret_cast<const float *>(info);
  const float nmin = f[4], nstep = f[5];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = nrm[i];
    uint32_t b = 0;
    if (nstep > 0.0f) {
      const float t = (v - nmin) / nstep;
      b = t >= 255.0f ? 255u : t > 0.0f ? (uint32_t)t : 0u;
      while (b > 0 && !(norm_edge(nmin, nstep, b) <= v)) --b;     // the cell's lower edge must not exceed the norm
    }
    nb[i] = (uint8_t)b;
  }
}

// LSQ pre-filter preprocessing of a base (O(n), independent of the queries): |c|^2 of the codebooks, the rows' cross-term
// norms rho = norm - sum_k |c_k[b_k]|^2 (float64, rounded down), their range and one byte per row.
// norm_buf: [n rounded up to 64] row bytes | 8 words of info | |c|^2 [16][256] f32 | rho [n] f32   (lsq_norm_bytes)
size_t lsq_norm_bytes(int64_t n) { return (((size_t)n + 63) & ~(size_t)63) + 32 + 16 * 256 * 4 + (size_t)n * 4; }

int lsq_norm_prepare(uint8_t *norm_buf, const uint8_t *codes, const float *centers, const float *row_bias, int64_t n,
                     int mp, int m_real, int d, hipStream_t stream) {
  const size_t nb_bytes = ((size_t)n + 63) & ~(size_t)63;
  uint32_t *info = reinterpret_cast<uint32_t *>(norm_buf + nb_bytes);
  float *cn = reinterpret_cast<float *>(info + 8);
  float *rho = cn + 16 * 256;
  RQ_HIP(hipMemsetAsync(info, 0xff, 4, stream));
  RQ_HIP(hipMemsetAsync(info + 1, 0, 4, stream));
  RQ_HIP(hipMemsetAsync(cn, 0, 16 * 256 * 4, stream));       // padding tables (m_real < m): |c|^2 = 0
  const uint32_t grid = (uint32_t)std::min<int64_t>(2048, (n + 255) / 256);
  hipLaunchKernelGGL(cnorm_kernel, dim3((m_real * 256 + 255) / 256), dim3(256), 0, stream, centers, m_real * 256, d, cn);
  hipLaunchKernelGGL(norm_residual_kernel, dim3(grid), dim3(256), 0, stream, row_bias, codes, (uint32_t)n, mp, m_real, cn, rho);
  hipLaunchKernelGGL(norm_minmax_kernel, dim3(grid), dim3(256), 0, stream, (const float *)rho, (uint32_t)n, info);
  hipLaunchKernelGGL(norm_info_kernel, dim3(1), dim3(1), 0, stream, info);
  hipLaunchKernelGGL(norm_quant_kernel, dim3(grid), dim3(256), 0, stream, (const float *)rho, (uint32_t)n, info, norm_buf);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int scan_launch(const ScanPlan &pl, float *dists, uint32_t *ids, uint64_t *keys, uint64_t *part, const uint8_t *codes,
                const float *centers, const float *queries, int64_t n, int64_t nq, int m, int d, int K,
                uint32_t id_offset, int id_base, uint32_t *work_counter, uint64_t *cand,
                hipStream_t stream, int lut_mode, const float *row_bias, uint8_t *norm_buf, const uint32_t *perm,
                bool norm_ready) {
  ScanParams p;
  p.perm = perm;
  p.samp_end = 0;           // set per row width below (launch_scan): positions below it hold a sample block in every 16 (order_sample_stride())
  p.codes = codes; p.centers = centers; p.queries = queries;
  p.n = (uint32_t)n; p.nq = (uint32_t)nq; p.sub = d / m; p.d = d; p.K = K;
  p.m_real = m;
  m = scan_padded_m(m);   // `codes` already has this row width (dev_linscan pads when needed)
  p.lut_mode = lut_mode; p.row_bias = row_bias;
  p.id_offset = id_offset; p.id_base = id_base;
  p.nslices = pl.nslices; p.rows_per_slice = pl.rows_per_slice; p.ngroups = pl.ngroups; p.whole = pl.whole;
  p.xcd_mode = pl.xcd ? 1u : 0u;
  {
    // pacing round = the items of whole windows, at least as many as the XCD has resident workgroups (fewer would leave
    // workgroups idle); slack = a quarter of a round.  Measured at 1e9 rows x 8 bytes, 1024 queries, k = 100 (128 items per
    // window, 64 workgroups per XCD; round 2's plan: 191.7 ms, 599 GB fetched per launch = 75x the code bytes) and on the
    // 1.25e8-row shard that one of 8 GPUs holds (round 2: 23.85 ms, 60x):
    //   round 128, slack 0 / 8 / 24: 195.8 / 194.7 / 195.1 ms, 6.5x / 7.1x / 7.0x;   slack 32: 185.9 ms, 19x   <- shipped
    //   round  64, slack 0 / 16 / 32: 199.8 / 195.8 / 189.0 ms, 6.0x / 7.0x / 10.7x
    //   no pacing: 185.8 ms, 65x (the pack dissolves after a few windows)
    //   shard: round 128 / slack 32: 23.5 ms, 8.9x;  round 64 / slack 32: 24.5 ms, 7.9x;  slack 0: 24.3 ms, 6.7x
    // Tighter pacing fetches less and costs more idle time than the L2 hits give back; the shipped point is the one that is
    // faster than round 2 at both sizes.
    const uint32_t per_xcd = std::max(1u, pl.grid / 8u);
    const int rd = tuning("SCAN_XCD_ROUND", 0);
    p.xcd_round = rd > 0 ? (uint32_t)rd : pl.ngroups * ((per_xcd + pl.ngroups - 1u) / pl.ngroups);
    const int sl = tuning("SCAN_XCD_SLACK", -1);
    p.xcd_slack = sl >= 0 ? (uint32_t)sl : p.xcd_round / 4u;
  }
  // chunk pacing inside the rounds (round 6, SCAN_PACE = 1: off by default -- EXPERIMENTS.md section 9 has the trade-off curve)
#if RQ_SCAN_PACE_BUILD      // (compiled out of the shipped library: the mere presence of the call in the streaming loop cost k = 1 1.8 %, the shard 2 %)
  p.pace = (pl.xcd && tuning("SCAN_PACE", 0)) ? work_counter + 64 : nullptr;
  p.pace_lag = (uint32_t)std::max(0, tuning("SCAN_PACE_LAG", 2));
  p.pace_votes = (uint32_t)std::max(1, tuning("SCAN_PACE_VOTES", 1));
#endif
  p.cap = pl.cap; p.trigger = pl.trigger; p.p2 = pl.p2; p.scratch_keys = pl.scratch_keys;
  p.sample = pl.sample;
  p.sample_rt = (uint32_t)tuning("SCAN_SAMPLE_RT", 4096);
  if (tuning("SCAN_SAMPLE", 0) > 0) p.sample_rt = pl.sample;      // an explicit SCAN_SAMPLE rules both
  p.srank_mul = (uint32_t)tuning("SCAN_SRANK_MUL", 2);
  p.retune_z = tuning("SCAN_RETUNE_Z", 6);
  p.retune_min_k = tuning("SCAN_RETUNE_MIN_K", 1);
  p.retune_div = std::max(2, tuning("SCAN_RETUNE_DIV", 8));
  p.work_counter = work_counter; p.cand = cand;
  p.gtab = reinterpret_cast<float4 *>(reinterpret_cast<char *>(cand) + pl.gtab_off);
  p.bkt = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(cand) + pl.bkt_off);
  p.bigk = pl.bigk ? (tuning("SCAN_SS_MAP", 1) ? 1 : 2) : 0;      // 2: sorted splitters only (tests, A/B)
  // the bucket finish needs BF_NB words of LDS per query of the group in the (dead) table space
  // (below k = 16 a query holds a few dozen candidates: select + sort of those costs 2.4 % of a k = 1 scan, the bucket finish's fixed
  // work -- counters, the look, two votes -- 4.0 %)
  p.bfin = (!pl.bigk && K >= 16 && (size_t)pl.scratch_keys * 8 >= (size_t)pl.qg * (BF_NB * 4 + 64)) ? tuning("SCAN_BUCKET_FINISH", 1) : 0;
  p.filter = (lut_mode != LUT_LSQ && !row_bias && tuning("SCAN_FILTER", 1)) ? 1 : 0;
  p.norm_bytes = nullptr; p.norm_info = nullptr; p.cnorm = nullptr;
  if (lut_mode == LUT_LSQ && row_bias && norm_buf && (m == 8 || m == 16) && tuning("SCAN_FILTER", 1) &&
      tuning("SCAN_FILTER_LSQ", 1)) {
    // the O(n) preprocessing of the base -- unless the caller holds a prepared base (rq_lsq_prepare: paid once, not per call)
    if (!norm_ready) RQ_TRY(lsq_norm_prepare(norm_buf, codes, centers, row_bias, n, m, p.m_real, d, stream));
    const size_t nb_bytes = ((size_t)n + 63) & ~(size_t)63;
    uint32_t *info = reinterpret_cast<uint32_t *>(norm_buf + nb_bytes);
    float *cn = reinterpret_cast<float *>(info + 8);
    p.norm_bytes = norm_buf;
    p.norm_info = reinterpret_cast<const float *>(info) + 4;
    p.cnorm = cn;
    p.filter = 1;
  }
  p.stats = tuning("SCAN_STATS", 0) ? reinterpret_cast<unsigned long long *>(work_counter + 16) : nullptr;
  p.dists = dists; p.ids = ids; p.keys = keys; p.part = part;
#if RQ_SCAN_PACE_BUILD
  if (p.pace) RQ_HIP(hipMemsetAsync(work_counter + 64, 0, WS_COUNTER_BYTES - 256, stream));
#endif
  RQ_HIP(hipMemsetAsync(work_counter, 0, p.stats ? 256 : 16 * sizeof(uint32_t), stream));   // [0..7] tickets, [8..15] finished items, per XCD
  switch (m) {
    case 2: return launch_scan<2>(p, pl, stream);
    case 4: return launch_scan<4>(p, pl, stream);
    case 8: return launch_scan<8>(p, pl, stream);
    case 16: return launch_scan<16>(p, pl, stream);
    case 32: return launch_scan<32>(p, pl, stream);
    case 64: return launch_scan<64>(p, pl, stream);
  }
  return fail(RQ_EUNSUPPORTED, "m=%d", m);
}

int merge_launch(float *dists, uint32_t *ids, uint64_t *keys_out, const uint64_t *keys_in, int64_t nq,
                 int P, int K, int id_base, hipStream_t stream) {
  MergeParams p;
  p.keys_in = keys_in; p.nq = (uint32_t)nq; p.P = (uint32_t)P; p.K = K; p.id_base = id_base;
  p.p2 = next_pow2((uint32_t)K);
  p.use_map = tuning("SCAN_SS_MAP", 1);
  p.dists = dists; p.ids = ids; p.keys_out = keys_out;
  if (P == 1) {
    const size_t total = (size_t)nq * K;
    hipLaunchKernelGGL(unpack_keys_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, p);
    RQ_HIP(hipGetLastError());
    return RQ_OK;
  }
  if (K > tuning("SCAN_SS_MIN_K", 1024)) {
    int dev = 0, num_cu = 256;
    RQ_HIP(hipGetDevice(&dev));
    RQ_HIP(hipDeviceGetAttribute(&num_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t grid = (uint32_t)std::min<int64_t>(nq, 2LL * num_cu);
    const size_t per_wg = (size_t)P * K;
    void *ws = nullptr;
    RQ_TRY(workspace(WS_MERGE, (size_t)grid * per_wg * (sizeof(uint64_t) + sizeof(uint16_t)) + 16, &ws, stream));
    uint64_t *scratch = (uint64_t *)ws;
    uint16_t *bkt = (uint16_t *)(scratch + (size_t)grid * per_wg);
    RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(merge_topk_big_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)SS_LDS_BYTES));
    hipLaunchKernelGGL(merge_topk_big_kernel, dim3(grid), dim3(SCAN_THREADS), SS_LDS_BYTES, stream, p, scratch, bkt);
    RQ_HIP(hipGetLastError());
    return RQ_OK;
  }
  constexpr int CTRL_BYTES = (sizeof(MergeCtrl) + 15) & ~15;
  size_t lds = CTRL_BYTES + (size_t)p.p2 * 8;
  RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(merge_topk_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(merge_topk_kernel, dim3((uint32_t)nq), dim3(MERGE_THREADS), lds, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// the row tiling rq_order.hip has to match: rows per lane and sub-step, rows of one wavefront tile (the shuffle granule),
// rows that meet in one LDS gather and the key bits per code byte that make such a gather conflict-free.
// Every LDS read width is served 32 lanes per pass on gfx950 (ds_read_b64: 256 B/clk; a dword gather moves half as much in
// the same passes -- measured again in round 4 with byte tables split into [k][quad][256] dwords read by ds_read2st64_b32
// and a 64-lane / 2-bit row order: 1.55 -> 2.65 ms), so the group is 32 rows and the window 32 values for all of them.
void scan_order_tiling(int mp, OrderTiling *t) {
  int r = 1, g = SCAN_THREADS, bk = SCAN_THREADS;
  switch (mp) {
#define RQ_OT(MM) case MM: r = ScanCfg<MM>::RPT; g = 64 * ScanCfg<MM>::RPT; bk = ScanCfg<MM>::BLK; break;
    RQ_OT(2) RQ_OT(4) RQ_OT(8) RQ_OT(16) RQ_OT(32) RQ_OT(64)
#undef RQ_OT
  }
  const int tg = tuning("ORDER_GRAN", 0);
  t->rpt = r;
  t->gran = tg > 0 ? tg : g;
  t->group = 32;
  t->cbits = 3;
  t->blk = bk;
}

int pad_codes_launch(uint8_t *dst, const uint8_t *src, int64_t n, int m, int mp, hipStream_t stream) {
  const size_t total = (size_t)n * mp;
  hipLaunchKernelGGL(pad_codes_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, stream, dst, src,
                     (size_t)n, m, mp);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int lut_launch(float *lut, const float *centers, const float *queries, int64_t nq, int m, int sub,
               hipStream_t stream) {
  const size_t total = (size_t)nq * m * 256;
  const int threads = 256;
  hipLaunchKernelGGL(adc_lut_kernel, dim3((uint32_t)((total + threads - 1) / threads)), dim3(threads), 0,
                     stream, lut, centers, queries, (uint32_t)nq, m, sub);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

int synth_codes_launch(uint8_t *codes, int64_t n, int m, uint64_t seed, int64_t row0, hipStream_t stream) {
  const size_t nbytes = (size_t)n * m;
  const size_t nthreads = (nbytes + 7) / 8;
  hipLaunchKernelGGL(synth_codes_kernel, dim3((uint32_t)((nthreads + 255) / 256)), dim3(256), 0, stream,
                     codes, nbytes, seed, (uint64_t)row0 * (uint64_t)m);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

}  // namespace rq
