// rq_topk.h -- exact top-k machinery shared by the ADC scan and the shard/GPU merge kernel.
//
// The reference keeps a full (dist,id) pair array per query and std::partial_sort's it
// (deps/src/linscan_aqd.cpp:78-92): the answer is the k smallest pairs in LEXICOGRAPHIC
// (dist, id) order.  Here a pair is packed into one uint64 key whose unsigned order is that
// lexicographic order:  key = ordered_bits(dist) << 32 | id.  Keys are unique (ids are), so
// "the k smallest keys" is a total-order statement and the result is independent of how the
// rows were partitioned over wavefronts, workgroups, slices or GPUs.
//
// Device routines (all threads of the workgroup must call them; they contain barriers):
//   radix_select  : k-th smallest key of an unsorted global buffer, 8 MSB-first 8-bit passes,
//                   for G independent query-lanes in lockstep (TPG threads each)
//   compact_leq   : copy the keys <= tau to the front of another buffer
//   radix_select_lds : the same on keys already in LDS (threshold sample, uint32 keys)
//   bitonic_sort_tiled : ascending sort of a power-of-two LDS array, wave-local stages without barriers
//   samplesort_topk : select + sort of the k smallest of cnt keys for large k, keys staying in global
//                   memory (splitters in LDS, bucket count / scatter, DPP row ranking per bucket)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rq {

constexpr uint64_t KEY_MAX = ~0ull;

// float -> uint32 whose unsigned order equals the float order (NaN-free inputs)
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  return __uint_as_float(u);
}
// d + 0.0f canonicalises -0 to +0 so key order == the reference's pair order (-0 == +0 there)
__device__ __forceinline__ uint64_t make_key(float d, uint32_t id) {
  return ((uint64_t)f2ord(d + 0.0f) << 32) | (uint64_t)id;
}
__device__ __forceinline__ float key_dist(uint64_t k) { return ord2f((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_id(uint64_t k) { return (uint32_t)k; }

template <int G>
struct SelState {
  uint64_t prefix[G];   // radix prefix found so far; after 8 passes the k-th smallest key
  uint32_t krem[G];     // rank still to resolve inside the current bucket (1-based)
  uint32_t newcnt[G];   // compaction cursor
  uint32_t bcnt[G];     // population of the bucket chosen in the last pass
  uint32_t vote[2];     // block_any() words (zeroed once per kernel)
  uint32_t hist[G][256];
};

// Workgroup-wide "does any thread have pred?" with ONE barrier and no static LDS.  HIP's __syncthreads_or/_and
// reserve 256 bytes of static LDS, which moves the dynamic LDS base off zero and keeps the compiler from folding
// LDS table offsets into the ds_read instructions of the scan's hot loop.  vote[2] lives in (dynamic) LDS and is
// zeroed at kernel start; `seq` is a per-thread, workgroup-uniform call counter (every thread makes the same
// calls).  Calls alternate between the two words, so a fast wavefront's next vote cannot overwrite the word a
// slow one is still reading; sequence numbers replace resets.  All threads of the workgroup must call it.
__device__ __forceinline__ bool block_any(bool pred, uint32_t *vote, uint32_t &seq) {
  ++seq;
  uint32_t *w = vote + (seq & 1u);
  if (__ballot(pred) != 0 && (threadIdx.x & 63) == 0) atomicMax(w, seq);
  __syncthreads();
  return *w == seq;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// hist[digit] += 1 for every lane with `on`.  Keys that share their high bytes (distances of one
// query, ids below 2^24) put a whole wavefront on ONE bin, and same-address LDS atomics serialise
// 64-way; that common case is detected with one ballot and served by a single add.
__device__ __forceinline__ void hist_add(uint32_t *hist, uint32_t digit, bool on) {
  const uint64_t todo = __ballot(on);
  if (!todo) return;
  const int leader = __ffsll((unsigned long long)todo) - 1;
  const uint32_t dl = __builtin_amdgcn_readlane(digit, leader);
  const uint64_t same = __ballot(on && digit == dl);
  if (same == todo) {            // the whole wavefront on one bin: one add of the population count
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[dl], (uint32_t)__popcll(same));
  } else if (on) {
    atomicAdd(&hist[digit], 1u);
  }
}

// k-th smallest (1-based k) of src[0..cnt) for query-lane g; result in st->prefix[g].
// `active`, `cnt`, `src`, `k` must be uniform inside a query-lane; cnt >= k when active.
// NPASS = 8 resolves the full 64-bit key; NPASS = 4 only its distance half (prefix then holds
// the k-th smallest distance in its top 32 bits, zeros below) -- enough for a threshold estimate.
template <int G, int TPG, int NPASS = 8>
__device__ __forceinline__ void radix_select(SelState<G> *st, const uint64_t *__restrict__ src,
                                             uint32_t cnt, uint32_t k, bool active, int g,
                                             int gi, uint32_t &vseq) {
  static_assert(TPG >= 64 && TPG % 64 == 0, "a query-lane is a whole number of wavefronts");
  if (gi == 0 && active) {
    st->prefix[g] = 0;
    st->krem[g] = k;
  }
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    const int shift = 56 - 8 * pass;
    for (int b = gi; b < 256; b += TPG) st->hist[g][b] = 0;
    __syncthreads();
    if (active) {
      const uint64_t pfx = st->prefix[g];
      // 8 independent loads in flight per thread: the keys live in L2, latency-bound otherwise
      constexpr int U = 8;
      for (uint32_t idx0 = gi; idx0 < cnt; idx0 += U * TPG) {
        uint64_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t idx = idx0 + u * TPG;
          key[u] = idx < cnt ? src[idx] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const bool match = (idx0 + u * TPG < cnt) &&
                             ((pass == 0) || ((key[u] >> (shift + 8)) == (pfx >> (shift + 8))));
          hist_add(&st->hist[g][0], (uint32_t)(key[u] >> shift) & 255u, match);
        }
      }
    }
    __syncthreads();
    if (active && gi < 64) {
      const uint32_t c0 = st->hist[g][gi * 4 + 0], c1 = st->hist[g][gi * 4 + 1];
      const uint32_t c2 = st->hist[g][gi * 4 + 2], c3 = st->hist[g][gi * 4 + 3];
      const uint32_t s = c0 + c1 + c2 + c3;
      const uint32_t incl = wave_incl_scan(s, gi);
      const uint32_t excl = incl - s;
      const uint32_t kk = st->krem[g];
      if (excl < kk && kk <= incl) {
        uint32_t r = kk - excl, b = 0;
        if (r > c0) { r -= c0; b = 1;
          if (r > c1) { r -= c1; b = 2;
            if (r > c2) { r -= c2; b = 3; } } }
        st->prefix[g] |= (uint64_t)(gi * 4 + b) << shift;
        st->krem[g] = r;
        st->bcnt[g] = b == 0 ? c0 : b == 1 ? c1 : b == 2 ? c2 : c3;
      }
    }
    if (NPASS == 8 && pass == 3) {
      // The distance half is resolved.  If every key of that distance is wanted (rank inside the bucket ==
      // its population -- always the case without distance ties) the answer is "all ids of this distance":
      // the id half needs no passes.  Uniform over the workgroup: all query-lanes must agree to stop.
      const bool done = !active || st->bcnt[g] == st->krem[g];
      if (!block_any(!done, st->vote, vseq)) {
        if (gi == 0 && active) st->prefix[g] |= 0xFFFFFFFFull;
        __syncthreads();
        return;
      }
    } else {
      __syncthreads();
    }
  }
}

// k-th smallest of keys[0..cnt) held in LDS (KeyT = uint32_t: NPASS 4, uint64_t: NPASS 8); same
// lock-step protocol as radix_select, but every pass is a handful of LDS reads per thread.
template <int G, int TPG, class KeyT>
__device__ __forceinline__ KeyT radix_select_lds(SelState<G> *st, const KeyT *keys, uint32_t cnt, uint32_t k,
                                                 bool active, int g, int gi) {
  constexpr int NPASS = (int)sizeof(KeyT);
  if (gi == 0 && active) {
    st->prefix[g] = 0;
    st->krem[g] = k;
  }
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    const int shift = 8 * (NPASS - 1 - pass);
    for (int b = gi; b < 256; b += TPG) st->hist[g][b] = 0;
    __syncthreads();
    if (active) {
      const KeyT pfx = (KeyT)st->prefix[g];
      for (uint32_t idx = gi; idx < cnt; idx += TPG) {
        const KeyT key = keys[idx];
        const bool match = (pass == 0) || ((uint64_t)(key >> shift) >> 8) == ((uint64_t)(pfx >> shift) >> 8);
        hist_add(&st->hist[g][0], (uint32_t)(key >> shift) & 255u, match);
      }
    }
    __syncthreads();
    if (active && gi < 64) {
      const uint32_t c0 = st->hist[g][gi * 4 + 0], c1 = st->hist[g][gi * 4 + 1];
      const uint32_t c2 = st->hist[g][gi * 4 + 2], c3 = st->hist[g][gi * 4 + 3];
      const uint32_t s = c0 + c1 + c2 + c3;
      const uint32_t incl = wave_incl_scan(s, gi);
      const uint32_t excl = incl - s;
      const uint32_t kk = st->krem[g];
      if (excl < kk && kk <= incl) {
        uint32_t r = kk - excl, b = 0;
        if (r > c0) { r -= c0; b = 1;
          if (r > c1) { r -= c1; b = 2;
            if (r > c2) { r -= c2; b = 3; } } }
        st->prefix[g] |= (uint64_t)(gi * 4 + b) << shift;
        st->krem[g] = r;
      }
    }
    __syncthreads();
  }
  return (KeyT)st->prefix[g];
}

// dst[0..) <- every key of src[0..cnt) that is <= tau (order not preserved); st->newcnt[g] = count.
template <int G, int TPG>
__device__ __forceinline__ void compact_leq(SelState<G> *st, const uint64_t *__restrict__ src,
                                            uint64_t *__restrict__ dst, uint32_t cnt, uint64_t tau,
                                            bool active, int g, int gi) {
  if (gi == 0) st->newcnt[g] = 0;
  __syncthreads();
  if (active) {
    const int lane = gi & 63;
    const uint32_t cnt_up = (cnt + 63u) & ~63u;  // keep whole waves in the ballot
    for (uint32_t idx = gi; idx < cnt_up; idx += TPG) {
      const uint64_t key = idx < cnt ? src[idx] : KEY_MAX;
      const bool take = idx < cnt && key <= tau;
      const uint64_t mask = __ballot(take);
      if (mask) {
        const int leader = __ffsll((unsigned long long)mask) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&st->newcnt[g], (uint32_t)__popcll(mask));
        base = __builtin_amdgcn_readlane(base, leader);
        if (take) dst[base + __popcll(mask & ((1ull << lane) - 1ull))] = key;
      }
    }
  }
  __syncthreads();
}

// one compare-exchange batch: comparators c0, c0+64, ... (up to 4 per lane), loads first
__device__ __forceinline__ void bitonic_batch(uint64_t *a, uint32_t c0, uint32_t cend, uint32_t j, uint32_t k) {
  uint32_t lo[4], hi[4];
  uint64_t x[4], y[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t c = c0 + 64u * u;
    lo[u] = ((c & ~(j - 1)) << 1) | (c & (j - 1));
    hi[u] = lo[u] | j;
    if (c < cend) { x[u] = a[lo[u]]; y[u] = a[hi[u]]; }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint32_t c = c0 + 64u * u;
    if (c < cend) {
      const bool up = (lo[u] & k) == 0;
      if ((x[u] > y[u]) == up) { a[lo[u]] = y[u]; a[hi[u]] = x[u]; }
    }
  }
}

// Two consecutive stages (j, j/2) of one merge phase k in ONE LDS round trip: the four elements
// b, b+j/2, b+j, b+j+j/2 (b with the bits j and j/2 clear) exchange among themselves only, and they lie in
// the same k-block, so one direction serves all four compare-exchanges.  Groups g0, g0+64 per lane.
__device__ __forceinline__ void bitonic_batch4(uint64_t *a, uint32_t g0, uint32_t gend, uint32_t j, uint32_t k) {
  const uint32_t jh = j >> 1;
  uint32_t b[2];
  uint64_t e[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const uint32_t g = g0 + 64u * u;
    b[u] = ((g & ~(jh - 1)) << 2) | (g & (jh - 1));
    if (g < gend) {
      e[u][0] = a[b[u]]; e[u][1] = a[b[u] + jh]; e[u][2] = a[b[u] + j]; e[u][3] = a[b[u] + j + jh];
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const uint32_t g = g0 + 64u * u;
    if (g < gend) {
      const bool up = (b[u] & k) == 0;
      auto cx = [up](uint64_t &x, uint64_t &y) {
        const bool sw = (x > y) == up;
        const uint64_t lo = sw ? y : x, hi = sw ? x : y;
        x = lo; y = hi;
      };
      cx(e[u][0], e[u][2]); cx(e[u][1], e[u][3]);   // stage j
      cx(e[u][0], e[u][1]); cx(e[u][2], e[u][3]);   // stage j/2
      a[b[u]] = e[u][0]; a[b[u] + jh] = e[u][1]; a[b[u] + j] = e[u][2]; a[b[u] + j + jh] = e[u][3];
    }
  }
}

// Ascending bitonic sort of a[0..p2) in LDS by a group of `nw` wavefronts (wave index w, lane).
// p2 is a power of two and uniform over the WORKGROUP (every thread of the block must call this:
// it contains __syncthreads()).  Wave w owns the segment [w*E, (w+1)*E), E = p2/nw_eff: every
// stage whose comparator span 2j fits in a segment is done by the owning wave alone with no block
// barrier (LDS operations of one wavefront complete in order), only the log2(nw_eff)*(...) wide
// stages synchronise the workgroup -- 1 of 55 stages at p2 = 1024 with 2 waves.
__device__ __forceinline__ void bitonic_sort_tiled(uint64_t *a, uint32_t p2, uint32_t nw, uint32_t w,
                                                   uint32_t lane, bool act) {
  uint32_t nw_eff = nw;
  while (nw_eff > 1 && p2 / nw_eff < 128) nw_eff >>= 1;   // keep >= 64 comparators per wave
  const uint32_t E = p2 / nw_eff;
  const bool mine = act && w < nw_eff;
  bool prev_wide = true;   // the caller's fill of a[] came from other waves
  // stages are taken two at a time (j, j/2) from the top of every merge phase: 36 LDS round trips instead
  // of 66 at p2 = 2048; a phase with an odd number of stages ends with the single stage j = 1
#pragma unroll 1
  for (uint32_t k = 2; k <= p2; k <<= 1) {
#pragma unroll 1
    for (uint32_t j = k >> 1; j > 0;) {
      const bool wide = 2 * j > E;
      if (wide || prev_wide) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (j >= 2) {
        if (mine) {
          // groups [w*E/4, (w+1)*E/4) touch only this wave's segment when !wide; for wide stages the same
          // split is simply a partition of all p2/4 groups
          const uint32_t gbeg = w * (E >> 2), gend = gbeg + (E >> 2);
          for (uint32_t g0 = gbeg + lane; g0 < gend; g0 += 128) bitonic_batch4(a, g0, gend, j, k);
        }
        j >>= 2;
      } else {
        if (mine) {
          const uint32_t cbeg = w * (E >> 1), cend = cbeg + (E >> 1);
          for (uint32_t c0 = cbeg + lane; c0 < cend; c0 += 256) bitonic_batch(a, c0, cend, j, k);
        }
        j = 0;
      }
      prev_wide = wide;
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Bucket finish (round 5; K <= 1024): "the n_out smallest of cnt keys, ascending" by ONE wavefront with no barrier.
//
// The cut to K (radix select: up to 8 passes over the candidates, each behind workgroup barriers) and the LDS bitonic sort
// (55 stages at 1024 keys) were 13 % of the headline scan, and all of it exposed (skipping both: 2.01 -> 1.76 ms).  A sorting
// network does O(log^2 n) work per key; the keys of one query, however, are distances from a narrow range [d_min, tau], so a
// MONOTONE map of the distance word onto BF_NB buckets (linear, then warped by a power: BfMap below) spreads them a few per
// bucket.  Per query:
//   0. min / max / mean of the distance words;  1. histogram (one LDS atomic per key);  2. exclusive scan: bucket starts, and
//   the bucket b* that holds rank n_out - 1 -- later buckets are dropped, which IS the cut;  3. kept keys scattered to their
//   bucket's range of `kbuf` (LDS; or `dst`, the query's other candidate buffer in L2, when they do not fit);  4. every kept
//   key counts the keys of its bucket below it (keys are unique: ranks are a permutation) and is emitted at start[b] + rank if
//   that is < n_out.  (The candidates held in registers across steps 0, 1, 3 -- 32 per lane, one read instead of three -- made
//   the compiler spill 392 registers in the CALLING kernel: 14.8 ms instead of 1.48.  Three streamed passes it is.)
// Exact for any input: the map is monotone in the key, so bucket order is key order, and inside a bucket the count decides.
// Ties in the distance (integer-valued tables, duplicated rows) land in one bucket and step 4 is O(bucket^2): a bucket of
// more than BF_MAX_BUCKET kept keys makes the routine give up BEFORE anything is written (false), and the caller runs the
// select + bitonic path.  `nxt`: BF_NB words of LDS private to the wavefront (LDS operations of one wavefront complete in
// order, so no barrier separates the steps); src and dst must not overlap.
// ------------------------------------------------------------------------------------------
constexpr uint32_t BF_NB = 512;
#ifndef RQ_BF_MAX_BUCKET
#define RQ_BF_MAX_BUCKET 48
#endif
#ifndef RQ_BF_MAX_S2
#define RQ_BF_MAX_S2 10
#endif
constexpr uint32_t BF_MAX_CNT = 4u * BF_NB;              // candidates of a query the bucket finish is tried on (4 per bucket on average)
constexpr uint32_t BF_MAX_BUCKET = RQ_BF_MAX_BUCKET;    // a kept bucket larger than this, or ...
constexpr uint32_t BF_MAX_S2 = RQ_BF_MAX_S2;            // ... sum of squared sizes of the kept buckets > BF_MAX_S2 * kept keys: give up

// Do the candidates of a query tie in distance a lot (rows that share their codes: clustered or duplicated data)?  The bucket
// paths rank the keys of a bucket against each other, which is quadratic in a tie group; inputs like that belong on the paths that
// order whole 64-bit keys (radix select + bitonic sort, sorted splitters).  A look at 64 keys: every lane stores {distance word,
// lane} into a table slot chosen by a hash of the distance word and reads the slot back -- a lane that finds its own distance
// under ANOTHER lane's number has a twin in the sample.  No atomics, no zeroing (a lane only reads the slot it has just written,
// so stale contents never count), hash collisions between different distances do not count either.  `tab`: 2^TB_LOG2 8-byte
// slots of LDS private to the wavefront.  On the bench data (no duplicated code rows) 0-2 of 64 lanes find a twin, on a base
// drawn from 1024 tight clusters 2-19 (median 7); the callers take >= BF_TIE_MIN as "tie-heavy".
#ifndef RQ_BF_TIE_MIN
#define RQ_BF_TIE_MIN 4
#endif
constexpr uint32_t BF_TIE_MIN = RQ_BF_TIE_MIN;
template <uint32_t TB_LOG2, uint32_t PER_LANE = 1>
__device__ __forceinline__ uint32_t bf_tie_twins(const uint64_t *__restrict__ src, uint32_t cnt, uint64_t *tab, uint32_t lane) {
  // (volatile: other lanes write the same slot -- the compiler must not forward the lane's own store to its load; LDS
  // operations of one wavefront complete in order, so every store precedes every load)
  volatile uint64_t *vt = tab;
  uint32_t hw[PER_LANE], slot[PER_LANE];
#pragma unroll
  for (uint32_t u = 0; u < PER_LANE; ++u) {
    const uint32_t i = lane + 64u * u;
    hw[u] = (uint32_t)(src[i < cnt ? i : 0u] >> 32);
    slot[u] = (hw[u] * 2654435761u) >> (32u - TB_LOG2);
  }
#pragma unroll
  for (uint32_t u = 0; u < PER_LANE; ++u)
    if (lane + 64u * u < cnt) vt[slot[u]] = ((uint64_t)hw[u] << 32) | (lane + 64u * u);
  uint32_t twins = 0u;
#pragma unroll
  for (uint32_t u = 0; u < PER_LANE; ++u) {
    const bool on = lane + 64u * u < cnt;
    const uint64_t got = vt[on ? slot[u] : 0u];
    twins += (uint32_t)__popcll(__ballot(on && (uint32_t)(got >> 32) == hw[u] && (uint32_t)got != lane + 64u * u));
  }
  return twins;
}

// Bucket of a key: x = (distance word - mn) / range in [0, 1], warped by x -> x^(2^psteps) (psteps squarings), times BF_NB.
// Every step is monotone in the key (unsigned subtract, int -> float, float multiplications of non-negative values, truncation),
// which is all exactness needs; the warp only evens the buckets out: the candidates of one query are the LOW tail of its distance
// distribution, whose density grows like a power of (d - d_min) -- on the bench data a linear map put 3 % of the keys into the
// last of 512 buckets.  With CDF(x) = x^p the mean of x is p / (p + 1); the caller picks the power of two next to that estimate.
struct BfMap {
  uint32_t mn;
  float inv;          // a little below 1 / range, so that x <= 1
  uint32_t psteps;    // wave-uniform
};
template <uint32_t NB>
__device__ __forceinline__ uint32_t bf_bucket_n(uint64_t key, const BfMap &m) {
  float y = (float)((uint32_t)(key >> 32) - m.mn) * m.inv;
#pragma unroll
  for (uint32_t i = 0; i < 5u; ++i) y = (i < m.psteps) ? y * y : y;
  return min((uint32_t)(y * ((float)NB - 0.5f)), NB - 1u);        // float -> uint truncates
}
__device__ __forceinline__ uint32_t bf_bucket(uint64_t key, const BfMap &m) { return bf_bucket_n<BF_NB>(key, m); }
// the warp's exponent from the mean of x (see BfMap): the power of two BELOW p = E / (1 - E) -- E >= 2/3, 4/5, 8/9, 16/17, 32/33.
// Rounding down, not to the nearest: too small an exponent leaves the top buckets a few times the average, too large a one
// folds the sparse low end of the range into bucket 0 -- and real tails are heavier there than the power law (sum of Gaussian
// tables: nearest put 54-72 keys into one bucket where the floor leaves 11; bench data: largest bucket 41 -> 23).
__device__ __forceinline__ uint32_t bf_psteps(float ex) {
  return ex >= 0.9697f ? 5u : ex >= 0.9412f ? 4u : ex >= 0.8889f ? 3u : ex >= 0.8f ? 2u : ex >= 0.6667f ? 1u : 0u;
}

// step 4 on the kept keys kb[0..kept) (kept >= 1), grouped by bucket (LDS or global); nxt[b] = end of bucket b, nxt[b - 1] its
// start, nxt[-1] = 0.  Every load is unconditional on a clamped index and all U of a step are issued together: written with
// `cond ? load : 0` the compiler branched around each load and waited for it alone -- 8 dependent LDS round trips per step.
template <uint32_t U, class KeyPtr, class Emit>
__device__ __forceinline__ void bf_rank_emit(KeyPtr kb, const uint32_t *nxt, uint32_t kept, uint32_t n_out, const BfMap &map,
                                             uint32_t lane, Emit emit) {
#pragma unroll 1
  for (uint32_t p0 = lane; p0 < kept; p0 += 64u * U) {
    uint64_t k[U];
    uint32_t lo[U], hi[U], r[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) k[u] = kb[min(p0 + 64u * u, kept - 1u)];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t b = bf_bucket(k[u], map);
      lo[u] = nxt[(int)b - 1];
      hi[u] = nxt[b];
    }
    uint32_t trips = 0u;
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      if (!(p0 + 64u * u < kept)) hi[u] = lo[u];          // nothing to count for a padding slot
      r[u] = lo[u];
      trips = max(trips, hi[u] - lo[u]);
    }
#pragma unroll 1
    for (uint32_t t = 0; t < trips; ++t) {
      uint64_t o[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) o[u] = kb[min(lo[u] + t, kept - 1u)];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) r[u] += (uint32_t)((lo[u] + t < hi[u]) & (o[u] < k[u]));
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u)
      if (p0 + 64u * u < kept && r[u] < n_out) emit(r[u], k[u]);
  }
}

// nxt: BF_NB words behind one more (nxt[-1]), kbuf: kcap keys -- LDS private to the wavefront (kept keys beyond kcap: through `dst` in global memory)
template <class Emit>
__device__ __forceinline__ bool bucket_finish_wave(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint32_t cnt,
                                                   uint32_t n_out, uint32_t *nxt, uint64_t *kbuf, uint32_t kcap, uint32_t lane,
                                                   Emit emit, unsigned long long *stats = nullptr) {
  // stats (optional, the workgroup's thread 0): cycles of [9] range + histogram, [10] scan + scatter, [11] rank + emit
#define BF_T() ((stats && threadIdx.x == 0) ? (unsigned long long)clock64() : 0ull)
#define BF_ADD(slot, t0) do { if (stats && threadIdx.x == 0) atomicAdd(&stats[slot], (unsigned long long)clock64() - (t0)); } while (0)
  unsigned long long t_s = BF_T();
  constexpr uint32_t U = 16;            // keys per lane in flight over the candidate list (L2 round trips are what these passes cost)
  n_out = min(n_out, cnt);
  if (n_out == 0u) return true;
#pragma unroll
  for (uint32_t b = lane; b < BF_NB; b += 64u) nxt[b] = 0u;
  if (lane == 0u) nxt[-1] = 0u;        // "the end of bucket -1": bucket 0 starts at 0 (bf_rank_emit reads it unconditionally)
  // ---- 0. range and mean of the distance words ----------------------------------------------------
  uint32_t mn = 0xffffffffu, mx = 0u;
  const uint32_t h0 = (uint32_t)(src[0] >> 32);
  float sum = 0.0f;                    // of (distance word - h0): only the warp's exponent is estimated from it
#pragma unroll 1
  for (uint32_t i0 = lane; i0 < cnt; i0 += 64u * U) {
    uint64_t k[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) k[u] = src[min(i0 + 64u * u, cnt - 1u)];      // (a repeated key changes neither min nor max)
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      const uint32_t hw = (uint32_t)(k[u] >> 32);
      mn = min(mn, hw);
      mx = max(mx, hw);
      sum += (i0 + 64u * u < cnt) ? (float)(int32_t)(hw - h0) : 0.0f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
    sum += __shfl_xor(sum, off);
  }
  const uint32_t range = mx - mn;
  BfMap map;
  map.mn = mn;
  // range == 0 (one distance): everything in bucket 0
  map.inv = range ? (1.0f / (float)range) * (1.0f - 9.5367431640625e-7f) : 0.0f;
  {
    // mean of x over the candidates (the padding lanes added (h0 - h0) = 0): E = p / (p + 1) under CDF(x) = x^p
    const float ex = range ? ((float)(int32_t)(h0 - mn) + sum / (float)cnt) / (float)range : 0.5f;
    map.psteps = bf_psteps(ex);
  }
  // ---- 1. histogram (the counters were zeroed above) ------------------------------------------------
#pragma unroll 1
  for (uint32_t i0 = lane; i0 < cnt; i0 += 64u * U) {
    uint64_t k[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) k[u] = src[min(i0 + 64u * u, cnt - 1u)];      // (unconditional: the U loads go out together)
#pragma unroll
    for (uint32_t u = 0; u < U; ++u)
      if (i0 + 64u * u < cnt) atomicAdd(&nxt[bf_bucket(k[u], map)], 1u);
  }
  BF_ADD(9, t_s);
  t_s = BF_T();
  // ---- 2. exclusive scan (lane l owns buckets [l * PER, (l + 1) * PER)), b*, the crowding test ----
  constexpr uint32_t PER = BF_NB / 64u;
  uint32_t c[PER], s = 0u;
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) { c[j] = nxt[lane * PER + j]; s += c[j]; }
  const uint32_t incl = wave_incl_scan(s, (int)lane);
  uint32_t run = incl - s, my_bstar = 0xffffffffu, my_kept = 0u, s2 = 0u;
  bool crowded = false;
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    const uint32_t end = run + c[j];
    if (run < n_out) {
      crowded |= c[j] > BF_MAX_BUCKET;
      s2 += c[j] * c[j];
      if (n_out <= end) { my_bstar = lane * PER + j; my_kept = end; }
    }
    nxt[lane * PER + j] = run;          // walks to the bucket's end during the scatter
    run = end;
  }
  if (__ballot(crowded)) return false;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s2 += (uint32_t)__shfl_xor((int)s2, off);
  if (s2 > BF_MAX_S2 * n_out) return false;      // ranking is a loop over a key's bucket: clumped distances make it the slow path
  const uint64_t owner = __ballot(my_bstar != 0xffffffffu);       // exactly one lane
  const int ol = __ffsll((unsigned long long)owner) - 1;
  const uint32_t bstar = (uint32_t)__builtin_amdgcn_readlane((int)my_bstar, ol);
  const uint32_t kept = (uint32_t)__builtin_amdgcn_readlane((int)my_kept, ol);      // keys of the buckets <= b*
  const bool in_lds = kept <= kcap;     // wave-uniform
  // ---- 3. scatter the kept keys (LDS when they fit the wavefront's share, else the query's other candidate buffer) ----
  auto scatter = [&](auto *out) {
#pragma unroll 1
    for (uint32_t i0 = lane; i0 < cnt; i0 += 64u * U) {
      uint64_t k[U];
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) k[u] = src[min(i0 + 64u * u, cnt - 1u)];      // (unconditional: the U loads go out together)
#pragma unroll
      for (uint32_t u = 0; u < U; ++u) {
        if (i0 + 64u * u < cnt) {
          const uint32_t b = bf_bucket(k[u], map);
          if (b <= bstar) out[atomicAdd(&nxt[b], 1u)] = k[u];
        }
      }
    }
  };
  // ---- 4. rank inside the bucket, emit ----------------------------------------------------------
  if (in_lds) {
    scatter(kbuf);
    BF_ADD(10, t_s);
    t_s = BF_T();
    bf_rank_emit<8>(kbuf, nxt, kept, n_out, map, lane, emit);
  } else {
    scatter(dst);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the wavefront's own stores, read back by its other lanes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    BF_ADD(10, t_s);
    t_s = BF_T();
    bf_rank_emit<4>(static_cast<const uint64_t *>(dst), nxt, kept, n_out, map, lane, emit);
  }
  BF_ADD(11, t_s);
#undef BF_T
#undef BF_ADD
  return true;
}

__host__ __device__ inline uint32_t next_pow2(uint32_t v) {
  uint32_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

// ------------------------------------------------------------------------------------------
// Exact "n_out smallest of cnt keys, ascending" for large n_out, by all NT threads of a block.
//
// The LDS bitonic sort needs next_pow2(K) keys of LDS and 105 read-modify-write stages at
// 16384 keys; for K in the thousands that dominated the scan.  This is a sample sort instead,
// with the keys staying in global memory (L2-resident: they were just written):
//   1. 2048 keys at stride cnt/2048 are sorted in LDS (16 KiB); 2047 of them split the key
//      space into 2048 buckets of ~cnt/2048 keys;
//   2. every key finds its bucket (11-step branch-free search in LDS) and counts itself;
//   3. an exclusive scan gives the bucket starts and the bucket b* that holds rank n_out-1 --
//      buckets behind b* are dropped, which replaces the radix-select cut;
//   4. the kept keys are scattered to their bucket's range of `dst`;
//   5. every kept key ranks itself inside its bucket by counting (keys are unique, so ranks
//      are a permutation) and is emitted at  start[b] + rank  if that is < n_out.
// The result is exact for any input; a skewed sample only makes step 5 slower (O(bucket^2)).
// src/dst: global, cnt keys each (dst is scratch); bkt: global scratch, cnt u16; lds:
// SS_LDS_BYTES.  All arguments are uniform over the block; contains __syncthreads().
// Keys equal to KEY_MAX are padding (short slices / shards): they are ignored, and the return
// value is the number of keys emitted, min(n_out, #real keys) -- the caller pads the rest.
// ------------------------------------------------------------------------------------------
#ifndef RQ_SS_UB
#define RQ_SS_UB 4
#endif
constexpr uint32_t SS_NS = 2048;
constexpr uint32_t SS_LDS_BYTES = SS_NS * 8 + SS_NS * 4 + 128 + SS_NS * 8;

template <int NT, class Emit>
__device__ __forceinline__ uint32_t samplesort_topk(const uint64_t *src, uint64_t *dst, uint16_t *bkt, uint32_t cnt,
                                                uint32_t n_out, unsigned char *lds, Emit emit,
                                                unsigned long long *stats = nullptr, bool use_map = true) {
  // stats (optional): cycles of [9] sample sort (map: range + mean), [10] bucket search (map: histogram), [11] scan + scatter (rank = rest)
#define SS_T() ((stats && threadIdx.x == 0) ? (unsigned long long)clock64() : 0ull)
#define SS_ADD(slot, t0) do { if (stats && threadIdx.x == 0) atomicAdd(&stats[slot], (unsigned long long)clock64() - (t0)); } while (0)
  constexpr uint32_t NS = SS_NS;
  constexpr int PER = NS / NT;
  static_assert(NS % NT == 0 && NT % 64 == 0 && NT / 64 <= 16, "thread split");
  uint64_t *smp = reinterpret_cast<uint64_t *>(lds);
  uint32_t *nxt = reinterpret_cast<uint32_t *>(lds + NS * 8);
  uint32_t *aux = nxt + NS;   // [0..15] wave sums, [16] b*, [17] end of b*
  uint64_t *tree = reinterpret_cast<uint64_t *>(lds + NS * 8 + NS * 4 + 128);   // splitters, BFS order
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();   // the previous user of `lds` is done
  // number of splitters for this input: buckets of ~6-12 keys (fewer splitters = a shorter sample sort and
  // search; the LDS areas are sized for the maximum, SS_NS)
  const uint32_t depth = cnt >= 12000u ? 11u : cnt >= 6000u ? 10u : cnt >= 3000u ? 9u : 8u;
  const uint32_t ns = 1u << depth;
  if (cnt <= NS) {
    // small input: the whole array is its own sample
    const uint32_t p2 = next_pow2(cnt);
    for (uint32_t i = tid; i < p2; i += NT) smp[i] = i < cnt ? src[i] : KEY_MAX;
    if (tid == 0) aux[18] = 0;
    bitonic_sort_tiled(smp, p2, NT / 64, wave, lane, true);
    for (uint32_t i = tid; i < cnt; i += NT)
      if (smp[i] != KEY_MAX && (i + 1 == p2 || smp[i + 1] == KEY_MAX)) aux[18] = i + 1;   // #real keys
    __syncthreads();
    n_out = min(n_out, aux[18]);
    for (uint32_t i = tid; i < n_out; i += NT) emit(i, smp[i]);
    return n_out;
  }
  unsigned long long t_s = SS_T();
  // ---- 1m / 2m. Round 5: buckets from a MAP of the distance word instead of sorted splitters ------------------------------
  // The candidates of one query are the low tail of its distance distribution; x = (distance word - min) / range warped by
  // x -> x^(2^psteps) (BfMap above: every step monotone in the key, the exponent from the mean of x) fills NS buckets as evenly
  // as random splitters do in the mean and more evenly in the spread (Poisson, not geometric sizes: sum of squares per key 8.9
  // against 15 at ~7 keys per bucket) -- without the sample sort and the 11-step search, 13 % of the kernel at k = 10000.
  // The map only sees the distance: rows that TIE in distance (duplicated codes, integer tables) share a bucket however many
  // they are, where the splitters -- whole 64-bit keys -- would part them by id.  So a bucket of more than SSM_MAX_BUCKET keys
  // among those that matter sends the query through the splitter path below; nothing has been written at that point.
  constexpr uint32_t SSM_MAX_BUCKET = 192;
#ifndef RQ_SSM_TIE_MIN
#define RQ_SSM_TIE_MIN 6
#endif
  constexpr uint32_t SSM_TIE_MIN = RQ_SSM_TIE_MIN;      // twins among 256 sampled keys that send the query to the splitters
  bool mapped = use_map;
  if (mapped) {       // tie-heavy input (a 256-key look, bf_tie_twins): straight to the splitters, which part ties by id
    if (wave == 0) {
      // (256 keys: at this K a tie group is a smaller share of the candidates; the splitters' space: 2048 slots)
      const uint32_t twins = bf_tie_twins<11, 4>(src, cnt, smp, lane);
      if (lane == 0) aux[21] = twins >= SSM_TIE_MIN ? 1u : 0u;
    }
    __syncthreads();
    mapped = aux[21] == 0u;
    __syncthreads();          // smp is written again below
  }
#pragma unroll 1
  for (;;) {
  if (mapped) {
    uint32_t *red = reinterpret_cast<uint32_t *>(smp);       // the splitters' space is free on this path: [16 w] per quantity
    float *fred = reinterpret_cast<float *>(smp);
    uint32_t mn = 0xffffffffu, mx = 0u, nreal = 0u;
    float sum = 0.0f;
    const uint32_t h0 = (uint32_t)(src[0] >> 32);
#pragma unroll 1
    for (uint32_t i0 = tid; i0 < cnt; i0 += NT * 8) {
      uint64_t k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) k[u] = src[min(i0 + (uint32_t)u * NT, cnt - 1u)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool real = i0 + (uint32_t)u * NT < cnt && k[u] != KEY_MAX;
        const uint32_t hw = (uint32_t)(k[u] >> 32);
        mn = real ? min(mn, hw) : mn;
        mx = real ? max(mx, hw) : mx;
        sum += real ? (float)(int32_t)(hw - h0) : 0.0f;
        nreal += (uint32_t)real;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
      sum += __shfl_xor(sum, off);
      nreal += (uint32_t)__shfl_xor((int)nreal, off);
    }
    if (lane == 0) { red[wave] = mn; red[16 + wave] = mx; fred[32 + wave] = sum; red[48 + wave] = nreal; }
    for (uint32_t i = tid; i < NS; i += NT) nxt[i] = 0;
    __syncthreads();
    for (uint32_t w = 0; w < (uint32_t)(NT / 64); ++w) {
      mn = min(mn, red[w]); mx = max(mx, red[16 + w]);
    }
    sum = 0.0f; nreal = 0u;
    for (uint32_t w = 0; w < (uint32_t)(NT / 64); ++w) { sum += fred[32 + w]; nreal += red[48 + w]; }
    BfMap map;
    const uint32_t range = mx - mn;
    map.mn = mn;
    map.inv = (nreal && range) ? (1.0f / (float)range) * (1.0f - 9.5367431640625e-7f) : 0.0f;
    map.psteps = (nreal && range) ? bf_psteps(((float)(int32_t)(h0 - mn) + sum / (float)nreal) / (float)range) : 0u;
    SS_ADD(9, t_s);
    t_s = SS_T();
#pragma unroll 1
    for (uint32_t i0 = tid; i0 < cnt; i0 += NT * 8) {
      uint64_t k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) k[u] = src[min(i0 + (uint32_t)u * NT, cnt - 1u)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t idx = i0 + (uint32_t)u * NT;
        if (idx < cnt) {
          const bool real = k[u] != KEY_MAX;
          const uint32_t b = bf_bucket_n<NS>(k[u], map);
          if (real) atomicAdd(&nxt[b], 1u);
          bkt[idx] = real ? (uint16_t)b : (uint16_t)0xFFFFu;
        }
      }
    }
    if (tid == 0) { aux[16] = 0; aux[17] = 0; aux[20] = 0; }       // aux[20]: a bucket that matters is crowded (set in the scan below)
  } else {
  for (uint32_t i = tid; i < NS; i += NT) {
    if (i < ns) smp[i] = src[(uint32_t)(((uint64_t)i * cnt) >> depth)];
    nxt[i] = 0;
  }
  if (tid == 0) { aux[16] = 0; aux[17] = 0; }
  bitonic_sort_tiled(smp, ns, NT / 64, wave, lane, true);
  // The ns-1 splitters smp[0..ns-2] go into breadth-first (Eytzinger) order: a binary search over
  // the SORTED array reads, at depth t, addresses that are all congruent modulo 2^(depth-t) keys --
  // up to 64 distinct addresses in one LDS bank; in BFS order a level is contiguous.
  for (uint32_t t = tid; t < ns; t += NT) {
    if (t) {
      const uint32_t lvl = 31u - (uint32_t)__builtin_clz(t), j = t - (1u << lvl);
      tree[t] = smp[(((2u * j + 1u) << (depth - 1u - lvl))) - 1u];
    }
  }
  __syncthreads();
  SS_ADD(9, t_s);
  t_s = SS_T();

  // ---- 2. bucket of every key (number of splitters that are <= key) ----------------------------
#pragma unroll 1
  for (uint32_t i0 = 0; i0 < cnt; i0 += NT * 4) {
    uint64_t k[4];
    uint32_t b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t idx = i0 + u * NT + tid;
      k[u] = idx < cnt ? src[idx] : KEY_MAX;
      b[u] = 1;
    }
#pragma unroll 1
    for (uint32_t lvl = 0; lvl < depth; ++lvl) {
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = 2u * b[u] + (uint32_t)(tree[b[u]] <= k[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] -= ns;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t idx = i0 + u * NT + tid;
      if (idx < cnt) {
        const bool real = k[u] != KEY_MAX;
        if (real) atomicAdd(&nxt[b[u]], 1u);
        bkt[idx] = real ? (uint16_t)b[u] : (uint16_t)0xFFFFu;
      }
    }
  }
  }  // splitter path
  __syncthreads();
  SS_ADD(10, t_s);
  t_s = SS_T();

  // ---- 3. exclusive scan of the bucket counts; b* = bucket holding rank n_out-1 -----------------
  {
    uint32_t c[PER], s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { c[j] = nxt[tid * PER + j]; s += c[j]; }
    const uint32_t incl = wave_incl_scan(s, (int)lane);
    if (lane == 63) aux[wave] = incl;
    __syncthreads();
    uint32_t run = incl - s, total = 0;
    for (uint32_t w = 0; w < (uint32_t)(NT / 64); ++w) {
      if (w < wave) run += aux[w];
      total += aux[w];
    }
    n_out = min(n_out, total);   // #real keys
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const uint32_t end = run + c[j];
      nxt[tid * PER + j] = run;
      if (run < n_out && n_out <= end) { aux[16] = tid * PER + j; aux[17] = end; }
      if (mapped && run < n_out && c[j] > SSM_MAX_BUCKET) aux[20] = 1u;
      run = end;
    }
    __syncthreads();
  }
  if (mapped && aux[20]) {      // (uniform) distance ties crowd a bucket: partition again, by whole keys
    mapped = false;
    __syncthreads();            // everybody has read aux[20] before the splitter path resets aux
    continue;
  }
  break;
  }  // map, then splitters if the map crowds
  const uint32_t bstar = aux[16];

  // ---- 4. scatter the kept keys; nxt[b] walks from the bucket's start to its end ----------------
  // (keys and bucket ids are loaded unconditionally, 8 per thread at a time: behind `if (idx < cnt)` every key cost two dependent
  // L2 round trips -- bucket id, then the key -- with four of them in flight)
#pragma unroll 1
  for (uint32_t i0 = tid; i0 < cnt; i0 += NT * 8) {
    uint64_t k[8];
    uint32_t b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t idx = min(i0 + (uint32_t)u * NT, cnt - 1u);
      k[u] = src[idx];
      b[u] = bkt[idx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + (uint32_t)u * NT < cnt && b[u] <= bstar) {
        const uint32_t pos = atomicAdd(&nxt[b[u]], 1u);
        dst[pos] = k[u];
      }
    }
  }
  __syncthreads();
  SS_ADD(11, t_s);
  // ---- 5. rank inside the bucket, emit ---------------------------------------------------------
  // One 16-lane DPP row per bucket (buckets average cnt/2048 ~ 8-13 keys): the row loads the
  // bucket 16 keys at a time, one key per lane, and every lane counts the keys below its own by
  // rotating the other chunk around the row -- one global load per 16 keys instead of one per
  // comparison.  nxt[b] is the bucket's end now, nxt[b-1] its start.
  {
    constexpr uint32_t NR = NT / 16, UB = RQ_SS_UB;   // rows per block; buckets a row has in flight
    const uint32_t row = tid >> 4, l16 = tid & 15;
    // r += #{rotations 1..15 of `other` around the 16-lane row that are < mine}: a 64-bit compare is
    // the borrow of (other - mine), so each rotation is sub / subb with the rotation folded into the
    // DPP operand, plus one add-with-carry.  (s_nop: DPP after a VALU write of EXEC needs 5 states.)
#define SS_ROTS(N)                                                                     \
    "v_sub_co_u32_dpp %1, vcc, %2, %4 row_ror:" #N " row_mask:0xf bank_mask:0xf\n"      \
    "v_subb_co_u32_dpp %1, vcc, %3, %5, vcc row_ror:" #N " row_mask:0xf bank_mask:0xf\n" \
    "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n"
#pragma unroll 1
    for (uint32_t b0 = row; b0 <= bstar; b0 += NR * UB) {
      uint32_t lo[UB], hi[UB];
      // the first TWO 16-key chunks of every bucket in flight are loaded up front: with random splitters the bucket sizes are
      // geometric (mean ~7.5), one bucket in ten has more than 16 keys, and a chunk fetched inside the loops below is a
      // dependent global load -- with 16 buckets per wavefront and trip almost every trip waited for two of them
      uint64_t first[UB], second[UB];
#pragma unroll
      for (uint32_t u = 0; u < UB; ++u) {
        const uint32_t b = b0 + u * NR;
        const bool on = b <= bstar;
        lo[u] = (on && b) ? nxt[b - 1] : 0u;
        hi[u] = on ? nxt[b] : 0u;
      }
#pragma unroll
      for (uint32_t u = 0; u < UB; ++u) {
        first[u] = (lo[u] + l16 < hi[u]) ? dst[lo[u] + l16] : KEY_MAX;
        second[u] = (lo[u] + 16u + l16 < hi[u]) ? dst[lo[u] + 16u + l16] : KEY_MAX;
      }
#pragma unroll
      for (uint32_t u = 0; u < UB; ++u) {
#pragma unroll 1
        for (uint32_t a = lo[u]; a < hi[u]; a += 16) {
          const bool have = a + l16 < hi[u];
          const uint64_t mine = (a == lo[u]) ? first[u] : (a == lo[u] + 16u) ? second[u] : (have ? dst[a + l16] : KEY_MAX);
          uint32_t r = lo[u];
#pragma unroll 1
          for (uint32_t c = lo[u]; c < hi[u]; c += 16) {
            const uint64_t other = (c == a) ? mine : (c == lo[u]) ? first[u] : (c == lo[u] + 16u) ? second[u]
                                   : ((c + l16 < hi[u]) ? dst[c + l16] : KEY_MAX);
            r += (uint32_t)(other < mine);
            const uint32_t olo = (uint32_t)other, ohi = (uint32_t)(other >> 32);
            const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
            uint32_t t;
#if !defined(RQ_SS_ABL) || RQ_SS_ABL != 1
            asm volatile("s_nop 4\n" SS_ROTS(1) SS_ROTS(2) SS_ROTS(3) SS_ROTS(4) SS_ROTS(5) SS_ROTS(6) SS_ROTS(7)
                         SS_ROTS(8) SS_ROTS(9) SS_ROTS(10) SS_ROTS(11) SS_ROTS(12) SS_ROTS(13) SS_ROTS(14) SS_ROTS(15)
                         : "+v"(r), "=&v"(t)
                         : "v"(olo), "v"(ohi), "v"(mlo), "v"(mhi)
                         : "vcc");
#endif
          }
#if defined(RQ_SS_ABL) && RQ_SS_ABL == 1
          r += l16;
#endif
#if !defined(RQ_SS_ABL) || RQ_SS_ABL != 2
          if (have && r < n_out) emit(r, mine);
#else
          if (have && r == 0xffffffffu) emit(r, mine);
#endif
        }
      }
    }
#undef SS_ROTS
  }
#undef SS_T
#undef SS_ADD
  return n_out;
}

}  // namespace rq
