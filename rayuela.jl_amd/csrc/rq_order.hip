// rq_order.hip -- bank-aware row order of a resident code matrix for the ADC scan (round 4).
//
// The scan's hot loop gathers one 8-byte table entry per code byte from a 256-entry LDS table: a ds_read_b64 of a
// 32-lane group costs as many passes as the fullest of the 32 eight-byte slot columns holds DISTINCT addresses
// (entry r sits in column r mod 32; equal addresses broadcast).  With the base rows in arrival order the 32 bytes of a
// group are random: 3.15 passes per gather, 63 % of the LDS cycles are conflict replays (profiles/r3_pmc_counters.md).
// But WHICH rows share a lane group is free -- the scan is a set operation and every key carries its row id.  If the 32
// rows of a group have byte k inside one aligned window of 32 consecutive values (the same top 3 bits), their gathers of
// table k hit 32 distinct columns or the same address: ONE pass.  A base of n rows has n/32 groups, i.e. log2(n/32) bits
// of freedom: a counting sort of the rows by the top 3 bits of their leading code bytes -- 15 bits = 5 of the 8 bytes at
// n = 1e6, 21 bits = 7 bytes on the 1.25e8-row shard of the SIFT1B configuration, all 8 bytes from 2^29 rows on -- makes
// those tables conflict-free (LDS-pass model, tools/rowperm_sim.py: 25.2 -> 15.5 passes per row at n = 1e6).
//
// Output: the permuted code matrix and perm[position] = original row.  Only rows that SURVIVE the threshold read perm
// (0.3-5 % of the rows), so keys carry original ids and the (dist, id) order of deps/src/linscan_aqd.cpp:91-97 is untouched:
// the answer is bit-identical for ANY permutation (tests/test_gpu_order.py).
//
// Two details the kernel's row tiling dictates (rq_scan.hip):
//  * lane j of half-wave h handles rows  tile*T + h*32*RPT + j*RPT + r  (T = 64*RPT, r < RPT), so the 32 consecutive rows
//    of the sort order that should meet in one gather are DEALT to those positions;
//  * the threshold estimates of a work item are statistics of ROWS, and the XCD windows / row slices treat any row range
//    alike.  A sorted base correlates position with content (measured: 487 of 1250 items fell back to the exact redo, 2.4 ->
//    9.5 ms; on clustered data even a shuffle of 1024-row granules was not enough).  So (a) one block in every
//    ORDER_SAMPLE_STRIDE (16) holds every 16th row of the base in ARRIVAL order -- the kernel visits a slice's sample blocks
//    first and takes both estimates from them --, and (b) the sorted rows are spread tile by tile (64 * RPT rows) by a Weyl
//    permutation g -> g * A mod G, so that any range of positions is a stratified sample of the sort order.
#include "rq_internal.h"

#include <cmath>

namespace rq {

struct OrderParams {
  const uint8_t *src;     // [n][mp]
  uint8_t *dst;           // [n][mp]
  uint32_t *perm;         // [n]   position -> original row
  uint32_t *rank;         // [n]   scratch: rank of the row inside its bucket
  uint32_t *hist;         // [nbins (+ tile totals behind, see order_bytes)]
  uint32_t n;
  int mp;                 // row width (2, 4, 8, 16, 32, 64)
  int ncoord;             // code bytes that take part in the key (<= 8)
  int nb[8];              // bits of byte c in the key (top bits of the byte)
  uint32_t nbins;
  uint32_t tile;          // T = 64 * RPT rows: one wavefront's rows of a sub-step
  uint32_t rpt;
  uint32_t group;         // rows that meet in one LDS gather: 32 (8-byte entries) or 64 (dword entries)
  uint32_t gran;          // shuffle granule (rows, multiple of tile)
  uint32_t ngran;         // full granules: n / gran
  uint32_t weyl;          // A: odd, coprime with ngran
  // SAMPLE BLOCKS (see order_sample_rows): the positions are cut into groups of `stride` blocks of `blk` rows; the first
  // block of each of the `sgroups` full groups holds every stride-th row of the base in ARRIVAL order, the sorted rows
  // fill the rest
  uint32_t stride, blk, sgroups;
  uint32_t rnd_rows;      // sample rows in all: sgroups * blk
  uint32_t nsorted;       // n - rnd_rows
  // GREEDY BALANCE (round 6, order_fine_greedy_kernel): the tables the sort key does not cover, LDS budget
  int gfree;              // number of free tables: the row's last 4 (8-byte rows) or 12 (16-byte rows) bytes; 0: plain order
  uint32_t glist;         // rows of a coarse bucket the LDS list holds (a bigger bucket takes the plain path)
  uint32_t gwaves;        // wavefronts of the workgroup that balance (LDS: 16 groups x gfree tables x 64 bytes each)
};

// Is row i a sample row?  It is sample row number i / stride.
__device__ __forceinline__ bool order_in_prefix(const OrderParams &p, uint32_t i) {
  return p.rnd_rows && i % p.stride == 0u && i / p.stride < p.rnd_rows;
}
// position of sample row r / of sorted index t
__device__ __forceinline__ uint32_t order_sample_pos(const OrderParams &p, uint32_t r) {
  return (r / p.blk) * p.stride * p.blk + r % p.blk;
}
__device__ __forceinline__ uint32_t order_sorted_pos(const OrderParams &p, uint32_t t) {
  if (!p.rnd_rows) return t;
  const uint32_t per = (p.stride - 1u) * p.blk;          // sorted rows per full group
  const uint32_t g = t / per;
  if (g >= p.sgroups) return p.sgroups * p.stride * p.blk + (t - p.sgroups * per);
  return g * p.stride * p.blk + p.blk + (t - g * per);
}

__device__ __forceinline__ void order_copy_row(const OrderParams &p, uint32_t row, uint32_t pos) {
  const uint8_t *a = p.src + (size_t)row * p.mp;
  uint8_t *b = p.dst + (size_t)pos * p.mp;
  if (p.mp == 8) *reinterpret_cast<uint64_t *>(b) = *reinterpret_cast<const uint64_t *>(a);
  else if (p.mp >= 16) {
    for (int o = 0; o < p.mp; o += 16) *reinterpret_cast<uint4 *>(b + o) = *reinterpret_cast<const uint4 *>(a + o);
  } else if (p.mp == 4) *reinterpret_cast<uint32_t *>(b) = *reinterpret_cast<const uint32_t *>(a);
  else *reinterpret_cast<uint16_t *>(b) = *reinterpret_cast<const uint16_t *>(a);
  p.perm[pos] = row;
}

constexpr int ORDER_SCAN_TILE = 16384;    // bins per scan workgroup (1024 threads x 16)

__device__ __forceinline__ uint32_t order_key_of(const OrderParams &p, uint32_t row) {
  // the first min(mp, 8) bytes of the row (rows are mp-byte aligned)
  uint64_t w;
  const uint8_t *r = p.src + (size_t)row * p.mp;
  if (p.mp >= 8) w = *reinterpret_cast<const uint64_t *>(r);
  else if (p.mp == 4) w = *reinterpret_cast<const uint32_t *>(r);
  else w = *reinterpret_cast<const uint16_t *>(r);
  uint32_t key = 0;
  for (int c = 0; c < p.ncoord; ++c) {
    const uint32_t b = (uint32_t)(w >> (8 * c)) & 0xffu;
    key = (key << p.nb[c]) | (b >> (8 - p.nb[c]));
  }
  return key;
}

// pass 1: bucket sizes, and every row's arrival rank inside its bucket (any order will do: the result of the scan does not
// depend on the permutation)
__global__ __launch_bounds__(256) void order_rank_kernel(OrderParams p) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x)
    if (!order_in_prefix(p, i)) p.rank[i] = atomicAdd(&p.hist[order_key_of(p, i)], 1u);
}

// pass 2a: exclusive scan inside tiles of ORDER_SCAN_TILE bins, tile totals to hist[nbins + tile]
__global__ __launch_bounds__(1024) void order_scan_tiles_kernel(uint32_t *hist, uint32_t nbins) {
  __shared__ uint32_t wsum[16];
  const uint32_t t0 = blockIdx.x * ORDER_SCAN_TILE + threadIdx.x * 16;
  uint32_t v[16], s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = t0 + i < nbins ? hist[t0 + i] : 0u; s += v[i]; }
  // inclusive scan of s over the 1024 threads: wave scan + scan of the 16 wave sums
  uint32_t x = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off);
    if ((threadIdx.x & 63) >= (uint32_t)off) x += y;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += wsum[w];
  uint32_t run = wbase + x - s;    // exclusive prefix of this thread's 16 bins inside the tile
#pragma unroll
  for (int i = 0; i < 16; ++i) { if (t0 + i < nbins) hist[t0 + i] = run; run += v[i]; }
  if (threadIdx.x == 1023) hist[nbins + blockIdx.x] = run;
}

// pass 2b: exclusive scan of the (<= 1024) tile totals, one workgroup
__global__ __launch_bounds__(1024) void order_scan_top_kernel(uint32_t *tot, uint32_t ntiles) {
  __shared__ uint32_t wsum[16];
  const uint32_t s = threadIdx.x < ntiles ? tot[threadIdx.x] : 0u;
  uint32_t x = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off);
    if ((threadIdx.x & 63) >= (uint32_t)off) x += y;
  }
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += wsum[w];
  if (threadIdx.x < ntiles) tot[threadIdx.x] = wbase + x - s;
}

// sorted rank -> position (see the header comment): `group` consecutive ranks become one lane group of the kernel --
// lane l of a wavefront handles rows tile * T + l * RPT + r (r < RPT), the group (b, r) is lanes [b * group, (b + 1) * group)
__device__ __forceinline__ uint32_t order_deal(const OrderParams &p, uint32_t s) {
  // s: rank among the sorted rows.  Shuffle and dealing act on that index space (granules and tiles of it stay aligned in
  // position space: block sizes are multiples of both); order_sorted_pos then steps over the sample blocks.
  uint32_t g = s / p.gran;
  if (g < p.ngran) {
    const uint32_t g2 = (uint32_t)(((uint64_t)g * p.weyl) % p.ngran);
    s = g2 * p.gran + (s - g * p.gran);
  }
  const uint32_t t = s / p.tile;
  if ((uint64_t)(t + 1) * p.tile <= p.nsorted) {            // (the ragged last tile stays in sort order)
    const uint32_t u = s - t * p.tile;
    const uint32_t gi = u / p.group, j = u - gi * p.group;   // group inside the tile, row inside the group
    const uint32_t b = gi / p.rpt, r = gi - b * p.rpt;
    s = t * p.tile + (b * p.group + j) * p.rpt + r;
  }
  return order_sorted_pos(p, s);
}

// pass 3: every row to its position, perm[position] = row
__global__ __launch_bounds__(256) void order_scatter_kernel(OrderParams p) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += gridDim.x * blockDim.x) {
    if (order_in_prefix(p, i)) { order_copy_row(p, i, order_sample_pos(p, i / p.stride)); continue; }
    const uint32_t key = order_key_of(p, i);
    const uint32_t s = p.hist[key] + p.hist[p.nbins + key / ORDER_SCAN_TILE] + p.rank[i];
    order_copy_row(p, i, order_deal(p, s));
  }
}

// ---- small bases (the ordering INSIDE a call): two levels, LDS histograms, 2 x 256 global atomics per workgroup ----------
// The one-level path above pays one returning device-scope atomic per row (1e6 rows: 48 us of its 100) and four launches.
// For keys of <= 18 bits the same order comes from a coarse level of 256 buckets (the key's top 8 bits) and a fine level
// inside each bucket:
//   order_coarse_count   per workgroup chunk: LDS histogram of the coarse byte -> ctot[256] (non-returning atomics)
//   order_coarse_scatter the chunk's rows reserve ranges of their coarse buckets (one returning atomic per workgroup and
//                        non-empty bucket) and write their ROW NUMBERS there
//   order_fine           one workgroup per coarse bucket: LDS counting sort of its rows by the remaining <= 10 key bits,
//                        rows dealt to their final positions (order_deal), perm written
constexpr int ORDER_COARSE = 256, ORDER_FINE_MAX = 1024, ORDER_SMALL_THREADS = 1024;

struct OrderSmall {
  uint32_t *ctot;        // [256] rows per coarse bucket, then [256] cursors
  uint32_t *idx;         // [n] row numbers grouped by coarse bucket
  uint32_t rows_per_wg;
  int fine_bits;
};

__device__ __forceinline__ void coarse_starts(uint32_t *starts /*LDS [257]*/, const uint32_t *ctot) {
  // exclusive scan of the 256 bucket sizes by the first 256 threads (4 wavefronts + their sums)
  __shared__ uint32_t wsum[4];
  const uint32_t tid = threadIdx.x;
  uint32_t v = 0, x = 0;
  if (tid < ORDER_COARSE) {
    v = ctot[tid];
    x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if ((tid & 63u) >= (uint32_t)off) x += y;
    }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = x;
  }
  __syncthreads();
  if (tid < ORDER_COARSE) {
    uint32_t base = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) base += wsum[w];
    starts[tid] = base + x - v;
    if (tid == ORDER_COARSE - 1) starts[ORDER_COARSE] = base + x;
  }
  __syncthreads();
}

__global__ __launch_bounds__(ORDER_SMALL_THREADS) void order_coarse_count_kernel(OrderParams p, OrderSmall q) {
  __shared__ uint32_t h[ORDER_COARSE];
  const uint32_t tid = threadIdx.x;
  if (tid < ORDER_COARSE) h[tid] = 0;
  __syncthreads();
  const uint32_t r0 = blockIdx.x * q.rows_per_wg, r1 = min(p.n, r0 + q.rows_per_wg);
  for (uint32_t i = r0 + tid; i < r1; i += ORDER_SMALL_THREADS)
    if (!order_in_prefix(p, i)) atomicAdd(&h[order_key_of(p, i) >> q.fine_bits], 1u);
  __syncthreads();
  if (tid < ORDER_COARSE && h[tid]) atomicAdd(&q.ctot[tid], h[tid]);
}

__global__ __launch_bounds__(ORDER_SMALL_THREADS) void order_coarse_scatter_kernel(OrderParams p, OrderSmall q) {
  __shared__ uint32_t h[ORDER_COARSE], base[ORDER_COARSE], starts[ORDER_COARSE + 1];
  const uint32_t tid = threadIdx.x;
  if (tid < ORDER_COARSE) h[tid] = 0;
  coarse_starts(starts, q.ctot);
  const uint32_t r0 = blockIdx.x * q.rows_per_wg, r1 = min(p.n, r0 + q.rows_per_wg);
  // local rank inside (chunk, bucket), then one global reservation per non-empty bucket
  constexpr int PER = 4;                       // rows_per_wg == PER * threads
  uint32_t c[PER], lr[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t i = r0 + u * ORDER_SMALL_THREADS + tid;
    c[u] = 0xffffffffu;
    if (i < r1) {
      if (order_in_prefix(p, i)) order_copy_row(p, i, order_sample_pos(p, i / p.stride));   // a sample row: arrival order, no sort
      else { c[u] = order_key_of(p, i) >> q.fine_bits; lr[u] = atomicAdd(&h[c[u]], 1u); }
    }
  }
  __syncthreads();
  if (tid < ORDER_COARSE && h[tid]) base[tid] = starts[tid] + atomicAdd(&q.ctot[ORDER_COARSE + tid], h[tid]);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const uint32_t i = r0 + u * ORDER_SMALL_THREADS + tid;
    if (c[u] != 0xffffffffu) q.idx[base[c[u]] + lr[u]] = i;
  }
}

__global__ __launch_bounds__(ORDER_SMALL_THREADS) void order_fine_kernel(OrderParams p, OrderSmall q) {
  __shared__ uint32_t f[ORDER_FINE_MAX], starts[ORDER_COARSE + 1], wsum[16];
  const uint32_t tid = threadIdx.x, nfine = 1u << q.fine_bits, fmask = nfine - 1u;
  for (uint32_t i = tid; i < nfine; i += ORDER_SMALL_THREADS) f[i] = 0;
  coarse_starts(starts, q.ctot);
  const uint32_t b0 = starts[blockIdx.x], b1 = starts[blockIdx.x + 1];
  for (uint32_t i = b0 + tid; i < b1; i += ORDER_SMALL_THREADS) atomicAdd(&f[order_key_of(p, q.idx[i]) & fmask], 1u);
  __syncthreads();
  {   // exclusive scan of the <= 1024 fine counts: one per thread
    const uint32_t v = tid < nfine ? f[tid] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if ((tid & 63u) >= (uint32_t)off) x += y;
    }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = x;
    __syncthreads();
    uint32_t wb = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) wb += wsum[w];
    if (tid < nfine) f[tid] = b0 + wb + x - v;        // the fine bucket's first sorted rank; walks to its end below
    __syncthreads();
  }
  for (uint32_t i = b0 + tid; i < b1; i += ORDER_SMALL_THREADS) {
    const uint32_t row = q.idx[i];
    const uint32_t s = atomicAdd(&f[order_key_of(p, row) & fmask], 1u);
    order_copy_row(p, row, order_deal(p, s));
  }
}


// ---- round 6: balance the rows of a sort bucket over the bucket's lane groups -------------------------------------------------
// The sort key makes the gathers of the tables it covers conflict-free and leaves the others at 3.15 passes (32 random bytes in
// 32 columns).  But a bucket of the 12-bit key holds ~8 lane groups' worth of rows (244 at 1e6 rows), and WHICH of them share a
// group is still free.  One wavefront per fine bucket deals the bucket's rows to its groups one by one (arrival order): the row
// goes to the group -- among those with room -- where it adds the least to sum over the free tables of (column load)^2; a row whose
// byte equals the last one stored in that column is free (same address: broadcast).  Measured on the bench codes (host probe,
// tools/greedy_order_probe.py): passes per row and lane group 15.5 -> 12.7 at m = 8 (the four free tables 3.15 -> 2.13 each),
// 40.7 -> 33.2 at m = 16; scan kernel k = 1 / 1000: m = 8 -8.0 / -4.0 %, m = 16 -9.1 / -6.6 %.  Any result is a permutation, so the
// scan's answer cannot depend on it (tests/test_gpu_order.py).
// Lanes: 16 groups x 4 lanes, lane (g, q) prices the free tables q, q + 4, q + 8 (the minimum over the groups is a DPP reduction).
// Buckets longer than 16 groups are cut into windows of 512 sorted
// ranks (16 groups); a group shared with the neighbouring bucket takes part with the slots this bucket owns in it.
constexpr uint32_t GREEDY_GROUPS = 16;

__device__ __forceinline__ uint32_t dpp_quad_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
  return v;
}
// minimum over the wavefront (unsigned), in every lane's return value: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then
// row_bcast 15 / 31 across them (GFX9 DPP); the lanes a shift does not reach keep the identity
__device__ __forceinline__ uint32_t dpp_wave_min(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// One chunk [c0, c1) of sorted ranks (inside one fine bucket, inside one 512-rank window): deal its rows to its lane groups.
// list: the coarse bucket's row numbers in fine-sorted order, list[s - b0] = row of sorted rank s; st: this wave's LDS state,
// [16 groups][T][32 columns] of {rows in the column, last byte stored there} as one 16-bit word.  MP = row bytes, TPL = tables
// per lane: the free tables are the last T = 4 * TPL bytes of the row, lane (g, q) prices bytes MP - T + q + 4 i (i < TPL), which
// sit in word (MP - T) / 4 + i of the row at byte q.
template <int MP, int TPL>
__device__ __forceinline__ void greedy_chunk(const OrderParams &p, const uint32_t *list, uint32_t b0, uint32_t c0, uint32_t c1,
                                             uint16_t *st) {
  constexpr uint32_t T = 4u * TPL, W0 = (MP - (int)T) / 4, NW = MP / 4;
  const uint32_t lane = threadIdx.x & 63u, gl = lane >> 2, q4 = lane & 3u;
  const uint32_t grp = p.group;
  const uint32_t g_first = c0 / grp, ng = (c1 - 1u) / grp - g_first + 1u, R = c1 - c0;
  if (ng < 2u || R < 4u) {          // nothing to choose
    for (uint32_t i = lane; i < R; i += 64u) order_copy_row(p, list[c0 - b0 + i], order_deal(p, c0 + i));
    return;
  }
  for (uint32_t i = lane; i < GREEDY_GROUPS * T * 32u / 2u; i += 64u) reinterpret_cast<uint32_t *>(st)[i] = 0u;
  const uint32_t glo = max(c0, (g_first + gl) * grp), ghi = min(c1, (g_first + gl + 1u) * grp);
  const uint32_t cap = (gl < ng && ghi > glo) ? ghi - glo : 0u;
  uint32_t next = glo;                 // the next free rank of this lane's group
  const uint32_t sh = 8u * q4;
  uint16_t *mine_st = st + (gl * T + q4) * 32u;          // table i of this lane: + i * 4 * 32
  __builtin_amdgcn_wave_barrier();
  // rows travel in batches of 64, one per lane, and are broadcast from there; the next batch is in flight during the deal
  uint32_t row = list[c0 - b0 + min(lane, R - 1u)];
  uint32_t w[NW];
  auto load_words = [&](uint32_t r, uint32_t (&o)[NW]) {
    const uint8_t *r8 = p.src + (size_t)r * MP;
    if constexpr (MP == 16) { const uint4 v = *reinterpret_cast<const uint4 *>(r8); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
    else { const uint2 v = *reinterpret_cast<const uint2 *>(r8); o[0] = v.x; o[1] = v.y; }
  };
  load_words(row, w);
  for (uint32_t b = 0; b < R; b += 64u) {
    const bool have = b + lane < R;
    const uint32_t nb = min(64u, R - b);
    uint32_t row_n = row, w_n[NW];
    if (b + 64u < R) row_n = list[c0 - b0 + min(b + 64u + lane, R - 1u)];
    load_words(row_n, w_n);
    uint32_t myrank = c0 + b + lane;
    for (uint32_t j = 0; j < nb; ++j) {
      uint32_t cost = 0, val[TPL], x[TPL];
      bool fresh[TPL];
#pragma unroll
      for (int i = 0; i < TPL; ++i) {
        const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)w[W0 + i], (int)j);      // word W0 + i of row j (uniform)
        val[i] = (cw >> sh) & 255u;
        x[i] = mine_st[i * 128 + (val[i] & 31u)];
      }
#pragma unroll
      for (int i = 0; i < TPL; ++i) {
        const uint32_t c = x[i] & 255u;
        fresh[i] = !(c != 0u && (x[i] >> 8) == val[i]);
        cost += fresh[i] ? 2u * c + 1u : 0u;
      }
      cost = dpp_quad_sum(cost);
      cost = (next < ghi && cap != 0u) ? min(cost, 1022u) : 1023u;
      const uint32_t best = dpp_wave_min((cost << 6) | lane);          // the cheapest group, lowest lane first
      const uint32_t wl = best & 63u;
      const uint32_t newrank = (uint32_t)__builtin_amdgcn_readlane((int)next, (int)wl);
      if (gl == (wl >> 2)) {
#pragma unroll
        for (int i = 0; i < TPL; ++i)
          mine_st[i * 128 + (val[i] & 31u)] = (uint16_t)(((x[i] & 255u) + (fresh[i] ? 1u : 0u)) | (val[i] << 8));
        next += 1u;
      }
      myrank = lane == j ? newrank : myrank;
      __builtin_amdgcn_wave_barrier();
    }
    if (have) order_copy_row(p, row, order_deal(p, myrank));
    row = row_n;
#pragma unroll
    for (uint32_t i = 0; i < NW; ++i) w[i] = w_n[i];
  }
}

__global__ __launch_bounds__(ORDER_SMALL_THREADS) void order_fine_greedy_kernel(OrderParams p, OrderSmall q) {
  __shared__ uint32_t f[ORDER_FINE_MAX], fst[ORDER_FINE_MAX + 1], starts[ORDER_COARSE + 1], wsum[16];
  extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];         // [glist] row list | [gwaves] balance state
  const uint32_t tid = threadIdx.x, nfine = 1u << q.fine_bits, fmask = nfine - 1u;
  for (uint32_t i = tid; i < nfine; i += ORDER_SMALL_THREADS) f[i] = 0;
  coarse_starts(starts, q.ctot);
  const uint32_t b0 = starts[blockIdx.x], b1 = starts[blockIdx.x + 1];
  const bool fits = b1 - b0 <= p.glist;
  for (uint32_t i = b0 + tid; i < b1; i += ORDER_SMALL_THREADS) atomicAdd(&f[order_key_of(p, q.idx[i]) & fmask], 1u);
  __syncthreads();
  {   // exclusive scan of the <= 1024 fine counts: one per thread
    const uint32_t v = tid < nfine ? f[tid] : 0u;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off);
      if ((tid & 63u) >= (uint32_t)off) x += y;
    }
    if ((tid & 63u) == 63u) wsum[tid >> 6] = x;
    __syncthreads();
    uint32_t wb = 0;
    for (uint32_t w = 0; w < (tid >> 6); ++w) wb += wsum[w];
    if (tid < nfine) { f[tid] = b0 + wb + x - v; fst[tid] = b0 + wb + x - v; }
    if (tid == nfine - 1u) fst[nfine] = b0 + wb + x;
    __syncthreads();
  }
  if (!fits) {        // a coarse bucket too big for the LDS list (clumped data): the plain order of the (shorter) key
    for (uint32_t i = b0 + tid; i < b1; i += ORDER_SMALL_THREADS) {
      const uint32_t row = q.idx[i];
      const uint32_t s = atomicAdd(&f[order_key_of(p, row) & fmask], 1u);
      order_copy_row(p, row, order_deal(p, s));
    }
    return;
  }
  uint32_t *list = dyn;
  for (uint32_t i = b0 + tid; i < b1; i += ORDER_SMALL_THREADS) {
    const uint32_t row = q.idx[i];
    list[atomicAdd(&f[order_key_of(p, row) & fmask], 1u) - b0] = row;
  }
  __syncthreads();
  const uint32_t wave = tid >> 6;
  if (wave >= p.gwaves) return;
  uint16_t *st = reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(dyn + p.glist) + (size_t)wave * GREEDY_GROUPS * (uint32_t)p.gfree * 64u);
  const uint32_t window = GREEDY_GROUPS * p.group;
  for (uint32_t fb = wave; fb < nfine; fb += p.gwaves) {
    const uint32_t s0 = fst[fb], s1 = fst[fb + 1];
    for (uint32_t c0 = s0; c0 < s1;) {
      // windows of 16 lane groups counted from the bucket's own first group: a bucket of <= ~480 rows is ONE window (cut at
      // global multiples of 512 ranks half of the 244-row buckets fell into two independent halves: 13.0 instead of 12.6 passes)
      const uint32_t c1 = min(s1, c0 / p.group * p.group + window);
      if (p.mp == 8) greedy_chunk<8, 1>(p, list, b0, c0, c1, st);
      else greedy_chunk<16, 3>(p, list, b0, c0, c1, st);
      c0 = c1;
    }
  }
}

static uint32_t gcd_u32(uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; }

// Key layout for n rows of mp bytes: `cbits` bits per leading code byte (the window 256 >> cbits = the bank columns one
// gather of `group` lanes can hit without a conflict) while log2(n / group) bits last; what is left over goes to the next
// byte (a wider window still lowers its passes).  0 bits: no ordering (tiny bases).
// Sample blocks of an ordered base.  The scan's second threshold estimate treats the first 1/8 of a slice as a random sample
// of it.  Sorted rows are anything but: a query's near neighbours share their leading code bytes and sit in a few buckets (on
// 1e6 rows from 1024 tight clusters the whole top-1000 is one or two 1024-row granules: 640 of 1250 items fell back to the
// exact redo, 4 -> 13.6 ms, whatever the shuffle granule).  So one block in every ORDER_SAMPLE_STRIDE (16) holds an
// every-16th-row sample of the base in ARRIVAL order -- what the estimate saw before there was any ordering -- and the kernel
// visits a slice's sample blocks first, re-estimates, then streams the sorted blocks (adc_scan_kernel: two_pass).  Any slice
// or XCD window that starts on a block boundary has its share of them; a sixteenth of the rows keeps the old conflict rate.
// (Measured, prepared SIFT1M base k = 1000 / 1.25e8-row shard: no sample blocks 1.94 / 15.1 ms -- and the cliff on clumped
// data --, stride 8: 2.05 / 16.3, 16: 2.02 / 15.8, 32: 2.11 / 15.8.)  The first estimate of SLICED items (a systematic row
// sample of the slice) was hit by the same clumping with 1024-row shuffle granules; with one wavefront tile per granule
// (128 rows at m = 8) a cluster's rows are dealt over the whole base and it holds.
// Returns the sample rows (a multiple of blk; 0: none) and, through *sgroups, the full groups of stride * blk positions.
int order_sample_stride() { return tuning("ORDER_SAMPLE_STRIDE", 16); }

uint32_t order_sample_rows(int64_t n, int blk, uint32_t *sgroups) {
  const int stride = order_sample_stride();
  uint32_t g = 0;
  if (stride >= 2 && blk > 0) g = (uint32_t)(n / ((int64_t)stride * blk));
  if (sgroups) *sgroups = g;
  return g * (uint32_t)blk;
}

// Does the greedy balance run for a base of n (sorted) rows of mp bytes whose full key would have `budget` bits?  Rows of 4, 8 or
// 16 bytes, the two-level path with at least 4 fine bits left after the key gives up 3, and coarse buckets (n / 256 rows) that fit
// the LDS list with room for skew.  out: {free tables, balancing wavefronts, list capacity (rows), dynamic LDS bytes}.
// A raw-pointer scan that orders its own scratch copy pays for the balance on EVERY call (0.11 ms per 1e6 rows of 8 bytes, 0.31 ms
// of 16 bytes, against 0.06 for the plain sort) and gains ~4-6 % of its scan: it pays from ORDER_GREEDY_MIN_NQ queries on (16384;
// measured at 1e4 queries: m = 8 +0.4 ... +2 %, m = 16 -0.7 ... +0.1 %).  dev_linscan announces its batch here; bases that are
// ordered ONCE (index handles, rq_dev_order_rows, ShardedIndex) leave it at 0 and always balance.
static thread_local int64_t g_order_call_nq = 0;
void order_set_call_queries(int64_t nq) { g_order_call_nq = nq; }

bool order_greedy_plan(int64_t n, int mp, int budget, uint32_t out[4]) {
  if (!tuning("ORDER_GREEDY", 1) || !tuning("ORDER_TWO_LEVEL", 1) || tuning("ORDER_CBITS", 0) > 0) return false;
  if (g_order_call_nq > 0 && g_order_call_nq < tuning("ORDER_GREEDY_MIN_NQ", 16384)) return false;
  if (mp != 8 && mp != 16) return false;
  const int total = budget - 3;
  if (total < 12 || total > 18) return false;      // (>= 4 tables under the key: the free tables are the row's last 4 / 12 bytes)
  const int gfree = mp == 8 ? 4 : 12;
  const size_t state = (size_t)GREEDY_GROUPS * gfree * 64;        // per balancing wavefront
  const size_t lds_max = 160 * 1024 - 12 * 1024;                 // static arrays of the kernel: ~9.5 KiB
  const size_t list_rows = (size_t)(n / 256) * 3 / 2 + 1024;     // 1.5 x the mean coarse bucket
  if (list_rows * 4 + state * 4 > lds_max) return false;
  const uint32_t waves = (uint32_t)std::min<size_t>(16, (lds_max - list_rows * 4) / state);
  if (out) { out[0] = (uint32_t)gfree; out[1] = waves; out[2] = (uint32_t)list_rows; out[3] = (uint32_t)(list_rows * 4 + waves * state); }
  return true;
}

int order_key_bits(int64_t n, int mp, const OrderTiling &t, int nb[8]) {
  for (int c = 0; c < 8; ++c) nb[c] = 0;
  if (n < 1024) return 0;
  n -= order_sample_rows(n, t.blk, nullptr);
  int budget = tuning("ORDER_BITS", 0);
  if (budget <= 0) budget = (int)std::floor(std::log2((double)n / (double)t.group) + 0.5);   // one bit per doubling of the group count
  budget = std::min(budget, 24);
  const int nc = std::min(mp, 8);
  int cb = tuning("ORDER_CBITS", 0);
  if (cb <= 0) cb = t.cbits;
  // round 6: where the greedy balance runs (order_greedy_plan) the key gives up one table's bits -- the buckets it leaves hold
  // ~8 lane groups of rows, and balancing those over ALL uncovered tables beats one more conflict-free table (15.5 -> 12.7 passes)
  if (order_greedy_plan(n, mp, budget, nullptr)) budget -= cb;
  int total = 0;
  for (int c = 0; c < nc && total < budget; ++c) { nb[c] = std::min(cb, budget - total); total += nb[c]; }
  // narrow rows (m <= 4): more bits per byte once every byte has its window
  for (int c = 0; c < nc && total < budget; ++c) { const int add = std::min(8 - nb[c], budget - total); nb[c] += add; total += add; }
  return total;
}

size_t order_scratch_bytes(int64_t n, int total_bits) {
  const size_t nbins = (size_t)1 << total_bits;
  const size_t ntiles = (nbins + ORDER_SCAN_TILE - 1) / ORDER_SCAN_TILE;
  return (size_t)n * 4 + (nbins + ntiles + 16) * 4;
}

// codes [n][mp] -> dst [n][mp] + perm [n]; scratch of order_scratch_bytes().  t: the scan tiling (scan_order_tiling).
int order_rows_launch(uint8_t *dst, uint32_t *perm, const uint8_t *src, int64_t n, int mp, void *scratch,
                      const OrderTiling &t, hipStream_t stream) {
  OrderParams p;
  const int rpt = t.rpt, gran = t.gran;
  const int total = order_key_bits(n, mp, t, p.nb);
  if (total <= 0 || total > 24) return fail(RQ_EINVAL, "order_rows: nothing to order (n=%lld)", (long long)n);
  p.gfree = 0; p.gwaves = 0; p.glist = 0;
  bool greedy_on = false;
  {   // did order_key_bits shorten the key for the greedy balance?  (same arithmetic: the budget of the sorted rows)
    const int64_t ns = n - order_sample_rows(n, t.blk, nullptr);
    int budget = tuning("ORDER_BITS", 0);
    if (budget <= 0) budget = (int)std::floor(std::log2((double)ns / (double)t.group) + 0.5);
    budget = std::min(budget, 24);
    greedy_on = order_greedy_plan(ns, mp, budget, nullptr) && total == budget - 3;
  }
  p.src = src; p.dst = dst; p.perm = perm;
  p.n = (uint32_t)n; p.mp = mp;
  p.ncoord = 0;
  for (int c = 0; c < 8; ++c) if (p.nb[c]) p.ncoord = c + 1;
  p.nbins = 1u << total;
  p.rank = reinterpret_cast<uint32_t *>(scratch);
  p.hist = p.rank + n;
  p.rpt = (uint32_t)rpt;
  p.group = (uint32_t)t.group;
  p.tile = 64u * (uint32_t)rpt;
  p.gran = (uint32_t)std::max(gran, (int)p.tile) / p.tile * p.tile;
  p.stride = (uint32_t)std::max(2, order_sample_stride());
  p.blk = (uint32_t)t.blk;
  p.rnd_rows = order_sample_rows(n, t.blk, &p.sgroups);
  p.nsorted = (uint32_t)n - p.rnd_rows;
  p.ngran = p.nsorted / p.gran;                 // full granules of the sorted index space
  p.weyl = 1;
  if (p.ngran > 2 && tuning("ORDER_SHUFFLE", 1)) {
    uint32_t a = (uint32_t)((double)p.ngran * 0.6180339887498949) | 1u;
    while (gcd_u32(a, p.ngran) != 1u) a += 2u;
    p.weyl = a % p.ngran;
  }
  if (total >= 9 && total <= 18 && tuning("ORDER_TWO_LEVEL", 1)) {
    // two levels: 256 coarse buckets, <= 1024 fine ones inside each (scratch: [n] row numbers | 512 counters)
    OrderSmall q;
    q.idx = reinterpret_cast<uint32_t *>(scratch);
    q.ctot = q.idx + n;
    q.fine_bits = total - 8;
    q.rows_per_wg = 4 * ORDER_SMALL_THREADS;       // (order_coarse_scatter_kernel: PER)
    const uint32_t nwg = (uint32_t)((n + q.rows_per_wg - 1) / q.rows_per_wg);
    RQ_HIP(hipMemsetAsync(q.ctot, 0, 2 * ORDER_COARSE * 4, stream));
    hipLaunchKernelGGL(order_coarse_count_kernel, dim3(nwg), dim3(ORDER_SMALL_THREADS), 0, stream, p, q);
    hipLaunchKernelGGL(order_coarse_scatter_kernel, dim3(nwg), dim3(ORDER_SMALL_THREADS), 0, stream, p, q);
    uint32_t gp[4];
    if (greedy_on && order_greedy_plan((int64_t)p.nsorted, mp, total + 3, gp)) {
      // the free tables: the last gfree bytes of the row (the key covers the leading ones; a partly covered byte is balanced too)
      p.gfree = (int)gp[0];
      p.gwaves = gp[1]; p.glist = gp[2];
      RQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(order_fine_greedy_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)gp[3]));
      hipLaunchKernelGGL(order_fine_greedy_kernel, dim3(ORDER_COARSE), dim3(ORDER_SMALL_THREADS), gp[3], stream, p, q);
    } else {
      hipLaunchKernelGGL(order_fine_kernel, dim3(ORDER_COARSE), dim3(ORDER_SMALL_THREADS), 0, stream, p, q);
    }
    RQ_HIP(hipGetLastError());
    return RQ_OK;
  }
  const uint32_t ntiles = (p.nbins + ORDER_SCAN_TILE - 1) / ORDER_SCAN_TILE;
  RQ_HIP(hipMemsetAsync(p.hist, 0, (size_t)(p.nbins + ntiles) * 4, stream));
  const uint32_t grid = (uint32_t)std::min<int64_t>((n + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(order_rank_kernel, dim3(grid), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(order_scan_tiles_kernel, dim3(ntiles), dim3(1024), 0, stream, p.hist, p.nbins);
  hipLaunchKernelGGL(order_scan_top_kernel, dim3(1), dim3(1024), 0, stream, p.hist + p.nbins, ntiles);
  hipLaunchKernelGGL(order_scatter_kernel, dim3(grid), dim3(256), 0, stream, p);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

// dst[i] = src[perm[i]]  (per-row side arrays of an ordered base: LSQ norms)
__global__ void gather_f32_kernel(float *__restrict__ dst, const float *__restrict__ src, const uint32_t *__restrict__ perm, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[perm[i]];
}

int gather_f32_launch(float *dst, const float *src, const uint32_t *perm, int64_t n, hipStream_t stream) {
  if (n <= 0) return RQ_OK;
  hipLaunchKernelGGL(gather_f32_kernel, dim3((uint32_t)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, dst, src, perm, (uint32_t)n);
  RQ_HIP(hipGetLastError());
  return RQ_OK;
}

}  // namespace rq
