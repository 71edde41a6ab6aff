// micro-benchmark: how fast can the byte pre-filter stream alone (no exact path, no top-k in the kernel)?
// Workgroup = NT threads, byte tables of QGB queries in LDS ([set of 8][k][256] uint2), persistent over
// (query group, row slice) items, codes [n][8] streamed 16 B per lane.  Alive rows are pushed to a per-wave
// LDS queue and dropped (the refine would take them from there).  Prints ms for n = 1e6 rows x nq = 1e4 queries.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/filter_lean.hip -o /tmp/filter_lean && /tmp/filter_lean
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

template <int NT, int QGB, int R, int KG = 0>
__global__ __launch_bounds__(NT) void filt_kernel(const uint8_t *__restrict__ codes, const uint2 *__restrict__ tabs, uint32_t n,
                                                  uint32_t ngroups, uint32_t nslices, uint32_t rows_per_slice,
                                                  uint32_t thr, unsigned long long *alive_out, uint32_t *counter) {
  constexpr int NS = QGB / 8;           // 8-query sets
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // R replicas of the table: lane l reads replica l % R, whose entries only occupy the 8-byte slots = l (mod R) of
  // a 256-byte bank row, so the 32/R lanes that share a replica spread over 32/R slots of their own
  uint2 *qt = reinterpret_cast<uint2 *>(smem);                           // [NS][8][256][R]
  uint32_t *queue = reinterpret_cast<uint32_t *>(qt + NS * 8 * 256 * R);  // [NT/64][256]
  __shared__ uint32_t s_item;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *myq = queue + __builtin_amdgcn_readfirstlane(wave * 256);
  unsigned long long alive_total = 0;
  const uint32_t nitems = ngroups * nslices;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_item = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= nitems) break;
    const uint32_t group = item % ngroups, slice = item / ngroups;
    for (int i = tid; i < NS * 8 * 256 * R; i += NT) qt[i] = tabs[(size_t)group * NS * 8 * 256 + i / R];
    const int rep = lane % R;
    const uint2 *__restrict__ gt = tabs + (size_t)group * NS * 8 * 256;
    __syncthreads();
    const uint32_t r_begin = slice * rows_per_slice, r_end = min(n, r_begin + rows_per_slice);
    uint32_t qtail = 0;
    constexpr int U = 2;
    for (uint32_t base = r_begin; base < r_end; base += NT * 2 * U) {
      uint4 wu[U];
      uint32_t amask = 0; (void)amask;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
        wu[u] = row0 + 2 <= r_end ? *reinterpret_cast<const uint4 *>(codes + (size_t)row0 * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t w[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w};
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          uint2 e[2][8];
#pragma unroll
          for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const uint32_t byte = (w[(r * 8 + k) >> 2] >> (8 * ((r * 8 + k) & 3))) & 0xffu;
              // KG > 0: the last KG sub-quantizers' tables are gathered through the vector-memory path (L1) instead of LDS
              if (k >= 8 - KG) e[r][k] = gt[(s * 8 + k) * 256 + byte];
              else e[r][k] = qt[((s * 8 + k) * 256 + byte) * R + rep];
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            uint32_t a0 = 0, a1 = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) { a0 += e[r][k].x; a1 += e[r][k].y; }
            constexpr uint32_t H = 0x80808080u;
            const uint32_t TC = (thr + 1u) * 0x01010101u;
            const uint32_t g0 = ((a0 | H) - TC) | a0, g1 = ((a1 | H) - TC) | a1;
            const bool cand = ((g0 & g1 & H) != H) && (row0 + r < r_end);
#ifdef DEFER_PUSH
            // round 4: one bit per (set, sub-step, row); the rows are queued once per block below
            amask |= cand ? (1u << ((s * U + u) * 2 + r)) : 0u;
#else
            const uint64_t mq = __ballot(cand);
            if (mq) {
              if (cand) myq[(qtail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u))) & 255u] = (row0 + r) | (s << 29);
              qtail += (uint32_t)__popcll(mq);
            }
#endif
          }
        }
      }
#ifdef DEFER_PUSH
      for (;;) {
        const bool has = amask != 0u;
        const uint64_t mq = __builtin_amdgcn_ballot_w64(has);
        if (!mq) break;
        const uint32_t b = (uint32_t)__builtin_ctz(amask | 0x80000000u);
        amask &= amask - 1u;
        if (has) myq[(qtail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u))) & 255u] = base + b;
        qtail += (uint32_t)__popcll(mq);
      }
#endif
    }
    alive_total += qtail;
  }
  if (lane == 0) atomicAdd(alive_out, alive_total);
}

// 16 queries per ds_read_b64: 4-bit table entries (values 0..7, so that the sum of two fits a nibble), 16 nibbles per
// 8-byte entry.  Per row: 8 gathers, pairwise nibble adds (4 x 2 words), unpack to byte sums (and / shift-and), accumulate,
// test 16 byte sums.  VERDICT r2 asked for this experiment after the 32-entry level-0 bound turned out useless.
template <int NT>
__global__ __launch_bounds__(NT) void filt_nib_kernel(const uint8_t *__restrict__ codes, const uint2 *__restrict__ tabs, uint32_t n,
                                                      uint32_t ngroups, uint32_t nslices, uint32_t rows_per_slice,
                                                      uint32_t thr, unsigned long long *alive_out, uint32_t *counter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint2 *qt = reinterpret_cast<uint2 *>(smem);                       // [8][256] entries of 16 nibbles
  uint32_t *queue = reinterpret_cast<uint32_t *>(qt + 8 * 256);
  __shared__ uint32_t s_item;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *myq = queue + __builtin_amdgcn_readfirstlane(wave * 256);
  unsigned long long alive_total = 0;
  const uint32_t nitems = ngroups * nslices;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_item = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= nitems) break;
    const uint32_t group = item % ngroups, slice = item / ngroups;
    for (int i = tid; i < 8 * 256; i += NT) { uint2 v = tabs[(size_t)group * 8 * 256 + i]; v.x &= 0x77777777u; v.y &= 0x77777777u; qt[i] = v; }
    __syncthreads();
    const uint32_t r_begin = slice * rows_per_slice, r_end = min(n, r_begin + rows_per_slice);
    uint32_t qtail = 0;
    constexpr int U = 2;
    for (uint32_t base = r_begin; base < r_end; base += NT * 2 * U) {
      uint4 wu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
        wu[u] = row0 + 2 <= r_end ? *reinterpret_cast<const uint4 *>(codes + (size_t)row0 * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t w[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w};
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
        uint2 e[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) e[r][k] = qt[k * 256 + ((w[(r * 8 + k) >> 2] >> (8 * ((r * 8 + k) & 3))) & 0xffu)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          uint32_t b[4] = {0, 0, 0, 0};      // byte sums: low nibbles of x, high nibbles of x, low of y, high of y
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            const uint32_t px = e[r][k].x + e[r][k + 1].x, py = e[r][k].y + e[r][k + 1].y;     // nibble sums <= 14
            b[0] += px & 0x0f0f0f0fu; b[1] += (px >> 4) & 0x0f0f0f0fu;
            b[2] += py & 0x0f0f0f0fu; b[3] += (py >> 4) & 0x0f0f0f0fu;
          }
          constexpr uint32_t H = 0x80808080u;
          const uint32_t TC = (thr + 1u) * 0x01010101u;
          uint32_t g = H;
#pragma unroll
          for (int i = 0; i < 4; ++i) g &= ((b[i] | H) - TC) | b[i];
          const bool cand = ((g & H) != H) && (row0 + r < r_end);
          const uint64_t mq = __ballot(cand);
          if (mq) {
            if (cand) myq[(qtail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u))) & 255u] = row0 + r;
            qtail += (uint32_t)__popcll(mq);
          }
        }
      }
    }
    alive_total += qtail;
  }
  if (lane == 0) atomicAdd(alive_out, alive_total);
}

// Cascade: level A = the nibble filter above for 16 queries; rows it lets through wait in a per-wave queue and get level B,
// 64 rows at a time: the byte tables of BOTH 8-query groups (2 x 8 ds_read_b64 per row), alive test per group, survivors
// counted (the exact evaluation would take them from there).  thrA tunes the share of rows that reach level B.
template <int NT>
__global__ __launch_bounds__(NT) void filt_casc_kernel(const uint8_t *__restrict__ codes, const uint2 *__restrict__ tabs, const uint2 *__restrict__ ntabs, uint32_t n,
                                                       uint32_t ngroups, uint32_t nslices, uint32_t rows_per_slice,
                                                       uint32_t thrA, uint32_t thrB, unsigned long long *alive_out, uint32_t *counter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint2 *qn = reinterpret_cast<uint2 *>(smem);                       // [8][256] nibble entries (16 queries)
  uint2 *qb = qn + 8 * 256;                                          // [2][8][256] byte entries (8 queries each)
  uint32_t *queue = reinterpret_cast<uint32_t *>(qb + 2 * 8 * 256);  // [NT/64][256] row ids
  __shared__ uint32_t s_item;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *myq = queue + __builtin_amdgcn_readfirstlane(wave * 256);
  unsigned long long alive_total = 0, reachedB = 0;
  const uint32_t nitems = ngroups * nslices;
  auto levelB = [&](uint32_t row, bool valid) -> uint32_t {
    const uint2 cw = valid ? *reinterpret_cast<const uint2 *>(codes + (size_t)row * 8) : make_uint2(0, 0);
    const uint32_t w[2] = {cw.x, cw.y};
    uint32_t hits = 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      uint2 e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = qb[(g * 8 + k) * 256 + ((w[k >> 2] >> (8 * (k & 3))) & 0xffu)];
      uint32_t a0 = 0, a1 = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { a0 += e[k].x; a1 += e[k].y; }
      constexpr uint32_t H = 0x80808080u;
      const uint32_t TC = (thrB + 1u) * 0x01010101u;
      const uint32_t g0 = ((a0 | H) - TC) | a0, g1 = ((a1 | H) - TC) | a1;
      hits += (valid && ((g0 & g1 & H) != H)) ? 1u : 0u;
    }
    return hits;
  };
  for (;;) {
    __syncthreads();
    if (tid == 0) s_item = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= nitems) break;
    const uint32_t group = item % ngroups, slice = item / ngroups;
    for (int i = tid; i < 8 * 256; i += NT) qn[i] = ntabs[(size_t)group * 8 * 256 + i];
    for (int i = tid; i < 2 * 8 * 256; i += NT) qb[i] = tabs[((size_t)group * 2 * 8 * 256 + i) % ((size_t)ngroups * 8 * 256)];
    __syncthreads();
    const uint32_t r_begin = slice * rows_per_slice, r_end = min(n, r_begin + rows_per_slice);
    uint32_t qtail = 0;
    constexpr int U = 2;
    for (uint32_t base = r_begin; base < r_end; base += NT * 2 * U) {
      uint4 wu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
        wu[u] = row0 + 2 <= r_end ? *reinterpret_cast<const uint4 *>(codes + (size_t)row0 * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t w[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w};
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
        uint2 e[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) e[r][k] = qn[k * 256 + ((w[(r * 8 + k) >> 2] >> (8 * ((r * 8 + k) & 3))) & 0xffu)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          uint32_t b[4] = {0, 0, 0, 0};
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            const uint32_t px = e[r][k].x + e[r][k + 1].x, py = e[r][k].y + e[r][k + 1].y;
            b[0] += px & 0x0f0f0f0fu; b[1] += (px >> 4) & 0x0f0f0f0fu;
            b[2] += py & 0x0f0f0f0fu; b[3] += (py >> 4) & 0x0f0f0f0fu;
          }
          constexpr uint32_t H = 0x80808080u;
          const uint32_t TC = (thrA + 1u) * 0x01010101u;
          uint32_t g = H;
#pragma unroll
          for (int i = 0; i < 4; ++i) g &= ((b[i] | H) - TC) | b[i];
          const bool cand = ((g & H) != H) && (row0 + r < r_end);
          const uint64_t mq = __ballot(cand);
          if (mq) {
            if (cand) myq[qtail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u))] = row0 + r;
            qtail += (uint32_t)__popcll(mq);
          }
        }
        while (qtail >= 64u) {          // level B for 64 queued rows
          qtail -= 64u;
          alive_total += levelB(myq[qtail + lane], true);
          reachedB += 1;
        }
      }
    }
    if (qtail) { alive_total += levelB(myq[lane < qtail ? lane : 0], lane < qtail); reachedB += (lane < qtail) ? 1 : 0; qtail = 0; }
  }
  atomicAdd(alive_out, alive_total);
  atomicAdd(alive_out + 1, reachedB);
}

template <int NT>
static void run_casc(const uint8_t *codes, const uint2 *tabs, const uint2 *ntabs, uint32_t n, uint32_t nq, uint32_t thrA, uint32_t thrB, int wgs_per_cu) {
  const uint32_t ngroups = nq / 16;
  const uint32_t grid = 256 * wgs_per_cu;
  uint32_t nslices = ngroups >= grid ? 1 : (grid + ngroups - 1) / ngroups;
  uint32_t rps = (n + nslices - 1) / nslices;
  rps = (rps + NT * 4 - 1) / (NT * 4) * (NT * 4);
  nslices = (n + rps - 1) / rps;
  unsigned long long *alive; uint32_t *counter;
  hipMalloc(&alive, 16); hipMalloc(&counter, 4);
  const size_t lds = (size_t)3 * 8 * 256 * 8 + (size_t)(NT / 64) * 256 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void *>(filt_casc_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  unsigned long long al[2] = {0, 0};
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(alive, 0, 16); hipMemset(counter, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL((filt_casc_kernel<NT>), dim3(grid), dim3(NT), lds, 0, codes, tabs, ntabs, n, ngroups, nslices, rps, thrA, thrB, alive, counter);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    hipMemcpy(al, alive, 16, hipMemcpyDeviceToHost);
  }
  printf("CASCADE NT=%4d WGs/CU=%d thrA=%u: %.3f ms for 1e10 (row, query) pairs; rows reaching level B %.2f%%, (row, 8-query group) alive after B %.3f%%  err=%s\n",
         NT, wgs_per_cu, thrA, best, 100.0 * (double)al[1] * (NT >= 0 ? 1.0 : 1.0) / ((double)n * (nq / 16)) * 1.0, 100.0 * (double)al[0] / ((double)n * (nq / 8)),
         hipGetErrorString(hipGetLastError()));
}

template <int NT>
static void run_nib(const uint8_t *codes, const uint2 *tabs, uint32_t n, uint32_t nq, uint32_t thr, int wgs_per_cu) {
  const uint32_t ngroups = nq / 16;
  const uint32_t grid = 256 * wgs_per_cu;
  uint32_t nslices = ngroups >= grid ? 1 : (grid + ngroups - 1) / ngroups;
  uint32_t rps = (n + nslices - 1) / nslices;
  rps = (rps + NT * 4 - 1) / (NT * 4) * (NT * 4);
  nslices = (n + rps - 1) / rps;
  unsigned long long *alive; uint32_t *counter;
  hipMalloc(&alive, 8); hipMalloc(&counter, 4);
  const size_t lds = (size_t)8 * 256 * 8 + (size_t)(NT / 64) * 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  unsigned long long al = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(alive, 0, 8); hipMemset(counter, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL((filt_nib_kernel<NT>), dim3(grid), dim3(NT), lds, 0, codes, tabs, n, ngroups, nslices, rps, thr, alive, counter);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    hipMemcpy(&al, alive, 8, hipMemcpyDeviceToHost);
  }
  printf("NIBBLE NT=%4d 16 queries per 8-byte gather, WGs/CU=%d slices=%u: %.3f ms for the same 1e10 (row, query) pairs  alive (row,set) share %.3f%%  err=%s\n", NT, wgs_per_cu, nslices, best,
         100.0 * (double)al / ((double)n * (nq / 16)), hipGetErrorString(hipGetLastError()));
}


// Round 6 (VERDICT r5 Next #7): 16 queries per gather with BYTE entries -- one ds_read_b128 per code byte, table [8][256] uint4
// (32 KiB per 16-query group).  Same adds per query as the 8-query kernel (4 dwords of 4 byte sums), half the addresses.  With
// ORDER4 the rows are sorted by the top 4 bits of their leading bytes (a 16-value window = the 16 sixteen-byte columns a
// 16-lane group of ds_read_b128 covers).
template <int NT>
__global__ __launch_bounds__(NT) void filt16_kernel(const uint8_t *__restrict__ codes, const uint4 *__restrict__ tabs, uint32_t n,
                                                    uint32_t ngroups, uint32_t nslices, uint32_t rows_per_slice,
                                                    uint32_t thr, unsigned long long *alive_out, uint32_t *counter) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4 *qt = reinterpret_cast<uint4 *>(smem);                        // [8][256] entries of 16 bytes
  uint32_t *queue = reinterpret_cast<uint32_t *>(qt + 8 * 256);
  __shared__ uint32_t s_item;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t *myq = queue + __builtin_amdgcn_readfirstlane(wave * 256);
  unsigned long long alive_total = 0;
  const uint32_t nitems = ngroups * nslices;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_item = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= nitems) break;
    const uint32_t group = item % ngroups, slice = item / ngroups;
    for (int i = tid; i < 8 * 256; i += NT) qt[i] = tabs[(size_t)group * 8 * 256 + i];
    __syncthreads();
    const uint32_t r_begin = slice * rows_per_slice, r_end = min(n, r_begin + rows_per_slice);
    uint32_t qtail = 0;
    constexpr int U = 2;
    for (uint32_t base = r_begin; base < r_end; base += NT * 2 * U) {
      uint4 wu[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
        wu[u] = row0 + 2 <= r_end ? *reinterpret_cast<const uint4 *>(codes + (size_t)row0 * 8) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t w[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w};
        const uint32_t row0 = base + u * NT * 2 + tid * 2;
#pragma unroll
        for (int r = 0; r < 2; ++r) {          // one row at a time: 8 x 16 bytes = 32 registers of entries
          uint4 e[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) e[k] = qt[k * 256 + ((w[(r * 8 + k) >> 2] >> (8 * ((r * 8 + k) & 3))) & 0xffu)];
          __builtin_amdgcn_sched_barrier(0);
          uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) { a0 += e[k].x; a1 += e[k].y; a2 += e[k].z; a3 += e[k].w; }
          constexpr uint32_t H = 0x80808080u;
          const uint32_t TC = (thr + 1u) * 0x01010101u;
          const uint32_t g = (((a0 | H) - TC) | a0) & (((a1 | H) - TC) | a1) & (((a2 | H) - TC) | a2) & (((a3 | H) - TC) | a3);
          const bool cand = ((g & H) != H) && (row0 + r < r_end);
          const uint64_t mq = __ballot(cand);
          if (mq) {
            if (cand) myq[(qtail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mq >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mq, 0u))) & 255u] = row0 + r;
            qtail += (uint32_t)__popcll(mq);
          }
        }
      }
    }
    alive_total += qtail;
  }
  if (lane == 0) atomicAdd(alive_out, alive_total);
}

template <int NT>
static void run16(const uint8_t *codes, const uint2 *tabs, uint32_t n, uint32_t nq, uint32_t thr, int wgs_per_cu) {
  const uint32_t ngroups = nq / 16;
  const uint32_t grid = 256 * wgs_per_cu;
  uint32_t nslices = ngroups >= grid ? 1 : (grid + ngroups - 1) / ngroups;
  uint32_t rps = (n + nslices - 1) / nslices;
  rps = (rps + NT * 4 - 1) / (NT * 4) * (NT * 4);
  nslices = (n + rps - 1) / rps;
  unsigned long long *alive; uint32_t *counter;
  hipMalloc(&alive, 8); hipMalloc(&counter, 4);
  const size_t lds = (size_t)8 * 256 * 16 + (size_t)(NT / 64) * 256 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void *>(filt16_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  unsigned long long al = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(alive, 0, 8); hipMemset(counter, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL((filt16_kernel<NT>), dim3(grid), dim3(NT), lds, 0, codes, reinterpret_cast<const uint4 *>(tabs), n, ngroups, nslices, rps, thr, alive, counter);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    hipMemcpy(&al, alive, 8, hipMemcpyDeviceToHost);
  }
  printf("B128 16 queries per ds_read_b128 NT=%4d WGs/CU=%d slices=%u lds=%zu KB: %.3f ms for the same 1e10 (row, query) pairs  alive (row,16-query set) share %.3f%%  err=%s\n",
         NT, wgs_per_cu, nslices, lds / 1024, best, 100.0 * (double)al / ((double)n * (nq / 16)), hipGetErrorString(hipGetLastError()));
}

template <int NT, int QGB, int R = 1, int KG = 0>
static void run(const uint8_t *codes, const uint2 *tabs, uint32_t n, uint32_t nq, uint32_t thr, int wgs_per_cu) {
  constexpr int NS = QGB / 8;
  const uint32_t ngroups = nq / QGB;
  const uint32_t grid = 256 * wgs_per_cu;
  uint32_t nslices = ngroups >= grid ? 1 : (grid + ngroups - 1) / ngroups;
  uint32_t rps = (n + nslices - 1) / nslices;
  rps = (rps + NT * 4 - 1) / (NT * 4) * (NT * 4);
  nslices = (n + rps - 1) / rps;
  unsigned long long *alive; uint32_t *counter;
  hipMalloc(&alive, 8); hipMalloc(&counter, 4);
  const size_t lds = (size_t)NS * 8 * 256 * 8 * R + (size_t)(NT / 64) * 256 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void *>(filt_kernel<NT, QGB, R, KG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  unsigned long long al = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipMemset(alive, 0, 8); hipMemset(counter, 0, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL((filt_kernel<NT, QGB, R, KG>), dim3(grid), dim3(NT), lds, 0, codes, tabs, n, ngroups, nslices, rps, thr, alive, counter);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
    hipMemcpy(&al, alive, 8, hipMemcpyDeviceToHost);
  }
  printf("KG=%d R=%d NT=%4d QGB=%2d WGs/CU=%d slices=%u lds=%zu KB: %.3f ms  alive (row,set) share %.3f%%  err=%s\n", KG, R, NT, QGB, wgs_per_cu, nslices, lds / 1024, best,
         100.0 * (double)al / ((double)n * (nq / 8)), hipGetErrorString(hipGetLastError()));
}

int main() {
  const uint32_t n = 1000000, nq = 10016;   // multiple of 32
  std::vector<uint8_t> hc((size_t)n * 8);
  for (auto &c : hc) c = (uint8_t)(rand() >> 8);
  if (const char *ob = getenv("ORDER")) {
    // round 4: bank-aware row order (csrc/rq_order.hip): counting sort by the top 3 bits of the leading code bytes, the 32
    // consecutive sorted rows of a lane group dealt to lanes (2 rows per lane), 1024-row granules left in sort order
    // (the micro-kernel has no threshold estimate that would need the shuffle)
    const int bits = atoi(ob);
    std::vector<uint64_t> key(n);
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; ++i) {
      uint64_t k = 0; int left = bits;
      const int per = getenv("ORDER_BITS_PER_BYTE") ? atoi(getenv("ORDER_BITS_PER_BYTE")) : 3;     // 4: 16-value windows (ds_read_b128 groups)
      for (int c = 0; c < 8 && left > 0; ++c) { const int nb = left < per ? left : per; k = (k << nb) | (hc[(size_t)i * 8 + c] >> (8 - nb)); left -= nb; }
      key[i] = (k << 32) | i; idx[i] = i;
    }
    std::sort(key.begin(), key.end());
    std::vector<uint8_t> h2(hc.size());
    for (uint32_t s = 0; s < n; ++s) {
      const uint32_t src = (uint32_t)key[s];
      uint32_t pos = s;
      const uint32_t t = s / 128;
      if ((t + 1) * 128 <= n) { const uint32_t u = s % 128, h = u / 64, v = u % 64, r = v / 32, j = v % 32; pos = t * 128 + h * 64 + j * 2 + r; }
      for (int c = 0; c < 8; ++c) h2[(size_t)pos * 8 + c] = hc[(size_t)src * 8 + c];
    }
    hc.swap(h2);
    printf("rows ordered by a %d-bit key\n", bits);
  }
  std::vector<uint8_t> ht((size_t)(nq / 8) * 8 * 256 * 8);
  for (auto &t : ht) t = (uint8_t)((rand() >> 8) % 28);        // entries 0..27: sums ~108 +- 23
  uint8_t *codes; uint2 *tabs;
  hipMalloc(&codes, hc.size()); hipMalloc(&tabs, ht.size());
  hipMemcpy(codes, hc.data(), hc.size(), hipMemcpyHostToDevice);
  hipMemcpy(tabs, ht.data(), ht.size(), hipMemcpyHostToDevice);
  std::vector<uint8_t> hn((size_t)(nq / 16) * 8 * 256 * 8);
  for (auto &t : hn) t = (uint8_t)(((rand() >> 8) % 8) | (((rand() >> 8) % 8) << 4));   // nibble entries 0..7
  uint2 *ntabs; hipMalloc(&ntabs, hn.size()); hipMemcpy(ntabs, hn.data(), hn.size(), hipMemcpyHostToDevice);
  const uint32_t thr = getenv("THR") ? atoi(getenv("THR")) : 62;   // ~ a few % of (row, set) pairs alive
  run<512, 8>(codes, tabs, n, nq, thr, 2);
  run<512, 8>(codes, tabs, n, nq, thr, 4);
  run16<512>(codes, tabs, n, nq, thr, 2);
  run16<512>(codes, tabs, n, nq, thr, 4);
  run16<256>(codes, tabs, n, nq, thr, 8);
  if (getenv("MICRO_B128_ONLY")) return 0;
  if (getenv("MICRO_QUICK")) {
    run<256, 8>(codes, tabs, n, nq, thr, 8);
    run<512, 16>(codes, tabs, n, nq, thr, 4);
    run<512, 16>(codes, tabs, n, nq, thr, 2);
    run<512, 32>(codes, tabs, n, nq, thr, 2);
    if (getenv("MICRO_FILT_ONLY")) return 0;
    run_nib<512>(codes, tabs, n, nq, 40, 2);
    run_nib<512>(codes, tabs, n, nq, 40, 4);
    for (uint32_t tA : {9u, 13u}) { run_casc<512>(codes, tabs, ntabs, n, nq, tA, 62, 2); run_casc<512>(codes, tabs, ntabs, n, nq, tA, 62, 4); }
    return 0;
  }
  run<1024, 8>(codes, tabs, n, nq, thr, 2);
  run<512, 16>(codes, tabs, n, nq, thr, 2);
  run<512, 16>(codes, tabs, n, nq, thr, 4);
  run<1024, 16>(codes, tabs, n, nq, thr, 2);
  run<512, 32>(codes, tabs, n, nq, thr, 2);
  run<1024, 32>(codes, tabs, n, nq, thr, 2);
  run<1024, 32>(codes, tabs, n, nq, thr, 1);
  for (uint32_t tA : {9u, 11u, 13u, 15u, 17u}) { run_casc<512>(codes, tabs, ntabs, n, nq, tA, 62, 2); run_casc<512>(codes, tabs, ntabs, n, nq, tA, 62, 4); run_casc<1024>(codes, tabs, ntabs, n, nq, tA, 62, 1); }
  run_nib<512>(codes, tabs, n, nq, 40, 2);
  run_nib<512>(codes, tabs, n, nq, 40, 4);
  run_nib<1024>(codes, tabs, n, nq, 40, 2);
  // L1 path for the last KG sub-quantizers
  run<512, 8, 1, 1>(codes, tabs, n, nq, thr, 2);
  run<512, 8, 1, 2>(codes, tabs, n, nq, thr, 2);
  run<512, 8, 1, 3>(codes, tabs, n, nq, thr, 2);
  run<512, 8, 1, 2>(codes, tabs, n, nq, thr, 4);
  run<512, 8, 1, 3>(codes, tabs, n, nq, thr, 4);
  run<512, 8, 1, 4>(codes, tabs, n, nq, thr, 4);
  run<512, 8, 2, 2>(codes, tabs, n, nq, thr, 4);
  run<512, 8, 4, 2>(codes, tabs, n, nq, thr, 2);
  // table replicas against bank conflicts (8 queries per workgroup: 16 KB x R)
  run<512, 8, 2>(codes, tabs, n, nq, thr, 4);
  run<512, 8, 4>(codes, tabs, n, nq, thr, 2);
  run<512, 8, 4>(codes, tabs, n, nq, thr, 4);
  run<512, 8, 8>(codes, tabs, n, nq, thr, 1);
  run<1024, 8, 8>(codes, tabs, n, nq, thr, 1);
  run<512, 16, 2>(codes, tabs, n, nq, thr, 2);
  run<512, 16, 4>(codes, tabs, n, nq, thr, 1);
  return 0;
}
