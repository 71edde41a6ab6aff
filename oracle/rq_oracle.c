/*
 * rq_oracle.c -- CPU restatement of Rayuela.jl's PQ/OPQ encode + ADC linear scan.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle for the HIP path and
 * the "port" leg of bench.py's cpu_baseline.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline may load it; the product library (librayuela_hip.so)
 * never links, loads or falls back to anything in oracle/.
 *
 * Parity status
 *   - ADC scan (oracle_linscan_aqd_query, oracle_adc_lut): PINNED.  Checked bit for
 *     bit (ids and distances) against the real reference deps/src/linscan_aqd.cpp
 *     compiled with its own flags into oracle/_ref/ (see oracle/Makefile) and against
 *     the committed fixtures in tests/golden/ that were generated from that build.
 *   - Encode / rotation (oracle_encode_pq, oracle_rotate_T): PARITY UNPINNED.  The
 *     reference encode is Julia on top of Distances.jl v0.8.0 + Clustering.jl v0.12.2
 *     + OpenBLAS (Manifest.toml:79-83,135-139); none of those sources are under
 *     /root/reference, no Julia exists in this image and no reference test holds a
 *     golden vector for it.  The arithmetic below restates the published algorithm of
 *     those two packages (GEMM-trick squared distance clamped at 0, strict-'<' argmin)
 *     with the one thing they leave to the BLAS build -- the f32 summation order --
 *     fixed to a k-ordered fmaf chain.
 *
 * Memory layouts are the C views of the Julia arrays (column-major d x n == row-major
 * [n][d]); see include/rayuela_hip.h.
 *
 * Build: gcc -O3 -fopenmp -mavx2 -mfma -ffp-contract=off  (oracle/Makefile).
 * -ffp-contract=off matters: the reference LUT arithmetic is an UNFUSED mul then add
 * (x86-64 baseline build, deps/build.jl:23); fmaf() is used only where named.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* src/utils.jl:179-203  splitarray(1:d, m): contiguous parts, the first d%m  */
/* parts carry one extra element.  offsets has m+1 entries (zero-based).       */
/* ------------------------------------------------------------------------- */
void oracle_splitarray(int d, int m, int *offsets) {
  int per = d / m, extra = d % m, pos = 0;
  for (int i = 0; i < m; i++) {
    offsets[i] = pos;
    pos += per + (i < extra ? 1 : 0);
  }
  offsets[m] = pos;
}

/* ------------------------------------------------------------------------- */
/* deps/src/linscan_aqd.cpp:66-74  per-query look-up table.                    */
/* lut[k*256+r] = sum_{s<subdim} (centers[(k*256+r)*subdim+s] - q[k*subdim+s])^2 */
/* accumulated sequentially in f32 from 0, mul and add NOT fused.              */
/* ------------------------------------------------------------------------- */
void oracle_adc_lut(float *lut, const float *centers, const float *query, int m,
                    int subdim) {
  for (int k = 0; k < m; k++) {
    const float *q = query + (size_t)k * subdim;
    for (int r = 0; r < 256; r++) {
      const float *c = centers + ((size_t)k * 256 + r) * subdim;
      float acc = 0.0f;
      for (int s = 0; s < subdim; s++) {
        float diff = c[s] - q[s];
        float sq = diff * diff;
        acc = acc + sq;
      }
      lut[k * 256 + r] = acc;
    }
  }
}

/* (dist, id) pairs ordered lexicographically -- what std::partial_sort on       */
/* pair<float,UINT32> yields in deps/src/linscan_aqd.cpp:91.                      */
typedef struct {
  float dist;
  uint32_t id;
} oracle_pair;

static inline int pair_less(oracle_pair a, oracle_pair b) {
  return (a.dist < b.dist) || (a.dist == b.dist && a.id < b.id);
}

static int pair_cmp_qsort(const void *pa, const void *pb) {
  oracle_pair a = *(const oracle_pair *)pa, b = *(const oracle_pair *)pb;
  if (pair_less(a, b)) return -1;
  if (pair_less(b, a)) return 1;
  return 0;
}

/* bounded max-heap holding the K lexicographically smallest pairs seen so far */
static void heap_sift_down(oracle_pair *h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, big = i;
    if (l < n && pair_less(h[big], h[l])) big = l;
    if (r < n && pair_less(h[big], h[r])) big = r;
    if (big == i) return;
    oracle_pair t = h[i];
    h[i] = h[big];
    h[big] = t;
    i = big;
  }
}

/* ------------------------------------------------------------------------- */
/* deps/src/linscan_aqd.cpp:37-114  linscan_aqd_query, same argument list.     */
/* dist_j = ((lut[0][b_j0] + lut[1][b_j1]) + ...) sequential f32 (:85-87);      */
/* result = the K smallest (dist, id) pairs in ascending lexicographic order    */
/* (:91-97), ids zero-based.  The reference gets there with a full pair array   */
/* and partial_sort in 1e7-row chunks; the global lexicographic top-K is the    */
/* same set, taken here with a bounded heap.  Requires 1 <= K <= N, B == 8*m.    */
/* ------------------------------------------------------------------------- */
void oracle_linscan_aqd_query(float *dists, uint32_t *res, const uint8_t *codes,
                              const float *centers, const float *queries, int N,
                              uint32_t NQ, int B, int K, int dim1codes,
                              int dim1queries, int subdim) {
  const int m = B / 8;
#pragma omp parallel
  {
    float *lut = (float *)malloc(sizeof(float) * (size_t)m * 256);
    oracle_pair *heap = (oracle_pair *)malloc(sizeof(oracle_pair) * (size_t)K);
#pragma omp for schedule(dynamic, 1)
    for (int64_t qi = 0; qi < (int64_t)NQ; qi++) {
      oracle_adc_lut(lut, centers, queries + (size_t)qi * dim1queries, m, subdim);
      int filled = 0;
      const uint8_t *pc = codes;
      for (int j = 0; j < N; j++, pc += dim1codes) {
        float acc = 0.0f;
        for (int k = 0; k < m; k++) acc = acc + lut[k * 256 + pc[k]];
        oracle_pair p = {acc, (uint32_t)j};
        if (filled < K) {
          heap[filled++] = p;
          if (filled == K)
            for (int i = K / 2 - 1; i >= 0; i--) heap_sift_down(heap, K, i);
        } else if (pair_less(p, heap[0])) {
          heap[0] = p;
          heap_sift_down(heap, K, 0);
        }
      }
      qsort(heap, (size_t)filled, sizeof(oracle_pair), pair_cmp_qsort);
      float *pd = dists + (size_t)qi * K;
      uint32_t *pr = res + (size_t)qi * K;
      for (int j = 0; j < filled; j++) {
        pd[j] = heap[j].dist;
        pr[j] = heap[j].id;
      }
      for (int j = filled; j < K; j++) {
        pd[j] = 0.0f;
        pr[j] = 0;
      }
    }
    free(lut);
    free(heap);
  }
}

/* ------------------------------------------------------------------------- */
/* deps/src/linscan_aqd_pairwise_byte.cpp:14-94 (mode 1, LSQ: T -= (2 q_k) c_k over the full  */
/* dimension, dist = sum_k T[h*k + b_k] then + dbnorms[row]) and :97-176 (mode 2, CQ:          */
/* T += (q_k - c_k)^2, no norms).  codebooks [m*h][d]; ids ONE-based (:76); top-k = the nn       */
/* smallest (dist, id) pairs in lexicographic order, as partial_sort on pair<float,int> gives.  */
/* ------------------------------------------------------------------------- */
void oracle_linscan_aq(float *dists, int *idx, const uint8_t *codes, const float *queries,
                       const float *codebooks, const float *dbnorms, int nqueries, int ncodes, int m,
                       int h, int d, int nn, int mode) {
#pragma omp parallel
  {
    float *t = (float *)malloc(sizeof(float) * (size_t)m * h);
    oracle_pair *heap = (oracle_pair *)malloc(sizeof(oracle_pair) * (size_t)nn);
#pragma omp for schedule(dynamic, 1)
    for (int qi = 0; qi < nqueries; qi++) {
      const float *q = queries + (size_t)qi * d;
      for (int j = 0; j < m * h; j++) {
        const float *c = codebooks + (size_t)j * d;
        float acc = 0.0f;
        if (mode == 1) {
          for (int k = 0; k < d; k++) {
            float two_q = 2 * q[k];
            float prod = two_q * c[k];
            acc = acc - prod;
          }
        } else {
          for (int k = 0; k < d; k++) {
            float diff = q[k] - c[k];
            float sq = diff * diff;
            acc = acc + sq;
          }
        }
        t[j] = acc;
      }
      int filled = 0;
      const uint8_t *pc = codes;
      for (int j = 0; j < ncodes; j++, pc += m) {
        float acc = 0.0f;
        for (int k = 0; k < m; k++) acc = acc + t[h * k + pc[k]];
        if (dbnorms) acc = acc + dbnorms[j];
        oracle_pair p = {acc, (uint32_t)(j + 1)};
        if (filled < nn) {
          heap[filled++] = p;
          if (filled == nn)
            for (int i = nn / 2 - 1; i >= 0; i--) heap_sift_down(heap, nn, i);
        } else if (pair_less(p, heap[0])) {
          heap[0] = p;
          heap_sift_down(heap, nn, 0);
        }
      }
      qsort(heap, (size_t)filled, sizeof(oracle_pair), pair_cmp_qsort);
      for (int j = 0; j < filled; j++) {
        dists[(size_t)qi * nn + j] = heap[j].dist;
        idx[(size_t)qi * nn + j] = (int)heap[j].id;
      }
    }
    free(t);
    free(heap);
  }
}

/* Full ADC distance row for one query (no top-K) -- used by tests to check     */
/* distances of arbitrary ids and by the merge tests.                            */
void oracle_adc_distances(float *out, const uint8_t *codes, const float *centers,
                          const float *query, int N, int m, int subdim) {
  float *lut = (float *)malloc(sizeof(float) * (size_t)m * 256);
  oracle_adc_lut(lut, centers, query, m, subdim);
  const uint8_t *pc = codes;
  for (int j = 0; j < N; j++, pc += m) {
    float acc = 0.0f;
    for (int k = 0; k < m; k++) acc = acc + lut[k * 256 + pc[k]];
    out[j] = acc;
  }
  free(lut);
}

/* ------------------------------------------------------------------------- */
/* src/OPQ.jl:26  RX = R' * X.                                                  */
/* C views: X [n][d], R [d][d] with Rc[i][k] = R[k,i] (Julia column-major),     */
/* RX [n][d];  RX[j][i] = sum_k Rc[i][k] * X[j][k] as a k-ordered fmaf chain     */
/* from +0 (bitwise what gfx950 f32 MFMA / v_fmac_f32 produce).                  */
/* ------------------------------------------------------------------------- */
void oracle_rotate_T(float *RX, const float *R, const float *X, int d, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; j++) {
    const float *x = X + (size_t)j * d;
    float *o = RX + (size_t)j * d;
    for (int i = 0; i < d; i++) {
      const float *r = R + (size_t)i * d;
      float acc = 0.0f;
      for (int k = 0; k < d; k++) acc = fmaf(r[k], x[k], acc);
      o[i] = acc;
    }
  }
}

/* ------------------------------------------------------------------------- */
/* src/PQ.jl:18-48  quantize_pq.  Per subspace i (src/utils.jl:179-203 split):  */
/*   Distances.pairwise(SqEuclidean(), C[i], Xs)  (src/PQ.jl:40):                */
/*       v_k = max( fl( fl(sa_k + sb) - 2*g_k ), 0 )                             */
/*       g_k = <c_k, x>, sa_k = |c_k|^2, sb = |x|^2                              */
/*   Clustering.update_assignments!  (src/PQ.jl:41): first-index argmin, '<'.    */
/* Canonical order: g, sa, sb are each a k-ordered fmaf chain from +0.           */
/* X [n][d]; C = concatenation over i of [h][sub_i] blocks (== cat(C...,dims=3)  */
/* when d % m == 0); codes [n][m] u8 ZERO-based (Julia adds 1, widens to Int16). */
/* costs (optional, may be NULL) [n][m] receives the winning v.                  */
/* ------------------------------------------------------------------------- */
void oracle_encode_pq(uint8_t *codes, float *costs, const float *X, const float *C,
                      int64_t n, int d, int m, int h) {
  int *off = (int *)malloc(sizeof(int) * (size_t)(m + 1));
  oracle_splitarray(d, m, off);
  /* transposed codebooks Ct[i][s][k] so the k loop vectorises; sa[i][k] */
  size_t *cbase = (size_t *)malloc(sizeof(size_t) * (size_t)(m + 1));
  cbase[0] = 0;
  for (int i = 0; i < m; i++)
    cbase[i + 1] = cbase[i] + (size_t)h * (size_t)(off[i + 1] - off[i]);
  float *Ct = (float *)malloc(sizeof(float) * cbase[m]);
  float *sa = (float *)malloc(sizeof(float) * (size_t)m * h);
  for (int i = 0; i < m; i++) {
    int sub = off[i + 1] - off[i];
    const float *Ci = C + cbase[i];
    for (int k = 0; k < h; k++) {
      float acc = 0.0f;
      for (int s = 0; s < sub; s++) {
        float c = Ci[(size_t)k * sub + s];
        Ct[cbase[i] + (size_t)s * h + k] = c;
        acc = fmaf(c, c, acc);
      }
      sa[(size_t)i * h + k] = acc;
    }
  }
#pragma omp parallel
  {
    float *g = (float *)malloc(sizeof(float) * (size_t)h);
#pragma omp for schedule(static)
    for (int64_t j = 0; j < n; j++) {
      const float *x = X + (size_t)j * d;
      for (int i = 0; i < m; i++) {
        int sub = off[i + 1] - off[i];
        const float *xs = x + off[i];
        const float *ct = Ct + cbase[i];
        const float *sai = sa + (size_t)i * h;
        float sb = 0.0f;
        for (int s = 0; s < sub; s++) sb = fmaf(xs[s], xs[s], sb);
        for (int k = 0; k < h; k++) g[k] = 0.0f;
        for (int s = 0; s < sub; s++) {
          float xv = xs[s];
          const float *row = ct + (size_t)s * h;
          for (int k = 0; k < h; k++) g[k] = fmaf(row[k], xv, g[k]);
        }
        int best = 0;
        float bestv = 0.0f;
        for (int k = 0; k < h; k++) {
          float t = sai[k] + sb;
          float v = t - 2.0f * g[k]; /* 2*g exact: same bits fused or not */
          v = v > 0.0f ? v : 0.0f;
          if (k == 0 || v < bestv) {
            best = k;
            bestv = v;
          }
        }
        codes[(size_t)j * m + i] = (uint8_t)best;
        if (costs) costs[(size_t)j * m + i] = bestv;
      }
    }
    free(g);
  }
  free(Ct);
  free(sa);
  free(cbase);
  free(off);
}

/* Every distance of quantize_pq's dmat (src/PQ.jl:40), not only the winners: U [n][m][h] receives the UNCLAMPED canonical
 * value  u_k = fl( fl(sa_k + sb) - 2 g_k )  (v_k = max(u_k, 0)), same chains as oracle_encode_pq.  Even splits only
 * (d % m == 0).  Test aid for the bf16 filter's margin (tests/test_gpu_encode_margin.py). */
void oracle_pq_distmat(float *U, const float *X, const float *C, int64_t n, int d, int m, int h) {
  const int sub = d / m;
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < n; j++) {
    for (int i = 0; i < m; i++) {
      const float *xs = X + (size_t)j * d + (size_t)i * sub;
      float sb = 0.0f;
      for (int s = 0; s < sub; s++) sb = fmaf(xs[s], xs[s], sb);
      for (int k = 0; k < h; k++) {
        const float *c = C + ((size_t)i * h + k) * sub;
        float sa = 0.0f, g = 0.0f;
        for (int s = 0; s < sub; s++) { sa = fmaf(c[s], c[s], sa); g = fmaf(c[s], xs[s], g); }
        const float t = sa + sb;
        U[((size_t)j * m + i) * h + k] = t - 2.0f * g;
      }
    }
  }
}

/* src/OPQ.jl:19-27  quantize_opq(X,R,C) = quantize_pq(R'X, C) */
void oracle_encode_opq(uint8_t *codes, const float *X, const float *R, const float *C,
                       int64_t n, int d, int m, int h) {
  float *RX = (float *)malloc(sizeof(float) * (size_t)n * d);
  oracle_rotate_T(RX, R, X, d, n);
  oracle_encode_pq(codes, NULL, RX, C, n, d, m, h);
  free(RX);
}

/* ------------------------------------------------------------------------- */
/* src/RVQ.jl:18-66  quantize_rvq(X, C): for i = 1..m                           */
/*   dmat = pairwise(SqEuclidean(), C[i], Xr)  (:37)  -- full-dimensional, so   */
/*   the same GEMM-trick distance + strict-'<' argmin as quantize_pq with ONE   */
/*   sub-quantizer of width d (same canonical fmaf chains, PARITY UNPINNED like */
/*   oracle_encode_pq);  Xr .-= C[i][:, B[i]]  (:56), a plain f32 subtraction.  */
/* C [m][h][d]; codes [n][m] zero-based; counts [m][h] (optional) = :43-47      */
/* `counts`; Xr_out [n][d] (optional) = the final residual.                     */
/* ------------------------------------------------------------------------- */
void oracle_encode_rvq(uint8_t *codes, uint32_t *counts, float *Xr_out, const float *X, const float *C,
                       int64_t n, int d, int m, int h) {
  float *Xr = (float *)malloc(sizeof(float) * (size_t)n * d);
  uint8_t *stage = (uint8_t *)malloc((size_t)n);
  memcpy(Xr, X, sizeof(float) * (size_t)n * d);
  if (counts) memset(counts, 0, sizeof(uint32_t) * (size_t)m * h);
  for (int i = 0; i < m; i++) {
    const float *Ci = C + (size_t)i * h * d;
    oracle_encode_pq(stage, NULL, Xr, Ci, n, d, 1, h);
    for (int64_t j = 0; j < n; j++) {
      const float *c = Ci + (size_t)stage[j] * d;
      float *x = Xr + (size_t)j * d;
      for (int s = 0; s < d; s++) x[s] = x[s] - c[s];
      codes[(size_t)j * m + i] = stage[j];
      if (counts) counts[(size_t)i * h + stage[j]]++;
    }
  }
  if (Xr_out) memcpy(Xr_out, Xr, sizeof(float) * (size_t)n * d);
  free(stage);
  free(Xr);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
