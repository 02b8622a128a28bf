// ann_recall_kernels.h -- exact-rank scoring of returned neighbours, the GPU counterpart of the
// reference's recall driver /root/reference/test_correctness.c:169-262.
//
// The reference sorts ALL n distances per query (qsort) and looks up the rank of every guessed neighbour.  The
// rank is simply the number of points strictly closer than the guess, so no sort is needed: one brute-force pass
// computes every (query, point) distance -- with the same exact tree as the query path -- and, for the rare
// points that beat a query's farthest guess, bumps a small per-query histogram.  Ties count as "not closer"
// (the reference's qsort places them arbitrarily).
#pragma once
#include "ann_query_kernels.h"

// guess_dist[q][j] = distance of query q to its j-th guess (+inf for ids >= n); gmax[q] = the largest finite-or-inf of them
template <int D>
__global__ __launch_bounds__(256) void recall_guess_dist_kernel(const FT *__restrict__ points, u32 n, int d, int Q, int k,
                                                                const FT *__restrict__ y, const size_t *__restrict__ guess,
                                                                FT *__restrict__ gdist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int q = blockIdx.x * wpb + w;
  if (q >= Q) return;
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    const int p = lane % L::LPR, g = lane / L::LPR;
    VT a[L::C];
    const VT *yp = reinterpret_cast<const VT *>(y + (size_t)q * D) + p;
#pragma unroll
    for (int c = 0; c < L::C; c++) a[c] = yp[c * L::LPR];
    for (int j0 = 0; j0 < k; j0 += L::RPW) {
      const int j = j0 + g;
      const size_t id = j < k ? guess[(size_t)q * k + j] : 0;
      const bool ok = j < k && id < n;
      const VT *rp = reinterpret_cast<const VT *>(points + (ok ? id : 0) * D) + p;
      VT b[L::C];
#pragma unroll
      for (int c = 0; c < L::C; c++) b[c] = rp[c * L::LPR];
      const FT dist = row_reduce<D, ROW_SQDIFF>(a, b);
      if (j < k && p == 0) gdist[(size_t)q * k + j] = ok ? dist : ft_inf();
    }
  } else {
    FT *yq = reinterpret_cast<FT *>(smem) + (size_t)w * 2 * d, *m = yq + d;
    for (int z = lane; z < d; z += ANN_WAVE) yq[z] = y[(size_t)q * d + z];
    wave_lds_sync();
    for (int j = 0; j < k; j++) {
      const size_t id = guess[(size_t)q * k + j];
      const bool ok = id < n;
      const FT dist = row_reduce_generic<ROW_SQDIFF>(d, yq, points + (ok ? id : 0) * (size_t)d, m);
      if (lane == 0) gdist[(size_t)q * k + j] = ok ? dist : ft_inf();
    }
  }
}

#define ANN_RECALL_TILE 64  // point rows per workgroup tile

// hist[q][c] += 1 for every point closer to query q than its farthest guess, c = number of guesses at least as
// close as the point... precisely c = #{j : gdist[q][j] <= dist}; then rank[q][j] = sum_{c <= j'} ... (host side).
// self != 0: point q is skipped for query q (scoring precomp's graph).
template <int D>
__global__ __launch_bounds__(256) void recall_scan_kernel(const FT *__restrict__ points, u32 n, int d, int Q, int k,
                                                          const FT *__restrict__ y, const FT *__restrict__ gdist,
                                                          int self, unsigned long long *__restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const u32 row0 = blockIdx.x * ANN_RECALL_TILE;
  const u32 rows = min((u32)ANN_RECALL_TILE, n - row0);
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    VT *tile = reinterpret_cast<VT *>(smem);  // [TILE][D/VEC]
    const VT *src = reinterpret_cast<const VT *>(points + (size_t)row0 * D);
    for (u32 i = threadIdx.x; i < rows * (D / ANN_VEC); i += blockDim.x) tile[i] = src[i];
    __syncthreads();
    const int p = lane % L::LPR, g = lane / L::LPR;
    for (int q = blockIdx.y * wpb + w; q < Q; q += gridDim.y * wpb) {
      VT a[L::C];
      const VT *yp = reinterpret_cast<const VT *>(y + (size_t)q * D) + p;
#pragma unroll
      for (int c = 0; c < L::C; c++) a[c] = yp[c * L::LPR];
      const FT *gq = gdist + (size_t)q * k;
      FT gfar = gq[0];
      for (int j = 1; j < k; j++) gfar = gq[j] > gfar ? gq[j] : gfar;  // guesses need not be sorted
      for (u32 r0 = 0; r0 < rows; r0 += L::RPW) {
        const u32 r = r0 + g;
        const bool act = r < rows && !(self && row0 + r == (u32)q);
        const VT *rp = tile + (size_t)(r < rows ? r : 0) * (D / ANN_VEC) + p;
        VT b[L::C];
#pragma unroll
        for (int c = 0; c < L::C; c++) b[c] = rp[c * L::LPR];
        const FT dist = row_reduce<D, ROW_SQDIFF>(a, b);
        if (act && p == 0 && dist < gfar) {  // rare
          int c = 0;
          for (int j = 0; j < k; j++) c += gq[j] <= dist;
          atomicAdd(&hist[(size_t)q * (k + 1) + c], 1ull);
        }
      }
    }
  } else {
    FT *yq = reinterpret_cast<FT *>(smem) + (size_t)w * 2 * d, *m = yq + d;
    for (int q = blockIdx.y * wpb + w; q < Q; q += gridDim.y * wpb) {
      for (int z = lane; z < d; z += ANN_WAVE) yq[z] = y[(size_t)q * d + z];
      wave_lds_sync();
      const FT *gq = gdist + (size_t)q * k;
      FT gfar = gq[0];
      for (int j = 1; j < k; j++) gfar = gq[j] > gfar ? gq[j] : gfar;
      for (u32 r = 0; r < rows; r++) {
        if (self && row0 + r == (u32)q) continue;
        const FT dist = row_reduce_generic<ROW_SQDIFF>(d, yq, points + (size_t)(row0 + r) * d, m);
        if (lane == 0 && dist < gfar) {
          int c = 0;
          for (int j = 0; j < k; j++) c += gq[j] <= dist;
          atomicAdd(&hist[(size_t)q * (k + 1) + c], 1ull);
        }
      }
      wave_lds_sync();
    }
  }
}
