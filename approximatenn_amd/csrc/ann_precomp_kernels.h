// ann_precomp_kernels.h -- HIP kernels of precomp() that are not shared with the query path (gfx950).
//
//   rows_add0 / rows_addn / means_finish / centre   column means by the reference's row tree and
//                                                    centring                  (alg.c:122-128,367-369)
//   hash_rows       rotate -> permute/pad -> FWHT -> rotate -> select -> sign hash, one wave per row,
//                   the row living in LDS                                     (run_initial, alg.c:154-183)
//   bases_rows      the inverse chain applied to the unit vectors             (save_vecs, alg.c:189-217)
//   bucket_*        counting sort of the points into the [2^ds][pm] table      (second_half, alg.c:252-267)
//
// The distance/top-k part of precomp (second_half's compdists + sort_and_uniq, det_results) reuses
// stage1_select / finalize1 / row_dists / exact_select from ann_query_kernels.h.
#pragma once
#include "ann_device.h"

// ---------------------------------------------------------------------------------- column means
// add_rows_step_0, compute.cl:15-21: r[x][y] = (a[x][y] + a[x+n/2][y]) + g, g = a[n-1][y] if n odd, x == 0.
__global__ void rows_add0_kernel(size_t n, size_t d, const FT *__restrict__ a, FT *__restrict__ r) {
  const size_t half = n / 2, total = half * d;
  const FT zero = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    FT g = ((n & 1) && e < d) ? a[(n - 1) * d + e] : zero;
    r[e] = a[e] + a[e + total] + g;
  }
}
// add_rows_step_n, compute.cl:26-31: r[x][y] = r[x][y] + (r[x+m/2][y] + g), x < m/2.
__global__ void rows_addn_kernel(size_t m, size_t d, FT *r) {
  const size_t half = m / 2, total = half * d;
  const FT zero = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    FT g = ((m & 1) && e < d) ? r[(m - 1) * d + e] : zero;
    r[e] = r[e] + (r[e + total] + g);
  }
}
// divide_by_length, compute.cl:36-39 (size_t length converted to ftype).
__global__ void means_finish_kernel(size_t n, size_t d, const FT *__restrict__ r, FT *__restrict__ means) {
  size_t y = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y < d) means[y] = r[y] / (FT)n;
}
// subtract_off, compute.cl:44-49.
__global__ void centre_kernel(size_t n, size_t d, const FT *__restrict__ pts, const FT *__restrict__ means,
                              FT *__restrict__ out) {
  const size_t total = n * d;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x)
    out[e] = pts[e] - means[e % d];
}

// --------------------------------------------------------------------------- random transform
struct XformDev {     // one try's transform (make_ortho_info, alg.c:59-74), device arrays
  const u32 *b_i, *b_j;   // [rots_b][rlb] coordinate pairs of the pre-Walsh Givens rotations
  const FT *b_c, *b_s;    // cos/sin of their angles, rounded from double libm on the host (Q11)
  const u32 *a_i, *a_j;   // [rots_a][rla] post-Walsh
  const FT *a_c, *a_s;
  const u32 *perm_b, *perm_ai;  // [d_max]
  int rots_b, rlb, rots_a, rla;
  int d, d_max, ds, l;    // l = log2(d_max)
};

// apply_rotation (compute.cl:55-68) for one row in LDS; swap = the inverse (i/j exchanged).
__device__ inline void givens_lds(FT *row, const u32 *ci, const u32 *cj, const FT *cs, const FT *sn,
                                  int len, bool swap) {
  for (int y = lane_id(); y < len; y += ANN_WAVE) {
    u32 k = swap ? cj[y] : ci[y], l = swap ? ci[y] : cj[y];
    FT c = cs[y], s = sn[y];
    FT vk = row[k], vl = row[l];
    FT q = vk * c - vl * s;
    FT r = vk * s + vl * c;
    row[k] = q;
    row[l] = r;
  }
  wave_lds_sync();
}

// walsh (alg.c:112-120) + apply_walsh_step (compute.cl:101-122) for one row of 2^l entries in LDS.
__device__ inline void fwht_lds(FT *a, int l) {
  const int half = 1 << (l - 1);
  for (int step = 0; step < l; step++) {
    const FT div = (FT)(step % 2 + 1);
    const bool scale = step == 0 && (l & 1);
#ifdef USE_FLOAT
    const FT rsr = (FT)(1.0 / __builtin_sqrt(2.0));
#else
    const FT rsr = 1.0 / __builtin_sqrt(2.0);
#endif
    for (int b = lane_id(); b < half; b += ANN_WAVE) {
      int hi = (b >> step) << step, lo = b ^ hi;
      int ia = hi << 1 | lo, ib = ia | 1 << step;
      FT x = a[ia], y = a[ib];
      FT p = (x + y) / div, q = (x - y) / div;
      if (scale) p = p * rsr, q = q * rsr;
      a[ia] = p;
      a[ib] = q;
    }
    wave_lds_sync();
  }
}

// run_initial: code[x] = sign hash of the row's ds low coordinates.  Every wave takes ANN_HASH_ROWS consecutive rows
// and walks them through the chain TOGETHER: each step (a rotation, the permutation, one FWHT level ...) is applied to
// all of its rows before the wave-level LDS fence, so the ~20 dependent LDS round trips of the chain are paid once per
// ANN_HASH_ROWS rows instead of once per row (one row per wave: 8.1 ms per try at cfg3, LDS-latency-bound).
#ifndef ANN_HASH_ROWS
#define ANN_HASH_ROWS 4
#endif
__device__ __forceinline__ void givens_rows(FT *base, size_t stride, int rows, const u32 *ci, const u32 *cj, const FT *cs,
                                            const FT *sn, int len) {
  for (int e = lane_id(); e < len * rows; e += ANN_WAVE) {
    const int r = e / len, y = e - r * len;
    FT *row = base + (size_t)r * stride;
    const u32 k = ci[y], l = cj[y];
    const FT c = cs[y], s = sn[y];
    const FT vk = row[k], vl = row[l];
    const FT q = vk * c - vl * s;
    const FT rr = vk * s + vl * c;
    row[k] = q;
    row[l] = rr;
  }
  wave_lds_sync();
}

__global__ __launch_bounds__(256) void hash_rows_kernel(XformDev X, size_t n, const FT *__restrict__ centred,
                                                        u32 *__restrict__ codes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = ANN_HASH_ROWS;
  const int lane = lane_id(), w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const size_t x0 = ((size_t)blockIdx.x * wpb + w) * R;
  if (x0 >= n) return;  // whole wave leaves together
  const int rows = (int)min((size_t)R, n - x0);
  const size_t stride = (size_t)X.d + X.d_max + X.ds;  // per row: work[d] | wide[d_max] | low[ds]
  FT *base = reinterpret_cast<FT *>(smem) + (size_t)w * R * stride;
  for (int e = lane; e < rows * X.d; e += ANN_WAVE) {
    const int r = e / X.d, z = e - r * X.d;
    base[(size_t)r * stride + z] = centred[(x0 + r) * X.d + z];
  }
  wave_lds_sync();
  for (int t = 0; t < X.rots_b; t++)
    givens_rows(base, stride, rows, X.b_i + t * X.rlb, X.b_j + t * X.rlb, X.b_c + t * X.rlb, X.b_s + t * X.rlb, X.rlb);
  for (int e = lane; e < rows * X.d_max; e += ANN_WAVE) {  // apply_permutation, compute.cl:77-85
    const int r = e / X.d_max, y = e - r * X.d_max;
    const u32 src = X.perm_b[y];
    FT *row = base + (size_t)r * stride;
    row[X.d + y] = src < (u32)X.d ? row[src] : (FT)0;
  }
  wave_lds_sync();
  {  // walsh (alg.c:112-120) + apply_walsh_step (compute.cl:101-122) on wide[]
    const int half = 1 << (X.l - 1);
#ifdef USE_FLOAT
    const FT rsr = (FT)(1.0 / __builtin_sqrt(2.0));
#else
    const FT rsr = 1.0 / __builtin_sqrt(2.0);
#endif
    for (int step = 0; step < X.l; step++) {
      const FT div = (FT)(step % 2 + 1);
      const bool scale = step == 0 && (X.l & 1);
      for (int e = lane; e < rows * half; e += ANN_WAVE) {
        const int r = e / half, b = e - r * half;
        FT *a = base + (size_t)r * stride + X.d;
        const int hi = (b >> step) << step, lo = b ^ hi;
        const int ia = hi << 1 | lo, ib = ia | 1 << step;
        const FT xx = a[ia], yy = a[ib];
        FT p = (xx + yy) / div, q = (xx - yy) / div;
        if (scale) p = p * rsr, q = q * rsr;
        a[ia] = p;
        a[ib] = q;
      }
      wave_lds_sync();
    }
  }
  for (int t = 0; t < X.rots_a; t++)
    givens_rows(base + X.d, stride, rows, X.a_i + t * X.rla, X.a_j + t * X.rla, X.a_c + t * X.rla, X.a_s + t * X.rla, X.rla);
  for (int e = lane; e < rows * X.d_max; e += ANN_WAVE) {  // apply_perm_inv, compute.cl:88-96
    const int r = e / X.d_max, y = e - r * X.d_max;
    const u32 dst = X.perm_ai[y];
    FT *row = base + (size_t)r * stride;
    if (dst < (u32)X.ds) row[X.d + X.d_max + dst] = row[X.d + y];
  }
  wave_lds_sync();
  // compute_signs, compute.cl:223-231: coordinate 0 is the most significant bit (ds <= 32 here)
  for (int r = 0; r < rows; r++) {
    const FT *low = base + (size_t)r * stride + X.d + X.d_max;
    const bool neg = lane < X.ds && (ft_bits(low[lane < X.ds ? lane : 0]) >> (sizeof(FT) * 8 - 1));
    const u64 m = __ballot(neg);
    const u32 code = X.ds ? (__brev((u32)m) >> (32 - X.ds)) : 0u;
    if (lane == 0) codes[x0 + r] = code;
  }
}

// save_vecs for one basis row per wave: bases[row][0..d).
__global__ __launch_bounds__(256) void bases_rows_kernel(XformDev X, FT *__restrict__ bases) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  const int row = blockIdx.x * wpb + w;
  if (row >= X.ds) return;
  FT *wide = reinterpret_cast<FT *>(smem) + (size_t)w * (X.d + X.d_max);
  FT *out = wide + X.d_max;
  for (int y = lane; y < X.d_max; y += ANN_WAVE) {
    u32 src = X.perm_ai[y];
    wide[y] = src < (u32)X.ds ? (FT)(src == (u32)row) : (FT)0;
  }
  wave_lds_sync();
  for (int r = X.rots_a - 1; r >= 0; r--)
    givens_lds(wide, X.a_i + r * X.rla, X.a_j + r * X.rla, X.a_c + r * X.rla, X.a_s + r * X.rla, X.rla, true);
  fwht_lds(wide, X.l);
  for (int y = lane; y < X.d_max; y += ANN_WAVE) {
    u32 dst = X.perm_b[y];
    if (dst < (u32)X.d) out[dst] = wide[y];
  }
  wave_lds_sync();
  for (int r = X.rots_b - 1; r >= 0; r--)
    givens_lds(out, X.b_i + r * X.rlb, X.b_j + r * X.rlb, X.b_c + r * X.rlb, X.b_s + r * X.rlb, X.rlb, true);
  for (int z = lane; z < X.d; z += ANN_WAVE) bases[(size_t)row * X.d + z] = out[z];
}

// ------------------------------------------------------------------------------- bucket tables
__global__ void fill_u32_kernel(size_t count, u32 value, u32 *out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (size_t)gridDim.x * blockDim.x)
    out[e] = value;
}
__global__ void fill_ft_kernel(size_t count, FT value, FT *out) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (size_t)gridDim.x * blockDim.x)
    out[e] = value;
}
__global__ void bucket_count_kernel(size_t n, const u32 *__restrict__ codes, u32 *cnt) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
    atomicAdd(&cnt[codes[j]], 1u);
}
__global__ void max_u32_kernel(size_t count, const u32 *__restrict__ v, u32 *out) {
  u32 m = 0;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (size_t)gridDim.x * blockDim.x)
    m = max(m, v[e]);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m = max(m, (u32)__shfl_xor(m, s));
  if (lane_id() == 0) atomicMax(out, m);
}
// scatter in arrival order; bucket_order_kernel then puts each bucket in the reference's order
__global__ void bucket_place_kernel(size_t n, u32 pm, const u32 *__restrict__ codes, u32 *cursor, u32 *table) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
    u32 c = codes[j];
    u32 pos = atomicAdd(&cursor[c], 1u);
    table[(size_t)c * pm + pos] = (u32)j;
  }
}
// ids descending inside each bucket (alg.c:265-266 fills ascending j from the back, Q8); padding n stays behind
__global__ void bucket_order_kernel(size_t nbuckets, u32 pm, const u32 *__restrict__ cnt, u32 *table) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  u32 *row = table + b * pm;
  const u32 c = cnt[b];
  for (u32 i = 1; i < c; i++) {
    u32 v = row[i];
    u32 j = i;
    while (j > 0 && row[j - 1] < v) {
      row[j] = row[j - 1];
      j--;
    }
    row[j] = v;
  }
}
