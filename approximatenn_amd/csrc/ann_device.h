// ann_device.h -- device-side building blocks shared by the query and precomp kernels (gfx950 only).
//
// Everything here is exact IEEE arithmetic in the reference's operation order; the library is built
// with -ffp-contract=off so that no multiply-add is fused.  One library is built per
// precision: FT is float under -DUSE_FLOAT, double otherwise (/root/reference/ftype.h:3-9).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef uint64_t u64;

#ifdef USE_FLOAT
typedef float FT;
typedef u32 UB;      // bit pattern of an FT
typedef float4 VT;   // one 16-byte chunk
#define ANN_VEC 4
#else
typedef double FT;
typedef u64 UB;
typedef double2 VT;
#define ANN_VEC 2
#endif

#define ANN_WAVE 64
#define ANN_ID_NONE 0xFFFFFFFFu

__device__ __forceinline__ UB ft_bits(FT x) { return __builtin_bit_cast(UB, x); }
__device__ __forceinline__ FT ft_from_bits(UB b) { return __builtin_bit_cast(FT, b); }
__device__ __forceinline__ FT ft_inf() {
#ifdef USE_FLOAT
  return __builtin_bit_cast(float, 0x7F800000u);
#else
  return __builtin_bit_cast(double, 0x7FF0000000000000ull);
#endif
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// number of set bits of a 64-bit ballot below this lane
__device__ __forceinline__ u32 mask_rank(u64 m) {
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
}
// inclusive prefix sum over the lanes of a wave
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
#pragma unroll
  for (int o = 1; o < ANN_WAVE; o <<= 1) {
    u32 t = __shfl_up(v, o);
    if (lane_id() >= o) v += t;
  }
  return v;
}
// LDS traffic between lanes of ONE wave: DS ops of a wave execute in order, so a wave-scope
// fence (compiler + memory ordering) is all that is needed.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Value of lane (i + N) of the same 16-lane row, N in 1..15, by DPP (row_shl:N) -- no LDS crossbar traffic.
// Lanes whose source falls outside the row read 0; callers only use lanes whose source is inside.
template <int N>
__device__ __forceinline__ float lane_up(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xF, 0xF, true));
}
template <int N>
__device__ __forceinline__ double lane_up(double v) {
  const u64 b = __builtin_bit_cast(u64, v);
  const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x100 + N, 0xF, 0xF, true);
  const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x100 + N, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((u64)hi << 32) | lo);
}
// partner value for the tree level that pairs lane position p with p + M (M a power of two)
template <int M>
__device__ __forceinline__ FT tree_partner(FT v) {
  if constexpr (M <= 8)
    return lane_up<M>(v);  // groups of LPR <= 16 lanes are aligned inside a 16-lane DPP row
  else
    return __shfl_xor(v, M);
}

// ------------------------------------------------------------------ candidate keys
// A candidate is (squared distance, point id).  Distances are >= +0, so their bit patterns order like
// the values; (dist_bits, id) ordered lexicographically is the total order used by the selection path.
#ifdef USE_FLOAT
typedef u64 Key;
__device__ __forceinline__ Key key_make(FT d, u32 id) { return ((u64)ft_bits(d) << 32) | id; }
__device__ __forceinline__ FT key_dist(Key k) { return ft_from_bits((u32)(k >> 32)); }
__device__ __forceinline__ u32 key_id(Key k) { return (u32)k; }
__device__ __forceinline__ bool key_less(Key a, Key b) { return a < b; }
__device__ __forceinline__ bool key_eq(Key a, Key b) { return a == b; }
__device__ __forceinline__ Key key_max() { return ~0ull; }
__device__ __forceinline__ Key key_shfl_xor(Key k, int m) {
  u32 lo = __shfl_xor((u32)k, m), hi = __shfl_xor((u32)(k >> 32), m);
  return ((u64)hi << 32) | lo;
}
#else
struct Key {
  u64 d;
  u64 i;
};
__device__ __forceinline__ Key key_make(FT d, u32 id) { return Key{ft_bits(d), id}; }
__device__ __forceinline__ FT key_dist(Key k) { return ft_from_bits(k.d); }
__device__ __forceinline__ u32 key_id(Key k) { return (u32)k.i; }
__device__ __forceinline__ bool key_less(Key a, Key b) { return a.d < b.d || (a.d == b.d && a.i < b.i); }
__device__ __forceinline__ bool key_eq(Key a, Key b) { return a.d == b.d && a.i == b.i; }
__device__ __forceinline__ Key key_max() { return Key{~0ull, ~0ull}; }
__device__ __forceinline__ Key key_shfl_xor(Key k, int m) {
  Key o;
  o.d = ((u64)__shfl_xor((u32)(k.d >> 32), m) << 32) | __shfl_xor((u32)k.d, m);
  o.i = __shfl_xor((u32)k.i, m);
  return o;
}
#endif

__device__ __forceinline__ Key wave_min_key(Key k) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    Key o = key_shfl_xor(k, m);
    if (key_less(o, k)) k = o;
  }
  return k;
}

// Select the up-to-`want` smallest DISTINCT keys of buf[0..cnt) (LDS, one wave), ascending, into
// out[0..m) (LDS, disjoint from buf) and return m.  Identical keys (= the same point reached through
// several buckets) collapse because every pass only looks at keys strictly above the previous pick.
__device__ inline int wave_select_smallest(const Key *buf, int cnt, int want, Key *out) {
  const int lane = lane_id();
  Key prev = key_max();
  int m = 0;
  for (; m < want; m++) {
    Key best = key_max();
    for (int i = lane; i < cnt; i += ANN_WAVE) {
      Key c = buf[i];
      if ((m == 0 || key_less(prev, c)) && key_less(c, best)) best = c;
    }
    best = wave_min_key(best);
    if (key_eq(best, key_max())) break;
    if (lane == 0) out[m] = best;
    prev = best;
  }
  wave_lds_sync();
  return m;
}

// ------------------------------------------------------------------ wave-resident top-K
// An ascending list of the (up to 64) smallest DISTINCT keys seen so far lives in ONE register per lane: lane i holds
// the i-th smallest, key_max() where there is none yet.  Inserting a wave-uniform key is a compare, a one-lane wave
// shift (DPP wave_shr:1 -- no LDS crossbar) and two selects; the K-th smallest (the admission threshold) is a
// v_readlane.  Replaces an LDS candidate buffer + periodic K-pass selection where K <= 64.
__device__ __forceinline__ u32 lane_prev_u32(u32 v) {  // value of lane i-1; lane 0 gets 0
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
#ifdef USE_FLOAT
__device__ __forceinline__ Key key_lane_prev(Key k) {
  return ((u64)lane_prev_u32((u32)(k >> 32)) << 32) | lane_prev_u32((u32)k);
}
__device__ __forceinline__ Key key_readlane(Key k, int src) {  // src wave-uniform
  return ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(k >> 32), src) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)k, src);
}
#else
__device__ __forceinline__ Key key_lane_prev(Key k) {
  Key o;
  o.d = ((u64)lane_prev_u32((u32)(k.d >> 32)) << 32) | lane_prev_u32((u32)k.d);
  o.i = lane_prev_u32((u32)k.i);
  return o;
}
__device__ __forceinline__ Key key_readlane(Key k, int src) {
  Key o;
  o.d = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(k.d >> 32), src) << 32) | (u32)__builtin_amdgcn_readlane((int)(u32)k.d, src);
  o.i = (u32)__builtin_amdgcn_readlane((int)(u32)k.i, src);
  return o;
}
#endif
__device__ __forceinline__ void wave_topk_insert(Key &mine, Key k) {  // k wave-uniform
  if (__ballot(key_eq(mine, k))) return;  // already listed (the same point reached through another bucket)
  const Key up = key_lane_prev(mine);     // lane 0 sees the zero key
  if (key_less(k, mine)) mine = key_less(k, up) ? up : k;
}
// Offer the keys of the lanes in `mm` (a ballot) to the list; tau = current K1-th smallest (admission threshold).
__device__ __forceinline__ void wave_topk_offer(Key &mine, Key &tau, Key key, u64 mm, int K1) {
  while (mm) {
    const int src = __builtin_ctzll(mm);
    mm &= mm - 1;
    const Key k = key_readlane(key, src);
    if (key_less(k, tau)) {
      wave_topk_insert(mine, k);
      tau = key_readlane(mine, K1 - 1);
    }
  }
}

// Two registers per lane: lanes 0..63 of `lo` hold ranks 0..63, of `hi` ranks 64..127 (K <= 128).  A key that enters
// `lo` pushes lo's last entry (lane 63) to the front of `hi`.
__device__ __forceinline__ void wave_topk_shift_in(Key &mine, Key k) {  // k wave-uniform, known not to be listed
  const Key up = key_lane_prev(mine);
  if (key_less(k, mine)) mine = key_less(k, up) ? up : k;
}
__device__ __forceinline__ void wave_topk_insert2(Key &lo, Key &hi, Key k) {  // k wave-uniform
  if (__ballot(key_eq(lo, k) || key_eq(hi, k))) return;
  const Key last = key_readlane(lo, ANN_WAVE - 1);  // key_max() while lo is not full: then k < last
  if (key_less(k, last)) {
    wave_topk_shift_in(lo, k);
    wave_topk_shift_in(hi, last);  // smaller than everything in hi; a no-op for key_max()
  } else {
    wave_topk_shift_in(hi, k);
  }
}
__device__ __forceinline__ Key wave_topk_kth2(Key lo, Key hi, int K1) {
  return K1 <= ANN_WAVE ? key_readlane(lo, K1 - 1) : key_readlane(hi, K1 - 1 - ANN_WAVE);
}
__device__ __forceinline__ void wave_topk_offer2(Key &lo, Key &hi, Key &tau, Key key, u64 mm, int K1) {
  while (mm) {
    const int src = __builtin_ctzll(mm);
    mm &= mm - 1;
    const Key k = key_readlane(key, src);
    if (key_less(k, tau)) {
      wave_topk_insert2(lo, hi, k);
      tau = wave_topk_kth2(lo, hi, K1);
    }
  }
}

// ------------------------------------------------------------------ row layout for power-of-two d
// A row of D elements is read by LPR lanes as C chunks of 16 bytes per lane; lane position p reads
// chunks p, p+LPR, ... so that every load instruction covers whole contiguous 128-byte pieces.
// Element index z = VEC*(p + LPR*c) + j.  The pairwise tree of compute.cl:160-167 (Q4) pairs z with
// z + s/2 for s = D, D/2, ..., 2: first chunk c with c + C/2 (in-lane), then lane p with p ^ (LPR/2)
// (cross-lane), finally j with j + VEC/2 (in-lane).  fp add commutes, so the xor butterfly produces in
// lane position 0 exactly the value the serial in-place tree leaves in m[0].
template <int D>
struct RowLay {
  static constexpr int CHUNKS = D / ANN_VEC;
  static constexpr int LPR0 = CHUNKS < 8 ? CHUNKS : 8;
  static constexpr int LPR = (CHUNKS / 4 > LPR0) ? (CHUNKS / 4 > 64 ? 64 : CHUNKS / 4) : LPR0;
  static constexpr int C = CHUNKS / LPR;    // chunks per lane
  static constexpr int RPW = ANN_WAVE / LPR;  // rows per wave pass
  static_assert(D >= 16 && (D & (D - 1)) == 0, "fast layout needs a power of two >= 16");
  static_assert(C >= 1 && C <= 4 && C * LPR * ANN_VEC == D, "unsupported row length");
};

enum { ROW_SQDIFF = 0, ROW_PRODUCT = 1 };

// One row against the lane's slice `a` of the left operand.  MODE ROW_SQDIFF: sum (a-b)^2 (compute.cl:
// 147-149); ROW_PRODUCT: sum a*b (compute.cl:268-275), with the reference's "+ 0" kept in every tree
// node because it turns -0 into +0 and the hash reads the raw sign bit (compute.cl:165-166,229).
// The result is valid in lane position 0 of each LPR-lane group ONLY (the cross-lane levels pull the partner's
// value downwards with DPP row shifts; the other lanes end up with partial sums nobody reads).
template <int D, int MODE>
__device__ __forceinline__ FT row_reduce(const VT (&a)[RowLay<D>::C], const VT (&b)[RowLay<D>::C]) {
  typedef RowLay<D> L;
  FT e[L::C][ANN_VEC];
#pragma unroll
  for (int c = 0; c < L::C; c++) {
    const FT *pa = reinterpret_cast<const FT *>(&a[c]);
    const FT *pb = reinterpret_cast<const FT *>(&b[c]);
#pragma unroll
    for (int j = 0; j < ANN_VEC; j++) {
      if (MODE == ROW_SQDIFF) {
        FT df = pa[j] - pb[j];
        e[c][j] = df * df;
      } else {
        e[c][j] = pa[j] * pb[j];
      }
    }
  }
  const FT zero = 0;
#pragma unroll
  for (int h = L::C / 2; h >= 1; h >>= 1)
#pragma unroll
    for (int c = 0; c < h; c++)
#pragma unroll
      for (int j = 0; j < ANN_VEC; j++)
        e[c][j] = (MODE == ROW_PRODUCT) ? e[c][j] + (e[c + h][j] + zero) : e[c][j] + e[c + h][j];
#define ANN_TREE_LEVEL(M)                                                                  \
  if constexpr (L::LPR / 2 >= (M)) {                                                       \
    _Pragma("unroll") for (int j = 0; j < ANN_VEC; j++) {                                   \
      FT o = tree_partner<(M)>(e[0][j]);                                                   \
      e[0][j] = (MODE == ROW_PRODUCT) ? e[0][j] + (o + zero) : e[0][j] + o;                 \
    }                                                                                      \
  }
  ANN_TREE_LEVEL(32) ANN_TREE_LEVEL(16) ANN_TREE_LEVEL(8) ANN_TREE_LEVEL(4) ANN_TREE_LEVEL(2) ANN_TREE_LEVEL(1)
#undef ANN_TREE_LEVEL
#pragma unroll
  for (int h = ANN_VEC / 2; h >= 1; h >>= 1)
#pragma unroll
    for (int j = 0; j < h; j++)
      e[0][j] = (MODE == ROW_PRODUCT) ? e[0][j] + (e[0][j + h] + zero) : e[0][j] + e[0][j + h];
  return e[0][0];
}

// ------------------------------------------------------------------ row layout for d = oc * 2^a chunks
// Any d that is a multiple of the 16-byte chunk (VEC elements): write d/VEC = oc * C with C = 2^a (a as large as
// possible, C <= 8) and oc odd.  `oc` lanes share a row; lane position p holds chunks p, p+oc, ..., p+(C-1)oc, i.e.
// elements z = VEC*(p + oc*c) + j.  The top a levels of the tree pair z with z + s/2 where s/2 is a multiple of
// oc*VEC -> chunk c with c + C/2, ... : in-lane, exactly as in RowLay.  What remains are m = oc*VEC partial sums,
// element VEC*p + j in lane p: the literal tree on those (odd levels included) runs across the lanes with
// shuffles whose lane offset and source register are wave-uniform per (level, j).  Result valid in lane position 0.
// Kernels encode this layout as a NEGATIVE template argument D = -C; oc comes at run time (P.d / (VEC*C)).
template <int D, bool POW2 = (D > 0)>
struct RowChunks;
template <int D>
struct RowChunks<D, true> {
  static constexpr int C = RowLay<D>::C;
};
// D < 0 encodes this layout: -D = 16*OC + C.  OC = 0: `oc` comes at run time (any odd-factor count up to 64, tail by
// shuffles).  OC in {3, 5}: `oc` is static, the groups of OC lanes sit inside the 16-lane DPP rows (16/OC groups per
// row, the remaining lanes idle) and the whole tail is unrolled at compile time with DPP row shifts -- no LDS crossbar,
// no run-time control flow (the reference drivers' default d = 80 is OC = 5: C = 4 float chunks / 8 double chunks).
// D = ANN_D_UNALIGNED: d is NOT a multiple of the 16-byte chunk (d = 33, 50, 77 ...): ceil(d/VEC) lanes share a row, each
// loads its VEC elements one by one (rows are not 16-byte aligned), zeros beyond d, and the literal tree starts at d.
#define ANN_D_UNALIGNED (-241)
// D = ANN_D_FOLD2 / ANN_D_FOLD3: as ANN_D_UNALIGNED everywhere (element-wise loads work for aligned rows too), except in
// the selection gathers, which fold the first 2 / 3 tree levels into a lane (ann_query_kernels.h: gather_fold).  Own
// codes, not a run-time branch: the folded code's registers would cost the other row lengths their occupancy.
#define ANN_D_FOLD2 (-243)
#define ANN_D_FOLD3 (-244)
#define ANN_D_FOLD4 (-245)
// D = ANN_D_FOLD4G / ANN_D_FOLD5G: rows that have NO lanes-per-row layout (more than 64 chunks that do not split evenly:
// d = 300 float, d = 150 double): every kernel runs its any-d code (literal tree through LDS) except the selection
// gathers, which fold 4 / 5 levels (16 / 32 leaves per lane).
#define ANN_D_FOLD4G (-246)
#define ANN_D_FOLD5G (-247)
// levels of the literal tree folded into a lane for a row of d elements: halve until <= 16 values remain (0: none)
__host__ __device__ inline int ann_fold_levels(int d) {
  int s = d, L = 0;
  while (L < 5 && s > 16) s >>= 1, L++;
  return s > 64 ? 0 : L;
}
template <int D>
struct OcCode {
  static constexpr bool GEN = D == ANN_D_FOLD4G || D == ANN_D_FOLD5G;  // the any-d code paths outside the gathers
  static constexpr int FOLD = D == ANN_D_FOLD2 ? 2 : D == ANN_D_FOLD3 ? 3 : (D == ANN_D_FOLD4 || D == ANN_D_FOLD4G) ? 4 :
                              D == ANN_D_FOLD5G ? 5 : 0;
  static constexpr bool UA = D == ANN_D_UNALIGNED || (FOLD > 0 && !GEN);
  static constexpr int C = (D < 0 && !GEN && !UA) ? ((-D) % 16) : 1;  // 16-byte chunks per lane (1 in the element-wise layouts)
  static constexpr int OC = (D < 0 && !GEN && !UA) ? ((-D) / 16) : 0;
  // Candidate rows are read once per query batch, hence non-temporal loads -- unless a row is not a whole number of
  // 128-byte lines: its lines are then touched by two different load instructions, and with non-temporal loads the second
  // touch goes to memory again.  Pure random-row gather, 320-byte rows (d = 80 float, the reference drivers' default):
  // 4.5-4.75 TB/s of row bytes non-temporal, 5.2-5.3 TB/s cached (tools/readbw.hip); 512-byte rows: 6.8 vs 6.3.
  static constexpr bool NT_ROWS = OC == 0 || (OC * C * 16) % 128 == 0;
};
template <int D>
struct RowChunks<D, false> {
  static constexpr int C = OcCode<D>::C;
};
// lane -> (row of the wave pass, position in the row's lane group) for the layout encoded by D < 0
template <int D>
struct OcLanes {
  int oc, rpw, g, p;
  bool valid;
  __device__ __forceinline__ OcLanes(int d, int lane) {
    constexpr int C = OcCode<D>::C, OC = OcCode<D>::OC;
    if constexpr (OC > 0) {
      constexpr int GPR = 16 / OC;  // groups per 16-lane DPP row
      const int l = lane & 15;
      oc = OC, rpw = 4 * GPR, g = (lane >> 4) * GPR + l / OC, p = l % OC, valid = l < GPR * OC;
    } else {
      oc = OcCode<D>::UA ? (d + ANN_VEC - 1) / ANN_VEC : d / (ANN_VEC * C);
      rpw = ANN_WAVE / oc, g = lane / oc, p = lane - g * oc, valid = g < rpw;
    }
  }
};

// v[r] of lane `src`, r wave-uniform.  Written as a uniform switch (one shuffle per case): a select chain over the
// registers is turned by the compiler into a store/load through scratch memory.
__device__ __forceinline__ FT shfl_reg(const FT (&v)[ANN_VEC], int r, int src) {
  FT o;
  switch (__builtin_amdgcn_readfirstlane(r)) {
    case 0: o = __shfl(v[0], src); break;
    case 1: o = __shfl(v[1], src); break;
#if ANN_VEC > 2
    case 2: o = __shfl(v[2], src); break;
    default: o = __shfl(v[3], src); break;
#else
    default: o = __shfl(v[1], src); break;
#endif
  }
  return o;
}

// Static tail: the literal tree (compute.cl:160-167) over S values, value z = VEC*p + j in e[j] of lane p, one level per
// template instance; the partner z + h lives in lane p + (j+h)/VEC, register (j+h)%VEC -- both compile-time constants.
template <int MODE, int S, int J>
__device__ __forceinline__ void oc_tail_elem(const FT (&old)[ANN_VEC], FT (&e)[ANN_VEC], int p, FT g) {
  constexpr int h = S >> 1, hh = J + h, off = hh / ANN_VEC, reg = hh % ANN_VEC;
  const FT zero = 0;
  FT o;
  if constexpr (off == 0) o = old[reg]; else o = lane_up<off>(old[reg]);
  const int z = ANN_VEC * p + J;
  if (z < h) e[J] = old[J] + (o + ((z == 0) ? g : zero));  // positions >= h are only read at this level
}
template <int MODE, int S>
__device__ __forceinline__ void oc_tail(FT (&e)[ANN_VEC], int p) {
  if constexpr ((S >> 1) > 0) {
    FT old[ANN_VEC];
#pragma unroll
    for (int j = 0; j < ANN_VEC; j++) old[j] = e[j];
    FT g = 0;
    if constexpr (S & 1) {  // g = m[S-1], added into z == 0 only
      constexpr int zz = S - 1, goff = zz / ANN_VEC, greg = zz % ANN_VEC;
      if constexpr (goff == 0) g = old[greg]; else g = lane_up<goff>(old[greg]);
    }
    oc_tail_elem<MODE, S, 0>(old, e, p, g);
    oc_tail_elem<MODE, S, 1>(old, e, p, g);
#if ANN_VEC > 2
    oc_tail_elem<MODE, S, 2>(old, e, p, g);
    oc_tail_elem<MODE, S, 3>(old, e, p, g);
#endif
    oc_tail<MODE, (S >> 1)>(e, p);
  }
}

template <int C, int MODE, int OC = 0>
__device__ __forceinline__ FT row_reduce_oc(const VT (&a)[C], const VT (&b)[C], int oc, int p, int s0 = 0) {  // s0 > 0: tree length (unaligned d)
  FT e[C][ANN_VEC];
#pragma unroll
  for (int c = 0; c < C; c++) {
    const FT *pa = reinterpret_cast<const FT *>(&a[c]);
    const FT *pb = reinterpret_cast<const FT *>(&b[c]);
#pragma unroll
    for (int j = 0; j < ANN_VEC; j++) {
      if (MODE == ROW_SQDIFF) {
        FT df = pa[j] - pb[j];
        e[c][j] = df * df;
      } else {
        e[c][j] = pa[j] * pb[j];
      }
    }
  }
  const FT zero = 0;
#pragma unroll
  for (int h = C / 2; h >= 1; h >>= 1)
#pragma unroll
    for (int c = 0; c < h; c++)
#pragma unroll
      for (int j = 0; j < ANN_VEC; j++)
        e[c][j] = (MODE == ROW_PRODUCT) ? e[c][j] + (e[c + h][j] + zero) : e[c][j] + e[c + h][j];
  // tail: literal tree (compute.cl:160-167) over m = oc*VEC values, value z = VEC*p + j lives in e[0][j] of lane p
  if constexpr (OC > 0) {
    oc_tail<MODE, OC * ANN_VEC>(e[0], p);
    return e[0][0];
  }
  const int lane = lane_id();
  for (int s = s0 > 0 ? s0 : oc * ANN_VEC; s >> 1; s >>= 1) {
    const int h = s >> 1;
    FT g = zero;
    if (s & 1) {  // g = m[s-1], added into z == 0 only; every lane takes part in the shuffle
      const int zz = s - 1;
      g = shfl_reg(e[0], zz % ANN_VEC, lane - p + zz / ANN_VEC);
    }
#pragma unroll
    for (int j = 0; j < ANN_VEC; j++) {
      const int hh = j + h;  // partner of z = VEC*p + j is z + h = VEC*(p + hh/VEC) + hh%VEC
      const FT o = shfl_reg(e[0], hh % ANN_VEC, lane + hh / ANN_VEC);
      const int z = ANN_VEC * p + j;
      if (z < h) e[0][j] = e[0][j] + (o + ((z == 0) ? g : zero));  // values >= h are only read at this level
    }
  }
  return e[0][0];
}

// Any d: the whole wave works on one row, staging the d terms in LDS scratch m[d] and running the
// in-place tree literally (odd s term included).  a = left operand (LDS or global), b = row (global).
// Returns the sum in every lane.
template <int MODE>
__device__ inline FT row_reduce_generic(int d, const FT *a, const FT *b, FT *m) {
  const int lane = lane_id();
  const FT zero = 0;
  for (int z = lane; z < d; z += ANN_WAVE) {
    if (MODE == ROW_SQDIFF) {
      FT df = a[z] - b[z];
      m[z] = df * df;
    } else {
      m[z] = a[z] * b[z];
    }
  }
  wave_lds_sync();
  for (int s = d; s >> 1; s >>= 1) {
    int h = s >> 1;
    for (int z = lane; z < h; z += ANN_WAVE) {
      FT g = ((s & 1) && z == 0) ? m[s - 1] : zero;
      m[z] = m[z] + (m[z + h] + g);
    }
    wave_lds_sync();
  }
  FT r = m[0];
  wave_lds_sync();
  return r;
}

// ------------------------------------------------------------------ the reference's "sort" network
// floor(log2(x)), 0 for x == 0 (algc.c:13-22).
__host__ __device__ inline int ann_lg(size_t x) {
  int r = 0;
  while (x >>= 1) r++;
  return r;
}
// Entries of a row of length L the top-k stage can ever look at: the sorted prefix
// P = 2^floor(log2 L), the k entries that are output, plus one more id for the duplicate test
// (compute.cl:212-217).  Rows shorter than 16 get a clipped 16-wide network, keep them whole.
__host__ __device__ inline size_t ann_need_len(size_t L, size_t k) {
  if (L < 16) return L;
  size_t P = (size_t)1 << ann_lg(L);
  size_t m = (P > k ? P : k) + 1;
  return m < L ? m : L;
}

// do_sort (alg.c:137-144) on one row held in LDS or global memory, executed by one workgroup.
// L is the reference's row length (it clips pairs with ib >= L); only indices < ann_need_len are touched.
// Pair pr is owned by thread pr % blockDim.x.  For sub-steps with stride 2^ss <= 32 the 64 pairs of one wave
// only touch that wave's own 128 entries (in every pass of blockDim.x pairs), so consecutive such sub-steps
// need a wave-level fence, not a workgroup barrier: 33 instead of 78 barriers per sort at 4096 entries.
template <bool IN_LDS, typename KP, typename IP>
__device__ inline void block_sort_net(size_t L, KP key, IP ids) {
  const int lk = ann_lg(L);
  const u32 npairs = 8u << (lk > 4 ? lk - 4 : 0);
  for (int s = 0; s < lk; s++)
    for (int ss = s; ss >= 0; ss--) {
      for (u32 pr = threadIdx.x; pr < npairs; pr += blockDim.x) {
        u32 hi = (pr >> ss) << ss, lo = pr ^ hi;
        u32 ia = hi << 1 | lo;
        if (ss == s) lo = (1u << ss) - lo - 1;
        u32 ib = hi << 1 | 1u << ss | lo;
        if (ib < L) {
          FT ka = key[ia], kb = key[ib];
          if (ka > kb) {  // strict: ties and NaN never swap (compute.cl:198-203)
            u32 ta = ids[ia], tb = ids[ib];
            key[ia] = kb, key[ib] = ka;
            ids[ia] = tb, ids[ib] = ta;
          }
        }
      }
      if (IN_LDS && ss > 0 && ss <= 5)
        wave_lds_sync();  // the next sub-step (stride 2^(ss-1)) stays inside this wave's entries
      else
        __syncthreads();
    }
}

// sort_and_uniq (alg.c:224-230): network, kill the first of each adjacent equal-id pair, network.
// `len` = number of stored entries (>= ann_need_len(L,k)).
template <bool IN_LDS, typename KP, typename IP>
__device__ inline void block_topk_stage(size_t L, size_t len, KP key, IP ids) {
  block_sort_net<IN_LDS>(L, key, ids);
  const FT inf = ft_inf();
  // every y reads ids only and writes its own key: no hazard inside the pass
  for (size_t y = threadIdx.x; y + 1 < len; y += blockDim.x)
    if (ids[y] == ids[y + 1]) key[y] = key[y] + inf;
  __syncthreads();
  block_sort_net<IN_LDS>(L, key, ids);
}
