// ann_query_kernels.h -- HIP kernels of the query()/det_results hot path (gfx950).
//
// What the reference does with ~250 OpenCL launches and a materialised [Q][L1][d] tensor
// (/root/reference/alg.c:458-519, 303-337) is done here by a handful of kernels:
//
//   codes_kernel        y - means, projection onto bases, sign hash            (alg.c:462-492)
//   stage1_select       candidate ids from the bucket tables (per-bucket segment words), row gather, squared L2 in
//                       the exact tree order, and selection of the k+1 smallest distinct (dist,id) keys
//                       -- no sort network, no materialised distances          (alg.c:493-500,308-312);
//                       optionally (small batches) the whole of stage 2 in the workgroup's tail (FusedTail)
//   stage1_bucket       precomp's second_half, one workgroup per bucket        (alg.c:245-290)
//   finalize1           proves the selection equals sort/rdups/sort (no ties between different ids
//                       among the k+1 best, >= k finite, an +inf inside the sorted prefix) or flags
//                       the query for the exact path
//   row_dists           exact path / stage 2: every needed slot's id and distance, in slot order
//                       (shufcomp alg.c:438-452, supercharge compute.cl:252-263, compdists alg.c:233-242)
//   exact_select        the reference's network + rdups + network, literally    (alg.c:224-230)
//   merge_finalize /    multi-GPU (owner protocol): G sorted candidate lists -> the k globally best + the selection
//   final_select        proof; min over the devices' partial stage-2 rows + the reference's network
//
// Only slots below ann_need_len() are ever produced.
#pragma once
#include "ann_device.h"
#include "ann_tie.h"

// Row passes fetched before the first is reduced in stage 2 / row_dists, in 16-byte chunks per lane.  1 = one pass at a
// time.  Measured with 8: nothing at cfg3 (stage 2 is one 282-MB burst on HBM, not per-pass latency) and -19 % on the
// fused small-batch kernel (registers); kept as an A/B knob.
#ifndef ANN_ROWS_INFLIGHT
#define ANN_ROWS_INFLIGHT 1
#endif

struct TryInfo {   // one try (= one random projection) of the index
  const u32 *tab;  // [2^ds][pm] bucket table, ids descending then padding n  (alg.c:261-266)
  const uint2 *seg;  // [2^ds] per bucket: x = first owned position | owned count << 16, y = valid count (see build_seg_kernel)
  const uint4 *segx; // [2^ds][2] 32-byte records of a small shard: header + the first 7 owned ids inline (build_segx_kernel), or NULL
  u32 pm;          // par_maxes[t]
  u32 off;         // first slot of this try's block in the candidate row     (alg.c:484-488,449)
  u32 end;         // off + (ds+1)*pm
  u32 magic;       // floor(2^32/pm)+1: slot/pm by multiply-high (exact while slot*pm < 2^32)
};

struct QParams {
  const FT *points;  // rows [lo,hi) of the point matrix, row-major, d elements each
  const TryInfo *tries;
  const u32 *graph;  // [n][k]
  const FT *means;   // [d]
  const FT *bases;   // [T][ds][d]
  u32 n, lo, hi;
  int d, k, T, ds;
  u32 L1, P1, Lc1, L2, Lc2;
  u32 q0, qn;        // stage 1: this launch covers queries [q0, q0 + qn); workgroup b takes q0 + b, q0 + b + grid ...
                     // (grid == qn: one query per workgroup; a smaller, PERSISTENT grid leaves wave slots to other streams)
  u32 fixed;         // opt-in non-parity mode (annhip_index_set_fixed): a query reads ITS OWN codes (Q2 undone); the host
                     // also sets P1 = Lc1 = L1 (every slot is a candidate, Q1 undone) and no network decides an order
};

#define ANN_S1_CHUNK 1024  // slots whose valid ids one wave stages in LDS at a time
#ifndef ANN_S1_PREFETCH
#define ANN_S1_PREFETCH 3  // row passes per wave kept in flight + 1 (gather_select)
#endif
#ifndef ANN_S1_PREFETCH_OC
#define ANN_S1_PREFETCH_OC 2  // the same for the oc-lanes-per-row layouts (d = 80, 96, 160 ...): d = 80 measured 2 / 3 / 4: 69.8 / 66.1 / 63 % of peak (124 / 143 VGPRs)
#endif

// One 16-byte chunk of a gathered point row.  NT: query batches read each candidate row once and never again,
// so the load is marked non-temporal and does not displace the re-used bucket tables / graph in L2 and the
// Infinity Cache (measured at cfg3: stage-1 kernel 1.244 -> 1.166 ms).  precomp keeps cached loads: there the
// same rows are candidates of every bucket-mate.
template <bool NT>
__device__ __forceinline__ VT load_row_chunk(const VT *p) {
  if constexpr (NT) {
    typedef FT native_vec __attribute__((ext_vector_type(ANN_VEC)));
    native_vec v = __builtin_nontemporal_load(reinterpret_cast<const native_vec *>(p));
    return __builtin_bit_cast(VT, v);
  } else {
    return *p;
  }
}

// Chunk `ci` of a row of d elements in the layouts D < 0.  Aligned layouts: one 16-byte load (NT as above); the
// unaligned one: element by element, zeros beyond d (never read by the tree, which starts at d).
template <int D, bool NT>
__device__ __forceinline__ VT oc_load_chunk(const FT *row, int ci, int d) {
  if constexpr (OcCode<D>::UA) {
    VT v;
    FT *o = reinterpret_cast<FT *>(&v);
#pragma unroll
    for (int j = 0; j < ANN_VEC; j++) {
      const int z = ci * ANN_VEC + j;
      o[j] = z < d ? (NT ? __builtin_nontemporal_load(row + z) : row[z]) : (FT)0;
    }
    return v;
  } else {
    return load_row_chunk<NT>(reinterpret_cast<const VT *>(row) + ci);
  }
}
template <int D>
__device__ __forceinline__ int oc_tree_len(int d) { return OcCode<D>::UA ? d : 0; }

// ------------------------------------------------------------------------------------------ codes
// code[q*T+t] (the reference's WRITE layout; stage 1 reads it back as [i*Q+x]).
//
// Power-of-two d: one workgroup owns one try t and a run of ANN_CODES_QPB queries.  The try's ds projection
// rows are staged in LDS once (ds*d*s bytes, 10 KB at cfg3) and every wave then streams queries against them:
// the rows are read from LDS instead of being re-fetched from L2 for every (query, try) pair.
// D = 0 (any d): one wave per (query, try), literal tree through LDS.
#define ANN_CODES_QPB 64
template <int D>
__global__ __launch_bounds__(256) void codes_kernel(QParams P, int Q, const FT *__restrict__ y,
                                                    u32 *__restrict__ codes, u32 *__restrict__ zero_me) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // the batch's flagged-query counter is reset here: a launch of its own cost ~5 us of every step
  if (zero_me && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_me = 0;
  const int lane = lane_id(), w = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    const int t = blockIdx.y;
    const int q0 = blockIdx.x * ANN_CODES_QPB, q1 = min(Q, q0 + ANN_CODES_QPB);
    VT *rows = reinterpret_cast<VT *>(smem);  // [ds][D/VEC]
    const VT *src = reinterpret_cast<const VT *>(P.bases + (size_t)t * P.ds * D);
    for (int i = threadIdx.x; i < P.ds * (D / ANN_VEC); i += blockDim.x) rows[i] = src[i];
    __syncthreads();
    const int p = lane % L::LPR, g = lane / L::LPR;
    VT mean[L::C];
    const VT *mp = reinterpret_cast<const VT *>(P.means) + p;
#pragma unroll
    for (int c = 0; c < L::C; c++) mean[c] = mp[c * L::LPR];
    for (int q = q0 + w; q < q1; q += wpb) {
      VT a[L::C];
      const VT *yp = reinterpret_cast<const VT *>(y + (size_t)q * D) + p;
#pragma unroll
      for (int c = 0; c < L::C; c++) {
        VT yv = yp[c * L::LPR];
        FT *o = reinterpret_cast<FT *>(&a[c]);
        const FT *py = reinterpret_cast<const FT *>(&yv), *pm = reinterpret_cast<const FT *>(&mean[c]);
#pragma unroll
        for (int j = 0; j < ANN_VEC; j++) o[j] = py[j] - pm[j];  // subtract_off, compute.cl:44-49
      }
      u32 code = 0;
      for (int s0 = 0; s0 < P.ds; s0 += L::RPW) {
        const int s = s0 + g;
        const bool act = s < P.ds;
        const VT *bp = rows + (size_t)(act ? s : 0) * (D / ANN_VEC) + p;
        VT b[L::C];
#pragma unroll
        for (int c = 0; c < L::C; c++) b[c] = bp[c * L::LPR];
        FT v = row_reduce<D, ROW_PRODUCT>(a, b);
        u32 sign = (u32)(ft_bits(v) >> (sizeof(FT) * 8 - 1));
        if (act && p == 0 && sign) code |= 1u << (P.ds - 1 - s);  // coord 0 = MSB, compute.cl:223-231
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) code |= __shfl_xor(code, m);
      if (lane == 0) codes[(size_t)q * P.T + t] = code;
    }
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    constexpr int C = OcCode<D>::C, OC = OcCode<D>::OC;
    const OcLanes<D> ol(P.d, lane);
    const int oc = ol.oc, rpw = ol.rpw, g = ol.g, p = ol.p;
    const long item = (long)blockIdx.x * wpb + w;
    const bool live = item < (long)Q * P.T;
    const int q = live ? (int)(item / P.T) : 0, t = live ? (int)(item % P.T) : 0;
    VT a[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
      VT yv = oc_load_chunk<D, false>(y + (size_t)q * P.d, p + c * oc, P.d), mv = oc_load_chunk<D, false>(P.means, p + c * oc, P.d);
      FT *o = reinterpret_cast<FT *>(&a[c]);
      const FT *py = reinterpret_cast<const FT *>(&yv), *pm = reinterpret_cast<const FT *>(&mv);
#pragma unroll
      for (int j = 0; j < ANN_VEC; j++) o[j] = py[j] - pm[j];
    }
    u32 code = 0;
    for (int s0 = 0; s0 < P.ds; s0 += rpw) {
      const int sidx = s0 + g;
      const bool act = ol.valid && sidx < P.ds;
      const FT *brow = P.bases + ((size_t)t * P.ds + (act ? sidx : 0)) * P.d;
      VT b[C];
#pragma unroll
      for (int c = 0; c < C; c++) b[c] = oc_load_chunk<D, false>(brow, p + c * oc, P.d);
      FT v = row_reduce_oc<C, ROW_PRODUCT, OC>(a, b, oc, p, oc_tree_len<D>(P.d));
      u32 sign = (u32)(ft_bits(v) >> (sizeof(FT) * 8 - 1));
      if (act && p == 0 && sign) code |= 1u << (P.ds - 1 - sidx);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) code |= __shfl_xor(code, m);
    if (live && lane == 0) codes[item] = code;
  } else {
    const long item = (long)blockIdx.x * wpb + w;
    const bool live = item < (long)Q * P.T;
    const int q = live ? (int)(item / P.T) : 0, t = live ? (int)(item % P.T) : 0;
    u32 code = 0;
    const int d = P.d;
    FT *u = reinterpret_cast<FT *>(smem) + (size_t)w * 2 * d, *m = u + d;
    for (int z = lane; z < d; z += ANN_WAVE) u[z] = y[(size_t)q * d + z] - P.means[z];
    wave_lds_sync();
    for (int s = 0; s < P.ds; s++) {
      FT v = row_reduce_generic<ROW_PRODUCT>(d, u, P.bases + ((size_t)t * P.ds + s) * d, m);
      code = code << 1 | (u32)(ft_bits(v) >> (sizeof(FT) * 8 - 1));
    }
    if (live && lane == 0) codes[item] = code;
  }
}

// Lane-per-query form of the hash for short power-of-two rows (d * s <= 512 bytes: d <= 128 float, 64 double).
// codes_kernel spreads ONE row over 32 lanes and pays five cross-lane tree levels (a DPP move and two adds each) per
// dot product: ~12 wave instructions per dot, 39 us per cfg3 batch.  Here a lane owns a whole query: its centred
// row sits in registers (d of them), a projection row comes out of LDS as broadcast reads, and the pairwise tree of
// compute.cl:160-167 runs literally, in the lane, on d/2 partial sums (level 1 fused with the products; the
// reference's "+ 0" in every node kept, see row_reduce) -- ~3 d lane instructions per dot and no cross-lane traffic.
// One workgroup = (try, 64 queries); its ANN_LPQ_WAVES waves split the try's ds projections between them (a wave per
// (try, 64 queries) leaves 1 100 long waves for 1 024 SIMDs at cfg3: the kernel then lasts two wave times).
#ifndef ANN_LPQ_WAVES
#define ANN_LPQ_WAVES 2
#endif
template <int D>
__global__ __launch_bounds__(64 * ANN_LPQ_WAVES) void codes_lpq_kernel(QParams P, int Q, const FT *__restrict__ y,
                                                                       u32 *__restrict__ codes, u32 *__restrict__ zero_me) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (zero_me && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_me = 0;
  constexpr int NC = D / ANN_VEC;  // 16-byte chunks per row
  const int lane = lane_id(), t = blockIdx.y, w = threadIdx.x >> 6;
  VT *rows = reinterpret_cast<VT *>(smem);  // [ds][NC], then the means [NC], then the waves' partial codes
  VT *mean = rows + (size_t)P.ds * NC;
  u32 *pcode = reinterpret_cast<u32 *>(mean + NC);  // [ANN_LPQ_WAVES][64]
  const VT *src = reinterpret_cast<const VT *>(P.bases + (size_t)t * P.ds * D);
  for (int i = threadIdx.x; i < P.ds * NC; i += blockDim.x) rows[i] = src[i];
  for (int i = threadIdx.x; i < NC; i += blockDim.x) mean[i] = reinterpret_cast<const VT *>(P.means)[i];
  __syncthreads();
  const int q = blockIdx.x * ANN_WAVE + lane;
  const bool live = q < Q;
  const VT *yp = reinterpret_cast<const VT *>(y + (size_t)(live ? q : Q - 1) * D);
  FT a[D];
#pragma unroll
  for (int c = 0; c < NC; c++) {
    const VT yv = yp[c], mv = mean[c];
    const FT *py = reinterpret_cast<const FT *>(&yv), *pm = reinterpret_cast<const FT *>(&mv);
#pragma unroll
    for (int j = 0; j < ANN_VEC; j++) a[c * ANN_VEC + j] = py[j] - pm[j];  // subtract_off, compute.cl:44-49
  }
  const FT zero = 0;
  u32 code = 0;
  const int sper = (P.ds + ANN_LPQ_WAVES - 1) / ANN_LPQ_WAVES, s_lo = w * sper, s_hi = min(P.ds, s_lo + sper);
#pragma unroll 1
  for (int s = s_lo; s < s_hi; s++) {
    const VT *b = rows + (size_t)s * NC;
    FT m[D / 2];
#pragma unroll
    for (int c = 0; c < NC / 2; c++) {  // products + the tree's first level: z with z + D/2
      // (the scheduler would otherwise hoist all d/4 LDS reads of the row to the top: d more live registers)
      if (c % 4 == 0 && c) __builtin_amdgcn_sched_barrier(0);
      const VT b0 = b[c], b1 = b[c + NC / 2];
      const FT *p0 = reinterpret_cast<const FT *>(&b0), *p1 = reinterpret_cast<const FT *>(&b1);
#pragma unroll
      for (int j = 0; j < ANN_VEC; j++) {
        const int z = c * ANN_VEC + j;
        m[z] = a[z] * p0[j] + (a[z + D / 2] * p1[j] + zero);
      }
    }
#pragma unroll
    for (int h = D / 4; h >= 1; h >>= 1)
#pragma unroll
      for (int z = 0; z < h; z++) m[z] = m[z] + (m[z + h] + zero);
    const u32 sign = (u32)(ft_bits(m[0]) >> (sizeof(FT) * 8 - 1));
    code |= sign << (P.ds - 1 - s);  // coord 0 = MSB, compute.cl:223-231
  }
  pcode[w * ANN_WAVE + lane] = code;
  __syncthreads();
  if (w == 0 && live) {
#pragma unroll
    for (int ww = 1; ww < ANN_LPQ_WAVES; ww++) code |= pcode[ww * ANN_WAVE + lane];
    codes[(size_t)q * P.T + t] = code;
  }
}

// id stored in slot j of query x's candidate row (compute_which, compute.cl:238-246; layout SURVEY Q9).
// `tries`/`qcode` live in LDS.  ti is a cursor the caller may keep between increasing j.
__device__ __forceinline__ u32 slot_id(const TryInfo *tries, const u32 *qcode, u32 j, int &ti) {
  while (j >= tries[ti].end) ti++;
  const TryInfo tr = tries[ti];
  u32 r = j - tr.off;
  u32 yy = __umulhi(r, tr.magic);
  u32 z = r - yy * tr.pm;
  u32 b = qcode[ti] ^ (yy ? 1u << (yy - 1) : 0u);
  return tr.tab[(size_t)b * tr.pm + z];
}

// ---------------------------------------------------------------------------------- stage1_select
// Bucket rows hold their ids in descending order followed by padding (Q8), so (a) the valid ids of a bucket are
// a prefix and (b) the ids a device owns (a contiguous id range) are ONE contiguous segment of the row.
// build_seg_kernel records, per bucket, where that segment starts, how long it is, and how many valid ids the
// bucket has.  The scan of stage 1 then touches only ids it will gather instead of every slot: at cfg3
// 168 segment words + 1.4k ids per query instead of 4096 slot ids (and 1/G of the ids on each of G shards).
// It also verifies the layout assumption; a table that violates it (not produced by precomp) sets *bad and the
// host falls back to the slot scan.
__global__ void build_seg_kernel(size_t nbuckets, u32 pm, const u32 *__restrict__ tab, u32 n, u32 lo, u32 hi,
                                 uint2 *__restrict__ seg, u32 *__restrict__ bad) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  const u32 *row = tab + b * pm;
  u32 ca = 0, zs = 0, co = 0, prev = 0xFFFFFFFFu;
  bool ok = true, ended = false, own_ended = false;
  for (u32 z = 0; z < pm; z++) {
    const u32 id = row[z];
    if (id >= n) {
      ended = true;
      continue;
    }
    if (ended || id >= prev) ok = false;  // valid ids must be a strictly descending prefix
    prev = id;
    ca++;
    if (id >= lo && id < hi) {
      if (own_ended) ok = false;
      if (!co) zs = z;
      co++;
    } else if (co) {
      own_ended = true;
    }
  }
  if (!ok || pm > 0xFFFFu) atomicAdd(bad, 1u);
  seg[b] = make_uint2(zs | (co << 16), ca);
}

// A device that owns a small part of the rows (1/4, 1/8 ...) finds about one owned id per bucket: reading the segment
// word and then the id costs two dependent random fetches per (try, neighbour) run, and the replicated scan becomes
// ~20 % of the stage-1 time at 8 shards.  For such shards every bucket gets ONE 32-byte record:
//   word 0 = first owned position | owned count << 8 | valid count << 16   (par_maxes <= 255 checked on the host)
//   words 1..7 = the first 7 owned ids (descending, as in the table row)
// so a run costs one 32-byte fetch; buckets with more than 7 owned ids (rare on a small shard) read the rest from the
// table row.  Built from the segment words, which have verified the layout.
#define ANN_SEGX_INLINE 7
__global__ void build_segx_kernel(size_t nbuckets, u32 pm, const u32 *__restrict__ tab, const uint2 *__restrict__ seg,
                                  uint4 *__restrict__ segx) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  const uint2 sg = seg[b];
  const u32 zs = sg.x & 0xFFFFu, co = sg.x >> 16, ca = sg.y;
  u32 w[8];
  w[0] = zs | (co << 8) | (ca << 16);
  const u32 *row = tab + b * pm + zs;
#pragma unroll
  for (int j = 0; j < ANN_SEGX_INLINE; j++) w[1 + j] = (u32)j < co ? row[j] : ANN_ID_NONE;
  segx[2 * b] = make_uint4(w[0], w[1], w[2], w[3]);
  segx[2 * b + 1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// Per-wave state of the running selection of the k+1 smallest distinct keys.
struct SelState {
  Key *kbuf, *kout;  // LDS: cap keys / K1 keys
  int kcnt, K1, cap;
  Key tau;           // keys >= tau cannot be among the k+1 smallest any more
};

__device__ __forceinline__ void sel_shrink(SelState &S) {
  wave_lds_sync();
  const int m = wave_select_smallest(S.kbuf, S.kcnt, S.K1, S.kout);
  for (int i = lane_id(); i < m; i += ANN_WAVE) S.kbuf[i] = S.kout[i];
  if (m == S.K1) S.tau = S.kout[S.K1 - 1];
  S.kcnt = m;
  wave_lds_sync();
}

// ---- folded layout for row lengths that are not a power of two (d = 100, 200, 384, 768, 33, 77 ...).
// The literal tree (compute.cl:160-167) halves s = d, d>>1, ... and at each level adds m[z + h] (and, into z = 0, the
// left-over m[s-1] of an odd s) to m[z].  Its first L levels are folded INTO a lane: lane z < s_L of a row's lane group
// holds the 2^L leaves z + sum of a subset of {h_1..h_L} and reduces them with the same additions in the same order
// (a bit-butterfly: level l combines the leaves that differ in bit l-1); the left-over of an odd level is a regular
// subtree of the lower levels and is computed by lane 0 alone.  What remains are s_L <= 16 values, one per lane, whose
// tree takes one or two shuffles per level.  Against the lanes-per-row layout with shuffles from the first level on
// (row_reduce_oc, OC = 0): d = 100 float needs 5 shuffles per 5 rows instead of ~30 per 2 rows.
struct FoldPlan {
  int L, sL;
  int h[5], odd[5], sprev[5];
  __device__ __forceinline__ explicit FoldPlan(int d) {
    int s = d;
    L = 0;
    while (L < 5 && s > 16) {
      sprev[L] = s, h[L] = s >> 1, odd[L] = s & 1;
      s >>= 1;
      L++;
    }
    sL = s;
    if (sL > 64) L = 0;  // not representable: the caller keeps its own layout
  }
};

template <int LV>
__device__ __forceinline__ void gather_fold(const QParams &P, const FoldPlan &fp, const u32 *list, int cnt, int alias, u32 x,
                                            const FT *yrow, SelState &S) {
  constexpr int NL = 1 << LV;
  const int lane = lane_id(), d = P.d, sL = fp.sL, rpw = ANN_WAVE / sL;
  const int g = lane / sL, z = lane - g * sL;
  const bool valid = g < rpw;
  const FT zero = 0;
  int off[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    off[i] = 0;
#pragma unroll
    for (int l = 0; l < LV; l++)
      if ((i >> l) & 1) off[i] += fp.h[l];
  }
  FT a[NL], ax[NL];  // the query's leaves; ax[(1 << l) + i]: leaf i of the left-over subtree of level l (lane z == 0 only)
#pragma unroll
  for (int i = 0; i < NL; i++) a[i] = yrow[z + off[i]], ax[i] = zero;
  if (z == 0) {
#pragma unroll
    for (int l = 0; l < LV; l++)
      if (fp.odd[l]) {
#pragma unroll
        for (int i = 0; i < (1 << l); i++) ax[(1 << l) + i] = yrow[fp.sprev[l] - 1 + off[i]];
      }
  }
  FT bn[NL];
  u32 idn = 0;
  if (cnt > 0) {
    idn = list[(valid && g < cnt) ? g : 0];
    const FT *rp = P.points + (size_t)(idn - P.lo) * d + z;
#pragma unroll
    for (int i = 0; i < NL; i++) bn[i] = __builtin_nontemporal_load(rp + off[i]);
  }
  for (int base = 0; base < cnt; base += rpw) {
    FT e[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const FT t = a[i] - bn[i];
      e[i] = t * t;
    }
    const u32 id = idn;
    const bool act = valid && base + g < cnt && !(alias && id == x);
    FT gx[LV];
#pragma unroll
    for (int l = 0; l < LV; l++) gx[l] = zero;
    // left-overs of the odd levels: regular subtrees of the levels below, lane 0 of each group only.  (Loading them
    // with the row and computing them in every lane was tried: the extra VALU work costs more than the divergence.)
    if (z == 0) {
      const FT *rp0 = P.points + (size_t)(id - P.lo) * d;
#pragma unroll
      for (int l = 0; l < LV; l++)
        if (fp.odd[l]) {
          FT t[NL];  // 2^l of them are used (every bound below is a constant once the loops are unrolled)
#pragma unroll
          for (int i = 0; i < NL; i++)
            if (i < (1 << l)) {
              const FT df = ax[(1 << l) + i] - __builtin_nontemporal_load(rp0 + fp.sprev[l] - 1 + off[i]);
              t[i] = df * df;
            }
#pragma unroll
          for (int j = 0; j < LV; j++)
#pragma unroll
            for (int i = 0; i < NL; i += 2 << j)
              if (j < l && i < (1 << l)) t[i] = t[i] + (t[i + (1 << j)] + zero);
          gx[l] = t[0];
        }
    }
    const int nb = base + rpw;
    if (nb < cnt) {  // next pass in flight while this one is reduced
      idn = list[(valid && nb + g < cnt) ? nb + g : nb];
      const FT *rp = P.points + (size_t)(idn - P.lo) * d + z;
#pragma unroll
      for (int i = 0; i < NL; i++) bn[i] = __builtin_nontemporal_load(rp + off[i]);
    }
#pragma unroll
    for (int l = 0; l < LV; l++)
#pragma unroll
      for (int i = 0; i < NL; i += 2 << l) e[i] = e[i] + (e[i + (1 << l)] + (i == 0 ? gx[l] : zero));
    FT v = e[0];
    for (int s = sL; s >> 1; s >>= 1) {  // the remaining tree: value z in lane z of the group
      const int h = s >> 1;
      const FT o = __shfl(v, lane + h);
      FT gg = zero;
      if (s & 1) gg = __shfl(v, lane - z + s - 1);
      if (z < h) v = v + (o + (z == 0 ? gg : zero));
    }
    const Key key = key_make(v, id);
    const bool pass = act && z == 0 && key_less(key, S.tau);
    const u64 mm = __ballot(pass);
    if (mm) {
      if (pass) S.kbuf[S.kcnt + mask_rank(mm)] = key;
      S.kcnt += __popcll(mm);
      if (S.kcnt + rpw > S.cap) sel_shrink(S);
    }
  }
}

// B) gather the rows listed in list[0..cnt) (LDS), squared L2 to the query in the reference's tree order, keep
// the keys that can still matter.  D > 0: LPR lanes per row, the next pass is prefetched while this one is reduced.
template <int D>
__device__ __forceinline__ void gather_select(const QParams &P, const u32 *list, int cnt, int alias, u32 x,
                                              const VT (&a)[RowChunks<D>::C],
                                              const FT *yq, FT *scratch, SelState &S, const FT *yrow = NULL) {
  const int lane = lane_id();
  if constexpr (D > 0) {
    // PF row buffers per lane form a ring: while pass i is reduced, the loads of passes i+1 .. i+PF-1 are in flight
    // (each pass = RPW rows = 64 lanes x C x 16 B).  The loop is unrolled by PF so that every buffer has a fixed
    // register home; a buffer is re-loaded right after it has been reduced.
    typedef RowLay<D> L;
    constexpr int PF = ANN_S1_PREFETCH;
    const int p = lane % L::LPR, g = lane / L::LPR;
    VT buf[PF][L::C];
    u32 idb[PF];
#pragma unroll
    for (int s = 0; s < PF; s++) {
      const int first = s * L::RPW;
      idb[s] = 0;
      if (first < cnt) {
        idb[s] = list[first + g < cnt ? first + g : first];
        const VT *rp = reinterpret_cast<const VT *>(P.points + (size_t)(idb[s] - P.lo) * D) + p;
#pragma unroll
        for (int c = 0; c < L::C; c++) buf[s][c] = load_row_chunk<true>(rp + c * L::LPR);
      }
    }
    for (int base = 0; base < cnt; base += PF * L::RPW) {
#pragma unroll
      for (int s = 0; s < PF; s++) {
        const int cur = base + s * L::RPW;
        if (cur < cnt) {  // wave-uniform
          const u32 id = idb[s];
          const bool act = cur + g < cnt && !(alias && id == x);
          const FT dist = row_reduce<D, ROW_SQDIFF>(a, buf[s]);
          const int nb = cur + PF * L::RPW;
          if (nb < cnt) {
            idb[s] = list[nb + g < cnt ? nb + g : nb];
            const VT *rp = reinterpret_cast<const VT *>(P.points + (size_t)(idb[s] - P.lo) * D) + p;
#pragma unroll
            for (int c = 0; c < L::C; c++) buf[s][c] = load_row_chunk<true>(rp + c * L::LPR);
          }
          const Key key = key_make(dist, id);
          const bool pass = act && p == 0 && key_less(key, S.tau);
          const u64 mm = __ballot(pass);
          if (mm) {
            if (pass) S.kbuf[S.kcnt + mask_rank(mm)] = key;
            S.kcnt += __popcll(mm);
            if (S.kcnt + L::RPW > S.cap) sel_shrink(S);
          }
        }
      }
    }
  } else if constexpr (OcCode<D>::FOLD > 0) {  // the first tree levels inside a lane (host: layout_code chose this d for it)
    const FoldPlan fp(P.d);
    gather_fold<OcCode<D>::FOLD>(P, fp, list, cnt, alias, x, yrow, S);
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    // the same ring of PF row buffers as the power-of-two layout: passes i+1 .. i+PF-1 are in flight while pass i is
    // reduced (one pass = rpw rows of oc lanes x C chunks; d = 80: 12 rows = 3.8 KB).  With a single pass of lookahead
    // the d = 80 gather ran at 2.7 waves per SIMD's worth of requests: 4.0 TB/s against 5.8 at d = 64.
    constexpr int C = OcCode<D>::C, OC = OcCode<D>::OC;
    constexpr int PF = ANN_S1_PREFETCH_OC;
    const OcLanes<D> ol(P.d, lane);
    const int oc = ol.oc, rpw = ol.rpw, g = ol.g, p = ol.p;  // lanes with !ol.valid have no row
    VT buf[PF][C];
    u32 idb[PF];
#pragma unroll
    for (int s = 0; s < PF; s++) {
      const int first = s * rpw;
      idb[s] = 0;
      if (first < cnt) {
        idb[s] = list[(ol.valid && first + g < cnt) ? first + g : first];
        const FT *rp = P.points + (size_t)(idb[s] - P.lo) * P.d;
#pragma unroll
        for (int c = 0; c < C; c++) buf[s][c] = oc_load_chunk<D, OcCode<D>::NT_ROWS>(rp, p + c * oc, P.d);
      }
    }
    for (int base = 0; base < cnt; base += PF * rpw) {
#pragma unroll
      for (int s = 0; s < PF; s++) {
        const int cur = base + s * rpw;
        if (cur < cnt) {  // wave-uniform
          const u32 id = idb[s];
          const bool act = ol.valid && cur + g < cnt && !(alias && id == x);
          const FT dist = row_reduce_oc<C, ROW_SQDIFF, OC>(a, buf[s], oc, p, oc_tree_len<D>(P.d));
          const int nb = cur + PF * rpw;
          if (nb < cnt) {
            idb[s] = list[(ol.valid && nb + g < cnt) ? nb + g : nb];
            const FT *rp = P.points + (size_t)(idb[s] - P.lo) * P.d;
#pragma unroll
            for (int c = 0; c < C; c++) buf[s][c] = oc_load_chunk<D, OcCode<D>::NT_ROWS>(rp, p + c * oc, P.d);
          }
          const Key key = key_make(dist, id);
          const bool pass = act && p == 0 && key_less(key, S.tau);
          const u64 mm = __ballot(pass);
          if (mm) {
            if (pass) S.kbuf[S.kcnt + mask_rank(mm)] = key;
            S.kcnt += __popcll(mm);
            if (S.kcnt + rpw > S.cap) sel_shrink(S);
          }
        }
      }
    }
  } else {
    for (int r = 0; r < cnt; r++) {
      const u32 id = list[r];
      if (alias && id == x) continue;  // wave-uniform
      const FT dist = row_reduce_generic<ROW_SQDIFF>(P.d, yq, P.points + (size_t)(id - P.lo) * P.d, scratch);
      const Key key = key_make(dist, id);
      if (key_less(key, S.tau)) {  // wave-uniform
        if (lane == 0) S.kbuf[S.kcnt] = key;
        S.kcnt++;
        if (S.kcnt + 1 > S.cap) sel_shrink(S);
      }
    }
  }
  wave_lds_sync();
}

// Stage 2 of ONE query on the calling workgroup (det_results second half, alg.c:314-327): row of len2 entries =
// the stage-1 top-k (`top`, LDS, ascending keys) followed by the graph neighbours of each of them (supercharge,
// compute.cl:252-263); distances of the new slots gathered here (owned, valid, not the excluded self: else +inf);
// the reference's network + rdups + network in LDS (alg.c:224-230); first k entries to out_ids/out_dist[x].
// t_* and cnt2p are LDS scratch (len2 entries each; *cnt2p must be 0 on entry).  Returns the rows gathered.
template <int D, typename IdOut>
__device__ __forceinline__ u32 stage2_in_workgroup(const QParams &P, u32 x, int alias, const VT (&a)[RowChunks<D>::C],
                                                   const FT *yq, FT *scratch, const Key *top, int k, u32 len2,
                                                   u32 *t_ids, u32 *t_slot, u32 *t_gid, FT *t_dist, u32 *cnt2p,
                                                   IdOut *__restrict__ out_ids, FT *__restrict__ out_dist) {
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  for (u32 j = threadIdx.x; j < len2; j += blockDim.x) {
    u32 id;
    if (j < (u32)k) {
      id = key_id(top[j]);
      t_dist[j] = key_dist(top[j]);
    } else {
      const u32 parent = key_id(top[j / k - 1]), z = j % k;
      id = parent < P.n ? P.graph[(size_t)parent * k + z] : (P.graph[z] | P.n);  // supercharge, Q7
      const bool ok = id < P.n && !(alias && id == x) && id >= P.lo && id < P.hi;
      if (ok) {
        const u32 pos = atomicAdd(cnt2p, 1u);
        t_slot[pos] = j;
        t_gid[pos] = id;
      } else {
        t_dist[j] = ft_inf();
      }
    }
    t_ids[j] = id;
  }
  __syncthreads();
  const int cnt2 = (int)*cnt2p;
  // U row passes are fetched before the first is reduced: written as load / reduce / store per pass the loop pays one
  // memory round trip per pass (the LDS store of a distance orders it before the next pass's LDS read of an id) --
  // 28 dependent round trips for the 55 rows of a cfg3 query on its one wave.
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    constexpr int U = (ANN_ROWS_INFLIGHT / L::C) > 0 ? (ANN_ROWS_INFLIGHT / L::C) : 1;
    const int p = lane % L::LPR, g = lane / L::LPR;
    for (int base0 = w * L::RPW; base0 < cnt2; base0 += W * L::RPW * U) {
      VT b[U][L::C];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = base0 + u * W * L::RPW + g;
        if (base0 + u * W * L::RPW < cnt2) {  // wave-uniform: passes beyond the list are neither loaded nor reduced
          const u32 id = t_gid[r < cnt2 ? r : base0];
          const VT *rp = reinterpret_cast<const VT *>(P.points + (size_t)(id - P.lo) * D) + p;
#pragma unroll
          for (int c = 0; c < L::C; c++) b[u][c] = load_row_chunk<true>(rp + c * L::LPR);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = base0 + u * W * L::RPW + g;
        if (base0 + u * W * L::RPW < cnt2) {
          const FT dist = row_reduce<D, ROW_SQDIFF>(a, b[u]);
          if (r < cnt2 && p == 0) t_dist[t_slot[r]] = dist;
        }
      }
    }
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    constexpr int C = OcCode<D>::C, OC = OcCode<D>::OC;
    constexpr int U = C <= 2 ? 4 : 2;
    const OcLanes<D> ol(P.d, lane);
    const int oc = ol.oc, rpw = ol.rpw, g = ol.g, p = ol.p;
    for (int base0 = w * rpw; base0 < cnt2; base0 += W * rpw * U) {
      VT b[U][C];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = base0 + u * W * rpw + g;
        if (base0 + u * W * rpw < cnt2) {  // wave-uniform
          const u32 id = t_gid[(ol.valid && r < cnt2) ? r : base0];
          const FT *rp = P.points + (size_t)(id - P.lo) * P.d;
#pragma unroll
          for (int c = 0; c < C; c++) b[u][c] = oc_load_chunk<D, OcCode<D>::NT_ROWS>(rp, p + c * oc, P.d);
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int r = base0 + u * W * rpw + g;
        if (base0 + u * W * rpw < cnt2) {
          const FT dist = row_reduce_oc<C, ROW_SQDIFF, OC>(a, b[u], oc, p, oc_tree_len<D>(P.d));
          if (ol.valid && r < cnt2 && p == 0) t_dist[t_slot[r]] = dist;
        }
      }
    }
  } else {
    for (int r = w; r < cnt2; r += W) {
      const FT dist = row_reduce_generic<ROW_SQDIFF>(P.d, yq, P.points + (size_t)(t_gid[r] - P.lo) * P.d, scratch);
      if (lane == 0) t_dist[t_slot[r]] = dist;
    }
  }
  __syncthreads();
  block_topk_stage<true>((size_t)k * (k + 1), len2, t_dist, t_ids);  // sort_and_uniq, alg.c:327
  for (int t = threadIdx.x; t < k; t += blockDim.x) {
    out_ids[(size_t)x * k + t] = t_ids[t];
    out_dist[(size_t)x * k + t] = t_dist[t];
  }
  return (u32)cnt2;
}

// Stage 2 as ONE kernel (single-device query path, stage-2 rows that fit LDS): per query the row assembly, the <= k*k
// neighbour gathers, the network and the size_t ids -- what row_dists<GRAPH> + exact_select + widen_ids did with three
// launches and two round trips of the [Q][Lc2] rows through HBM (cfg3: 58 + 33 + 6 us -> one ~60 us kernel).
// IdOut = size_t: query() results (the ABI's ids); u32: precomp's graph rows.  Query x = xbase + blockIdx.x; outputs are
// indexed by x.
template <int D, typename IdOut>
__global__ __launch_bounds__(128) void stage2_fused_kernel(QParams P, int Q, const FT *__restrict__ y, int alias,
                                                           const u32 *__restrict__ top_id, const FT *__restrict__ top_dist,
                                                           u32 len2, IdOut *__restrict__ out_ids,
                                                           FT *__restrict__ out_dist,
                                                           unsigned long long *__restrict__ rows_done, u32 xbase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6;
  const u32 x = xbase + blockIdx.x;
  const int k = P.k;
  unsigned char *sp = smem;
  Key *top = reinterpret_cast<Key *>(sp);     sp += sizeof(Key) * (size_t)k;
  u32 *t_ids = reinterpret_cast<u32 *>(sp);   sp += sizeof(u32) * (size_t)len2;
  u32 *t_slot = reinterpret_cast<u32 *>(sp);  sp += sizeof(u32) * (size_t)len2;
  u32 *t_gid = reinterpret_cast<u32 *>(sp);   sp += sizeof(u32) * (size_t)len2;
  u32 *cnt2 = reinterpret_cast<u32 *>(sp);    sp += sizeof(u32) * 4;
  sp = smem + (((sp - smem) + 15) & ~(size_t)15);
  FT *t_dist = reinterpret_cast<FT *>(sp);    sp += sizeof(FT) * (size_t)len2;
  sp = smem + (((sp - smem) + 15) & ~(size_t)15);
  FT *yq = reinterpret_cast<FT *>(sp);  // generic d only: [d] + 2*[d]
  for (int t = threadIdx.x; t < k; t += blockDim.x) top[t] = key_make(top_dist[(size_t)x * k + t], top_id[(size_t)x * k + t]);
  if (threadIdx.x == 0) *cnt2 = 0;
  if constexpr (D == 0 || OcCode<D>::GEN)
    for (int z = threadIdx.x; z < P.d; z += blockDim.x) yq[z] = y[(size_t)x * P.d + z];
  VT a[RowChunks<D>::C];
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    const VT *yp = reinterpret_cast<const VT *>(y + (size_t)x * D) + (lane % L::LPR);
#pragma unroll
    for (int c = 0; c < L::C; c++) a[c] = yp[c * L::LPR];
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    const OcLanes<D> ol(P.d, lane);
#pragma unroll
    for (int c = 0; c < OcCode<D>::C; c++) a[c] = oc_load_chunk<D, false>(y + (size_t)x * P.d, ol.p + c * ol.oc, P.d);
  }
  __syncthreads();
  const u32 got = stage2_in_workgroup<D, IdOut>(P, x, alias, a, yq, yq + (size_t)(1 + w) * P.d, top, k, len2, t_ids, t_slot,
                                                t_gid, t_dist, cnt2, out_ids, out_dist);
  if (rows_done && threadIdx.x == 0) atomicAdd(&rows_done[(x & 63u) * 8u], (unsigned long long)got);
}

// Stage 2 by SELECTION for rows too long for the fused kernel's LDS row (k >= 32: Lc2 > 1024; cfg5: 8 193 entries).
// What row_dists<GRAPH> + exact_select do with a [Q][Lc2] round trip through HBM and a P2-entry network per query is
// done by the machinery of stage 1: the workgroup's waves split the first P2 slots of the row (the only ones the
// reference's network orders, Q1), derive the slot ids (supercharge, compute.cl:252-263), gather the owned valid rows
// and keep the k+1 smallest distinct keys; slots [0,k) enter with the distances stage 1 gave them.  The output is the
// first k keys when the finalize1 argument holds for this row: at least k finite distinct keys, no distance shared
// among the kept ones (a tie between different ids is ordered by the network, Q17), and -- only when the sorted prefix
// holds no +inf at all, which is the rule here -- k+1 keys were found: the duplicate test at P2-1 reads slot P2's id
//  and can only ever kill the LARGEST entry of the prefix, which is among the first k outputs only if fewer
// than k+1 distinct keys exist.  Everything else is appended to `flist` and takes the literal path afterwards.
template <int D, typename IdOut>
__global__ __launch_bounds__(256) void stage2_select_kernel(QParams P, int Q, const FT *__restrict__ y, int alias,
                                                            const u32 *__restrict__ top_id, const FT *__restrict__ top_dist,
                                                            u32 P2, int K1, int cap, IdOut *__restrict__ out_ids,
                                                            FT *__restrict__ out_dist, u32 *__restrict__ flist,
                                                            u32 *__restrict__ fcount,
                                                            unsigned long long *__restrict__ exact_total,
                                                            unsigned long long *__restrict__ rows_done, u32 xbase) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const u32 x = xbase + blockIdx.x;
  const int k = P.k;
  // ---- LDS carve-up (mirrored by stage2_select_lds_bytes on the host)
  unsigned char *sp = smem;
  Key *kbuf_all = reinterpret_cast<Key *>(sp);   sp += sizeof(Key) * (size_t)W * cap;
  Key *kout_all = reinterpret_cast<Key *>(sp);   sp += sizeof(Key) * (size_t)W * K1;
  Key *mbuf = reinterpret_cast<Key *>(sp);       sp += sizeof(Key) * (size_t)W * K1;
  Key *top = reinterpret_cast<Key *>(sp);        sp += sizeof(Key) * (size_t)k;
  u32 *list_all = reinterpret_cast<u32 *>(sp);   sp += sizeof(u32) * (size_t)W * ANN_S1_CHUNK;
  int *mcnt = reinterpret_cast<int *>(sp);       sp += sizeof(int) * (size_t)W;
  u32 *cnts = reinterpret_cast<u32 *>(sp);       sp += sizeof(u32) * 4;  // [0] finite entries of the prefix [1] rows gathered
  sp = smem + (((sp - smem) + 15) & ~(size_t)15);
  FT *yq = reinterpret_cast<FT *>(sp);  // generic d only: [d] + W*[d]
  u32 *list = list_all + (size_t)w * ANN_S1_CHUNK;

  for (int t = threadIdx.x; t < k; t += blockDim.x) top[t] = key_make(top_dist[(size_t)x * k + t], top_id[(size_t)x * k + t]);
  if (threadIdx.x < 4) cnts[threadIdx.x] = 0;
  if constexpr (D == 0 || OcCode<D>::GEN)
    for (int z = threadIdx.x; z < P.d; z += blockDim.x) yq[z] = y[(size_t)x * P.d + z];
  VT a[RowChunks<D>::C];
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    const VT *yp = reinterpret_cast<const VT *>(y + (size_t)x * D) + (lane % L::LPR);
#pragma unroll
    for (int c = 0; c < L::C; c++) a[c] = yp[c * L::LPR];
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    const OcLanes<D> ol(P.d, lane);
#pragma unroll
    for (int c = 0; c < OcCode<D>::C; c++) a[c] = oc_load_chunk<D, false>(y + (size_t)x * P.d, ol.p + c * ol.oc, P.d);
  }
  __syncthreads();

  SelState S;
  S.kbuf = kbuf_all + (size_t)w * cap, S.kout = kout_all + (size_t)w * K1;
  S.kcnt = 0, S.K1 = K1, S.cap = cap, S.tau = key_max();
  FT *scratch = yq + (size_t)(1 + w) * P.d;
  u32 nfin = 0, vown = 0;
  int cnt = 0;
  const u32 per = (((P2 + W - 1) / W) + 63u) & ~63u;  // this wave's slice of the sorted prefix [0, P2)
  const u32 s0 = min(P2, (u32)w * per), s1 = min(P2, s0 + per);
  for (u32 base = s0; base < s1; base += ANN_WAVE) {
    const u32 j = base + lane;
    bool direct = false, ok = false;
    Key dk = key_max();
    u32 id = ANN_ID_NONE;
    if (j < s1) {
      if (j < (u32)k) {  // the stage-1 result itself, with the distance it already has
        dk = top[j];
        direct = key_dist(dk) < ft_inf();
      } else {
        const u32 parent = key_id(top[j / k - 1]), z = j % k;
        id = parent < P.n ? P.graph[(size_t)parent * k + z] : (P.graph[z] | P.n);  // supercharge, Q7
        ok = id < P.n && !(alias && id == x) && id >= P.lo && id < P.hi;
      }
    }
    nfin += __popcll(__ballot(direct || ok));
    if (base < (u32)k) {  // wave-uniform: only the first passes of wave 0 hold direct keys
      if (S.kcnt + ANN_WAVE > S.cap) sel_shrink(S);
      const bool push = direct && key_less(dk, S.tau);
      const u64 dm = __ballot(push);
      if (push) S.kbuf[S.kcnt + mask_rank(dm)] = dk;
      S.kcnt += __popcll(dm);
    }
    const u64 mm = __ballot(ok);
    if (ok) list[cnt + mask_rank(mm)] = id;
    cnt += __popcll(mm);
    if (cnt + ANN_WAVE > ANN_S1_CHUNK) {
      wave_lds_sync();
      vown += cnt;
      gather_select<D>(P, list, cnt, alias, x, a, yq, scratch, S, y + (size_t)x * P.d);
      cnt = 0;
    }
  }
  wave_lds_sync();
  vown += cnt;
  gather_select<D>(P, list, cnt, alias, x, a, yq, scratch, S, y + (size_t)x * P.d);

  {  // this wave's survivors -> merge buffer
    const int m = wave_select_smallest(S.kbuf, S.kcnt, K1, S.kout);
    for (int i = lane; i < m; i += ANN_WAVE) mbuf[(size_t)w * K1 + i] = S.kout[i];
    if (lane == 0) {
      mcnt[w] = m;
      atomicAdd(&cnts[0], nfin);
      atomicAdd(&cnts[1], vown);
    }
  }
  __syncthreads();
  if (w == 0) {
    int total = 0;
    for (int ww = 0; ww < W; ww++) {  // cap >= W*K1 (host guarantees)
      const int m = mcnt[ww];
      for (int i = lane; i < m; i += ANN_WAVE) S.kbuf[total + i] = mbuf[(size_t)ww * K1 + i];
      total += m;
    }
    wave_lds_sync();
    const int m = wave_select_smallest(S.kbuf, total, K1, S.kout);
    bool bad = m < k;
    for (int t = lane; t + 1 < m; t += ANN_WAVE)
      if (ft_bits(key_dist(S.kout[t])) == ft_bits(key_dist(S.kout[t + 1]))) bad = true;
    if (m >= k && !(key_dist(S.kout[k - 1]) < ft_inf())) bad = true;
    if (P.L2 > P2 && cnts[0] >= P2 && m < K1) bad = true;
    const bool reject = !P.fixed && __ballot(bad) != 0;
    if (reject) {
      if (lane == 0) {
        flist[atomicAdd(fcount, 1u)] = x;
        if (exact_total) atomicAdd(exact_total, 1ull);
      }
    } else {  // (fixed mode: the k smallest distinct keys in (distance, id) order, (+inf, n) where fewer exist)
      for (int t = lane; t < k; t += ANN_WAVE) {
        out_ids[(size_t)x * k + t] = t < m ? (IdOut)key_id(S.kout[t]) : (IdOut)P.n;
        out_dist[(size_t)x * k + t] = t < m ? key_dist(S.kout[t]) : ft_inf();
      }
    }
    if (rows_done && lane == 0) atomicAdd(&rows_done[(x & 63u) * 8u], (unsigned long long)cnts[1]);
  }
}

// One workgroup per query; its waves split the work on the first P1 slots of the candidate row.  Per wave:
//   A) SEG: for its share of the (try, hamming-neighbour) runs below P1, read the bucket's segment word and copy
//      the owned ids into an LDS list (prefix sum over the lanes' counts, then a balanced copy);
//      !SEG (fallback): read every slot id of its slice and ballot-compact the owned valid ones;
//   B) gather_select on the list whenever it fills, and at the end.
// The waves' survivors are merged by wave 0.  Output per query: K1 = k+1 ascending distinct keys (padded with
// (+inf, ANN_ID_NONE)); nv_tot = valid slots below P1 on ANY device -- with SEG an upper bound that ignores the
// self exclusion (only used for finalize1's "is there an +inf in the prefix" test, where an over-estimate merely
// sends a query to the exact path); nv_own = rows this device gathered.
// Optional fused tail of stage1_select (single-device query path): instead of handing the k+1 candidates to
// finalize1 / row_dists<GRAPH> / exact_select / widen_ids (four more launches, each with its drain and fill), the
// query's own workgroup applies the finalize1 test, and -- unless the query has to take the exact path -- builds the
// stage-2 row (supercharge, compute.cl:252-263), gathers the <= k*k neighbour-of-neighbour rows, runs the reference's
// network on the L2-entry row in LDS and writes the final size_t ids and distances.  Rejected queries are appended to
// `flist`; the host runs the separate exact path for them afterwards (device-driven, normally zero rows).
struct FusedTail {
  int enabled;          // 0: classic path (multi-GPU staged calls, precomp); 1: stage 2 in the workgroup's tail;
                        // 2: finalize1's test in the tail (accepted: top_id/top_dist, rejected: flist) -- no finalize1 launch
  u32 len2;             // Lc2 = ann_need_len(L2, k)
  size_t *out_ids;      // [Q][k]
  FT *out_dist;         // [Q][k]
  u32 *flist, *fcount;  // rejected queries
  unsigned long long *exact_total;
  u32 *top_id;          // enabled == 2: [Q][k]
  FT *top_dist;
};

template <int D, int SEG, bool FUSED>  // SEG: 0 = slot scan, 1 = segment words + table rows, 2 = inline 32-byte records
__global__ __launch_bounds__(256) void stage1_select_kernel(QParams P, int Q, const FT *__restrict__ y,
                                                            int alias, const u32 *__restrict__ codes,
                                                            int K1, int cap, u32 runs_used,
                                                            FT *__restrict__ cand_dist,
                                                            u32 *__restrict__ cand_id,
                                                            u32 *__restrict__ nv_tot,
                                                            u32 *__restrict__ nv_own, FusedTail F,
                                                            Key *__restrict__ cand_key) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  // ---- LDS carve-up (mirrored by stage1_lds_bytes on the host)
  unsigned char *sp = smem;
  Key *kbuf_all = reinterpret_cast<Key *>(sp);           sp += sizeof(Key) * (size_t)W * cap;
  Key *kout_all = reinterpret_cast<Key *>(sp);           sp += sizeof(Key) * (size_t)W * K1;
  Key *mbuf = reinterpret_cast<Key *>(sp);               sp += sizeof(Key) * (size_t)W * K1;
  TryInfo *tries = reinterpret_cast<TryInfo *>(sp);      sp += sizeof(TryInfo) * (size_t)P.T;
  const u32 **rptr_all = reinterpret_cast<const u32 **>(sp);  sp += sizeof(u32 *) * (size_t)W * ANN_WAVE;
  u32 *list_all = reinterpret_cast<u32 *>(sp);           sp += sizeof(u32) * (size_t)W * ANN_S1_CHUNK;
  u32 *pref_all = reinterpret_cast<u32 *>(sp);           sp += sizeof(u32) * (size_t)W * ANN_WAVE;
  u32 *qcode = reinterpret_cast<u32 *>(sp);              sp += sizeof(u32) * (size_t)P.T;
  int *mcnt = reinterpret_cast<int *>(sp);               sp += sizeof(int) * (size_t)W;
  u32 *cnts = reinterpret_cast<u32 *>(sp);               sp += sizeof(u32) * 4;  // [0] valid [1] gathered [2] tail list [3] rejected
  const size_t tl = FUSED ? F.len2 : 0;
  u32 *t_ids = reinterpret_cast<u32 *>(sp);              sp += sizeof(u32) * tl;
  u32 *t_slot = reinterpret_cast<u32 *>(sp);             sp += sizeof(u32) * tl;
  u32 *t_gid = reinterpret_cast<u32 *>(sp);              sp += sizeof(u32) * tl;
  sp = smem + (((sp - smem) + 15) & ~(size_t)15);
  FT *t_dist = reinterpret_cast<FT *>(sp);               sp += sizeof(FT) * tl;
  sp = smem + (((sp - smem) + 15) & ~(size_t)15);
  FT *yq = reinterpret_cast<FT *>(sp);  // generic d only: [d] + W*[d]
  u32 *list = list_all + (size_t)w * ANN_S1_CHUNK;
  u32 *pref = pref_all + (size_t)w * ANN_WAVE;
  const u32 **rptr = rptr_all + (size_t)w * ANN_WAVE;

  for (u32 xi = blockIdx.x; xi < P.qn; xi += gridDim.x) {  // one query per pass (a single pass unless the grid is persistent)
  const u32 x = xi + P.q0;
  if (xi != blockIdx.x) __syncthreads();  // the previous query's LDS state has been consumed
  for (int i = threadIdx.x; i < P.T; i += blockDim.x) {
    tries[i] = P.tries[i];
    qcode[i] = P.fixed ? codes[(size_t)x * P.T + i] : codes[(size_t)i * Q + x];  // Q2: read layout [try][query]
  }
  if (threadIdx.x < 4) cnts[threadIdx.x] = 0;
  if constexpr (D == 0 || OcCode<D>::GEN)
    for (int z = threadIdx.x; z < P.d; z += blockDim.x) yq[z] = y[(size_t)x * P.d + z];
  __syncthreads();

  SelState S;
  S.kbuf = kbuf_all + (size_t)w * cap, S.kout = kout_all + (size_t)w * K1;
  S.kcnt = 0, S.K1 = K1, S.cap = cap, S.tau = key_max();
  FT *scratch = yq + (size_t)(1 + w) * P.d;
  u32 vtot = 0, vown = 0;

  // the query row, as this lane's slice
  VT a[RowChunks<D>::C];
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    const VT *yp = reinterpret_cast<const VT *>(y + (size_t)x * D) + (lane % L::LPR);
#pragma unroll
    for (int c = 0; c < L::C; c++) a[c] = yp[c * L::LPR];
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    const OcLanes<D> ol(P.d, lane);
#pragma unroll
    for (int c = 0; c < OcCode<D>::C; c++) a[c] = oc_load_chunk<D, false>(y + (size_t)x * P.d, ol.p + c * ol.oc, P.d);
  }

  int cnt = 0;
  if constexpr (SEG == 2) {
    const u32 ds1 = (u32)P.ds + 1u;
    const u32 per = (runs_used + W - 1) / W;  // runs of this wave: [r0, r1)
    const u32 r0 = min(runs_used, (u32)w * per), r1 = min(runs_used, r0 + per);
    for (u32 rb = r0; rb < r1; rb += ANN_WAVE) {
      const u32 r = rb + lane;
      u32 c = 0, va = 0;
      uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
      const u32 *src = NULL;
      if (r < r1) {
        const u32 i = r / ds1, yy = r - i * ds1;
        const TryInfo tr = tries[i];
        const u32 b = qcode[i] ^ (yy ? 1u << (yy - 1) : 0u);  // compute_which, compute.cl:243-245
        const uint4 *rec = tr.segx + (size_t)b * 2;
        v0 = rec[0], v1 = rec[1];
        const u32 zs = v0.x & 0xFFu, co = (v0.x >> 8) & 0xFFu, ca = (v0.x >> 16) & 0xFFu;
        const u32 zlim = min(tr.pm, P.P1 - (tr.off + yy * tr.pm));  // slots of this run below P1 (Q1)
        va = min(ca, zlim);
        const u32 ze = min(zs + co, zlim);
        c = ze > zs ? ze - zs : 0u;
        src = tr.tab + (size_t)b * tr.pm + zs;
      }
      vtot += va;
      const u32 incl = wave_incl_scan(c), base = incl - c;
      const u32 total = (u32)__builtin_amdgcn_readfirstlane((int)__shfl(incl, ANN_WAVE - 1));
      // lanes [lo, lo + nfit) are appended per round: as many runs as still fit the list (normally all 64 at once)
      for (u32 lo = 0; lo < ANN_WAVE;) {
        const u32 base_lo = (u32)__builtin_amdgcn_readfirstlane((int)__shfl(base, (int)lo));
        const u32 room = (u32)ANN_S1_CHUNK - (u32)cnt;
        const bool fits = (u32)lane >= lo && incl - base_lo <= room;  // prefix sums are monotone: `fits` is a lane prefix
        const u32 nfit = (u32)__popcll(__ballot(fits));
        if (fits && c) {
          u32 *dst = list + cnt + (base - base_lo);
          if (c > 0) dst[0] = v0.y;
          if (c > 1) dst[1] = v0.z;
          if (c > 2) dst[2] = v0.w;
          if (c > 3) dst[3] = v1.x;
          if (c > 4) dst[4] = v1.y;
          if (c > 5) dst[5] = v1.z;
          if (c > 6) dst[6] = v1.w;
          for (u32 j = ANN_SEGX_INLINE; j < c; j++) dst[j] = src[j];  // rare on a small shard
        }
        const u32 end = lo + nfit;
        const u32 upto = end < ANN_WAVE ? (u32)__builtin_amdgcn_readfirstlane((int)__shfl(base, (int)(end < ANN_WAVE ? end : 0))) : total;
        cnt += (int)(upto - base_lo);
        lo = end;
        if (lo < ANN_WAVE) {  // the next run does not fit any more: drain the list (a run is at most pm <= 255 ids)
          wave_lds_sync();
          vown += cnt;
          gather_select<D>(P, list, cnt, alias, x, a, yq, scratch, S, y + (size_t)x * P.d);
          cnt = 0;
        }
      }
      wave_lds_sync();
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) vtot += __shfl_xor(vtot, m);
  } else if constexpr (SEG == 1) {
    const u32 ds1 = (u32)P.ds + 1u;
    const u32 per = (runs_used + W - 1) / W;  // runs of this wave: [r0, r1)
    const u32 r0 = min(runs_used, (u32)w * per), r1 = min(runs_used, r0 + per);
    for (u32 rb = r0; rb < r1; rb += ANN_WAVE) {
      const u32 r = rb + lane;
      u32 c = 0, va = 0;
      const u32 *src = NULL;
      if (r < r1) {
        const u32 i = r / ds1, yy = r - i * ds1;
        const TryInfo tr = tries[i];
        const u32 b = qcode[i] ^ (yy ? 1u << (yy - 1) : 0u);  // compute_which, compute.cl:243-245
        const uint2 sg = tr.seg[b];
        const u32 zs = sg.x & 0xFFFFu, co = sg.x >> 16, ca = sg.y;
        const u32 zlim = min(tr.pm, P.P1 - (tr.off + yy * tr.pm));  // slots of this run below P1 (Q1)
        va = min(ca, zlim);
        const u32 ze = min(zs + co, zlim);
        c = ze > zs ? ze - zs : 0u;
        src = tr.tab + (size_t)b * tr.pm + zs;
      }
      vtot += va;
      const u32 incl = wave_incl_scan(c);
      const u32 total = __shfl(incl, ANN_WAVE - 1);
      pref[lane] = incl - c;
      rptr[lane] = src;
      wave_lds_sync();
      for (u32 done = 0; done < total;) {  // balanced copy of `total` ids into the list, list-capacity pieces
        const u32 take = min((u32)ANN_S1_CHUNK - (u32)cnt, total - done);
        for (u32 e = done + lane; e < done + take; e += ANN_WAVE) {
          int lo_ = 0, hi_ = ANN_WAVE - 1;  // last run j with pref[j] <= e (it has c_j > 0)
          while (lo_ < hi_) {
            const int mid = (lo_ + hi_ + 1) >> 1;
            if (pref[mid] <= e) lo_ = mid; else hi_ = mid - 1;
          }
          list[cnt + (e - done)] = rptr[lo_][e - pref[lo_]];
        }
        cnt += take, done += take;
        if (cnt == ANN_S1_CHUNK) {
          wave_lds_sync();
          vown += cnt;
          gather_select<D>(P, list, cnt, alias, x, a, yq, scratch, S, y + (size_t)x * P.d);
          cnt = 0;
        }
      }
      wave_lds_sync();
    }
    // one wave-level sum of the per-lane valid counts
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) vtot += __shfl_xor(vtot, m);
  } else {
    // fallback: every slot of this wave's slice of [0, P1)
    const u32 per = (((P.P1 + W - 1) / W) + 63u) & ~63u;
    const u32 s0 = min(P.P1, (u32)w * per), s1 = min(P.P1, s0 + per);
    int ti = 0;
    for (u32 base = s0; base < s1; base += ANN_WAVE) {
      const u32 j = base + lane;
      u32 id = ANN_ID_NONE;
      if (j < s1) id = slot_id(tries, qcode, j, ti);
      const bool ok = id < P.n && !(alias && id == x);
      const bool own = ok && id >= P.lo && id < P.hi;
      vtot += __popcll(__ballot(ok));
      const u64 mm = __ballot(own);
      if (own) list[cnt + mask_rank(mm)] = id;
      cnt += __popcll(mm);
      if (cnt + ANN_WAVE > ANN_S1_CHUNK) {
        wave_lds_sync();
        vown += cnt;
        gather_select<D>(P, list, cnt, alias, x, a, yq, scratch, S, y + (size_t)x * P.d);
        cnt = 0;
      }
    }
  }
  wave_lds_sync();
  vown += cnt;
  gather_select<D>(P, list, cnt, alias, x, a, yq, scratch, S, y + (size_t)x * P.d);

  // ---- this wave's survivors -> merge buffer
  {
    const int m = wave_select_smallest(S.kbuf, S.kcnt, K1, S.kout);
    for (int i = lane; i < m; i += ANN_WAVE) mbuf[(size_t)w * K1 + i] = S.kout[i];
    if (lane == 0) {
      mcnt[w] = m;
      atomicAdd(&cnts[0], vtot);
      atomicAdd(&cnts[1], vown);
    }
  }
  __syncthreads();
  if (w == 0) {
    int total = 0;
    for (int ww = 0; ww < W; ww++) {  // cap >= W*K1 (host guarantees)
      const int m = mcnt[ww];
      for (int i = lane; i < m; i += ANN_WAVE) S.kbuf[total + i] = mbuf[(size_t)ww * K1 + i];
      total += m;
    }
    wave_lds_sync();
    const int m = wave_select_smallest(S.kbuf, total, K1, S.kout);
    if constexpr (!FUSED) {
      if (cand_key) {  // sharded hosts: one packed (dist,id) key per candidate, the unit their exchange moves
        for (int i = lane; i < K1; i += ANN_WAVE) cand_key[(size_t)x * K1 + i] = i < m ? S.kout[i] : key_make(ft_inf(), ANN_ID_NONE);
      } else {
        for (int i = lane; i < K1; i += ANN_WAVE) {
          cand_dist[(size_t)x * K1 + i] = i < m ? key_dist(S.kout[i]) : ft_inf();
          cand_id[(size_t)x * K1 + i] = i < m ? key_id(S.kout[i]) : ANN_ID_NONE;
        }
      }
      if (lane == 0) {
        nv_tot[x] = cnts[0];
        nv_own[x] = cnts[1];  // per-query count; never a same-address atomic from every workgroup (fan-in ~12 ns each)
      }
      if (F.enabled == 2) {  // finalize1's test here (see finalize1_kernel): one launch and its drain less per step
        const int k = K1 - 1;
        bool bad = m < k;
        for (int t = lane; t + 1 < m; t += ANN_WAVE)
          if (ft_bits(key_dist(S.kout[t])) == ft_bits(key_dist(S.kout[t + 1]))) bad = true;
        if (m >= k && !(key_dist(S.kout[k - 1]) < ft_inf())) bad = true;
        if (P.L1 > P.P1 && cnts[0] >= P.P1) bad = true;
        if (__ballot(bad) != 0) {
          if (lane == 0) {
            F.flist[atomicAdd(F.fcount, 1u)] = x;
            if (F.exact_total) atomicAdd(F.exact_total, 1ull);
          }
        } else {
          for (int t = lane; t < k; t += ANN_WAVE) {
            F.top_id[(size_t)x * k + t] = key_id(S.kout[t]);
            F.top_dist[(size_t)x * k + t] = key_dist(S.kout[t]);
          }
        }
      }
    } else {
      // the finalize1 test (see finalize1_kernel): >= k finite distinct keys, no shared distance, an +inf in the prefix
      const int k = K1 - 1;
      bool bad = m < k;
      for (int t = lane; t + 1 < m; t += ANN_WAVE)
        if (ft_bits(key_dist(S.kout[t])) == ft_bits(key_dist(S.kout[t + 1]))) bad = true;
      if (m >= k && !(key_dist(S.kout[k - 1]) < ft_inf())) bad = true;
      if (P.L1 > P.P1 && cnts[0] >= P.P1) bad = true;
      const bool reject = __ballot(bad) != 0;
      if (reject && cand_dist)  // the exact path's tie shortcut (ann_tie.h) wants the candidate list of a rejected query
        for (int i = lane; i < K1; i += ANN_WAVE) {
          cand_dist[(size_t)x * K1 + i] = i < m ? key_dist(S.kout[i]) : ft_inf();
          cand_id[(size_t)x * K1 + i] = i < m ? key_id(S.kout[i]) : ANN_ID_NONE;
        }
      if (lane == 0) {
        cnts[3] = reject ? 1u : 0u;
        if (reject) {
          F.flist[atomicAdd(F.fcount, 1u)] = x;
          atomicAdd(F.exact_total, 1ull);
        }
      }
    }
  }
  if constexpr (FUSED) {
  __syncthreads();
  if (cnts[3]) {  // exact path later; only the statistics are written here
    if (threadIdx.x == 0) nv_own[x] = cnts[1];
    continue;
  }
  // ---- fused stage 2 (det_results second half, alg.c:314-327) on this query's own workgroup
  {
    const u32 cnt2 = stage2_in_workgroup<D, size_t>(P, x, alias, a, yq, scratch, kout_all /* wave 0's sorted survivors */,
                                                    K1 - 1, F.len2, t_ids, t_slot, t_gid, t_dist, &cnts[2], F.out_ids, F.out_dist);
    if (threadIdx.x == 0) nv_own[x] = cnts[1] + cnt2;  // rows gathered for this query, both stages
  }
  }  // FUSED
  }  // queries of this workgroup
}

// ---------------------------------------------------------------------------------- stage1_bucket
// precomp's second_half (alg.c:245-290), bucket-centric.  Every point of a bucket has the same hash code, hence the
// same candidate row (its own bucket + the d_short Hamming-1 buckets, compute.cl:238-246): the per-point kernel
// re-gathers those ~200 rows for each of the ~10 members.  Here one workgroup owns one bucket: it stages the
// candidate rows in LDS tile by tile (read once from HBM) and every wave scores its share of the members against the
// tile, each member keeping its own running selection in LDS.  Same distances, same selection, same outputs as
// stage1_select with alias = 1 (self excluded); nv_tot is exact here (valid slots minus the member itself).
// Host-side conditions (else the per-point kernel is used): power-of-two d, pm members' selection state + one tile
// fit LDS.
// One tile = ANN_BK_TILE_CHUNKS 16-byte chunks of candidate rows (16 KB: 32 rows at d = 128 float).
// What bounds this kernel is the LDS, not HBM (counters, round 2: LDS busy 66 % of the kernel, waves 60 % waiting):
//  * each member's running selection is a wave-resident top-K (ann_device.h: one key per lane, DPP shift insert) --
//    the LDS candidate buffer and its K-pass shrink (132 LDS-crossbar shuffles each) are gone; between tiles the
//    list rests in LDS (K1 keys per member).
// (Double-buffering the tile was tried and is slower: 0.72 -> 0.92 s at cfg3 -- occupancy matters more here.)
#ifndef ANN_BK_TILE_CHUNKS
#define ANN_BK_TILE_CHUNKS 1024
#endif
#ifndef ANN_BK_MEMBERS
#define ANN_BK_MEMBERS 1  // members scored per pass over the tile (measured at cfg3: 1 -> 0.53 s, 2 -> 0.57, 3 -> 0.72, 4 -> 0.77:
                          // fewer LDS reads per distance do not pay once the selection is out of the LDS; balance and occupancy do)
#endif
#ifndef ANN_BK_WAVES
#define ANN_BK_WAVES 4
#endif
#ifndef ANN_BK_WAVES_HI
#define ANN_BK_WAVES_HI 8  // the two-key variant (long lists, few workgroups per CU): cfg5 precomp 40.9 s with 4 waves, 36.5 s with 8
#endif
#define ANN_BK_MAX_RUNS 64
// HI: K1 up to 128, two keys per lane (ann_device.h: wave_topk_insert2).  mgroup: members whose lists fit the LDS at a
// time; a bucket with more members walks the candidate tiles once per group (still members/groups times fewer row
// reads than the per-point kernel: cfg5, K1 = 101, 16-byte keys: groups of ~33 of up to 118 members).
template <int D, bool HI>
__global__ __launch_bounds__(64 * (HI ? ANN_BK_WAVES_HI : ANN_BK_WAVES)) void stage1_bucket_kernel(QParams P, int K1, u32 list_cap, u32 mgroup,
                                                            FT *__restrict__ cand_dist, u32 *__restrict__ cand_id,
                                                            u32 *__restrict__ nv_tot, u32 *__restrict__ nv_own,
                                                            u32 brem, u32 bmod) {
  typedef RowLay<D> L;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const TryInfo tr = P.tries[0];  // one-try view
  const u32 b = blockIdx.x, pm = tr.pm;
  if (bmod > 1 && b % bmod != brem) return;  // sharded precomp: this bucket is scored by another rank
  const u32 members = tr.seg[b].y;  // valid ids of this bucket (in precomp the owned range is everything)
  if (members == 0) return;
  constexpr int CH = D / ANN_VEC;        // chunks per row
  constexpr int ROWV = CH + 1;           // tile row stride in 16-byte units (+1: spreads rows over LDS banks)
  constexpr int TROWS = ANN_BK_TILE_CHUNKS / CH >= 8 ? ANN_BK_TILE_CHUNKS / CH : 8;  // rows per tile (>= one wave pass)
  unsigned char *sp = smem;
  VT *tile = reinterpret_cast<VT *>(sp);                 sp += sizeof(VT) * (size_t)TROWS * ROWV;
  Key *klist = reinterpret_cast<Key *>(sp);              sp += sizeof(Key) * (size_t)mgroup * K1;  // per member of the group: its K1 best so far
  u32 *clist = reinterpret_cast<u32 *>(sp);              sp += sizeof(u32) * (size_t)list_cap;  // candidate ids, slot order
  u32 *roff = reinterpret_cast<u32 *>(sp);               sp += sizeof(u32) * (ANN_BK_MAX_RUNS + 1);
  const u32 *mem_ids = tr.tab + (size_t)b * pm;  // members, descending ids
  const u32 ds1 = (u32)P.ds + 1u;
  // valid ids of every run below P1 (the first seg.y entries of the neighbour bucket's row): counts, then offsets
  if (threadIdx.x < ds1) {
    const u32 yy = threadIdx.x, start = yy * pm;
    const u32 nb = b ^ (yy ? 1u << (yy - 1) : 0u);
    roff[yy + 1] = start < P.P1 ? min(tr.seg[nb].y, min(pm, P.P1 - start)) : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    roff[0] = 0;
    for (u32 yy = 0; yy < ds1; yy++) roff[yy + 1] += roff[yy];
  }
  __syncthreads();
  const u32 total = roff[ds1];  // valid slots below P1: the same for every member
  for (u32 e = threadIdx.x; e < ds1 * pm; e += blockDim.x) {
    const u32 yy = e / pm, z = e - yy * pm;
    if (z < roff[yy + 1] - roff[yy]) {
      const u32 nb = b ^ (yy ? 1u << (yy - 1) : 0u);
      clist[roff[yy] + z] = tr.tab[(size_t)nb * pm + z];
    }
  }

  const int p = lane % L::LPR, g = lane / L::LPR;
  for (u32 g0 = 0; g0 < members; g0 += mgroup) {
  const u32 gmem = min(mgroup, members - g0);  // members [g0, g0 + gmem) this round
  __syncthreads();  // clist complete / the previous group's results written
  for (u32 e = threadIdx.x; e < gmem * (u32)K1; e += blockDim.x) klist[e] = key_max();
  for (u32 r0 = 0; r0 < total; r0 += TROWS) {
    const u32 rows = min((u32)TROWS, total - r0);
    __syncthreads();  // previous tile fully consumed (first tile: the lists are initialised)
    for (u32 e = threadIdx.x; e < rows * CH; e += blockDim.x) {  // whole 128-byte pieces per load instruction
      const u32 r = e / CH, c = e - r * CH;
      tile[(size_t)r * ROWV + c] = load_row_chunk<true>(reinterpret_cast<const VT *>(P.points + (size_t)clist[r0 + r] * D) + c);
    }
    __syncthreads();
    // every wave scores its members, NM at a time, against the tile (a missing member of the last group repeats the
    // group's first one and is masked out)
    constexpr int NM = ANN_BK_MEMBERS;
    for (u32 m0 = (u32)NM * w; m0 < gmem; m0 += (u32)NM * W) {
      u32 mi[NM], xi[NM];
      VT a[NM][L::C];
      Key mine[NM], hi[NM], tau[NM];
#pragma unroll
      for (int j = 0; j < NM; j++) {
        mi[j] = m0 + j < gmem ? m0 + j : m0;
        xi[j] = mem_ids[g0 + mi[j]];
        const VT *yp = reinterpret_cast<const VT *>(P.points + (size_t)xi[j] * D) + p;
#pragma unroll
        for (int c = 0; c < L::C; c++) a[j][c] = yp[c * L::LPR];
        mine[j] = lane < K1 ? klist[(size_t)mi[j] * K1 + lane] : key_max();
        hi[j] = (HI && ANN_WAVE + lane < K1) ? klist[(size_t)mi[j] * K1 + ANN_WAVE + lane] : key_max();
        tau[j] = HI ? wave_topk_kth2(mine[j], hi[j], K1) : key_readlane(mine[j], K1 - 1);
      }
      for (u32 base = 0; base < rows; base += L::RPW) {
        const u32 r = base + g;
        const bool inb = r < rows;
        const u32 id = clist[r0 + (inb ? r : base)];
        VT bv[L::C];
        const VT *tp = tile + (size_t)(inb ? r : base) * ROWV + p;
#pragma unroll
        for (int c = 0; c < L::C; c++) bv[c] = tp[c * L::LPR];
        const bool ok = inb && p == 0;
#pragma unroll
        for (int j = 0; j < NM; j++) {
          const Key key = key_make(row_reduce<D, ROW_SQDIFF>(a[j], bv), id);
          const u64 mm = __ballot(ok && (j == 0 || m0 + j < gmem) && id != xi[j] && key_less(key, tau[j]));
          if (mm) {
            if constexpr (HI) wave_topk_offer2(mine[j], hi[j], tau[j], key, mm, K1);
            else wave_topk_offer(mine[j], tau[j], key, mm, K1);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NM; j++)
        if (j == 0 || m0 + j < gmem) {
          if (lane < K1) klist[(size_t)mi[j] * K1 + lane] = mine[j];
          if (HI && ANN_WAVE + lane < K1) klist[(size_t)mi[j] * K1 + ANN_WAVE + lane] = hi[j];
        }
    }
  }
  __syncthreads();
  // results: each member's list is already ascending and distinct
  for (u32 e = threadIdx.x; e < gmem * (u32)K1; e += blockDim.x) {
    const u32 m = e / K1, i = e - m * K1, x = mem_ids[g0 + m];
    const Key kk = klist[e];
    const bool have = !key_eq(kk, key_max());
    cand_dist[(size_t)x * K1 + i] = have ? key_dist(kk) : ft_inf();
    cand_id[(size_t)x * K1 + i] = have ? key_id(kk) : ANN_ID_NONE;
    if (i == 0) {
      nv_tot[x] = total - 1u;  // the member itself sits in its own bucket's run, below P1
      nv_own[x] = total - 1u;
    }
  }
  }  // member groups
}

// -------------------------------------------------------------------------------------- finalize1
// Decide, per query, whether the K1 = k+1 smallest distinct keys determine the reference's stage-1
// output.  They do when (a) at least k of them exist and are finite, (b) no two of them share a distance
// (a tie between different ids is ordered by the network), and (c) the sorted prefix holds at
// least one +inf entry or the row ends at P1 (otherwise the duplicate test at P1-1 reads slot P1's id).
// Then the output is simply the first k keys.  Otherwise the query is appended to `flist`.
#define ANN_ID_SKIP 0xFFFFFFFDu  // cand_id[x][0] of a row another rank of a sharded precomp is responsible for
__global__ void mark_rows_kernel(size_t rows, u32 stride, u32 value, u32 *out) {
  for (size_t x = (size_t)blockIdx.x * blockDim.x + threadIdx.x; x < rows; x += (size_t)gridDim.x * blockDim.x)
    out[x * stride] = value;
}
__global__ void finalize1_kernel(int Q, int k, int K1, u32 L1, u32 P1, const FT *__restrict__ cand_dist,
                                 const u32 *__restrict__ cand_id, const u32 *__restrict__ nv_tot,
                                 u32 *__restrict__ top_id, FT *__restrict__ top_dist, int ostride,
                                 int ooff, u32 *__restrict__ flist, u32 *__restrict__ fcount,
                                 unsigned long long *__restrict__ exact_total, u32 fixed_n = 0) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= Q) return;
  const FT *cd = cand_dist + (size_t)x * K1;
  const u32 *ci = cand_id + (size_t)x * K1;
  if (ci[0] == ANN_ID_SKIP) return;  // not this rank's row: neither accepted nor flagged
  bool flag = (u32)k > P1 || K1 != k + 1;
  int m = 0;
  while (m < K1 && ci[m] != ANN_ID_NONE) m++;
  if (fixed_n) {  // fixed mode (fixed_n = n): the k smallest distinct keys as they are, (+inf, n) where fewer exist
    for (int t = 0; t < k; t++) {
      top_id[(size_t)x * ostride + ooff + t] = t < m ? ci[t] : fixed_n;
      top_dist[(size_t)x * ostride + ooff + t] = t < m ? cd[t] : ft_inf();
    }
    return;
  }
  if (m < k) flag = true;
  if (!flag) {
    if (!(cd[k - 1] < ft_inf())) flag = true;
    for (int t = 0; t + 1 < m; t++)
      if (ft_bits(cd[t]) == ft_bits(cd[t + 1])) flag = true;
    if (L1 > P1 && nv_tot[x] >= P1) flag = true;
  }
  if (flag) {
    flist[atomicAdd(fcount, 1u)] = (u32)x;
    if (exact_total) atomicAdd(exact_total, 1ull);
  } else {
    for (int t = 0; t < k; t++) {
      top_id[(size_t)x * ostride + ooff + t] = ci[t];
      top_dist[(size_t)x * ostride + ooff + t] = cd[t];
    }
  }
}

// -------------------------------------------------------------------------------------- row_dists
// Ids and distances of the first `len` slots of a row, in slot order; one workgroup per row.
//   MODE_TABLE: stage-1 row of query x = qidx[blockIdx.x]            (len = Lc1)
//   MODE_GRAPH: stage-2 row: slots [0,k) = current top-k with their distances, slot (y+1)k+z = z-th
//               graph neighbour of top[y] (sentinel parents give graph[0][z] | n, Q7)   (len = Lc2)
// Slots this device does not own, sentinels and the excluded self row get +inf; a multi-GPU caller
// min-reduces the distance rows across devices before exact_select.
//   MODE_GRAPH_DIST: stage-2 row of a point-sharded host: distances of slots [k, len) only, no ids (the query's owner
//               derives them itself), row stride len - k; queries whose top-k is flagged (ANN_ID_FLAG) are skipped
//               and appended to flist
enum { MODE_TABLE = 0, MODE_GRAPH = 1, MODE_GRAPH_DIST = 2 };
#define ANN_ID_FLAG 0xFFFFFFFEu  // top_id[x][0] of a query whose stage 1 has to be redone by the exact path
#define ANN_RD_CHUNK 2048

template <int D, int MODE>
__global__ __launch_bounds__(256) void row_dists_kernel(QParams P, int Q, const FT *__restrict__ y,
                                                        int alias, const u32 *__restrict__ codes,
                                                        const u32 *__restrict__ qidx, u32 xbase,
                                                        u32 len, const u32 *__restrict__ top_id,
                                                        const FT *__restrict__ top_dist,
                                                        u32 *__restrict__ ids_out,
                                                        FT *__restrict__ dist_out,
                                                        unsigned long long *__restrict__ rows_done,
                                                        const u32 *__restrict__ live_rows, u32 nrows,
                                                        u32 chunk, u32 live_off, u32 flat_split) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // Device-driven launches (live_rows != NULL) use a small persistent grid that walks the device-side row
  // count: launching one workgroup per POSSIBLE row just to exit cost ~50 us per launch at Q = 10k.
  // live_off: this launch covers entries [live_off, live_off + nrows) of the counted list (bounded workspace).
  if (live_rows) nrows = min(nrows, max(*live_rows, live_off) - live_off);
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  // flat_split > 0 (device-driven launches of long rows): a ONE-dimensional persistent grid over (row, part) items,
  // part = which flat_split-th of the row's chunks -- a single flagged row still spreads over flat_split workgroups, a
  // thousand of them keep the whole grid busy, and the launch that finds nothing to do is a few hundred workgroups.
  const u32 nparts = flat_split ? flat_split : gridDim.y;
  const u32 nitems = flat_split ? nrows * flat_split : nrows;
  for (u32 item = blockIdx.x; item < nitems; item += gridDim.x) {
  const u32 row = flat_split ? item / flat_split : item;
  const u32 part = flat_split ? item % flat_split : blockIdx.y;
  const u32 x = qidx ? qidx[row] : xbase + row;
  if constexpr (MODE == MODE_GRAPH_DIST) {
    if (top_id[(size_t)x * P.k] == ANN_ID_FLAG) {  // workgroup-uniform
      if (threadIdx.x == 0 && part == 0) ids_out[1 + atomicAdd(&ids_out[0], 1u)] = x;  // ids_out = {count, list...}
      continue;
    }
  }
  constexpr bool DIST_ONLY = MODE == MODE_GRAPH_DIST;
  const u32 skip = DIST_ONLY ? (u32)P.k : 0u;  // columns [0, skip) are not stored
  u32 *ids_row = DIST_ONLY ? NULL : ids_out + (size_t)row * len;
  FT *dist_row = dist_out + (size_t)row * (len - skip) - skip;
  unsigned char *sp = smem;
  u32 *lslot = reinterpret_cast<u32 *>(sp);           sp += sizeof(u32) * chunk;  // chunk <= ANN_RD_CHUNK
  u32 *lid = reinterpret_cast<u32 *>(sp);             sp += sizeof(u32) * chunk;
  TryInfo *tries = reinterpret_cast<TryInfo *>(sp);   sp += sizeof(TryInfo) * (size_t)P.T;
  u32 *qcode = reinterpret_cast<u32 *>(sp);           sp += sizeof(u32) * (size_t)P.T;
  u32 *lcount = reinterpret_cast<u32 *>(sp);          sp += sizeof(u32) * 4;
  sp = smem + (((sp - smem) + 15) & ~(size_t)15);
  FT *yq = reinterpret_cast<FT *>(sp);

  if (MODE == MODE_TABLE)
    for (int i = threadIdx.x; i < P.T; i += blockDim.x) {
      tries[i] = P.tries[i];
      qcode[i] = codes[(size_t)i * Q + x];
    }
  if constexpr (D == 0 || OcCode<D>::GEN)
    for (int z = threadIdx.x; z < P.d; z += blockDim.x) yq[z] = y[(size_t)x * P.d + z];
  VT a[RowChunks<D>::C];
  if constexpr (D > 0) {
    typedef RowLay<D> L;
    const VT *yp = reinterpret_cast<const VT *>(y + (size_t)x * D) + (lane % L::LPR);
#pragma unroll
    for (int c = 0; c < L::C; c++) a[c] = yp[c * L::LPR];
  } else if constexpr (D < 0 && !OcCode<D>::GEN) {
    const OcLanes<D> ol(P.d, lane);
#pragma unroll
    for (int c = 0; c < OcCode<D>::C; c++) a[c] = oc_load_chunk<D, false>(y + (size_t)x * P.d, ol.p + c * ol.oc, P.d);
  }
  u32 gathered = 0;
  // nparts workgroups share one row: each takes every nparts-th chunk of its slots
  for (u32 c0 = part * chunk; c0 < len; c0 += nparts * chunk) {
    const u32 c1 = min(len, c0 + chunk);
    if (threadIdx.x == 0) lcount[0] = 0;
    __syncthreads();
    for (u32 j = c0 + threadIdx.x; j < c1; j += blockDim.x) {
      u32 id;
      bool given = false;
      if (MODE == MODE_TABLE) {
        int ti = 0;
        id = slot_id(tries, qcode, j, ti);
      } else {
        const u32 k = P.k;
        if (j < k) {
          if (DIST_ONLY) continue;
          id = top_id[(size_t)x * k + j];
          given = true;
        } else {
          const u32 parent = top_id[(size_t)x * k + (j / k - 1)];
          const u32 z = j % k;
          id = parent < P.n ? P.graph[(size_t)parent * k + z] : (P.graph[z] | P.n);
        }
      }
      if (!DIST_ONLY) ids_row[j] = id;
      if (given) {
        dist_row[j] = top_dist[(size_t)x * P.k + j];
      } else {
        const bool own = id < P.n && !(alias && id == x) && id >= P.lo && id < P.hi;
        if (own) {
          const u32 pos = atomicAdd(&lcount[0], 1u);
          lslot[pos] = j;
          lid[pos] = id;
        } else {
          dist_row[j] = ft_inf();
        }
      }
    }
    __syncthreads();
    const int cnt = (int)lcount[0];
    gathered += cnt;
    if constexpr (D > 0) {
      typedef RowLay<D> L;
      constexpr int U = (ANN_ROWS_INFLIGHT / L::C) > 0 ? (ANN_ROWS_INFLIGHT / L::C) : 1;  // row passes in flight (see stage2_in_workgroup)
      const int p = lane % L::LPR, g = lane / L::LPR;
      for (int base0 = w * L::RPW; base0 < cnt; base0 += W * L::RPW * U) {
        VT b[U][L::C];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int r = base0 + u * W * L::RPW + g;
          if (base0 + u * W * L::RPW < cnt) {  // wave-uniform
            const u32 id = lid[r < cnt ? r : base0];
            const VT *rp = reinterpret_cast<const VT *>(P.points + (size_t)(id - P.lo) * D) + p;
#pragma unroll
            for (int c = 0; c < L::C; c++) b[u][c] = load_row_chunk<true>(rp + c * L::LPR);
          }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int r = base0 + u * W * L::RPW + g;
          if (base0 + u * W * L::RPW < cnt) {
            const FT dist = row_reduce<D, ROW_SQDIFF>(a, b[u]);
            if (r < cnt && p == 0) dist_row[lslot[r]] = dist;
          }
        }
      }
    } else if constexpr (D < 0 && !OcCode<D>::GEN) {
      constexpr int C = OcCode<D>::C, OC = OcCode<D>::OC;
      const OcLanes<D> ol(P.d, lane);
      const int oc = ol.oc, rpw = ol.rpw, g = ol.g, p = ol.p;
      for (int base = w * rpw; base < cnt; base += W * rpw) {
        const int r = base + g;
        const bool act = ol.valid && r < cnt;
        const u32 id = lid[act ? r : base];
        const FT *rp = P.points + (size_t)(id - P.lo) * P.d;
        VT b[C];
#pragma unroll
        for (int c = 0; c < C; c++) b[c] = oc_load_chunk<D, OcCode<D>::NT_ROWS>(rp, p + c * oc, P.d);
        const FT dist = row_reduce_oc<C, ROW_SQDIFF, OC>(a, b, oc, p, oc_tree_len<D>(P.d));
        if (act && p == 0) dist_row[lslot[r]] = dist;
      }
    } else {
      FT *m = yq + (size_t)(1 + w) * P.d;
      for (int r = w; r < cnt; r += W) {
        const u32 id = lid[r];
        const FT dist = row_reduce_generic<ROW_SQDIFF>(P.d, yq, P.points + (size_t)(id - P.lo) * P.d, m);
        if (lane == 0) dist_row[lslot[r]] = dist;
      }
    }
    __syncthreads();
  }
  // statistics: 64 counters on separate 64-byte lines; one counter for every workgroup would serialise the
  // launch behind ~12 ns per atomic (measured: +110 us on a 10k-row launch)
  if (rows_done && threadIdx.x == 0) atomicAdd(&rows_done[(row & 63u) * 8u], (unsigned long long)gathered);
  __syncthreads();  // LDS is re-used by the next row of this workgroup
  }
}

// ------------------------------------------------------------------------------- stage2_dist_multi
// row_dists<MODE_GRAPH_DIST> for many queries per workgroup (sharded hosts, power-of-two d).  One workgroup per
// query is mostly overhead there: a rank owns 1/G of the rows, i.e. ~7 of a cfg3 query's 55 stage-2 slots at G = 8,
// behind three dependent round trips (top ids -> graph row -> point rows) -- 80k workgroups, 0.10 ms for 47 us worth of
// rows.  Here a 256-thread workgroup takes QPB queries: their (len - k) * QPB slots are derived side by side (the
// loads of different queries are independent), the owned ones of ALL of them go through one gather loop.
// Same outputs as row_dists<MODE_GRAPH_DIST>: dist_out[x][len - k] (+inf for slots this device does not own, sentinels,
// the excluded self row), flagged queries skipped and appended to flagged = {count, list...}.
#define ANN_S2M_CAP 2048  // slots per workgroup (LDS lists)
template <int D>
__global__ __launch_bounds__(256) void stage2_dist_multi_kernel(QParams P, int Q, const FT *__restrict__ y, int alias,
                                                                u32 len, u32 qpb, const u32 *__restrict__ top_id,
                                                                FT *__restrict__ dist_out, u32 *__restrict__ flagged,
                                                                unsigned long long *__restrict__ rows_done) {
  static_assert(D > 0, "power-of-two row layout only");
  typedef RowLay<D> L;
  __shared__ u32 lq[ANN_S2M_CAP], lslot[ANN_S2M_CAP], lid[ANN_S2M_CAP];  // owned slots: local query, slot, point id
  __shared__ u32 lcount;
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const u32 k = (u32)P.k, per = len - k;
  const u32 x0 = blockIdx.x * qpb, nq = min(qpb, (u32)Q - x0);
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  for (u32 e = threadIdx.x; e < nq * per; e += blockDim.x) {
    const u32 ql = e / per, j = k + (e - ql * per), x = x0 + ql;
    const u32 first = top_id[(size_t)x * k];
    if (first == ANN_ID_FLAG) {  // the exact path redoes this query: listed once, no distances
      if (j == k) flagged[1 + atomicAdd(&flagged[0], 1u)] = x;
      continue;
    }
    const u32 parent = top_id[(size_t)x * k + (j / k - 1)];
    const u32 z = j % k;
    const u32 id = parent < P.n ? P.graph[(size_t)parent * k + z] : (P.graph[z] | P.n);  // supercharge, Q7
    const bool own = id < P.n && !(alias && id == x) && id >= P.lo && id < P.hi;
    if (own) {
      const u32 pos = atomicAdd(&lcount, 1u);
      lq[pos] = ql, lslot[pos] = j - k, lid[pos] = id;
    } else {
      dist_out[(size_t)x * per + (j - k)] = ft_inf();
    }
  }
  __syncthreads();
  const int cnt = (int)lcount;
  const int p = lane % L::LPR, g = lane / L::LPR;
  constexpr int U = (4 / L::C) > 0 ? (4 / L::C) : 1;  // row passes in flight
  for (int base0 = w * L::RPW; base0 < cnt; base0 += W * L::RPW * U) {
    VT a[U][L::C], b[U][L::C];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = base0 + u * W * L::RPW + g;
      if (base0 + u * W * L::RPW < cnt) {  // wave-uniform
        const int rr = r < cnt ? r : base0;
        const VT *yp = reinterpret_cast<const VT *>(y + (size_t)(x0 + lq[rr]) * D) + p;
        const VT *rp = reinterpret_cast<const VT *>(P.points + (size_t)(lid[rr] - P.lo) * D) + p;
#pragma unroll
        for (int c = 0; c < L::C; c++) a[u][c] = yp[c * L::LPR], b[u][c] = load_row_chunk<true>(rp + c * L::LPR);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int r = base0 + u * W * L::RPW + g;
      if (base0 + u * W * L::RPW < cnt) {
        const FT dist = row_reduce<D, ROW_SQDIFF>(a[u], b[u]);
        if (r < cnt && p == 0) dist_out[(size_t)(x0 + lq[r]) * per + lslot[r]] = dist;
      }
    }
  }
  if (rows_done && threadIdx.x == 0) atomicAdd(&rows_done[(blockIdx.x & 63u) * 8u], (unsigned long long)cnt);
}

// ----------------------------------------------------------------------------------- exact_select
// sort_and_uniq (alg.c:224-230) on `len` stored entries of a row of reference length L, then the first k
// entries out.  One workgroup per row.  USE_LDS: the row is staged in LDS, otherwise the network runs in
// place in global memory (rows too long for LDS; the workgroup owns the row).
// Tie path (ann_tie.h): when stage 1's candidate list of the row's query is given, wave 0 first tries to derive the
// result from the class bits of the row; only rows it cannot take run the network.  NW = 64-bit words per lane
// (P <= 4096 * NW), 0 = no tie path in this instantiation.
struct TieArgs {
  const FT *cand_d;   // [Q][K1] ascending distinct keys of stage 1 (NULL: no tie path)
  const u32 *cand_i;
  int K1;
  unsigned long long *resolved;  // statistics: rows answered by the tie path
  int derive;         // cand_d == NULL: derive the list from the row itself (tie_derive_list)
};
template <bool USE_LDS, int NW>
__global__ __launch_bounds__(1024) void exact_select_kernel(u32 L, u32 len, u32 in_stride, int k,
                                                           u32 *__restrict__ ids_in,
                                                           FT *__restrict__ dist_in,
                                                           const u32 *__restrict__ qidx, u32 xbase,
                                                           u32 *__restrict__ out_id,
                                                           FT *__restrict__ out_dist, int ostride,
                                                           int ooff, const u32 *__restrict__ live_rows,
                                                           u32 nrows, size_t *__restrict__ out64, u32 live_off, TieArgs T) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (live_rows) nrows = min(nrows, max(*live_rows, live_off) - live_off);  // persistent grid over the device-side row count
  for (u32 row = blockIdx.x; row < nrows; row += gridDim.x) {
  const u32 x = qidx ? qidx[row] : xbase + row;
  u32 *gi = ids_in + (size_t)row * in_stride;
  FT *gd = dist_in + (size_t)row * in_stride;
  if constexpr (NW > 0) {
    if (T.cand_d || T.derive) {  // workgroup-uniform
      const size_t o = (size_t)x * ostride + ooff;
      const FT *cd = T.cand_d ? T.cand_d + (size_t)x * T.K1 : NULL;
      const u32 *ci = T.cand_d ? T.cand_i + (size_t)x * T.K1 : NULL;
      if (!T.cand_d && L >= 16 && T.K1 <= ANN_WAVE) {
        FT *dcd;
        u32 *dci;
        tie_derive_list((u32)1 << ann_lg(L), T.K1, gi, gd, smem + ann_tie_lds_bytes(NW), &dcd, &dci);
        cd = dcd, ci = dci;
      }
      const bool done = cd && tie_resolve<NW>(L, k, T.K1, gi, gd, cd, ci, smem,
                                              out64 ? NULL : out_id + o, out64 ? out64 + o : NULL, out_dist + o, T.resolved);
      if (done) {
        if (threadIdx.x == 0 && T.resolved) atomicAdd(T.resolved, 1ull);
        continue;
      }
    }
  }
  if (USE_LDS) {
    FT *sd = reinterpret_cast<FT *>(smem);
    u32 *si = reinterpret_cast<u32 *>(sd + len);
    for (u32 j = threadIdx.x; j < len; j += blockDim.x) sd[j] = gd[j], si[j] = gi[j];
    __syncthreads();
    block_topk_stage<true>(L, len, sd, si);
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
      if (out64) out64[(size_t)x * ostride + ooff + t] = si[t];  // straight to the ABI's size_t ids
      else out_id[(size_t)x * ostride + ooff + t] = si[t];
      out_dist[(size_t)x * ostride + ooff + t] = sd[t];
    }
  } else {
    block_topk_stage<false>(L, len, gd, gi);
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
      if (out64) out64[(size_t)x * ostride + ooff + t] = gi[t];
      else out_id[(size_t)x * ostride + ooff + t] = gi[t];
      out_dist[(size_t)x * ostride + ooff + t] = gd[t];
    }
  }
  __syncthreads();
  }
}

// ---------------------------------------------------------------- point-sharded hosts, owner protocol
// Queries are dealt to OWNER devices in contiguous slices of qs.  After the all-to-all of stage-1 candidates the
// owner holds, for each of its nq queries, G ascending lists of K1 distinct packed keys (ids disjoint across
// devices): in[g][xl][K1], xl = query - qbase.  One thread per query merges them to the K1 globally smallest and
// applies finalize1's proof (see finalize1_kernel).  Accepted: top_id/top_dist[xl][0..k).  Rejected:
// top_id[xl][0] = ANN_ID_FLAG -- every device sees the flag after the all-gather of top_id and joins the exact path.
__global__ __launch_bounds__(256) void merge_finalize_kernel(int G, int nq, u32 qbase, u32 qs, int K1, int k, u32 L1, u32 P1,
                                                             const Key *__restrict__ in, const u32 *__restrict__ nv_tot,
                                                             u32 *__restrict__ top_id, FT *__restrict__ top_dist,
                                                             unsigned long long *__restrict__ exact_total) {
  // one wave per query: all G*K1 keys are fetched at once into LDS (no dependent loads), then the K1 smallest are
  // selected by the same wave-level selection stage 1 uses
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  const u32 xl = blockIdx.x * W + w;
  if (xl >= qs) return;  // whole wave
  u32 *ti = top_id + (size_t)xl * k;
  FT *td = top_dist + (size_t)xl * k;
  if (xl >= (u32)nq) {  // padding slot of the last slice: defined content for the collectives, never read as a query
    for (int t = lane; t < k; t += ANN_WAVE) ti[t] = ANN_ID_NONE, td[t] = ft_inf();
    return;
  }
  Key *buf = reinterpret_cast<Key *>(smem) + (size_t)w * (G + 1) * K1, *out = buf + (size_t)G * K1;
  const int tot = G * K1;
  for (int e = lane; e < tot; e += ANN_WAVE) {
    const int g = e / K1, t = e - g * K1;
    buf[e] = in[((size_t)g * qs + xl) * K1 + t];  // padding keys (+inf, NONE) sort last and are dropped below
  }
  wave_lds_sync();
  int m = wave_select_smallest(buf, tot, K1, out);
  while (m > 0 && key_id(out[m - 1]) == ANN_ID_NONE) m--;  // wave-uniform (LDS reads of the same address)
  bool bad = (u32)k > P1 || K1 != k + 1 || m < k;
  for (int t = lane; t + 1 < m; t += ANN_WAVE)
    if (ft_bits(key_dist(out[t])) == ft_bits(key_dist(out[t + 1]))) bad = true;  // two ids at one distance (Q17)
  if (m >= k && !(key_dist(out[k - 1]) < ft_inf())) bad = true;
  if (L1 > P1 && nv_tot[qbase + xl] >= P1) bad = true;
  const bool flag = __ballot(bad) != 0;
  for (int t = lane; t < k; t += ANN_WAVE) {
    ti[t] = (flag && t == 0) ? ANN_ID_FLAG : (t < m ? key_id(out[t]) : ANN_ID_NONE);
    td[t] = t < m ? key_dist(out[t]) : ft_inf();
  }
  if (flag && lane == 0 && exact_total) atomicAdd(exact_total, 1ull);
}

// Flagged queries in ascending order (every device derives the identical list from the all-gathered top ids):
// flist = {n_listed = min(total, cap), total, list[cap]}.  Two passes over chunks of ANN_FLAG_CHUNK queries: per-chunk counts,
// then every chunk places its flagged queries behind the chunks before it (ordered ballot compaction).
#define ANN_FLAG_CHUNK 256  // small workgroups: these run beside the gather of another batch
__global__ __launch_bounds__(ANN_FLAG_CHUNK) void flag_count_kernel(int Q, int k, const u32 *__restrict__ top_all,
                                                                   u32 *__restrict__ chunk_cnt) {
  __shared__ u32 total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const int x = blockIdx.x * ANN_FLAG_CHUNK + threadIdx.x;
  const bool f = x < Q && top_all[(size_t)x * k] == ANN_ID_FLAG;
  const u64 m = __ballot(f);
  if (lane_id() == 0 && m) atomicAdd(&total, (u32)__popcll(m));
  __syncthreads();
  if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = total;
}
__global__ __launch_bounds__(ANN_FLAG_CHUNK) void flag_place_kernel(int Q, int k, const u32 *__restrict__ top_all,
                                                                   const u32 *__restrict__ chunk_cnt, u32 cap,
                                                                   u32 *__restrict__ flist) {
  __shared__ u32 wsum[ANN_FLAG_CHUNK / ANN_WAVE];
  __shared__ u32 before_s, all_s;
  const int lane = lane_id(), w = threadIdx.x >> 6;
  if (threadIdx.x == 0) before_s = 0, all_s = 0;
  __syncthreads();
  u32 mine = 0, every = 0;
  for (u32 c = threadIdx.x; c < gridDim.x; c += blockDim.x) {
    const u32 v = chunk_cnt[c];
    every += v;
    if (c < blockIdx.x) mine += v;
  }
#pragma unroll
  for (int s_ = 32; s_ >= 1; s_ >>= 1) mine += __shfl_xor(mine, s_), every += __shfl_xor(every, s_);
  if (lane == 0 && every) atomicAdd(&before_s, mine), atomicAdd(&all_s, every);
  const int x = blockIdx.x * ANN_FLAG_CHUNK + threadIdx.x;
  const bool f = x < Q && top_all[(size_t)x * k] == ANN_ID_FLAG;
  const u64 m = __ballot(f);
  if (lane == 0) wsum[w] = (u32)__popcll(m);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) flist[0] = min(all_s, cap), flist[1] = all_s;
  if (f) {
    u32 pos = before_s + mask_rank(m);
    for (int ww = 0; ww < w; ww++) pos += wsum[ww];
    if (pos < cap) flist[2 + pos] = (u32)x;
  }
}

// After the exact path rewrote the flagged rows of top_all / top_d_all (rows indexed by query): the owner copies
// its own queries' rows into its slices (top_id / top_dist [qs][k]) for the final step.
__global__ void patch_owner_kernel(const u32 *__restrict__ flist, u32 qbase, u32 qs, int k, const u32 *__restrict__ top_all,
                                   const FT *__restrict__ top_d_all, u32 *__restrict__ top_id, FT *__restrict__ top_dist) {
  const u32 nl = flist[0];
  for (u32 e = blockIdx.x * blockDim.x + threadIdx.x; e < nl * (u32)k; e += gridDim.x * blockDim.x) {
    const u32 x = flist[2 + e / k], t = e % k;
    if (x >= qbase && x < qbase + qs) {
      top_id[(size_t)(x - qbase) * k + t] = top_all[(size_t)x * k + t];
      top_dist[(size_t)(x - qbase) * k + t] = top_d_all[(size_t)x * k + t];
    }
  }
}

// The owner's final step: stage-2 row of query qbase+xl = its top-k (ids, distances) followed by the neighbours of
// the top-k (supercharge, compute.cl:252-263; ids derived here, distances = min over the G devices' partial rows
// in2[g][xl][len-k], +inf where a device does not own the slot), then the reference's network + rdups + network
// (alg.c:224-230) in LDS; first k entries out.  Flagged queries are left to the repair pass.
__global__ __launch_bounds__(1024) void final_select_kernel(int G, int nq, u32 qbase, u32 qs, u32 n, int k, u32 L, u32 len,
                                                           const u32 *__restrict__ graph,
                                                           const u32 *__restrict__ top_id,
                                                           const FT *__restrict__ top_dist,
                                                           const FT *__restrict__ in2, u32 *__restrict__ out_id,
                                                           FT *__restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  FT *sd = reinterpret_cast<FT *>(smem);
  u32 *si = reinterpret_cast<u32 *>(sd + len);
  const u32 w2 = len - (u32)k;
  for (u32 xl = blockIdx.x; xl < qs; xl += gridDim.x) {
    const u32 *ti = top_id + (size_t)xl * k;
    if (xl >= (u32)nq || ti[0] == ANN_ID_FLAG) {  // workgroup-uniform
      for (int t = threadIdx.x; t < k; t += blockDim.x) {
        out_id[(size_t)xl * k + t] = xl >= (u32)nq ? ANN_ID_NONE : ANN_ID_FLAG;
        out_dist[(size_t)xl * k + t] = ft_inf();
      }
      continue;
    }
    for (u32 j = threadIdx.x; j < len; j += blockDim.x) {
      if (j < (u32)k) {
        si[j] = ti[j], sd[j] = top_dist[(size_t)xl * k + j];
      } else {
        const u32 parent = ti[j / k - 1], z = j % k;
        si[j] = parent < n ? graph[(size_t)parent * k + z] : (graph[z] | n);  // Q7
        FT best = ft_inf();
        for (int g = 0; g < G; g++) {
          const FT v = in2[((size_t)g * qs + xl) * w2 + (j - k)];
          best = v < best ? v : best;
        }
        sd[j] = best;
      }
    }
    __syncthreads();
    block_topk_stage<true>(L, len, sd, si);
    for (int t = threadIdx.x; t < k; t += blockDim.x) {
      out_id[(size_t)xl * k + t] = si[t];
      out_dist[(size_t)xl * k + t] = sd[t];
    }
    __syncthreads();
  }
}

// u32 ids -> the ABI's size_t ids
__global__ void widen_ids_kernel(size_t count, const u32 *__restrict__ in, size_t *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = in[i];
}
__global__ void narrow_ids_kernel(size_t count, const size_t *__restrict__ in, u32 *__restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (u32)in[i];
}
