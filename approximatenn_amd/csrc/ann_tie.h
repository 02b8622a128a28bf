// ann_tie.h -- sort_and_uniq (alg.c:224-230) for a row whose k+1 best candidates contain ONE run of equal distances
// between different ids, WITHOUT sorting the row (gfx950 only; one wave per row).
//
// Such a query cannot be answered by selection alone: the reference's network (compute.cl:188-203, strict `>`) never
// swaps equal keys, so the order of tied entries -- and through rdups (compute.cl:212-217: the FIRST of two adjacent
// equal ids is killed) even HOW MANY copies of a tied id survive -- is whatever the comparator sequence makes of the
// row.  The literal path (row_dists + exact_select) runs that sequence on all P = 2^floor(log2 L) entries: 78
// sub-steps on 4 096 (key, id) pairs, ~80 us for one workgroup.  But the trajectory of a tied entry depends on very
// little.  With v the tied distance, call an entry below / tied / above by its distance.  A compare-exchange of
// positions (a < b) swaps iff key[a] > key[b]; for a tied entry at a that is "the partner is BELOW", for a tied entry
// at b "the partner is ABOVE", two tied entries never swap.  And the array of classes evolves by the same network on its
// own (swap iff class[a] > class[b]; exchanging equal classes changes nothing).  By the 0-1 principle that is two bit
// arrays, NB[j] = !(key[j] < v) and AB[j] = key[j] > v, under (a, b) -> (a & b, a | b).  So:
//   1. class bits of the row in slot order, 64 positions per lane and 64-bit word; the tied entries (pos, id), <= 64;
//   2. push both through the network: the bit arrays with in-word shifts (strides < 64) or one word exchanged between
//      lanes (strides >= 64), every tied entry by looking its partner's bit up; all of it in registers, the network
//      unrolled with compile-time strides.  That yields the tied entries' order
//      after the first do_sort; everything below v is sorted by distance (one id per distance there, copies adjacent),
//      so rdups' effect is known everywhere it matters;
//   3. class bits of the array AFTER rdups (killed entries are +inf = above), the surviving tied entries pushed
//      through the network again: their order after the second do_sort;
//   4. output = the distinct keys below v, the tied survivors in that order, the distinct keys above v; first k.
// Anything that does not fit (a second tie run, > 64 tied entries, no +inf in the sorted prefix, NaN, k + 1 > 64, a
// failed consistency check) returns false and the caller runs the literal network.  tools/tie_model.py is the same
// algorithm in Python, checked against the literal network on random rows.
#pragma once
#include "ann_device.h"

#define ANN_TIE_MAX 64       // tied entries tracked (one per lane)
#define ANN_TIE_BELOW 2048   // entries below the tied distance (copies included) kept for the multiplicity count

// positions of a 64-bit word whose index has bit T clear
template <int T>
__device__ __forceinline__ constexpr u64 tie_m() {
  constexpr u64 M[6] = {0x5555555555555555ull, 0x3333333333333333ull, 0x0F0F0F0F0F0F0F0Full,
                        0x00FF00FF00FF00FFull, 0x0000FFFF0000FFFFull, 0x00000000FFFFFFFFull};
  return M[T];
}

// LDS bytes tie_resolve<NW> needs
__host__ __device__ inline size_t ann_tie_lds_bytes(int nw) {
  return 2 * sizeof(u64) * 64 * (size_t)nw + sizeof(u32) * (4 * ANN_TIE_MAX + ANN_TIE_BELOW + 8);
}
// ... and tie_derive_list in front of it (placed BEHIND tie_resolve's region): the valid keys of the sorted prefix, the
// waves' partial lists (up to 16 waves), the list itself as distances + ids
__host__ __device__ inline size_t ann_tie_derive_bytes(size_t P) {
  return sizeof(Key) * (P + 17 * (size_t)ANN_WAVE) + (sizeof(FT) + sizeof(u32)) * ANN_WAVE + 128;
}

// exchange neighbouring blocks of 2^T bits
template <int T>
__device__ __forceinline__ u64 tie_swap_level(u64 x) {
  return ((x >> (1 << T)) & tie_m<T>()) | ((x & tie_m<T>()) << (1 << T));
}
// reverse the bits inside every block of 2^M bits: r[j] = x[j ^ (2^M - 1)]
template <int M>
__device__ __forceinline__ u64 tie_blockrev(u64 x) {
  if constexpr (M == 6) return __brevll(x);
  else if constexpr (M <= 3) {
    x = tie_swap_level<0>(x);
    if constexpr (M > 1) x = tie_swap_level<1>(x);
    if constexpr (M > 2) x = tie_swap_level<2>(x);
    return x;
  } else {  // the whole word reversed, then the ORDER of the blocks restored
    x = __brevll(x);
    x = tie_swap_level<5>(x);
    if constexpr (M == 4) x = tie_swap_level<4>(x);
    return x;
  }
}

// compare-exchange (a, b) -> (a & b, a | b) of a bit array held one 64-bit word per lane, at stride 2^SS inside the word;
// FLIP: the first sub-step of a merge level, partner = index ^ (2^(SS+1) - 1) (compute.cl:191: lo reversed)
template <int SS, bool FLIP>
__device__ __forceinline__ u64 tie_word_step(u64 x) {
  constexpr u64 M = tie_m<SS>();
  if constexpr (FLIP) {
    const u64 r = tie_blockrev<SS + 1>(x);
    return (x & r & M) | ((x | r) & ~M);
  } else {
    const u64 lo = x & M, hi = (x >> (1 << SS)) & M;
    return (lo & hi) | ((lo | hi) << (1 << SS));
  }
}

// One sub-step (merge level S, stride 2^SS < 64: inside a word) of do_sort's network (alg.c:137-144) on the class bits
// AB/NB (position j = bit j & 63 of word (j >> 12) of lane (j >> 6) & 63) and on one tracked tied entry per lane, then
// the sub-steps SS-1 .. 0 of the level.  Everything stays in registers; a tracked entry fetches the word that holds
// its partner from the lane that has it.
template <int NW, int S, int SS>
__device__ __forceinline__ void tie_sub(u64 (&AB)[NW], u64 (&NB)[NW], bool tracked, u32 &pos) {
  static_assert(SS < 6 && SS <= S, "in-word strides only");
  constexpr bool FLIP = SS == S;
  constexpr u32 mask = FLIP ? ((2u << S) - 1u) : (1u << SS);
  {  // the tracked entries: exchanged with a partner BELOW when they are the pair's lower index, ABOVE when the upper.
     // The partner sits on the other side of its pair, so ONE word per lane answers both questions: NB on upper-index
     // positions, AB on lower-index positions.
    const u32 q = pos ^ mask;
    const bool isa = !((pos >> SS) & 1u);
    const u32 src = (q >> 6) & 63u;
    u64 word = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const u64 cw = (NB[w] & ~tie_m<SS>()) | (AB[w] & tie_m<SS>());
      const u32 lo = __shfl((u32)cw, (int)src), hi = __shfl((u32)(cw >> 32), (int)src);
      if (NW == 1 || (q >> 12) == (u32)w) word = ((u64)hi << 32) | lo;
    }
    const u32 bit = (u32)(word >> (q & 63u)) & 1u;
    if (tracked && bit == (isa ? 0u : 1u)) pos = q;
  }
#pragma unroll
  for (int w = 0; w < NW; w++) AB[w] = tie_word_step<SS, FLIP>(AB[w]), NB[w] = tie_word_step<SS, FLIP>(NB[w]);
  if constexpr (SS > 0) tie_sub<NW, S, SS - 1>(AB, NB, tracked, pos);
}

// A sub-step at stride 2^ss >= 64 (whole words exchanged between lanes; between a lane's own words from 2^12 on), s and
// ss at run time: one copy of this code serves every such sub-step.
template <int NW>
__device__ __forceinline__ void tie_cross(int s, int ss, u64 (&AB)[NW], u64 (&NB)[NW], bool tracked, u32 &pos) {
  const bool flip = ss == s;
  const u32 mask = flip ? ((2u << s) - 1u) : (1u << ss);
  const u32 lane = (u32)lane_id();
  {
    const u32 q = pos ^ mask;
    const bool isa = !((pos >> ss) & 1u);
    const u32 src = (q >> 6) & 63u;
    u64 word = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const bool bside = ss < 12 ? ((lane >> (ss - 6)) & 1u) != 0 : (((u32)w >> (ss - 12)) & 1u) != 0;
      const u64 cw = bside ? NB[w] : AB[w];
      const u32 lo = __shfl((u32)cw, (int)src), hi = __shfl((u32)(cw >> 32), (int)src);
      if (NW == 1 || (q >> 12) == (u32)w) word = ((u64)hi << 32) | lo;
    }
    const u32 bit = (u32)(word >> (q & 63u)) & 1u;
    if (tracked && bit == (isa ? 0u : 1u)) pos = q;
  }
  if (NW == 1 || ss < 12) {
    const int B = ss - 6;
    const u32 lm = flip ? ((2u << B) - 1u) : (1u << B);
    const bool aside = !((lane >> B) & 1u);
#pragma unroll
    for (int w = 0; w < NW; w++) {
      u64 oa = __shfl((unsigned long long)AB[w], (int)(lane ^ lm)), ob = __shfl((unsigned long long)NB[w], (int)(lane ^ lm));
      if (flip) oa = __brevll(oa), ob = __brevll(ob);
      AB[w] = aside ? (AB[w] & oa) : (AB[w] | oa);
      NB[w] = aside ? (NB[w] & ob) : (NB[w] | ob);
    }
  } else {
    const int B = ss - 12;
    const u32 wm = flip ? ((2u << B) - 1u) : (1u << B);
    u64 na[NW], nb[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
      u64 oa = 0, ob = 0;
#pragma unroll
      for (int v = 0; v < NW; v++)
        if ((u32)v == ((u32)w ^ wm)) oa = AB[v], ob = NB[v];
      if (flip) {  // partner position: word ^ wm, lane ^ 63, bit ^ 63
        oa = __brevll(__shfl((unsigned long long)oa, (int)(lane ^ 63u)));
        ob = __brevll(__shfl((unsigned long long)ob, (int)(lane ^ 63u)));
      }
      const bool aside = !(((u32)w >> B) & 1u);
      na[w] = aside ? (AB[w] & oa) : (AB[w] | oa);
      nb[w] = aside ? (NB[w] & ob) : (NB[w] | ob);
    }
#pragma unroll
    for (int w = 0; w < NW; w++) AB[w] = na[w], NB[w] = nb[w];
  }
}

// merge levels S < 6 (the whole level inside a word), compile-time strides
template <int NW, int S>
__device__ __forceinline__ void tie_levels_inword(int lk, u64 (&AB)[NW], u64 (&NB)[NW], bool tracked, u32 &pos) {
  if (S < lk) {
    tie_sub<NW, S, S>(AB, NB, tracked, pos);
    if constexpr (S + 1 < 6) tie_levels_inword<NW, S + 1>(lk, AB, NB, tracked, pos);
  }
}

// The network on 2^lk positions applied to the class bits and the tracked entries.  Code that runs once per sub-step
// is bound by instruction fetch when every sub-step is its own straight-line copy (78 copies: 57 us for the two
// passes); so only the 21 sub-steps of the levels below 64 are, the strides >= 64 share one run-time copy and the six
// in-word strides that end every higher level share another.
template <int NW>
__device__ __forceinline__ void tie_net_sim(int lk, u64 (&AB)[NW], u64 (&NB)[NW], bool tracked, u32 &pos) {
  tie_levels_inword<NW, 0>(lk, AB, NB, tracked, pos);
  for (int s = 6; s < lk; s++) {
    for (int ss = s; ss >= 6; ss--) tie_cross<NW>(s, ss, AB, NB, tracked, pos);
    tie_sub<NW, 6, 5>(AB, NB, tracked, pos);  // strides 32 .. 1, straight
  }
}

// The candidate list of a row from the row itself -- for callers that have no stage-1 list of the row's query (a sharded
// host's reduced rows: only the query's owner ever held the merged list; stage-2 rows; the staged API): the K1 smallest
// DISTINCT (distance, id) keys among the finite entries of the row's first P slots, ascending, as cd / ci in LDS, padded
// with (+inf, ANN_ID_NONE).  One workgroup: all waves compact the finite keys into LDS, every wave selects the K1
// smallest of its share (wave_select_smallest), wave 0 merges.  `lds`: ann_tie_derive_bytes(P) bytes.
__device__ inline void tie_derive_list(u32 P, int K1, const u32 *__restrict__ gi, const FT *__restrict__ gd,
                                       unsigned char *lds, FT **cd_out, u32 **ci_out) {
  const int lane = lane_id(), w = threadIdx.x >> 6, W = blockDim.x >> 6;
  Key *keys = reinterpret_cast<Key *>(lds);
  Key *part = keys + P;                    // [W][K1], K1 <= 64, W <= 16
  Key *fin = part + 16 * ANN_WAVE;         // [K1]
  FT *cd = reinterpret_cast<FT *>(fin + ANN_WAVE);
  u32 *ci = reinterpret_cast<u32 *>(cd + ANN_WAVE);
  u32 *cnt = ci + ANN_WAVE;                // [0] keys [1..W] the waves' list lengths
  if (threadIdx.x < 20) cnt[threadIdx.x] = 0;
  __syncthreads();
  for (u32 j0 = (u32)w * ANN_WAVE; j0 < P; j0 += (u32)W * ANN_WAVE) {
    const u32 j = j0 + lane;
    const FT dj = j < P ? gd[j] : ft_inf();
    const bool ok = dj < ft_inf();
    const u64 m = __ballot(ok);
    if (m) {
      u32 base = 0;
      if (lane == 0) base = atomicAdd(&cnt[0], (u32)__popcll(m));
      base = __shfl(base, 0);
      if (ok) keys[base + mask_rank(m)] = key_make(dj, gi[j]);
    }
  }
  __syncthreads();
  const int n = (int)cnt[0], per = (n + W - 1) / W, a = min(n, w * per), b = min(n, a + per);
  const int mw = wave_select_smallest(keys + a, b - a, K1, part + (size_t)w * ANN_WAVE);
  if (lane == 0) cnt[1 + w] = (u32)mw;
  __syncthreads();
  if (w == 0) {
    int tot = 0;  // the partial lists, packed into the (now free) front of `keys`
    for (int ww = 0; ww < W; ww++) {
      const int mm = (int)cnt[1 + ww];
      for (int i = lane; i < mm; i += ANN_WAVE) keys[tot + i] = part[(size_t)ww * ANN_WAVE + i];
      tot += mm;
    }
    wave_lds_sync();
    const int m = wave_select_smallest(keys, tot, K1, fin);
    for (int t = lane; t < ANN_WAVE; t += ANN_WAVE) {
      cd[t] = t < m ? key_dist(fin[t]) : ft_inf();
      ci[t] = t < m ? key_id(fin[t]) : ANN_ID_NONE;
    }
  }
  __syncthreads();
  *cd_out = cd, *ci_out = ci;
}

// Wave 0's part of tie_resolve (below): LDS holds the class bits LA/LB, the tied positions, the positions below v and
// the counters; dk/ik = this lane's key of the candidate list.
template <int NW>
__device__ inline bool tie_finish(u32 L, int k, int m, int t0, int t1, FT v, FT dk, u32 ik, const u32 *__restrict__ gi,
                                  unsigned char *lds, u32 *out_id, size_t *out_id64, FT *out_dist, unsigned long long *ts) {
#ifdef ANN_TIE_PROFILE
#define TIE_TS(i) do { if (ts && threadIdx.x == 0) ts[i] = wall_clock64(); } while (0)
#else
#define TIE_TS(i) do { } while (0)
#endif
  const u32 lane = (u32)lane_id();
  const int lk = ann_lg(L);
  const u32 P = 1u << lk;
  u64 *LA = reinterpret_cast<u64 *>(lds);
  u64 *LB = LA + 64 * NW;
  u32 *tpos = reinterpret_cast<u32 *>(LB + 64 * NW);
  u32 *tid = tpos + ANN_TIE_MAX, *ord = tid + ANN_TIE_MAX, *ord2 = ord + ANN_TIE_MAX, *bl = ord2 + ANN_TIE_MAX;
  u32 *ctl = bl + ANN_TIE_BELOW;
  const u32 nt = ctl[0], nbelow = ctl[1];
  if (ctl[2] || nt < 2 || nt > ANN_TIE_MAX || nbelow > ANN_TIE_BELOW) return false;
  if (L > P && !ctl[3]) return false;  // rdups at P-1 would read the unsorted id at P
  u64 AB[NW], NB[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) AB[w] = LA[w * 64 + lane], NB[w] = LB[w * 64 + lane];
  // the ids of the few entries that matter, one round trip
  if (lane < nt) tid[lane] = gi[tpos[lane]];
  for (u32 u = lane; u < nbelow; u += ANN_WAVE) bl[u] = gi[bl[u]];  // in place: entry u is read and written by one lane
  wave_lds_sync();

  TIE_TS(3);
  // ---- copies of every key below v (one id per distance there: sorted by distance = copies adjacent)
  u32 c = 0;
  if ((int)lane < t0)
    for (u32 u = 0; u < nbelow; u++) c += bl[u] == ik ? 1u : 0u;
  const u32 cum = wave_incl_scan(c);
  if (__ballot((int)lane < t0 && c == 0)) return false;
  if ((t0 ? (u32)__shfl(cum, t0 - 1) : 0u) != nbelow) return false;
  const u32 nb = nbelow;

  TIE_TS(4);
  // ---- the two do_sorts, one copy of the network code: pass 0 = where do the tied entries land, then rdups and the
  // array after it as class bits; pass 1 = where do the surviving tied entries land
  u32 pos = lane < nt ? tpos[lane] : 0u;
  bool tracked = lane < nt, surv = false;
  u32 oid = ANN_ID_NONE, st = 0;
  for (int pass = 0; pass < 2; pass++) {
    tie_net_sim<NW>(lk, AB, NB, tracked, pos);
    TIE_TS(5 + 2 * pass);
    if (pass == 0) {
      ord[lane] = ANN_ID_NONE;
      wave_lds_sync();
      const u32 r1 = pos - nb;
      if (__ballot(lane < nt && r1 >= nt)) return false;
      if (lane < nt) ord[r1] = tid[lane];
      wave_lds_sync();
      oid = ord[lane];
      const u32 onx = ord[(lane + 1) & 63u];
      if (__ballot(lane < nt && oid == ANN_ID_NONE)) return false;  // two entries on one position: cannot happen
      // rdups: the first of two adjacent equal ids is killed; the entry after the tied run has another distance
      surv = lane < nt && (lane + 1 == nt || oid != onx);
      st = (u32)__popcll(__ballot(surv));
      // the array after rdups as class bits: survivors below v, tied survivors, everything else above
#pragma unroll
      for (int w = 0; w < NW; w++) LA[w * 64 + lane] = ~0ull, LB[w * 64 + lane] = ~0ull;
      wave_lds_sync();
      if ((int)lane < t0) {
        const u32 p = cum - 1u;
        atomicAnd(reinterpret_cast<u32 *>(LA) + (p >> 5), ~(1u << (p & 31u)));
        atomicAnd(reinterpret_cast<u32 *>(LB) + (p >> 5), ~(1u << (p & 31u)));
      }
      if (surv) {
        const u32 p = nb + lane;
        atomicAnd(reinterpret_cast<u32 *>(LA) + (p >> 5), ~(1u << (p & 31u)));
      }
      wave_lds_sync();
#pragma unroll
      for (int w = 0; w < NW; w++) AB[w] = LA[w * 64 + lane], NB[w] = LB[w * 64 + lane];
      pos = nb + lane;
      tracked = surv;
      TIE_TS(6);
    } else {
      ord2[lane] = ANN_ID_NONE;
      wave_lds_sync();
      const u32 r2 = pos - (u32)t0;
      if (__ballot(surv && r2 >= st)) return false;
      if (surv) ord2[r2] = oid;
      wave_lds_sync();
      if (__ballot(lane < st && ord2[lane] == ANN_ID_NONE)) return false;
    }
  }

  // ---- the first k entries
  const int t = (int)lane;
  const int above = t1 + 1 + (t - t0 - (int)st);  // list index of an output beyond the tied survivors
  const u32 ia = __shfl(ik, above & 63);
  const FT da = __shfl(dk, above & 63);
  if (__ballot(t < k && t >= t0 + (int)st && (above >= m || !(da < ft_inf())))) return false;
  if (t < k) {
    u32 oi;
    FT od;
    if (t < t0) oi = ik, od = dk;
    else if (t < t0 + (int)st) oi = ord2[t - t0], od = v;
    else oi = ia, od = da;
    if (out_id64) out_id64[t] = oi;
    else out_id[t] = oi;
    out_dist[t] = od;
  }
  TIE_TS(8);
  return true;
}

// One WORKGROUP (any number of whole waves; the row scan is shared by all of them, the rest runs on wave 0).  L: the
// reference's row length; the row's first min(L, P + 1) entries (ids gi, distances gd) are in global memory in slot
// order.  cd/ci: the K1 = k+1 smallest distinct (distance, id) keys of the row's first P slots, ascending, padded with
// (+inf, ANN_ID_NONE) -- stage 1's output.  On success the first k entries of sort_and_uniq's result are written to
// out_id / out_id64 (whichever is non-NULL) and out_dist and true is returned (workgroup-uniform; the function
// contains workgroup barriers: every thread has to call it).
template <int NW>
__device__ inline bool tie_resolve(u32 L, int k, int K1, const u32 *__restrict__ gi, const FT *__restrict__ gd,
                                   const FT *__restrict__ cd, const u32 *__restrict__ ci, unsigned char *lds,
                                   u32 *out_id, size_t *out_id64, FT *out_dist, unsigned long long *ts = NULL) {
#ifdef ANN_TIE_PROFILE  // debug builds: 100 MHz timestamps of the phases into ts[1..]
#define TIE_TS(i) do { if (ts && threadIdx.x == 0) ts[i] = wall_clock64(); } while (0)
#else
#define TIE_TS(i) do { } while (0)
#endif
  const u32 lane = (u32)lane_id();
  TIE_TS(1);
  const int lk = ann_lg(L);
  const u32 P = 1u << lk;
  if (L < 16 || K1 != k + 1 || K1 > ANN_WAVE || (u32)k > P || P > 4096u * NW) return false;
  u64 *LA = reinterpret_cast<u64 *>(lds);
  u64 *LB = LA + 64 * NW;
  u32 *tpos = reinterpret_cast<u32 *>(LB + 64 * NW);
  u32 *tid = tpos + ANN_TIE_MAX, *ord = tid + ANN_TIE_MAX, *ord2 = ord + ANN_TIE_MAX, *bl = ord2 + ANN_TIE_MAX;
  u32 *ctl = bl + ANN_TIE_BELOW;  // [0] tied entries [1] entries below v [2] NaN seen [3] +inf seen [4] result

  // ---- the candidate list: exactly one run [t0, t1] of equal distances, >= k keys, the k-th finite (every wave
  // derives the same facts from the same list)
  const bool has_key = (int)lane < K1;
  const FT dk = has_key ? cd[lane] : ft_inf();
  const u32 ik = has_key ? ci[lane] : ANN_ID_NONE;
  const int m = __popcll(__ballot(ik != ANN_ID_NONE));
  const FT dn = __shfl_down(dk, 1);
  const u64 E = __ballot((int)lane + 1 < m && ft_bits(dk) == ft_bits(dn));
  if (m < k || !E) return false;
  if (!(__shfl(dk, k - 1) < ft_inf())) return false;
  const int t0 = __builtin_ctzll(E), run = __popcll(E), t1 = t0 + run;
  if ((E >> t0) != (run == 64 ? ~0ull : ((1ull << run) - 1ull))) return false;
  const FT v = __shfl(dk, t0);

  TIE_TS(2);
  // ---- class bits of the row in slot order; the tied entries; the entries below v.  All waves of the workgroup:
  // word i (positions 64 i ..) is one coalesced load and four ballots, by wave i mod W (one wave alone: 11 us of 30).
  for (u32 i = threadIdx.x; i < 64u * NW; i += blockDim.x) LA[i] = ~0ull, LB[i] = ~0ull;
  if (threadIdx.x < 5) ctl[threadIdx.x] = 0;
  __syncthreads();
  {
    const u32 nwords = (P + 63u) >> 6, W = blockDim.x >> 6, w0 = threadIdx.x >> 6;
    constexpr u32 UNR = 16;  // loads in flight per lane: the 64 words of a 4 096-entry row in ONE round trip on four waves
    for (u32 ib = w0; ib < nwords; ib += W * UNR) {
      FT dbuf[UNR];
#pragma unroll
      for (u32 u = 0; u < UNR; u++) {
        const u32 j = (ib + u * W) * 64u + lane;
        dbuf[u] = (ib + u * W < nwords && j < P) ? gd[j] : ft_inf();
      }
#pragma unroll
      for (u32 u = 0; u < UNR; u++) {
        const u32 i = ib + u * W, j = i * 64u + lane;
        if (i < nwords) {  // wave-uniform
          const bool valid = j < P;
          const FT dj = dbuf[u];
          const bool below = dj < v, eq = valid && dj == v;
          const u64 ma = __ballot(dj > v || !valid), mnb = __ballot(!below), me = __ballot(eq), mb = __ballot(below);
          const bool nan = __ballot(dj != dj) != 0, inf = __ballot(valid && dj == ft_inf()) != 0;
          if (lane == 0) {
            LA[i] = ma, LB[i] = mnb;  // word i = lane i & 63, word i >> 6 of that lane
            if (nan) ctl[2] = 1;
            if (inf) ctl[3] = 1;
          }
          if (me) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&ctl[0], (u32)__popcll(me));
            base = __shfl(base, 0);
            const u32 idx = base + mask_rank(me);
            if (eq && idx < ANN_TIE_MAX) tpos[idx] = j;
          }
          if (mb) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&ctl[1], (u32)__popcll(mb));
            base = __shfl(base, 0);
            const u32 idx = base + mask_rank(mb);
            if (below && idx < ANN_TIE_BELOW) bl[idx] = j;
          }
        }
      }
    }
  }
  __syncthreads();
  bool ok = false;
  if (threadIdx.x < ANN_WAVE) ok = tie_finish<NW>(L, k, m, t0, t1, v, dk, ik, gi, lds, out_id, out_id64, out_dist, ts);
  if (threadIdx.x == 0) ctl[4] = ok ? 1u : 0u;
  __syncthreads();
  ok = ctl[4] != 0;
  __syncthreads();  // the LDS is the caller's again
  return ok;
}
