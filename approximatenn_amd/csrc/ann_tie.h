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
//   2. push both through the network: the bit arrays with in-word shifts (strides < 64) or one word exchanged through
//      LDS (strides >= 64), every tied entry by looking its partner's bit up.  That yields the tied entries' order
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

// positions of a 64-bit word whose index has bit t clear
__device__ const u64 ann_tie_m[6] = {0x5555555555555555ull, 0x3333333333333333ull, 0x0F0F0F0F0F0F0F0Full,
                                     0x00FF00FF00FF00FFull, 0x0000FFFF0000FFFFull, 0x00000000FFFFFFFFull};

// LDS bytes tie_resolve<NW> needs
__host__ __device__ inline size_t ann_tie_lds_bytes(int nw) {
  return 2 * sizeof(u64) * 64 * (size_t)nw + sizeof(u32) * (4 * ANN_TIE_MAX + ANN_TIE_BELOW);
}

// compare-exchange of the bit array held one word per lane at stride 2^ss inside the word; flip: the first sub-step
// of a merge level, partner = index ^ (2^(s+1) - 1) (compute.cl:191: lo reversed)
__device__ __forceinline__ u64 tie_word_step(u64 x, int ss, bool flip) {
  const u64 M = ann_tie_m[ss];
  if (flip) {
    u64 r = x;  // r[j] = x[j ^ (2^(ss+1) - 1)]: reverse the bits inside every block of 2^(ss+1)
    for (int t = 0; t <= ss; t++) r = ((r >> (1 << t)) & ann_tie_m[t]) | ((r & ann_tie_m[t]) << (1 << t));
    return (x & r & M) | ((x | r) & ~M);
  }
  const int sh = 1 << ss;
  const u64 lo = x & M, hi = (x >> sh) & M;
  return (lo & hi) | ((lo | hi) << sh);
}

// The network of do_sort (alg.c:137-144) on 2^lk positions, applied to the class bits AB/NB (position j = bit j & 63 of
// word (j >> 12) of lane (j >> 6) & 63) and to one tracked tied entry per lane (`tracked`, position `pos`).
template <int NW>
__device__ inline void tie_net_sim(int lk, u64 (&AB)[NW], u64 (&NB)[NW], u64 *LA, u64 *LB, bool tracked, u32 &pos) {
  const u32 lane = (u32)lane_id();
  for (int s = 0; s < lk; s++)
    for (int ss = s; ss >= 0; ss--) {
      const bool flip = ss == s;
      const u32 mask = flip ? ((2u << s) - 1u) : (1u << ss);
#pragma unroll
      for (int w = 0; w < NW; w++) LA[w * 64 + lane] = AB[w], LB[w * 64 + lane] = NB[w];
      wave_lds_sync();
      if (tracked) {
        const u32 q = pos ^ mask;
        const bool isa = !((pos >> ss) & 1u);  // the lower index of the pair
        const u32 word = reinterpret_cast<const u32 *>(isa ? LB : LA)[q >> 5];
        const u32 bit = (word >> (q & 31u)) & 1u;
        if (bit == (isa ? 0u : 1u)) pos = q;   // partner below (we are a) / partner above (we are b): exchanged
      }
      if (ss < 6) {
#pragma unroll
        for (int w = 0; w < NW; w++) AB[w] = tie_word_step(AB[w], ss, flip), NB[w] = tie_word_step(NB[w], ss, flip);
      } else if (ss < 12) {
        const u32 pl = lane ^ (flip ? ((2u << (s - 6)) - 1u) : (1u << (ss - 6)));
        const bool aside = !((lane >> (ss - 6)) & 1u);
#pragma unroll
        for (int w = 0; w < NW; w++) {
          u64 oa = LA[w * 64 + pl], ob = LB[w * 64 + pl];
          if (flip) oa = __brevll(oa), ob = __brevll(ob);
          AB[w] = aside ? (AB[w] & oa) : (AB[w] | oa);
          NB[w] = aside ? (NB[w] & ob) : (NB[w] | ob);
        }
      } else {
        const u32 pl = flip ? lane ^ 63u : lane;
#pragma unroll
        for (int w = 0; w < NW; w++) {
          const u32 pw = (u32)w ^ (flip ? ((2u << (s - 12)) - 1u) : (1u << (ss - 12)));
          const bool aside = !(((u32)w >> (ss - 12)) & 1u);
          u64 oa = LA[pw * 64 + pl], ob = LB[pw * 64 + pl];
          if (flip) oa = __brevll(oa), ob = __brevll(ob);
          AB[w] = aside ? (AB[w] & oa) : (AB[w] | oa);
          NB[w] = aside ? (NB[w] & ob) : (NB[w] | ob);
        }
      }
      wave_lds_sync();
    }
}

// One wave.  L: the reference's row length; the row's first min(L, P + 1) entries (ids gi, distances gd) are in
// global memory in slot order.  cd/ci: the K1 = k+1 smallest distinct (distance, id) keys of the row's first P slots,
// ascending, padded with (+inf, ANN_ID_NONE) -- stage 1's output.  On success the first k entries of sort_and_uniq's
// result are written to out_id / out_id64 (whichever is non-NULL) and out_dist and true is returned (wave-uniform).
template <int NW>
__device__ inline bool tie_resolve(u32 L, int k, int K1, const u32 *__restrict__ gi, const FT *__restrict__ gd,
                                   const FT *__restrict__ cd, const u32 *__restrict__ ci, unsigned char *lds,
                                   u32 *out_id, size_t *out_id64, FT *out_dist) {
  const u32 lane = (u32)lane_id();
  const int lk = ann_lg(L);
  const u32 P = 1u << lk;
  if (L < 16 || K1 != k + 1 || K1 > ANN_WAVE || (u32)k > P || P > 4096u * NW) return false;
  u64 *LA = reinterpret_cast<u64 *>(lds);
  u64 *LB = LA + 64 * NW;
  u32 *tpos = reinterpret_cast<u32 *>(LB + 64 * NW);
  u32 *tid = tpos + ANN_TIE_MAX, *ord = tid + ANN_TIE_MAX, *ord2 = ord + ANN_TIE_MAX, *bl = ord2 + ANN_TIE_MAX;

  // ---- the candidate list: exactly one run [t0, t1] of equal distances, >= k keys, the k-th finite
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

  // ---- class bits of the row in slot order; the tied entries; the ids below v
  u64 AB[NW], NB[NW];
#pragma unroll
  for (int w = 0; w < NW; w++) AB[w] = NB[w] = ~0ull;
  u32 nt = 0, nbelow = 0;
  u64 badm = 0, infm = 0;
  const u32 nwords = (P + 63u) >> 6;
  for (u32 i = 0; i < nwords; i++) {
    const u32 j = i * 64u + lane;
    const bool valid = j < P;
    const FT dj = valid ? gd[j] : ft_inf();
    const bool below = dj < v, eq = valid && dj == v;
    const u64 ma = __ballot(dj > v || !valid), mnb = __ballot(!below), me = __ballot(eq), mb = __ballot(below);
    badm |= __ballot(dj != dj);
    infm |= __ballot(valid && dj == ft_inf());
#pragma unroll
    for (int w = 0; w < NW; w++)
      if (lane == (i & 63u) && (i >> 6) == (u32)w) AB[w] = ma, NB[w] = mnb;
    if (me | mb) {
      const u32 idj = valid ? gi[j] : ANN_ID_NONE;
      if (eq) {
        const u32 idx = nt + mask_rank(me);
        if (idx < ANN_TIE_MAX) tpos[idx] = j, tid[idx] = idj;
      }
      if (below) {
        const u32 idx = nbelow + mask_rank(mb);
        if (idx < ANN_TIE_BELOW) bl[idx] = idj;
      }
      nt += (u32)__popcll(me), nbelow += (u32)__popcll(mb);
    }
  }
  if (badm || nt < 2 || nt > ANN_TIE_MAX || nbelow > ANN_TIE_BELOW) return false;
  if (L > P && !infm) return false;  // rdups at P-1 would read the unsorted id at P
  wave_lds_sync();

  // ---- copies of every key below v (one id per distance there: sorted by distance = copies adjacent)
  u32 c = 0;
  if ((int)lane < t0)
    for (u32 u = 0; u < nbelow; u++) c += bl[u] == ik ? 1u : 0u;
  const u32 cum = wave_incl_scan(c);
  if (__ballot((int)lane < t0 && c == 0)) return false;
  if ((t0 ? (u32)__shfl(cum, t0 - 1) : 0u) != nbelow) return false;
  const u32 nb = nbelow;

  // ---- first do_sort: where do the tied entries land
  u32 pos = lane < nt ? tpos[lane] : 0u;
  tie_net_sim<NW>(lk, AB, NB, LA, LB, lane < nt, pos);
  ord[lane] = ANN_ID_NONE;
  wave_lds_sync();
  const u32 r1 = pos - nb;
  if (__ballot(lane < nt && r1 >= nt)) return false;
  if (lane < nt) ord[r1] = tid[lane];
  wave_lds_sync();
  const u32 oid = ord[lane], onx = ord[(lane + 1) & 63u];
  if (__ballot(lane < nt && oid == ANN_ID_NONE)) return false;  // two entries on one position: cannot happen
  // ---- rdups: the first of two adjacent equal ids is killed; the entry after the tied run has another distance
  const bool surv = lane < nt && (lane + 1 == nt || oid != onx);
  const u64 sm = __ballot(surv);
  const u32 st = (u32)__popcll(sm);

  // ---- the array after rdups as class bits: survivors below v, tied survivors, everything else above
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
  wave_lds_sync();

  // ---- second do_sort
  pos = nb + lane;
  tie_net_sim<NW>(lk, AB, NB, LA, LB, surv, pos);
  ord2[lane] = ANN_ID_NONE;
  wave_lds_sync();
  const u32 r2 = pos - (u32)t0;
  if (__ballot(surv && r2 >= st)) return false;
  if (surv) ord2[r2] = oid;
  wave_lds_sync();
  if (__ballot(lane < st && ord2[lane] == ANN_ID_NONE)) return false;

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
  return true;
}
