/*
 * oracle/ann_oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT.
 *
 * A clean-room, single-threaded CPU restatement of the precomp()/query() hot
 * path of marcusrussi/approximateNN, written from the behavioural spec in
 * SURVEY.md section 8.0/8.0.1 (every quirk Q1..Q19 reproduced on purpose).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg have something to CHECK the HIP path against.  Nothing
 * under approximatenn_amd/ may include, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function
 * here bit-for-bit against the reference itself compiled from
 * /root/reference (`make -C oracle ref` -> oracle/_ref/libref_{f32,f64}.so),
 * and tests/test_oracle_golden.py checks it against the committed vectors in
 * tests/golden/ that tests/golden/make_golden.py generated from that build.
 *
 * Unlike the reference it never materialises the [rows][cands][d] "diffs"
 * tensor: each distance is produced by one fused subtract/square/tree-sum.
 * The arithmetic (operation order, roundings) is exactly the reference's.
 *
 * Build: -DORACLE_F32 for the float build (ftype.h:3-9 of the reference),
 * default double.  Compile with -ffp-contract=off.
 *
 * Citations are file:line into /root/reference.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_F32
typedef float ft;
typedef uint32_t ubits;
#else
typedef double ft;
typedef uint64_t ubits;
#endif

/* Layout-compatible with save_t, ann.h:8-12. */
typedef struct {
  int tries;
  size_t n, k, d_short, d_long;
  size_t **which_par, *par_maxes, *graph;
  ft *row_means, *bases;
} oracle_save_t;

/* ---------------------------------------------------------------- helpers */

static size_t at_least_1(size_t v) { return v ? v : 1; }

/* floor(log2(x)), 0 for x == 0.  algc.c:13-22 ("lg"). */
unsigned oracle_lg(size_t x) {
  unsigned r = 0;
  while (x >>= 1)
    r++;
  return r;
}

/* rand_pr.c:8 / randNorm.c:7: uniform in [0,1) from libc random(). */
static double unit_draw(void) {
  return (double)(unsigned long)random() / ((double)RAND_MAX + 1);
}

/* Box-Muller with a cached second variate, randNorm.c:9-21.  The cache is
 * process-global like the reference's static. */
static double bm_cache;
static int bm_have = 0;
double oracle_rand_norm(void) {
  if (bm_have) {
    bm_have = 0;
    return bm_cache;
  }
  double radius = sqrt(log(unit_draw()) * -2);
  double theta = unit_draw() * M_PI * 2;
  bm_cache = radius * sin(theta);
  bm_have = 1;
  return radius * cos(theta);
}
void oracle_rand_norm_reset(void) { bm_have = 0; }

/* time_results.c:10-13 (genRand). */
void oracle_gen_rand(size_t count, ft *out) {
  for (size_t i = 0; i < count; i++)
    out[i] = (ft)oracle_rand_norm();
}

/* Partial Fisher-Yates, rand_pr.c:17-30: always draws d_pre times. */
size_t *oracle_rand_perm(size_t d_pre, size_t d_post) {
  size_t *p = malloc(sizeof(size_t) * d_post);
  for (size_t i = 0; i < d_post; i++)
    p[i] = i;
  for (size_t i = 0; i < d_pre; i++) {
    size_t j = (unsigned long)random() % (d_post - i) + i;
    size_t t = p[i];
    p[i] = p[j];
    p[j] = t;
  }
  return p;
}

/* ------------------------------------------------------- pairwise tree sum */

/* In-place tree of compute.cl:160-167 driven as alg.c:130-135 (Q4):
 * for s = len; s>>1; s>>=1:  m[z] = m[z] + (m[z+s/2] + g), z < s/2,
 * g = m[s-1] if s odd and z == 0, else 0.  Destroys m, returns m[0]. */
ft oracle_tree_sum(size_t len, ft *m) {
  for (size_t s = len; s >> 1; s >>= 1) {
    size_t h = s / 2;
    for (size_t z = 0; z < h; z++) {
      ft g = ((s & 1) && z == 0) ? m[s - 1] : 0;
      m[z] = m[z] + (m[z + h] + g);
    }
  }
  return m[0];
}

/* Squared distance with the +inf rules of compute.cl:135-151 (Q3,Q5,Q15).
 * `valid` = id < n and not the excluded self row. */
static ft sq_dist(size_t d, const ft *a, const ft *p, int valid, ft *scratch) {
  if (!valid)
    return (ft)INFINITY; /* d*d + inf with finite d */
  for (size_t z = 0; z < d; z++) {
    ft df = a[z] - p[z];
    scratch[z] = df * df;
  }
  return oracle_tree_sum(d, scratch);
}

/* ---------------------------------------------------- the "sort" network */

/* One row of do_sort (alg.c:137-144) = sort_two_step (compute.cl:181-206)
 * for every (step, sstep).  Only touches the first 2^floor(log2 L) entries
 * when L >= 16, and a clipped 16-wide network when L < 16 (Q1).  Swap iff
 * strictly greater (Q17). */
void oracle_sort_net(size_t L, size_t *ids, ft *key) {
  int lk = (int)oracle_lg(L);
  size_t items = (size_t)1 << (lk > 4 ? lk - 4 : 0);
  for (int s = 0; s < lk; s++)
    for (int ss = s; ss >= 0; ss--)
      for (size_t pr = 0; pr < items * 8; pr++) {
        size_t hi = (pr >> ss) << ss, lo = pr ^ hi;
        size_t ia = hi << 1 | lo;
        if (ss == s)
          lo = ((size_t)1 << ss) - lo - 1;
        size_t ib = hi << 1 | (size_t)1 << ss | lo;
        if (ib < L && key[ia] > key[ib]) {
          ft tk = key[ia];
          key[ia] = key[ib];
          key[ib] = tk;
          size_t ti = ids[ia];
          ids[ia] = ids[ib];
          ids[ib] = ti;
        }
      }
}

/* sort_and_uniq, alg.c:224-230 with rdups compute.cl:212-217 (Q6). */
void oracle_topk_stage(size_t L, size_t *ids, ft *key) {
  oracle_sort_net(L, ids, key);
  for (size_t y = 0; y + 1 < L; y++)
    if (ids[y] == ids[y + 1])
      key[y] += (ft)INFINITY;
  oracle_sort_net(L, ids, key);
}

/* ------------------------------------------- random orthogonal transforms */

typedef struct {
  size_t rots, len;
  size_t *ci, *cj; /* [rots][len] coordinate pairs */
  ft *ang;         /* [rots][len] */
} givens_set;

typedef struct {
  givens_set before, after;
  size_t *perm_b, *perm_ai; /* both length d_max */
} transform_t;

/* rand_rot, rand_pr.c:10-16, repeated `rots` times (alg.c:37-56). */
static givens_set draw_givens(size_t rots, size_t len, size_t dim) {
  givens_set g;
  g.rots = rots;
  g.len = len;
  g.ci = malloc(sizeof(size_t) * (rots * len + 1));
  g.cj = malloc(sizeof(size_t) * (rots * len + 1));
  g.ang = malloc(sizeof(ft) * (rots * len + 1));
  for (size_t r = 0; r < rots; r++) {
    size_t *sel = oracle_rand_perm(2 * len, dim);
    for (size_t i = 0; i < len; i++) {
      g.ci[r * len + i] = sel[2 * i];
      g.cj[r * len + i] = sel[2 * i + 1];
      g.ang[r * len + i] = (ft)(unit_draw() * M_PI);
    }
    free(sel);
  }
  return g;
}

/* make_ortho_info, alg.c:59-74: draw order is Q12. */
static transform_t draw_transform(size_t rlb, size_t rb, size_t rla, size_t ra,
                                  size_t ds, size_t d, size_t d_max) {
  transform_t t;
  t.before = draw_givens(rb, rlb, d);
  t.after = draw_givens(ra, rla, ds);
  t.perm_b = oracle_rand_perm(d, d_max);
  t.perm_ai = oracle_rand_perm(ds, d_max);
  return t;
}

static void free_transform(transform_t *t) {
  free(t->before.ci), free(t->before.cj), free(t->before.ang);
  free(t->after.ci), free(t->after.cj), free(t->after.ang);
  free(t->perm_b), free(t->perm_ai);
}

/* compute.cl:55-68 on one row; cos/sin via double libm then rounded (Q11). */
static void givens_row(ft *row, size_t k, size_t l, ft angle) {
  ft c = (ft)cos((double)angle), s = (ft)sin((double)angle);
  ft q = row[k] * c - row[l] * s;
  ft r = row[k] * s + row[l] * c;
  row[k] = q;
  row[l] = r;
}

/* walsh (alg.c:112-120) + apply_walsh_step (compute.cl:101-122) on one row
 * of length 2^l, l >= 4 (Q14). */
static void fwht_row(ft *a, unsigned l) {
  size_t len = (size_t)1 << l;
  for (unsigned step = 0; step < l; step++) {
    ft div = (ft)(step % 2 + 1);
    for (size_t b = 0; b < len / 2; b++) {
      size_t hi = (b >> step) << step, lo = b ^ hi;
      size_t ia = hi << 1 | lo, ib = ia | (size_t)1 << step;
      ft x = a[ia], y = a[ib];
      a[ia] = (x + y) / div;
      a[ib] = (x - y) / div;
    }
    if (step == 0 && (l & 1)) {
      ft scale = (ft)(1 / sqrt(2.0));
      for (size_t i = 0; i < len; i++)
        a[i] *= scale;
    }
  }
}

/* run_initial (alg.c:154-183) for one centred row -> ds low coordinates. */
static void forward_row(const transform_t *t, size_t ds, size_t d, size_t d_max,
                        const ft *src, ft *work, ft *wide, ft *low) {
  memcpy(work, src, sizeof(ft) * d);
  for (size_t r = 0; r < t->before.rots; r++)
    for (size_t y = 0; y < t->before.len; y++)
      givens_row(work, t->before.ci[r * t->before.len + y],
                 t->before.cj[r * t->before.len + y],
                 t->before.ang[r * t->before.len + y]);
  for (size_t y = 0; y < d_max; y++)
    wide[y] = t->perm_b[y] < d ? work[t->perm_b[y]] : 0;
  fwht_row(wide, oracle_lg(d_max));
  for (size_t r = 0; r < t->after.rots; r++)
    for (size_t y = 0; y < t->after.len; y++)
      givens_row(wide, t->after.ci[r * t->after.len + y],
                 t->after.cj[r * t->after.len + y],
                 t->after.ang[r * t->after.len + y]);
  for (size_t y = 0; y < d_max; y++)
    if (t->perm_ai[y] < ds)
      low[t->perm_ai[y]] = wide[y];
}

/* save_vecs (alg.c:189-217): inverse chain applied to unit vector e_row of
 * R^ds, giving one row of bases[try] (length d). */
static void inverse_row(const transform_t *t, size_t ds, size_t d, size_t d_max,
                        size_t row, ft *wide, ft *out) {
  for (size_t y = 0; y < d_max; y++)
    wide[y] = t->perm_ai[y] < ds ? (ft)(t->perm_ai[y] == row) : 0;
  for (size_t r = t->after.rots; r-- > 0;)
    for (size_t y = 0; y < t->after.len; y++)
      givens_row(wide, t->after.cj[r * t->after.len + y],
                 t->after.ci[r * t->after.len + y],
                 t->after.ang[r * t->after.len + y]);
  fwht_row(wide, oracle_lg(d_max));
  for (size_t y = 0; y < d_max; y++)
    if (t->perm_b[y] < d)
      out[t->perm_b[y]] = wide[y];
  for (size_t r = t->before.rots; r-- > 0;)
    for (size_t y = 0; y < t->before.len; y++)
      givens_row(out, t->before.cj[r * t->before.len + y],
                 t->before.ci[r * t->before.len + y],
                 t->before.ang[r * t->before.len + y]);
}

/* compute_signs, compute.cl:223-231 (Q10): coord 0 is the MSB, raw sign bit. */
static size_t sign_code(size_t ds, const ft *low) {
  size_t r = 0;
  for (size_t i = 0; i < ds; i++) {
    ubits b;
    memcpy(&b, low + i, sizeof b);
    r = r << 1 | (size_t)(b >> (sizeof(ft) * 8 - 1));
  }
  return r;
}

/* -------------------------------------------------------- det_results */

/* det_results, alg.c:303-337, for ONE query row whose stage-1 row has
 * already been through oracle_topk_stage.  top_ids/top_key are its first k
 * entries.  nbr_stride/nbr = the graph (query: save->graph stride k;
 * precomp: the merged matrix itself, stride k*tries, Q16). */
static void refine_row(size_t n, size_t k, size_t d, const ft *a, const ft *points,
                       int excl, size_t self, const size_t *top_ids,
                       const ft *top_key, const size_t *nbr, size_t nbr_stride,
                       size_t *row_ids, ft *row_key, ft *scratch) {
  size_t L2 = k * (k + 1);
  for (size_t z = 0; z < k; z++) {
    row_ids[z] = top_ids[z];
    row_key[z] = top_key[z];
  }
  /* supercharge, compute.cl:252-263 (Q7). */
  for (size_t y = 0; y < k; y++)
    for (size_t z = 0; z < k; z++) {
      size_t parent = top_ids[y];
      row_ids[(y + 1) * k + z] =
          parent < n ? nbr[parent * nbr_stride + z] : (nbr[z] | n);
    }
  for (size_t j = k; j < L2; j++) {
    size_t id = row_ids[j];
    int ok = id < n && !(excl && id == self);
    row_key[j] = sq_dist(d, a, points + (ok ? id : 0) * d, ok, scratch);
  }
  oracle_topk_stage(L2, row_ids, row_key);
}

/* ------------------------------------------------------------ precomp */

/* MK_NAME(precomp), alg.c:342-434.  Returns malloc'd size_t[n*k]; *dists_o
 * (if non-NULL) malloc'd ft[n*k]; fills *save (if non-NULL). */
size_t *oracle_precomp(size_t n, size_t k, size_t d, const ft *points, int tries,
                       size_t rots_before, size_t rot_len_before,
                       size_t rots_after, size_t rot_len_after,
                       oracle_save_t *save, ft **dists_o) {
  size_t ds = (size_t)ceil(log2((ft)n / k)); /* Q13 */
  size_t d_max = 1;
  while (d_max < d)
    d_max <<= 1;
  if (ds > d_max)
    ds = d_max;
  size_t T = (size_t)tries;

  /* column means by the row tree, alg.c:122-128 + compute.cl:15-39 (Q4). */
  size_t half = n / 2;
  ft *acc = malloc(sizeof(ft) * (half ? half : 1) * d);
  for (size_t x = 0; x < half; x++)
    for (size_t y = 0; y < d; y++) {
      ft g = ((n & 1) && x == 0) ? points[(n - 1) * d + y] : 0;
      acc[x * d + y] = points[x * d + y] + points[(x + half) * d + y] + g;
    }
  for (size_t m = n >> 1; m >> 1; m >>= 1)
    for (size_t x = 0; x < m / 2; x++)
      for (size_t y = 0; y < d; y++) {
        ft g = (x == 0 && (m & 1)) ? acc[(m - 1) * d + y] : 0;
        acc[x * d + y] += acc[(x + m / 2) * d + y] + g;
      }
  ft *means = malloc(sizeof(ft) * d);
  for (size_t y = 0; y < d; y++) {
    acc[y] /= n;
    means[y] = acc[y];
  }
  free(acc);
  ft *centred = malloc(sizeof(ft) * n * d);
  for (size_t x = 0; x < n; x++)
    for (size_t y = 0; y < d; y++)
      centred[x * d + y] = points[x * d + y] - means[y];

  if (save) {
    save->tries = tries;
    save->n = n;
    save->k = k;
    save->d_short = ds;
    save->d_long = d;
    save->row_means = malloc(sizeof(ft) * d);
    memcpy(save->row_means, means, sizeof(ft) * d);
    save->which_par = malloc(sizeof(size_t *) * T);
    save->par_maxes = malloc(sizeof(size_t) * T);
    save->bases = malloc(sizeof(ft) * T * ds * d);
  }
  free(means);

  /* all transforms are drawn before any other work (alg.c:387-392, Q12). */
  transform_t *tf = malloc(sizeof(transform_t) * T);
  for (size_t t = 0; t < T; t++)
    tf[t] = draw_transform(rot_len_before, rots_before, rot_len_after,
                           rots_after, ds, d, d_max);

  size_t **codes = malloc(sizeof(size_t *) * T);
  ft *work = malloc(sizeof(ft) * d), *wide = malloc(sizeof(ft) * d_max);
  ft *low = malloc(sizeof(ft) * (ds ? ds : 1));
  for (size_t t = 0; t < T; t++) {
    codes[t] = malloc(sizeof(size_t) * n);
    for (size_t x = 0; x < n; x++) {
      forward_row(&tf[t], ds, d, d_max, centred + x * d, work, wide, low);
      codes[t][x] = sign_code(ds, low);
    }
    if (save)
      for (size_t r = 0; r < ds; r++)
        inverse_row(&tf[t], ds, d, d_max, r, wide, save->bases + (t * ds + r) * d);
    free_transform(&tf[t]);
  }
  free(tf), free(work), free(wide), free(low), free(centred);

  /* second_half, alg.c:245-290, fused per point. */
  size_t W = k * T;
  size_t *merged_ids = malloc(sizeof(size_t) * n * W);
  ft *merged_key = malloc(sizeof(ft) * n * W);
  ft *scratch = malloc(sizeof(ft) * d);
  size_t nb = (size_t)1 << ds;
  for (size_t t = 0; t < T; t++) {
    size_t *cnt = calloc(nb, sizeof(size_t));
    for (size_t j = 0; j < n; j++)
      cnt[codes[t][j]]++;
    size_t pm = cnt[0];
    for (size_t b = 1; b < nb; b++)
      if (pm < cnt[b])
        pm = cnt[b];
    size_t *table = malloc(sizeof(size_t) * (pm ? pm : 1) * nb);
    for (size_t b = 0; b < nb; b++)
      for (size_t l = cnt[b]; l < pm; l++)
        table[b * pm + l] = n;
    for (size_t j = 0; j < n; j++) /* ascending j fills from the back (Q8) */
      table[codes[t][j] * pm + --cnt[codes[t][j]]] = j;
    free(cnt);

    size_t L = (ds + 1) * pm;
    size_t *row_ids = malloc(sizeof(size_t) * (L ? L : 1));
    ft *row_key = malloc(sizeof(ft) * (L ? L : 1));
    for (size_t x = 0; x < n; x++) {
      size_t code = codes[t][x];
      for (size_t y = 0; y <= ds; y++) {
        size_t b = code ^ (y ? (size_t)1 << (y - 1) : 0);
        for (size_t z = 0; z < pm; z++) {
          size_t id = table[b * pm + z];
          int ok = id < n && id != x;
          row_ids[y * pm + z] = id;
          row_key[y * pm + z] =
              sq_dist(d, points + x * d, points + (ok ? id : 0) * d, ok, scratch);
        }
      }
      oracle_topk_stage(L, row_ids, row_key);
      memcpy(merged_ids + x * W + t * k, row_ids, sizeof(size_t) * k);
      memcpy(merged_key + x * W + t * k, row_key, sizeof(ft) * k);
    }
    free(row_ids), free(row_key);
    if (save) {
      save->which_par[t] = table;
      save->par_maxes[t] = pm;
    } else
      free(table);
    free(codes[t]);
  }
  free(codes);

  /* det_results with precomputed distances, graph == merged (alg.c:419-422). */
  for (size_t x = 0; x < n; x++)
    oracle_topk_stage(W, merged_ids + x * W, merged_key + x * W);
  size_t L2 = k * (k + 1);
  size_t *result = malloc(sizeof(size_t) * n * k);
  ft *rdist = malloc(sizeof(ft) * n * k);
  size_t *row_ids = malloc(sizeof(size_t) * L2);
  ft *row_key = malloc(sizeof(ft) * L2);
  for (size_t x = 0; x < n; x++) {
    refine_row(n, k, d, points + x * d, points, 1, x, merged_ids + x * W,
               merged_key + x * W, merged_ids, W, row_ids, row_key, scratch);
    memcpy(result + x * k, row_ids, sizeof(size_t) * k);
    memcpy(rdist + x * k, row_key, sizeof(ft) * k);
  }
  free(row_ids), free(row_key), free(merged_ids), free(merged_key), free(scratch);
  if (dists_o)
    *dists_o = rdist;
  else
    free(rdist);
  if (save) { /* Q19: save keeps one copy, caller gets another */
    save->graph = result;
    result = malloc(sizeof(size_t) * n * k);
    memcpy(result, save->graph, sizeof(size_t) * n * k);
  }
  return result;
}

/* ------------------------------------- precomp, sampled (large configs) */

/* The index-building half of oracle_precomp for a SAMPLE of the rows: column
 * means (alg.c:360-369), the T transforms drawn from random() in the
 * reference's order (Q12; the same number of draws as a full precomp),
 * bases (save_vecs, alg.c:189-217) and the hash codes of the sampled rows
 * (run_initial, alg.c:154-183).  Lets a test check a GPU-built save_t at
 * BASELINE sizes in seconds: means and bases in full, and -- by looking the
 * sampled points up in which_par[t][code] -- the bucket tables by sample.
 * means_out ft[d], bases_out ft[T*ds*d], codes_out size_t[nrows*T]
 * (codes_out[i*T+t] = code of point rows[i] in try t).  Returns d_short. */
size_t oracle_precomp_tables_sample(size_t n, size_t k, size_t d, const ft *points,
                                    int tries, size_t rots_before, size_t rot_len_before,
                                    size_t rots_after, size_t rot_len_after,
                                    size_t nrows, const size_t *rows, ft *means_out,
                                    ft *bases_out, size_t *codes_out) {
  size_t ds = (size_t)ceil(log2((ft)n / k)); /* Q13 */
  size_t d_max = 1;
  while (d_max < d)
    d_max <<= 1;
  if (ds > d_max)
    ds = d_max;
  size_t T = (size_t)tries, half = n / 2;
  ft *acc = malloc(sizeof(ft) * (half ? half : 1) * d);
  for (size_t x = 0; x < half; x++)
    for (size_t y = 0; y < d; y++) {
      ft g = ((n & 1) && x == 0) ? points[(n - 1) * d + y] : 0;
      acc[x * d + y] = points[x * d + y] + points[(x + half) * d + y] + g;
    }
  for (size_t m = n >> 1; m >> 1; m >>= 1)
    for (size_t x = 0; x < m / 2; x++)
      for (size_t y = 0; y < d; y++) {
        ft g = (x == 0 && (m & 1)) ? acc[(m - 1) * d + y] : 0;
        acc[x * d + y] += acc[(x + m / 2) * d + y] + g;
      }
  for (size_t y = 0; y < d; y++) {
    acc[y] /= n;
    means_out[y] = acc[y];
  }
  free(acc);
  transform_t *tf = malloc(sizeof(transform_t) * T);
  for (size_t t = 0; t < T; t++)
    tf[t] = draw_transform(rot_len_before, rots_before, rot_len_after, rots_after, ds, d, d_max);
  ft *work = malloc(sizeof(ft) * d), *wide = malloc(sizeof(ft) * d_max);
  ft *low = malloc(sizeof(ft) * (ds ? ds : 1)), *cen = malloc(sizeof(ft) * d);
  for (size_t t = 0; t < T; t++) {
    for (size_t i = 0; i < nrows; i++) {
      for (size_t y = 0; y < d; y++)
        cen[y] = points[rows[i] * d + y] - means_out[y];
      forward_row(&tf[t], ds, d, d_max, cen, work, wide, low);
      codes_out[i * T + t] = sign_code(ds, low);
    }
    for (size_t r = 0; r < ds; r++)
      inverse_row(&tf[t], ds, d, d_max, r, wide, bases_out + (t * ds + r) * d);
    free_transform(&tf[t]);
  }
  free(tf), free(work), free(wide), free(low), free(cen);
  return ds;
}

/* Merged, sorted stage-1 row of point x (second_half per try, alg.c:245-290,
 * then det_results' first topk_stage over the k*T merged entries,
 * alg.c:308-312) from a COMPLETE index: which_par gives the candidates, the
 * point's own code in try t is the bucket it sits in (code_of[t][x]). */
static void merged_row_of(const oracle_save_t *save, const ft *points, size_t x,
                          size_t *const *code_of, size_t *ids, ft *key, ft *scratch) {
  size_t n = save->n, k = save->k, d = save->d_long, ds = save->d_short;
  size_t T = (size_t)save->tries, W = k * T;
  for (size_t t = 0; t < T; t++) {
    size_t pm = save->par_maxes[t], L = (ds + 1) * pm, code = code_of[t][x];
    size_t *row_ids = malloc(sizeof(size_t) * at_least_1(L));
    ft *row_key = malloc(sizeof(ft) * at_least_1(L));
    for (size_t y = 0; y <= ds; y++) {
      size_t b = code ^ (y ? (size_t)1 << (y - 1) : 0);
      for (size_t z = 0; z < pm; z++) {
        size_t id = save->which_par[t][b * pm + z];
        int ok = id < n && id != x;
        row_ids[y * pm + z] = id;
        row_key[y * pm + z] = sq_dist(d, points + x * d, points + (ok ? id : 0) * d, ok, scratch);
      }
    }
    oracle_topk_stage(L, row_ids, row_key);
    memcpy(ids + t * k, row_ids, sizeof(size_t) * k);
    memcpy(key + t * k, row_key, sizeof(ft) * k);
    free(row_ids), free(row_key);
  }
  oracle_topk_stage(W, ids, key);
}

/* Graph rows (and their squared distances) of the points rows[0..nrows) as
 * precomp returns them (alg.c:419-433), computed from a complete save_t: the
 * merged rows of each sampled point and of the k points its refinement reads
 * (Q16), then refine_row.  out_ids size_t[nrows*k], out_dists ft[nrows*k].
 * Returns 0, or -1 if a point is missing from a bucket table. */
int oracle_precomp_graph_rows(const oracle_save_t *save, const ft *points, size_t nrows,
                              const size_t *rows, size_t *out_ids, ft *out_dists) {
  size_t n = save->n, k = save->k, d = save->d_long, ds = save->d_short;
  size_t T = (size_t)save->tries, W = k * T, L2 = k * (k + 1), nb = (size_t)1 << ds;
  size_t **code_of = malloc(sizeof(size_t *) * T);
  int bad = 0;
  for (size_t t = 0; t < T; t++) {
    code_of[t] = malloc(sizeof(size_t) * n);
    for (size_t x = 0; x < n; x++)
      code_of[t][x] = (size_t)-1;
    size_t pm = save->par_maxes[t];
    for (size_t b = 0; b < nb; b++)
      for (size_t z = 0; z < pm; z++) {
        size_t id = save->which_par[t][b * pm + z];
        if (id < n)
          code_of[t][id] = b;
      }
    for (size_t x = 0; x < n; x++)
      bad |= code_of[t][x] == (size_t)-1;
  }
  ft *scratch = malloc(sizeof(ft) * d);
  size_t *mx_i = malloc(sizeof(size_t) * W), *par_i = malloc(sizeof(size_t) * W);
  ft *mx_k = malloc(sizeof(ft) * W), *par_k = malloc(sizeof(ft) * W);
  size_t *row_ids = malloc(sizeof(size_t) * L2);
  ft *row_key = malloc(sizeof(ft) * L2);
  for (size_t i = 0; i < nrows && !bad; i++) {
    size_t x = rows[i];
    merged_row_of(save, points, x, code_of, mx_i, mx_k, scratch);
    for (size_t z = 0; z < k; z++) {
      row_ids[z] = mx_i[z];
      row_key[z] = mx_k[z];
    }
    for (size_t y = 0; y < k; y++) { /* supercharge, compute.cl:252-263 (Q7), graph == merged rows (Q16) */
      size_t parent = mx_i[y];
      merged_row_of(save, points, parent < n ? parent : 0, code_of, par_i, par_k, scratch);
      for (size_t z = 0; z < k; z++)
        row_ids[(y + 1) * k + z] = parent < n ? par_i[z] : (par_i[z] | n);
    }
    for (size_t j = k; j < L2; j++) {
      size_t id = row_ids[j];
      int ok = id < n && id != x;
      row_key[j] = sq_dist(d, points + x * d, points + (ok ? id : 0) * d, ok, scratch);
    }
    oracle_topk_stage(L2, row_ids, row_key);
    memcpy(out_ids + i * k, row_ids, sizeof(size_t) * k);
    memcpy(out_dists + i * k, row_key, sizeof(ft) * k);
  }
  for (size_t t = 0; t < T; t++)
    free(code_of[t]);
  free(code_of), free(scratch), free(mx_i), free(mx_k), free(par_i), free(par_k);
  free(row_ids), free(row_key);
  return bad ? -1 : 0;
}

/* -------------------------------------------------------------- query */

/* Hash codes of a query batch in the reference's WRITE layout
 * codes[q*T + t] (alg.c:462-492).  Exposed for tests. */
void oracle_query_codes(const oracle_save_t *save, size_t ycnt, const ft *y,
                        size_t *codes) {
  size_t d = save->d_long, ds = save->d_short, T = (size_t)save->tries;
  ft *u = malloc(sizeof(ft) * d), *m = malloc(sizeof(ft) * d);
  ft *low = malloc(sizeof(ft) * (ds ? ds : 1));
  for (size_t q = 0; q < ycnt; q++) {
    for (size_t z = 0; z < d; z++)
      u[z] = y[q * d + z] - save->row_means[z];
    for (size_t t = 0; t < T; t++) {
      for (size_t s = 0; s < ds; s++) {
        const ft *b = save->bases + (t * ds + s) * d;
        for (size_t z = 0; z < d; z++)
          m[z] = u[z] * b[z];
        low[s] = oracle_tree_sum(d, m);
      }
      codes[q * T + t] = sign_code(ds, low);
    }
  }
  free(u), free(m), free(low);
}

/* Statistics the harness needs for the algorithmic-bytes model. */
typedef struct {
  size_t L1, P1, L2, P2;
  unsigned long long valid1, valid2; /* summed over queries */
} oracle_query_stats;

/* MK_NAME(query), alg.c:458-519.  `alias` != 0 reproduces the pointer-equality
 * self exclusion of compute.cl:144-146 when the caller passes y == points. */
size_t *oracle_query_ex(const oracle_save_t *save, const ft *points, size_t ycnt,
                        const ft *y, ft **dists_o, int alias,
                        oracle_query_stats *stats) {
  size_t n = save->n, k = save->k, d = save->d_long, ds = save->d_short;
  size_t T = (size_t)save->tries;
  size_t *codes = malloc(sizeof(size_t) * at_least_1(T * ycnt));
  oracle_query_codes(save, ycnt, y, codes);
  size_t M = 0;
  for (size_t t = 0; t < T; t++)
    M += save->par_maxes[t];
  size_t L1 = M * (ds + 1), L2 = k * (k + 1);
  size_t P1 = (size_t)1 << oracle_lg(L1), P2 = (size_t)1 << oracle_lg(L2);
  if (stats) {
    stats->L1 = L1, stats->P1 = P1, stats->L2 = L2, stats->P2 = P2;
    stats->valid1 = stats->valid2 = 0;
  }
  size_t *row_ids = malloc(sizeof(size_t) * (L1 ? L1 : 1));
  ft *row_key = malloc(sizeof(ft) * (L1 ? L1 : 1));
  size_t *r2_ids = malloc(sizeof(size_t) * L2);
  ft *r2_key = malloc(sizeof(ft) * L2);
  ft *scratch = malloc(sizeof(ft) * d);
  size_t *result = malloc(sizeof(size_t) * at_least_1(ycnt * k));
  ft *rdist = malloc(sizeof(ft) * at_least_1(ycnt * k));
  for (size_t x = 0; x < ycnt; x++) {
    const ft *a = y + x * d;
    size_t off = 0;
    for (size_t t = 0; t < T; t++) {
      size_t pm = save->par_maxes[t];
      size_t code = codes[t * ycnt + x]; /* Q2: read layout != write layout */
      for (size_t yy = 0; yy <= ds; yy++) {
        size_t b = code ^ (yy ? (size_t)1 << (yy - 1) : 0);
        for (size_t z = 0; z < pm; z++) {
          size_t slot = off + yy * pm + z;
          size_t id = save->which_par[t][b * pm + z];
          int ok = id < n && !(alias && id == x);
          row_ids[slot] = id;
          row_key[slot] = sq_dist(d, a, points + (ok ? id : 0) * d, ok, scratch);
          if (stats && ok && slot < P1)
            stats->valid1++;
        }
      }
      off += pm * (ds + 1);
    }
    oracle_topk_stage(L1, row_ids, row_key);
    if (stats) {
      for (size_t yy = 0; yy < k; yy++)
        for (size_t z = 0; z < k; z++) {
          size_t slot = (yy + 1) * k + z;
          size_t par = row_ids[yy];
          size_t id = par < n ? save->graph[par * k + z] : (save->graph[z] | n);
          if (slot < P2 && id < n && !(alias && id == x))
            stats->valid2++;
        }
    }
    refine_row(n, k, d, a, points, alias, x, row_ids, row_key, save->graph, k,
               r2_ids, r2_key, scratch);
    memcpy(result + x * k, r2_ids, sizeof(size_t) * k);
    memcpy(rdist + x * k, r2_key, sizeof(ft) * k);
  }
  free(codes), free(row_ids), free(row_key), free(r2_ids), free(r2_key);
  free(scratch);
  if (dists_o)
    *dists_o = rdist;
  else
    free(rdist);
  return result;
}

size_t *oracle_query(const oracle_save_t *save, const ft *points, size_t ycnt,
                     const ft *y, ft **dists_o) {
  return oracle_query_ex(save, points, ycnt, y, dists_o, y == points, NULL);
}

/* free_save, ann.c:25-34. */
void oracle_free_save(oracle_save_t *save) {
  for (int i = 0; i < save->tries; i++)
    free(save->which_par[i]);
  free(save->which_par);
  free(save->par_maxes);
  free(save->graph);
  free(save->row_means);
  free(save->bases);
}

void oracle_free(void *p) { free(p); }
