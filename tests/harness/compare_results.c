/* compare_results -- GPU (HIP backend) vs CPU (oracle) agreement, the counterpart of
 * /root/reference/compare_results.c, tightened as SURVEY 8(c) asks:
 *   - precomp mode: both sides start from srandom(seed') (compare_results.c:123-130); every save_t field is
 *     compared as cdiff_save does (graph/which_par/par_maxes exact, bases/row_means in ULPs/1024), and the
 *     returned ids and squared distances too;
 *   - query mode (-y/-z): ONE index (GPU-built), query on both sides (compare_results.c:106-116);
 *   - distances ARE compared (<= 1e-5 relative for float, the north-star bound; in practice bit-identical);
 *   - exit status 1 on any id difference or distance violation (the reference always exits 0).            */
#include <limits.h>

#include "harness_common.h"
#include "ann_hip.h"

#ifdef USE_FLOAT
#define ifabs abs
#else
#define ifabs labs
#endif

static size_t diffcount(size_t cnt, const size_t *p, const size_t *q) { /* compare_results.c:20-25 */
  size_t c = 0;
  for (size_t i = 0; i < cnt; i++) c += p[i] != q[i];
  return c;
}

static size_t dist_violations(size_t cnt, const ftype *a, const ftype *b, size_t *bit_diffs) {
  size_t bad = 0;
  *bit_diffs = 0;
  for (size_t i = 0; i < cnt; i++) {
    if (memcmp(a + i, b + i, sizeof(ftype))) (*bit_diffs)++;
    if (isinf(a[i]) || isinf(b[i])) {
      bad += a[i] != b[i];
      continue;
    }
    double den = fabs((double)b[i]) > 0 ? fabs((double)b[i]) : 1;
    bad += fabs((double)a[i] - (double)b[i]) / den > 1e-5;
  }
  return bad;
}

static double cdiff_save(const save_t *a, const save_t *b) { /* compare_results.c:152-171 */
  if (a->tries != b->tries || a->d_short != b->d_short || a->k != b->k || a->d_long != b->d_long || a->n != b->n)
    return (double)ULONG_MAX;
  double c = 0;
  for (size_t i = 0; i < a->n * a->k; i++) c += a->graph[i] != b->graph[i];
  for (size_t i = 0; i < a->tries * a->d_short * a->d_long; i++)
    c += ifabs(((i_ftype *)a->bases)[i] - ((i_ftype *)b->bases)[i]) / 1024.;
  for (size_t i = 0; i < a->d_long; i++)
    c += ifabs(((i_ftype *)a->row_means)[i] - ((i_ftype *)b->row_means)[i]) / 1024.;
  for (int i = 0; i < a->tries; i++) {
    if (a->par_maxes[i] != b->par_maxes[i]) return (double)ULONG_MAX;
    for (size_t j = 0; j < a->par_maxes[i] << a->d_short; j++) c += a->which_par[i][j] != b->which_par[i][j];
  }
  return c;
}

int main(int argc, char **argv) {
  opts_t o = parse_opts(argc, argv, "n:k:d:t:o:y:b:s:a:r:S:G:V:hvz", 3);
  if (o.use_y && !o.ycnt) o.ycnt = 50;
  srandom(o.seed);
  gpu_init();
  if (o.devices > 1 || o.vshards > 0) annhip_set_devices(o.devices, o.vshards);
  ftype *points = malloc(sizeof(ftype) * o.n * o.d);
  double score = 0;
  size_t dist_bad = 0, dist_bits = 0, id_bad = 0;
  if (o.use_y) {
    save_t save;
    oracle_gen_rand(o.n * o.d, points);
    free(precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL, 0));
    ftype *y = malloc(sizeof(ftype) * o.ycnt * o.d);
    for (size_t i = 0; i < o.reps; i++) {
      ftype *dg, *dc;
      size_t bits;
      oracle_gen_rand(o.ycnt * o.d, y);
      size_t *g = query(&save, points, o.ycnt, y, &dg, 0);
      size_t *c = oracle_query(&save, points, o.ycnt, y, &dc);
      size_t df = diffcount(o.ycnt * o.k, g, c);
      score += df, id_bad += df;
      dist_bad += dist_violations(o.ycnt * o.k, dg, dc, &bits);
      dist_bits += bits;
      free(g), free(c), free(dg), free(dc);
      if (o.verbose) printf("%zu ", i + 1), fflush(stdout);
    }
    free(y);
    free_save(&save);
  } else {
    for (size_t i = 0; i < o.reps; i++) {
      save_t sg, sc;
      ftype *dg, *dc;
      size_t bits;
      oracle_gen_rand(o.n * o.d, points);
      unsigned inner = (unsigned)random();
      srandom(inner);
      size_t *g = precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &sg, &dg, 0);
      srandom(inner);
      size_t *c = oracle_precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &sc, &dc);
      double sdiff = cdiff_save(&sg, &sc);
      size_t df = diffcount(o.n * o.k, g, c);
      score += sdiff, id_bad += df + (sdiff != 0);
      dist_bad += dist_violations(o.n * o.k, dg, dc, &bits);
      dist_bits += bits;
      free(g), free(c), free(dg), free(dc);
      free_save(&sg), free_save(&sc);
      if (o.verbose) printf("%zu ", i + 1), fflush(stdout);
    }
  }
  gpu_cleanup();
  free(points);
  if (o.verbose) putchar('\n');
  printf("Average diffs for %s: %g\n", o.use_y ? "query" : "comp", score / o.reps);
  printf("distance check: %zu outside 1e-5 relative, %zu not bit-identical\n", dist_bad, dist_bits);
  int ok = id_bad == 0 && dist_bad == 0;
  printf("%s\n", ok ? "PASS" : "FAIL");
  return ok ? 0 : 1;
}
