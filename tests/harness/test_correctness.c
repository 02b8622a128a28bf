/* test_correctness -- recall of the answers, the counterpart of /root/reference/test_correctness.c:92-141: same options,
 * same data (N(0,1) by Box-Muller on libc random(), with -S for a fixed seed), the same three numbers
 *     Average index score   (mean rank excess per neighbour:  (sum of ranks per row - k(k-1)/2) / k)
 *     Prob correct          (share of the guessed neighbours that are among the true k nearest)
 *     Max index score       (largest rank seen / k)
 * for precomp()'s graph (default) or for query() batches (-y / -z), through precomp()/query() of include/ann.h.
 * The ranks come from one brute-force pass on the GPU with the query path's exact distance arithmetic
 * (annhip_recall_ranks_host: rank = number of points STRICTLY closer) where the reference sorts all n distances per row on the
 * CPU (test_correctness.c:169-262; its distance loop starts at `z = step % 1`, i.e. always 0, and so counts one term twice
 * whenever a tree level is odd -- d = 80 has one -- which perturbs ITS ground truth: SURVEY 4, DESIGN.md "Quality").
 * -c scores the oracle's answers instead (CPU precomp/query), same scorer: the two columns must agree, the answers being
 * bit-identical.  A quality metric, not a parity test: parity is compare_results'.                                   */
#include "harness_common.h"
#include "ann_hip.h"

static void score(size_t n, size_t k, size_t d, const ftype *points, size_t rows, const ftype *y, const size_t *guess,
                  int self, double *sum, double *wrong, double *max) {
  unsigned long long *r = malloc(sizeof(unsigned long long) * rows * k);
  annhip_recall_ranks_host(n, d, k, points, rows, y, guess, self, r);
  double f = 0, g = 0;
  unsigned long long mx = 0;
  for (size_t i = 0; i < rows * k; i++) { /* cscore, test_correctness.c:246-262 */
    f += (double)r[i];
    g += r[i] >= k;
    if (r[i] > mx) mx = r[i];
  }
  *sum += f / rows, *wrong += g / rows / k, *max += (double)mx;
  free(r);
}

int main(int argc, char **argv) {
  opts_t o = parse_opts(argc, argv, "n:k:d:t:o:y:b:s:a:r:S:G:V:hvzc", 1);
  if (o.use_y && !o.ycnt) o.ycnt = 50;
  srandom(o.seed);
  gpu_init();
  if (o.devices > 1 || o.vshards > 0) annhip_set_devices(o.devices, o.vshards);
  double sc = 0, scb = 0, scc = 0;
  ftype *points = malloc(sizeof(ftype) * o.n * o.d);
  annhip_synth_reset();
  if (o.use_y) {
    save_t save;
    annhip_synth_randnorm(o.n * o.d, points);
    if (o.use_cpu) free(oracle_precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL));
    else free(precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL, 0));
    ftype *y = malloc(sizeof(ftype) * o.ycnt * o.d);
    for (size_t i = 0; i < o.reps; i++) {
      annhip_synth_randnorm(o.ycnt * o.d, y);
      size_t *g = o.use_cpu ? oracle_query(&save, points, o.ycnt, y, NULL) : query(&save, points, o.ycnt, y, NULL, 0);
      for (size_t j = 0; j < o.ycnt * o.k; j++)
        if (g[j] >= o.n) g[j] = 0; /* (n, +inf) fillers of a short candidate list: any wrong id will do for the scorer */
      score(o.n, o.k, o.d, points, o.ycnt, y, g, 0, &sc, &scb, &scc);
      free(g);
      if (o.verbose) printf("%zu ", i + 1), fflush(stdout);
    }
    free(y);
    free_save(&save);
  } else {
    for (size_t i = 0; i < o.reps; i++) {
      annhip_synth_randnorm(o.n * o.d, points);
      size_t *g = o.use_cpu ? oracle_precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, NULL, NULL)
                            : precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, NULL, NULL, 0);
      for (size_t j = 0; j < o.n * o.k; j++)
        if (g[j] >= o.n) g[j] = j / o.k; /* filler: the point itself is excluded by the scorer, so count it as wrong */
      score(o.n, o.k, o.d, points, o.n, points, g, 1, &sc, &scb, &scc);
      free(g);
      if (o.verbose) printf("%zu ", i + 1), fflush(stdout);
    }
  }
  gpu_cleanup();
  free(points);
  if (o.verbose) putchar('\n');
  printf("Average index score for %s (on %cPU): %g.\nProb correct: %g.\nMax index score: %g\n", o.use_y ? "query" : "comp",
         o.use_cpu ? 'C' : 'G', (sc / o.reps - o.k * (o.k - 1) / 2.) / o.k, 1 - scb / o.reps, scc / o.reps / o.k);
  return 0;
}
