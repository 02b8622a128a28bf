/* time_results -- wall-clock timing through the reference's host-pointer API, the counterpart of
 * /root/reference/time_results.c:92-141 (same timed region: around query() resp. precomp() only), with a
 * fixed seed (-S), queries/s, and -- in the same run -- the CPU column (oracle, 1 core) on the same index.
 * These are the PCIe-inclusive numbers (y goes up and ids/distances come back through host memory on every
 * call); bench.py reports the HBM-resident rate.                                                          */
#include "harness_common.h"
#include "ann_hip.h"

int main(int argc, char **argv) {
  opts_t o = parse_opts(argc, argv, "n:k:d:t:o:y:b:s:a:r:S:P:hzvc", 10);
  srandom(o.seed);
  if (!o.use_cpu) gpu_init();
  ftype *points = malloc(sizeof(ftype) * o.n * o.d), *dists;
  oracle_gen_rand(o.n * o.d, points);
  if (o.ycnt) {
    save_t save;
    double t0 = now_s();
    if (o.use_cpu)
      free(oracle_precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL));
    else
      free(precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL, 0));
    printf("precomp (with save) on %cPU: %.3f s\n", o.use_cpu ? 'C' : 'G', now_s() - t0);
    ftype *y = malloc(sizeof(ftype) * o.ycnt * o.d);
    double tg = 0, tc = 0, first = 0;
    size_t cpu_reps = 0;
    for (size_t i = 0; i < o.reps; i++) {
      oracle_gen_rand(o.ycnt * o.d, y);
      if (!o.use_cpu) {
        t0 = now_s();
        size_t *r = query(&save, points, o.ycnt, y, &dists, 0);
        double dt = now_s() - t0;
        if (i == 0) first = dt; else tg += dt;   /* first call = cold (index upload), reported apart */
        free(r), free(dists);
      }
      if (o.use_cpu || i == 0) { /* CPU column: every batch with -c, else one batch beside the GPU run */
        t0 = now_s();
        size_t *r = oracle_query(&save, points, o.ycnt, y, &dists);
        tc += now_s() - t0, cpu_reps++;
        free(r), free(dists);
      }
      if (o.verbose) printf("%zu ", i + 1), fflush(stdout);
    }
    if (o.verbose) putchar('\n');
    if (!o.use_cpu) {
      size_t warm = o.reps > 1 ? o.reps - 1 : 1;
      double avg = o.reps > 1 ? tg / warm : first;
      printf("Average time for query (on GPU, host buffers): %gs  => %.0f queries/s  (first call %gs)\n", avg,
             o.ycnt / avg, first);
    }
    if (!o.use_cpu && o.lanes > 0) {
      /* the same kind of batches through the pipelined host API: uploads, kernels and downloads overlap */
      annhip_index *ix = annhip_index_create(&save, points, 0, 0, save.n);
      annhip_stream *st = annhip_stream_open(ix, o.ycnt, o.lanes);
      size_t nb = o.reps * 4 + (size_t)o.lanes;
      ftype *ys = malloc(sizeof(ftype) * o.ycnt * o.d * (size_t)o.lanes);
      size_t *ids = malloc(sizeof(size_t) * o.ycnt * o.k);
      ftype *dd = malloc(sizeof(ftype) * o.ycnt * o.k);
      for (int l = 0; l < o.lanes; l++) oracle_gen_rand(o.ycnt * o.d, ys + (size_t)l * o.ycnt * o.d);
      long head = 0;
      double tp = 0;
      for (size_t b = 0; b < nb; b++) {
        if (b == (size_t)o.lanes) tp = now_s();  /* timed once the pipeline is full */
        long t = annhip_stream_submit(st, o.ycnt, ys + (b % o.lanes) * o.ycnt * o.d, 0);
        if (t < 0) {
          annhip_stream_collect(st, head++, ids, dd);
          t = annhip_stream_submit(st, o.ycnt, ys + (b % o.lanes) * o.ycnt * o.d, 0);
        }
      }
      while (head < (long)nb) annhip_stream_collect(st, head++, ids, dd);
      tp = now_s() - tp;
      double per = tp / (nb - o.lanes);
      printf("Average time for query (on GPU, host buffers, %d-lane pipeline): %gs  => %.0f queries/s\n", o.lanes, per,
             o.ycnt / per);
      annhip_stream_close(st);
      annhip_index_destroy(ix);
      free(ys), free(ids), free(dd);
    }
    printf("Average time for query (on CPU, oracle, 1 of %ld cores): %gs  => %.0f queries/s\n",
           sysconf(_SC_NPROCESSORS_ONLN), tc / cpu_reps, o.ycnt / (tc / cpu_reps));
    free(y);
    free_save(&save);
  } else {
    double t = 0;
    for (size_t i = 0; i < o.reps; i++) {
      save_t save;
      if (i) oracle_gen_rand(o.n * o.d, points);
      double t0 = now_s();
      size_t *r = o.use_cpu ? oracle_precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena,
                                             o.save_test ? &save : NULL, &dists)
                            : precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena,
                                      o.save_test ? &save : NULL, &dists, 0);
      t += now_s() - t0;
      if (o.save_test) free_save(&save);
      free(r), free(dists);
    }
    printf("Average time for %s (on %cPU): %gs\n", o.save_test ? "comp (with save)" : "comp (no save)",
           o.use_cpu ? 'C' : 'G', t / o.reps);
  }
  free(points);
  if (!o.use_cpu) gpu_cleanup();
  return 0;
}
