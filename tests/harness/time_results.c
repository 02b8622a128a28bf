/* time_results -- wall-clock timing through the reference's host-pointer API, the counterpart of
 * /root/reference/time_results.c:92-141 (same timed region: around query() resp. precomp() only; same data: N(0,1) by
 * Box-Muller on libc random(), points first, then precomp's draws, then one draw per batch), with a fixed seed (-S),
 * queries/s, the ALGORITHMIC bytes and GB/s of the stage-1 kernel against the 8 TB/s HBM peak, and -- in
 * the same run -- the CPU column on the same index: the reference's own query_cpu when oracle/_ref/libref_<prec>.so
 * (built from /root/reference by `make -C oracle ref`) is present, and the oracle, 1 core each.
 * These are the PCIe-inclusive numbers (y goes up and ids/distances come back through host memory on every call);
 * bench.py reports the HBM-resident rate.                                                                          */
#include <dlfcn.h>
#include <libgen.h>

#include "harness_common.h"
#include "ann_hip.h"

typedef size_t *(*query_cpu_fn)(const save_t *, const ftype *, size_t, const ftype *, ftype **);

/* the reference's CPU path, if its library travelled with the repo (oracle/_ref/, never part of the product) */
static query_cpu_fn load_reference(const char *argv0) {
  char buf[4096], self[4096];
  ssize_t len = readlink("/proc/self/exe", self, sizeof self - 1);
  if (len <= 0) snprintf(self, sizeof self, "%s", argv0); else self[len] = 0;
#ifdef USE_FLOAT
  snprintf(buf, sizeof buf, "%s/../../oracle/_ref/libref_f32.so", dirname(self));
#else
  snprintf(buf, sizeof buf, "%s/../../oracle/_ref/libref_f64.so", dirname(self));
#endif
  void *h = dlopen(buf, RTLD_NOW | RTLD_LOCAL);
  return h ? (query_cpu_fn)dlsym(h, "query_cpu") : NULL;
}

int main(int argc, char **argv) {
  opts_t o = parse_opts(argc, argv, "n:k:d:t:o:y:b:s:a:r:S:P:C:G:V:Fhzvc", 10);
  srandom(o.seed);
  if (!o.use_cpu) gpu_init();
  if (o.devices > 1 || o.vshards > 0) annhip_set_devices(o.devices, o.vshards);
  ftype *points = malloc(sizeof(ftype) * o.n * o.d), *dists;
  annhip_synth_reset();
  annhip_synth_randnorm(o.n * o.d, points);                 /* genRand, time_results.c:94 */
  const double HBM_PEAK = 8000.0;                            /* GB/s, MI355X HBM3E */
  if (o.ycnt) {
    save_t save;
    double t0 = now_s();
    if (o.use_cpu)
      free(oracle_precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL));
    else
      free(precomp(o.n, o.k, o.d, points, (int)o.tries, o.rb, o.rlenb, o.ra, o.rlena, &save, NULL, 0));
    printf("precomp (with save) on %cPU: %.3f s\n", o.use_cpu ? 'C' : 'G', now_s() - t0);
    ftype *y = malloc(sizeof(ftype) * o.ycnt * o.d), *y0 = malloc(sizeof(ftype) * o.ycnt * o.d);
    double tg = 0, first = 0;
    if (!o.use_cpu) annhip_host_profile(1);
    for (size_t i = 0; i < o.reps; i++) {
      annhip_synth_randnorm(o.ycnt * o.d, y);                /* time_results.c:103 */
      if (i == 0) memcpy(y0, y, sizeof(ftype) * o.ycnt * o.d);
      if (!o.use_cpu) {
        if (i == 1) {                                        /* statistics of the warm calls only */
          double dummy[8];
          annhip_host_stats(&save, dummy, 1);
        }
        t0 = now_s();
        size_t *r = query(&save, points, o.ycnt, y, &dists, 0);
        double dt = now_s() - t0;
        if (i == 0) first = dt; else tg += dt;   /* first call = cold (index upload), reported apart */
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
      double st[8];
      if (annhip_host_stats(&save, st, 0) == 0 && st[0] > 0 && st[5] > 0) {
        /* bytes/query = V1*d*s [row gathers] + P(L1)*4 [candidate ids] + d*s [query] + T*4 [codes]
         *              + (k+1)*(s+4) [candidates out], V1 = rows actually gathered per query (counted on the device) */
        const double s = sizeof(ftype), v1 = st[2] / st[5], kern_ms = st[1] / st[0];
        const double bpq = v1 * o.d * s + st[6] * 4 + o.d * s + o.tries * 4.0 + (o.k + 1) * (s + 4);
        const double gb = bpq * o.ycnt / 1e9, gbs = gb / (kern_ms * 1e-3);
        printf("stage-1 kernel (candidate gather + L2 + top-k): %.4f ms per %zu-query batch, %.1f rows gathered per query, "
               "%.3f GB algorithmic  => %.1f GB/s = %.1f %% of the %.0f GB/s HBM peak\n", kern_ms, o.ycnt, v1, gb, gbs,
               100 * gbs / HBM_PEAK, HBM_PEAK);
        printf("whole call (host buffers in and out, PCIe included): %.1f GB/s algorithmic = %.1f %% of the HBM peak\n",
               gb / avg, 100 * gb / avg / HBM_PEAK);
        /* rows sharded over several devices (-G / -V): every device's own gather against its own HBM */
        const int shards = annhip_host_shards(&save);
        for (int g = 0; shards > 1 && g < shards; g++) {
          double sg[8];
          if (annhip_host_stats_shard(&save, g, sg, 0) != 0 || sg[0] <= 0 || sg[5] <= 0) continue;
          const double v1g = sg[2] / sg[5], msg = sg[1] / sg[0];
          const double bg = (v1g * o.d * s + sg[6] * 4 + o.d * s + o.tries * 4.0 + (o.k + 1) * (s + 4)) * o.ycnt / 1e9;
          printf("  shard %d of %d: stage-1 kernel %.4f ms, %.1f rows gathered per query, %.3f GB  => %.1f GB/s = %.1f %% of its HBM peak\n",
                 g, shards, msg, v1g, bg, bg / (msg * 1e-3), 100 * bg / (msg * 1e-3) / HBM_PEAK);
        }
      }
      if (o.fp_cost)
        printf("residency-cache fingerprint per query() call: sampled (default) %.3f ms, strict (ANN_HIP_CACHE=strict, full "
               "content) %.1f ms\n", annhip_fingerprint_ms(&save, points, 0), annhip_fingerprint_ms(&save, points, 1));
    }
    if (!o.use_cpu && o.lanes > 0) {
      /* the same kind of batches through the pipelined host API: uploads, kernels and downloads overlap */
      annhip_index *ix = annhip_index_create(&save, points, 0, 0, save.n);
      annhip_stream *st = annhip_stream_open(ix, o.ycnt, o.lanes);
      size_t nb = o.reps * 4 + (size_t)o.lanes;
      ftype *ys = malloc(sizeof(ftype) * o.ycnt * o.d * (size_t)o.lanes);
      size_t *ids = malloc(sizeof(size_t) * o.ycnt * o.k);
      ftype *dd = malloc(sizeof(ftype) * o.ycnt * o.k);
      for (int l = 0; l < o.lanes; l++) annhip_synth_randnorm(o.ycnt * o.d, ys + (size_t)l * o.ycnt * o.d);
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
    /* CPU column, same index, 1 core.  -C n bounds the batch (the reference materialises ycnt*L1*d values and is
     * slow; results depend on the batch size, SURVEY Q2, so this is a timing of the same kind of work, not a check) */
    const long ncores = sysconf(_SC_NPROCESSORS_ONLN);
    size_t cq = o.cpu_queries ? (o.cpu_queries < o.ycnt ? o.cpu_queries : o.ycnt) : o.ycnt;
    query_cpu_fn ref = load_reference(argv[0]);
    if (ref) {
      size_t rq = o.cpu_queries ? cq : (cq < 128 ? cq : 128);
      t0 = now_s();
      size_t *r = ref(&save, points, rq, y0, &dists);
      double tr = now_s() - t0;
      free(r), free(dists);
      printf("Average time for query (on CPU, the reference's query_cpu, 1 of %ld cores, %zu-query batch): %gs  => %.1f queries/s\n",
             ncores, rq, tr, rq / tr);
    }
    t0 = now_s();
    size_t *r = oracle_query(&save, points, cq, y0, &dists);
    double tc = now_s() - t0;
    free(r), free(dists);
    printf("Average time for query (on CPU, oracle, 1 of %ld cores, %zu-query batch): %gs  => %.0f queries/s\n", ncores, cq,
           tc, cq / tc);
    free(y), free(y0);
    free_save(&save);
  } else {
    double t = 0;
    for (size_t i = 0; i < o.reps; i++) {
      save_t save;
      if (i) annhip_synth_randnorm(o.n * o.d, points);
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
