/* ann_dispatch.c -- precomp()/query()/free_save() for programs that link this library INSTEAD of the
 * reference's ann.o (/root/reference/ann.c:6-34).  The GPU branch is the HIP backend.  The CPU branch of
 * the reference (algc.o) is not part of this product: asking for it here fails loudly instead of silently
 * running something else.  A maintainer who wants both keeps the reference's ann.o + algc.o and links
 * libapproxnn_hip for precomp_gpu/query_gpu. */
#include <stdio.h>
#include <stdlib.h>

#include "algg.h"
#include "ann.h"
#include "ann_hip.h"

static void no_cpu_path(const char *what) {
  fprintf(stderr, "%s: use_cpu != 0 requested, but the CPU path (algc.c) is not part of the HIP backend\n", what);
  exit(1);
}

size_t *query(const save_t *save, const ftype *points, size_t ycnt, const ftype *y, ftype **dists, char use_cpu) {
  if (use_cpu) no_cpu_path("query");
  return query_gpu(save, points, ycnt, y, dists);
}

size_t *precomp(size_t n, size_t k, size_t d, const ftype *points, int tries, size_t rots_before,
                size_t rot_len_before, size_t rots_after, size_t rot_len_after, save_t *save, ftype **dists,
                char use_cpu) {
  if (use_cpu) no_cpu_path("precomp");
  return precomp_gpu(n, k, d, points, tries, rots_before, rot_len_before, rots_after, rot_len_after, save, dists);
}

/* every field is plain malloc memory (ann.c:25-34) */
void free_save(save_t *save) {
  annhip_cache_drop(save); /* the resident copy of this index goes with it */
  for (int t = 0; t < save->tries; t++) free(save->which_par[t]);
  free(save->which_par);
  free(save->par_maxes);
  free(save->graph);
  free(save->row_means);
  free(save->bases);
}
