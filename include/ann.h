/* Public surface kept from the reference (/root/reference/ann.h:8-12,46-49,61-62,65).
 *
 * save_t is the precomputed index.  Its layout is ABI: an index built by this library must be
 * consumable by the reference's query_cpu and the other way round
 * (/root/reference/compare_results.c:101-110), and every pointer field is plain malloc() memory
 * because free_save() free()s them (/root/reference/ann.c:25-34).
 *
 *   which_par[t] : size_t[2^d_short][par_maxes[t]]  bucket table of try t, ids descending, padded with n
 *   graph        : size_t[n][k]                      approximate kNN graph of the point set
 *   row_means    : ftype[d_long]                     column means of the points
 *   bases        : ftype[tries][d_short][d_long]     rows of the random projections
 */
#ifndef APPROXNN_HIP_ANN_H
#define APPROXNN_HIP_ANN_H
#include <stddef.h>
#include "ftype.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int tries;
  size_t n, k, d_short, d_long, **which_par, *par_maxes, *graph;
  ftype *row_means, *bases;
} save_t;

/* kNN graph of `points` (row-major [n][d]) and, if save != NULL, the index for later queries.
 * Returns malloc'd size_t[n*k]; *dists (if non-NULL) receives malloc'd squared distances ftype[n*k].
 * use_cpu != 0 asks for the reference's single-core CPU path, which is NOT part of this library:
 * link the reference's algc.o for it; the bundled dispatcher exits loudly instead. */
size_t *precomp(size_t n, size_t k, size_t d, const ftype *points, int tries,
                size_t rots_before, size_t rot_len_before, size_t rots_after,
                size_t rot_len_after, save_t *save, ftype **dists, char use_cpu);

/* k approximate nearest neighbours of each of the ycnt rows of y among `points` (the matrix the
 * index was built from).  Ownership of the results as for precomp(). */
size_t *query(const save_t *save, const ftype *points, size_t ycnt, const ftype *y,
              ftype **dists, char use_cpu);

void free_save(save_t *save);

#ifdef __cplusplus
}
#endif
#endif
