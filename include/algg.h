/* GPU backend entry points: exactly the two symbols the reference's dispatcher binds
 * (/root/reference/algg.h:5-11, called from /root/reference/ann.c:11,21).  Here they are
 * implemented with hand-written HIP kernels for gfx950 instead of OpenCL. */
#ifndef APPROXNN_HIP_ALGG_H
#define APPROXNN_HIP_ALGG_H
#include "ann.h"
#ifdef __cplusplus
extern "C" {
#endif

/* replaces query_gpu, /root/reference/algg.h:5-6 (body: alg.c:458-519 via alggp.c).
 * If y == points (pointer equality) every query excludes its own row, as compute.cl:144-146 does. */
size_t *query_gpu(const save_t *save, const ftype *points, size_t ycnt, const ftype *y,
                  ftype **dists_o);

/* replaces precomp_gpu, /root/reference/algg.h:7-11 (body: alg.c:342-434 via alggp.c).
 * Consumes libc random() in the reference's order, so srandom(s) gives the same index. */
size_t *precomp_gpu(size_t n, size_t k, size_t d, const ftype *points, int tries,
                    size_t rots_before, size_t rot_len_before, size_t rots_after,
                    size_t rot_len_after, save_t *save, ftype **dists_o);

#ifdef __cplusplus
}
#endif
#endif
