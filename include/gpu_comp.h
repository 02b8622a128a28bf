/* Device lifecycle symbols the reference's drivers call around precomp/query
 * (/root/reference/gpu_comp.h:11-13; callers time_results.c:87-88,134-135,
 * compare_results.c:94,137).  Re-issued without <CL/opencl.h>: the device is a HIP device. */
#ifndef APPROXNN_HIP_GPU_COMP_H
#define APPROXNN_HIP_GPU_COMP_H
#ifdef __cplusplus
extern "C" {
#endif

/* replaces gpu_init, /root/reference/gpu_comp.c:21-91: selects HIP device $ANN_HIP_DEVICE (default 0);
 * prints to stderr and exit(1)s when there is no gfx950-class device. Idempotent. */
void gpu_init(void);
/* replaces gpu_cleanup, /root/reference/gpu_comp.c:103-114: runs registered callbacks (LIFO),
 * drops every cached device-resident index. */
void gpu_cleanup(void);
/* replaces register_cleanup, /root/reference/gpu_comp.c:93-101: runs f at gpu_cleanup(), or
 * immediately when gpu_init() has not been called. */
void register_cleanup(void (*f)(void));

#ifdef __cplusplus
}
#endif
#endif
