/* Precision switch of the drop-in surface.
 * Replaces /root/reference/ftype.h:3-9: the stock build is double, -DUSE_FLOAT selects float.
 * One shared library is built per precision (libapproxnn_hip_f64.so / libapproxnn_hip_f32.so). */
#ifndef APPROXNN_HIP_FTYPE_H
#define APPROXNN_HIP_FTYPE_H
#ifdef USE_FLOAT
typedef float ftype;
typedef int i_ftype;
#else
typedef double ftype;
typedef long i_ftype;
#endif
#endif
