// ann_synth.cpp -- the reference drivers' synthetic data, for benchmarks and harnesses.  Host-only; no HIP calls.
//
// time_results.c / compare_results.c / test_correctness.c fill points and queries with iid N(0,1) values drawn by
// Box-Muller on libc random() (/root/reference/randNorm.c:9-21, genRand in time_results.c:10-13).  SURVEY 8(d) makes
// that stream the benchmark's input, so a driver seeded like the reference's sees the reference's data.  This is a
// restatement of the generator (same expressions, same draw order, same double libm calls, same rounding to ftype):
//     u1 = sqrt(log(U()) * -2);  u2 = U() * M_PI * 2;  return u1*cos(u2), then u1*sin(u2) on the next call
// with U() = random() / (RAND_MAX + 1).  The draws come from the CALLER's random() stream, strictly in order; only
// the transcendental part is spread over host threads (1.3 G values for cfg3's point matrix: ~50 s -> a few seconds).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/ann_hip.h"

namespace {
// the reference keeps the second value of a pair in a static between calls (randNorm.c:7,13-19)
bool g_have = false;
double g_next = 0;
inline double unit(long r) { return (double)(unsigned long)r / ((double)RAND_MAX + 1); }
}  // namespace

extern "C" void annhip_synth_reset(void) { g_have = false; }

extern "C" void annhip_synth_randnorm(size_t count, ftype *out) {
  size_t at = 0;
  if (count && g_have) {
    out[at++] = (ftype)g_next;
    g_have = false;
  }
  const size_t pairs = (count - at + 1) / 2;  // the last pair may leave one value pending for the next call
  const size_t CH = (size_t)1 << 21;
  std::vector<int32_t> r1(std::min(pairs, CH)), r2(std::min(pairs, CH));
  unsigned nthr = std::thread::hardware_concurrency();
  nthr = nthr < 1 ? 1 : nthr > 32 ? 32 : nthr;
  for (size_t p0 = 0; p0 < pairs; p0 += CH) {
    const size_t m = std::min(CH, pairs - p0);
    for (size_t i = 0; i < m; i++) {  // the stream itself is sequential: radius draw first, then the angle draw
      r1[i] = (int32_t)random();
      r2[i] = (int32_t)random();
    }
    ftype *dst = out + at + 2 * p0;
    const size_t room = count - at - 2 * p0;  // values still to write from this chunk on
    auto work = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) {
        const double u1 = sqrt(log(unit(r1[i])) * -2), u2 = unit(r2[i]) * M_PI * 2;
        // gcc -O2 (the reference's build, and the oracle's) fuses the sin/cos pair of randNorm.c:13-14 into ONE
        // sincos() call, and glibc's sincos differs from sin()/cos() in the last bit for ~0.1 % of the arguments:
        // calling it explicitly reproduces the reference binary's values whatever compiler builds this file.
        double sn, cs;
        sincos(u2, &sn, &cs);
        const double c = u1 * cs, s = u1 * sn;
        dst[2 * i] = (ftype)c;
        if (2 * i + 1 < room)
          dst[2 * i + 1] = (ftype)s;
        else
          g_next = s, g_have = true;  // only the very last pair of the call can get here
      }
    };
    if (m < 4096 || nthr == 1) {
      work(0, m);
    } else {
      std::vector<std::thread> th;
      const size_t per = (m + nthr - 1) / nthr;
      for (unsigned t = 0; t < nthr; t++) {
        const size_t lo = std::min(m, t * per), hi = std::min(m, lo + per);
        if (lo < hi) th.emplace_back(work, lo, hi);
      }
      for (auto &x : th) x.join();
    }
  }
}
