/* Shared by the two C harnesses (TEST TOOLS).  They are the counterparts of the reference's drivers
 * /root/reference/compare_results.c and /root/reference/time_results.c, linked against
 *   - libann_dispatch_<prec>.so + libapproxnn_hip_<prec>.so : the product, through precomp()/query() of ann.h
 *   - oracle/liboracle_<prec>.so                           : the CPU column / the checker (tests only)       */
#ifndef HARNESS_COMMON_H
#define HARNESS_COMMON_H
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "ann.h"
#include "gpu_comp.h"

/* oracle entry points (oracle/ann_oracle.c); oracle_save_t is layout-compatible with save_t */
extern size_t *oracle_precomp(size_t n, size_t k, size_t d, const ftype *points, int tries, size_t rb, size_t rlb,
                              size_t ra, size_t rla, save_t *save, ftype **dists_o);
extern size_t *oracle_query(const save_t *save, const ftype *points, size_t ycnt, const ftype *y, ftype **dists_o);
extern void oracle_gen_rand(size_t count, ftype *out);

typedef struct {
  size_t n, k, d, tries, reps, ycnt, rb, rlenb, ra, rlena, cpu_queries;
  unsigned seed;
  int verbose, use_y, use_cpu, save_test, lanes, devices, vshards, fp_cost;
} opts_t;

static void usage(const char *prog) {
  fprintf(stderr,
          "%s options (same letters as the reference's drivers, plus -S):\n"
          "\t-n points (1000)  -k neighbours (10)  -d dimension (80)  -t tries (10)  -o repetitions\n"
          "\t-b/-s pre-Walsh rotation count/size (6/1)  -a/-r post-Walsh rotation count/size (1/1)\n"
          "\t-y query count  -z (compare: -y 50; time: save the index)  -c CPU column only  -v verbose\n"
          "\t-S seed for srandom() (12345; the reference seeds with time(NULL))\n"
          "\t-P lanes (time_results only): also time the pipelined host API (annhip_stream_*) with that many lanes\n"
          "\t-G devices: shard the point rows over that many GPUs of this process (annhip_set_devices; RCCL exchanges)\n"
          "\t-V shards: the same with that many VIRTUAL shards on one GPU (loop-back exchanges)\n"
          "\t-F (time_results only): print what one residency-cache fingerprint costs (sampled default and strict)\n"
          "\t-C queries (time_results only): size of the batch the CPU column is timed on (default: -y for the oracle,\n"
          "\t   128 for the reference's query_cpu)\n", prog);
}

static opts_t parse_opts(int argc, char **argv, const char *letters, size_t default_reps) {
  opts_t o = {1000, 10, 80, 10, default_reps, 0, 6, 1, 1, 1, 0, 12345u, 0, 0, 0, 0, 0, 0, 0, 0};
  int c;
  opterr = 0;
  while ((c = getopt(argc, argv, letters)) != -1) switch (c) {
      case 'n': o.n = strtoul(optarg, NULL, 0); break;
      case 'k': o.k = strtoul(optarg, NULL, 0); break;
      case 'd': o.d = strtoul(optarg, NULL, 0); break;
      case 't': o.tries = strtoul(optarg, NULL, 0); break;
      case 'o': o.reps = strtoul(optarg, NULL, 0); break;
      case 'y': o.ycnt = strtoul(optarg, NULL, 0); o.use_y = 1; break;
      case 'z': o.use_y = 1; o.save_test = 1; break;
      case 'b': o.rb = strtoul(optarg, NULL, 0); break;
      case 's': o.rlenb = strtoul(optarg, NULL, 0); break;
      case 'a': o.ra = strtoul(optarg, NULL, 0); break;
      case 'r': o.rlena = strtoul(optarg, NULL, 0); break;
      case 'S': o.seed = (unsigned)strtoul(optarg, NULL, 0); break;
      case 'v': o.verbose = 1; break;
      case 'c': o.use_cpu = 1; break;
      case 'P': o.lanes = (int)strtol(optarg, NULL, 0); break;
      case 'C': o.cpu_queries = strtoul(optarg, NULL, 0); break;
      case 'G': o.devices = (int)strtol(optarg, NULL, 0); break;
      case 'V': o.vshards = (int)strtol(optarg, NULL, 0); break;
      case 'F': o.fp_cost = 1; break;
      default: usage(argv[0]); exit(c == 'h' ? 0 : 2);
    }
  return o;
}

static __attribute__((unused)) double now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + t.tv_nsec * 1e-9;
}
#endif
