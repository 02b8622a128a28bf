// ann_saveio.cpp -- on-disk format for save_t.  Host-only; no HIP calls.
//
// The reference keeps its index in memory only (/root/reference/ann.h:8-12 has no serialiser), which forces a
// rebuild per process and makes it impossible to ship an index between a CPU and a GPU box.  Format (little
// endian, one file):
//   char  magic[8] = "ANNSAVE1"
//   u32   ftype_bytes (4|8), u32 id_bytes (4 when n < 2^32-1, else 8)
//   u64   tries, n, k, d_short, d_long
//   u64   par_maxes[tries]
//   ftype row_means[d_long]; ftype bases[tries*d_short*d_long]
//   id    graph[n*k]
//   id    which_par[t][2^d_short * par_maxes[t]]   for t = 0..tries-1
//   u64   fnv1a64 of everything above
// Reading returns malloc'd fields exactly as precomp() fills them (free_save() releases them).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ann_hip.h"

namespace {
struct Hasher {
  uint64_t h = 1469598103934665603ull;
  void feed(const void *p, size_t nbytes) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < nbytes; i++) h = (h ^ b[i]) * 1099511628211ull;
  }
};

struct Writer {
  FILE *f;
  Hasher hs;
  bool ok = true;
  void put(const void *p, size_t nbytes) {
    if (!ok) return;
    hs.feed(p, nbytes);
    ok = fwrite(p, 1, nbytes, f) == nbytes;
  }
  void put_ids(const size_t *ids, size_t count, unsigned id_bytes) {
    if (id_bytes == 8) return put(ids, count * 8);
    const size_t chunk = 1 << 16;
    uint32_t *tmp = (uint32_t *)malloc(chunk * 4);
    for (size_t i = 0; i < count && ok; i += chunk) {
      size_t m = count - i < chunk ? count - i : chunk;
      for (size_t j = 0; j < m; j++) tmp[j] = (uint32_t)ids[i + j];
      put(tmp, m * 4);
    }
    free(tmp);
  }
};

struct Reader {
  FILE *f;
  Hasher hs;
  bool ok = true;
  void get(void *p, size_t nbytes) {
    if (!ok) return;
    ok = fread(p, 1, nbytes, f) == nbytes;
    if (ok) hs.feed(p, nbytes);
  }
  size_t *get_ids(size_t count, unsigned id_bytes) {
    size_t *out = (size_t *)malloc(sizeof(size_t) * (count ? count : 1));
    if (!out) return ok = false, (size_t *)NULL;
    if (id_bytes == 8) {
      get(out, count * 8);
      return out;
    }
    const size_t chunk = 1 << 16;
    uint32_t *tmp = (uint32_t *)malloc(chunk * 4);
    for (size_t i = 0; i < count && ok; i += chunk) {
      size_t m = count - i < chunk ? count - i : chunk;
      get(tmp, m * 4);
      for (size_t j = 0; j < m && ok; j++) out[i + j] = tmp[j];
    }
    free(tmp);
    return out;
  }
};

int fail(const char *what, const char *path) {
  fprintf(stderr, "approxnn_hip: %s: %s\n", what, path);
  return -1;
}
}  // namespace

extern "C" int annhip_save_write(const save_t *save, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return fail("cannot create index file", path);
  Writer w{f};
  const uint32_t fb = (uint32_t)sizeof(ftype), ib = save->n < 0xFFFFFFFFull ? 4u : 8u;
  const uint64_t dims[5] = {(uint64_t)save->tries, save->n, save->k, save->d_short, save->d_long};
  w.put("ANNSAVE1", 8);
  w.put(&fb, 4);
  w.put(&ib, 4);
  w.put(dims, sizeof dims);
  for (int t = 0; t < save->tries; t++) {
    uint64_t pm = save->par_maxes[t];
    w.put(&pm, 8);
  }
  w.put(save->row_means, sizeof(ftype) * save->d_long);
  w.put(save->bases, sizeof(ftype) * (size_t)save->tries * save->d_short * save->d_long);
  w.put_ids(save->graph, save->n * save->k, ib);
  for (int t = 0; t < save->tries; t++) w.put_ids(save->which_par[t], save->par_maxes[t] << save->d_short, ib);
  const uint64_t sum = w.hs.h;
  if (w.ok) w.ok = fwrite(&sum, 1, 8, f) == 8;
  const bool closed = fclose(f) == 0;
  return (w.ok && closed) ? 0 : fail("short write to index file", path);
}

extern "C" int annhip_save_read(const char *path, save_t *save) {
  FILE *f = fopen(path, "rb");
  if (!f) return fail("cannot open index file", path);
  Reader r{f};
  char magic[8];
  uint32_t fb = 0, ib = 0;
  uint64_t dims[5] = {0, 0, 0, 0, 0};
  r.get(magic, 8);
  r.get(&fb, 4);
  r.get(&ib, 4);
  r.get(dims, sizeof dims);
  if (!r.ok || memcmp(magic, "ANNSAVE1", 8)) return fclose(f), fail("not an ANNSAVE1 index file", path);
  if (fb != sizeof(ftype)) return fclose(f), fail("index file was written by the other precision build (ftype.h)", path);
  if ((ib != 4 && ib != 8) || dims[0] == 0 || dims[0] > 4096 || dims[3] > 40 || dims[2] == 0 || dims[1] <= dims[2])
    return fclose(f), fail("implausible header in index file", path);
  memset(save, 0, sizeof *save);
  save->tries = (int)dims[0];
  save->n = dims[1], save->k = dims[2], save->d_short = dims[3], save->d_long = dims[4];
  save->par_maxes = (size_t *)malloc(sizeof(size_t) * save->tries);
  save->which_par = (size_t **)calloc(save->tries, sizeof(size_t *));
  for (int t = 0; t < save->tries; t++) {
    uint64_t pm = 0;
    r.get(&pm, 8);
    save->par_maxes[t] = pm;
  }
  save->row_means = (ftype *)malloc(sizeof(ftype) * (save->d_long ? save->d_long : 1));
  r.get(save->row_means, sizeof(ftype) * save->d_long);
  const size_t nb = (size_t)save->tries * save->d_short * save->d_long;
  save->bases = (ftype *)malloc(sizeof(ftype) * (nb ? nb : 1));
  r.get(save->bases, sizeof(ftype) * nb);
  save->graph = r.ok ? r.get_ids(save->n * save->k, ib) : NULL;
  for (int t = 0; t < save->tries && r.ok; t++) save->which_par[t] = r.get_ids(save->par_maxes[t] << save->d_short, ib);
  const uint64_t want = r.hs.h;
  uint64_t got = 0;
  bool sum_ok = r.ok && fread(&got, 1, 8, f) == 8 && got == want;
  fclose(f);
  if (!sum_ok) {  // release whatever was allocated; the struct is left zeroed
    for (int t = 0; t < save->tries; t++) free(save->which_par[t]);
    free(save->which_par), free(save->par_maxes), free(save->graph), free(save->row_means), free(save->bases);
    memset(save, 0, sizeof *save);
    return fail("truncated or corrupted index file (checksum)", path);
  }
  return 0;
}
