// ann_saveio.cpp -- on-disk format for save_t.  Host-only; no HIP calls.
//
// The reference keeps its index in memory only (/root/reference/ann.h:8-12 has no serialiser), which forces a
// rebuild per process and makes it impossible to ship an index between a CPU and a GPU box.  Format (little
// endian, one file):
//   char  magic[8] = "ANNSAVE2"
//   u32   ftype_bytes (4|8), u32 id_bytes (4 when n < 2^32-1, else 8)
//   u64   tries, n, k, d_short, d_long
//   u64   par_maxes[tries]
//   ftype row_means[d_long]; ftype bases[tries*d_short*d_long]
//   id    graph[n*k]
//   id    which_par[t][2^d_short * par_maxes[t]]   for t = 0..tries-1
//   u64   checksum of everything above (four interleaved 64-bit multiply lanes over 8-byte words: ~5 GB/s on one
//         core, where the byte-wise FNV of format 1 managed ~0.7 GB/s on the 2-3 GB of a cfg3 index)
// Format 1 ("ANNSAVE1", written by the first round of this backend) is the same layout with an FNV-1a-64 checksum;
// it is still READ (behind the same size, checksum and id-range checks); files are always written as format 2.
// Reading checks the header against the FILE SIZE before it allocates anything, verifies the checksum, then checks
// every id against its range (which_par <= n, graph below the sentinel bound) -- a file that passes cannot make a kernel gather out of
// bounds.  It returns malloc'd fields exactly as precomp() fills them (free_save() releases them).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "../../include/ann_hip.h"

namespace {
struct Hasher {  // streaming; the result depends only on the byte sequence, not on how it was cut into feed() calls
  uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
  unsigned char pend[32];
  size_t npend = 0;
  uint64_t total = 0;
  inline void block(const unsigned char *b) {
    uint64_t w[4];
    memcpy(w, b, 32);
    for (int i = 0; i < 4; i++) {
      h[i] = (h[i] ^ w[i]) * 0xFF51AFD7ED558CCDull;
      h[i] ^= h[i] >> 29;
    }
  }
  void feed(const void *p, size_t nbytes) {
    const unsigned char *b = (const unsigned char *)p;
    total += nbytes;
    if (npend) {
      const size_t take = nbytes < 32 - npend ? nbytes : 32 - npend;
      memcpy(pend + npend, b, take);
      npend += take, b += take, nbytes -= take;
      if (npend < 32) return;
      block(pend);
      npend = 0;
    }
    for (; nbytes >= 32; b += 32, nbytes -= 32) block(b);
    if (nbytes) memcpy(pend, b, nbytes), npend = nbytes;
  }
  uint64_t done() {
    if (npend) {
      memset(pend + npend, 0, 32 - npend);
      block(pend);
      npend = 0;
    }
    uint64_t r = total;
    for (int i = 0; i < 4; i++) r = (r ^ h[i]) * 0xC4CEB9FE1A85EC53ull, r ^= r >> 32;
    return r;
  }
};

const size_t ID_CHUNK = (size_t)1 << 20;

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
    uint32_t *tmp = (uint32_t *)malloc(ID_CHUNK * 4);
    if (!tmp) return (void)(ok = false);
    for (size_t i = 0; i < count && ok; i += ID_CHUNK) {
      size_t m = count - i < ID_CHUNK ? count - i : ID_CHUNK;
      for (size_t j = 0; j < m; j++) tmp[j] = (uint32_t)ids[i + j];
      put(tmp, m * 4);
    }
    free(tmp);
  }
};

struct Fnv1a {  // the checksum of format 1
  uint64_t h = 1469598103934665603ull;
  void feed(const void *p, size_t nbytes) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < nbytes; i++) h = (h ^ b[i]) * 1099511628211ull;
  }
};

struct Reader {
  FILE *f;
  Hasher hs;
  Fnv1a old;
  int version = 2;  // set from the magic before anything is fed
  bool ok = true;
  void get(void *p, size_t nbytes) {
    if (!ok) return;
    ok = fread(p, 1, nbytes, f) == nbytes;
    if (ok) feed(p, nbytes);
  }
  void feed(const void *p, size_t nbytes) {
    if (version == 1) old.feed(p, nbytes);
    else hs.feed(p, nbytes);
  }
  uint64_t sum() { return version == 1 ? old.h : hs.done(); }
  // ids of one section, widened to size_t; *max_id = the largest id seen
  size_t *get_ids(size_t count, unsigned id_bytes, size_t *max_id) {
    size_t *out = (size_t *)malloc(sizeof(size_t) * (count ? count : 1));
    if (!out) return ok = false, (size_t *)NULL;
    size_t mx = 0;
    if (id_bytes == 8) {
      get(out, count * 8);
      for (size_t i = 0; i < count && ok; i++) mx = out[i] > mx ? out[i] : mx;
    } else {
      uint32_t *tmp = (uint32_t *)malloc(ID_CHUNK * 4);
      if (!tmp) ok = false;
      for (size_t i = 0; i < count && ok; i += ID_CHUNK) {
        size_t m = count - i < ID_CHUNK ? count - i : ID_CHUNK;
        get(tmp, m * 4);
        for (size_t j = 0; j < m && ok; j++) {
          out[i + j] = tmp[j];
          mx = tmp[j] > mx ? tmp[j] : mx;
        }
      }
      free(tmp);
    }
    *max_id = mx;
    return out;
  }
};

int fail(const char *what, const char *path) {
  fprintf(stderr, "approxnn_hip: %s: %s\n", what, path);
  return -1;
}

// a * b, or SIZE_MAX on overflow
size_t mul_sat(size_t a, size_t b) {
  size_t r;
  return __builtin_mul_overflow(a, b, &r) ? SIZE_MAX : r;
}
size_t add_sat(size_t a, size_t b) {
  size_t r;
  return __builtin_add_overflow(a, b, &r) ? SIZE_MAX : r;
}

void release(save_t *save) {
  if (save->which_par)
    for (int t = 0; t < save->tries; t++) free(save->which_par[t]);
  free(save->which_par), free(save->par_maxes), free(save->graph), free(save->row_means), free(save->bases);
  memset(save, 0, sizeof *save);
}
}  // namespace

extern "C" int annhip_save_write(const save_t *save, const char *path) {
  FILE *f = fopen(path, "wb");
  if (!f) return fail("cannot create index file", path);
  setvbuf(f, NULL, _IOFBF, 1 << 22);
  Writer w{f};
  const uint32_t fb = (uint32_t)sizeof(ftype), ib = save->n < 0xFFFFFFFFull ? 4u : 8u;
  const uint64_t dims[5] = {(uint64_t)save->tries, save->n, save->k, save->d_short, save->d_long};
  w.put("ANNSAVE2", 8);
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
  const uint64_t sum = w.hs.done();
  if (w.ok) w.ok = fwrite(&sum, 1, 8, f) == 8;
  const bool closed = fclose(f) == 0;
  return (w.ok && closed) ? 0 : fail("short write to index file", path);
}

extern "C" int annhip_save_read(const char *path, save_t *save) {
  memset(save, 0, sizeof *save);
  FILE *f = fopen(path, "rb");
  if (!f) return fail("cannot open index file", path);
  struct stat sb;
  if (fstat(fileno(f), &sb) != 0 || sb.st_size < 0) return fclose(f), fail("cannot stat index file", path);
  const size_t file_size = (size_t)sb.st_size;
  setvbuf(f, NULL, _IOFBF, 1 << 22);
  Reader r{f};
  char magic[8];
  uint32_t fb = 0, ib = 0;
  uint64_t dims[5] = {0, 0, 0, 0, 0};
  if (fread(magic, 1, 8, f) != 8 || (memcmp(magic, "ANNSAVE2", 8) && memcmp(magic, "ANNSAVE1", 8)))
    return fclose(f), fail("not an index file of this library (magic ANNSAVE2, or ANNSAVE1 of its first format)", path);
  r.version = magic[7] == '1' ? 1 : 2;
  r.feed(magic, 8);
  r.get(&fb, 4);
  r.get(&ib, 4);
  r.get(dims, sizeof dims);
  if (!r.ok) return fclose(f), fail("truncated index file (header)", path);
  if (fb != sizeof(ftype)) return fclose(f), fail("index file was written by the other precision build (ftype.h)", path);
  const size_t T = dims[0], n = dims[1], k = dims[2], ds = dims[3], d = dims[4];
  if ((ib != 4 && ib != 8) || T == 0 || T > 4096 || ds > 40 || k == 0 || n <= k || d == 0 || d > ((size_t)1 << 24) ||
      (ib == 4 && n >= 0xFFFFFFFFull))
    return fclose(f), fail("implausible header in index file", path);
  // every section must fit the file: nothing below allocates more than the file can fill
  const size_t head = 8 + 4 + 4 + sizeof dims;
  if (add_sat(head, mul_sat(T, 8)) > file_size) return fclose(f), fail("truncated index file (header)", path);
  save->tries = (int)T;
  save->n = n, save->k = k, save->d_short = ds, save->d_long = d;
  save->par_maxes = (size_t *)malloc(sizeof(size_t) * T);
  save->which_par = (size_t **)calloc(T, sizeof(size_t *));
  if (!save->par_maxes || !save->which_par) return fclose(f), release(save), fail("out of memory reading index file", path);
  size_t want_size = add_sat(head, T * 8);
  const size_t n_bases = mul_sat(mul_sat(T, ds), d), n_graph = mul_sat(n, k);
  want_size = add_sat(want_size, mul_sat(add_sat(d, n_bases), sizeof(ftype)));
  want_size = add_sat(want_size, mul_sat(n_graph, ib));
  bool sane = true;
  for (size_t t = 0; t < T; t++) {
    uint64_t pm = 0;
    r.get(&pm, 8);
    save->par_maxes[t] = pm;
    if (pm == 0 || pm > n) sane = false;
    want_size = add_sat(want_size, mul_sat(mul_sat(pm, (size_t)1 << ds), ib));
  }
  want_size = add_sat(want_size, 8);
  if (!r.ok || !sane || want_size != file_size)
    return fclose(f), release(save), fail("index file size does not match its header (truncated or corrupted)", path);
  save->row_means = (ftype *)malloc(sizeof(ftype) * d);
  save->bases = (ftype *)malloc(sizeof(ftype) * (n_bases ? n_bases : 1));
  if (!save->row_means || !save->bases) return fclose(f), release(save), fail("out of memory reading index file", path);
  r.get(save->row_means, sizeof(ftype) * d);
  r.get(save->bases, sizeof(ftype) * n_bases);
  size_t max_graph = 0, max_tab = 0;
  save->graph = r.ok ? r.get_ids(n_graph, ib, &max_graph) : NULL;
  for (size_t t = 0; t < T && r.ok; t++) {
    size_t mx = 0;
    save->which_par[t] = r.get_ids(save->par_maxes[t] << ds, ib, &mx);
    max_tab = mx > max_tab ? mx : max_tab;
  }
  const uint64_t want = r.sum();
  uint64_t got = 0;
  const bool sum_ok = r.ok && fread(&got, 1, 8, f) == 8 && got == want;
  fclose(f);
  if (!sum_ok) return release(save), fail("truncated or corrupted index file (checksum)", path);
  // Table entries are point ids or the padding n; graph entries are point ids or, for points with fewer than k
  // valid candidates, the reference's sentinels n / (graph[0][z] | n) (compute.cl:259-262, Q7), i.e. below 2^ceil(lg n)+1.
  // (The kernels test id < n before every gather; this is the loader's own line of defence.)
  size_t lim = 2;
  while (lim <= n) lim <<= 1;
  if (max_tab > n || max_graph >= lim) return release(save), fail("index file holds an id outside its range", path);
  return 0;
}
