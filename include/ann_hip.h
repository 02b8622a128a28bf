/* ann_hip.h -- the resident / staged C-ABI of the HIP backend.
 *
 * query_gpu()/precomp_gpu() (algg.h) are the drop-in symbols; they are thin wrappers over the calls
 * below, which keep the index resident in HBM between calls and expose the stages of the query path
 * so that a multi-GPU host (one process per GPU, point rows sharded) can put its RCCL exchanges between
 * them.  Plain pointers and sizes only; "dev" pointers are HIP device pointers of the current device.
 *
 * The reference has no counterpart for residency: its GPU path re-wraps points, graph and every bucket
 * table as CL_MEM_USE_HOST_PTR buffers on every call (/root/reference/alg.c:444-445,503-508).
 */
#ifndef APPROXNN_HIP_ANN_HIP_H
#define APPROXNN_HIP_ANN_HIP_H
#include <stddef.h>
#include <stdint.h>
#include "ann.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct annhip_index annhip_index;

/* "f32" or "f64": the precision this library was built for (ftype.h). */
const char *annhip_precision(void);

/* ---- index lifecycle ------------------------------------------------------------------------- */
/* Upload an index.  `points` holds rows [row_lo,row_hi) only (row-major, save->d_long each): the rows
 * this device owns.  points_on_device != 0: `points` is a device pointer that the index borrows (the
 * caller keeps it alive); otherwise it is host memory and is copied.  Single GPU: row_lo=0,row_hi=n. */
annhip_index *annhip_index_create(const save_t *save, const ftype *points, int points_on_device,
                                  size_t row_lo, size_t row_hi);
void annhip_index_destroy(annhip_index *ix);
/* out[0..11] = n, k, d, d_short, tries, L1, P1, Lc1, L2, P2, Lc2, sum(par_maxes)  (SURVEY 8 notation) */
void annhip_index_info(const annhip_index *ix, size_t out[12]);
/* All launches of this index go to `hip_stream` (a hipStream_t; NULL = the default stream). */
void annhip_index_set_stream(annhip_index *ix, void *hip_stream);
/* annhip_sh_stage1 launches its gather in `pieces` kernels over consecutive query ranges (default 1).  Same results;
 * the launch boundaries are where workgroups of other streams -- RCCL's in particular -- find free compute units. */
void annhip_index_set_gather_pieces(annhip_index *ix, int pieces);
/* annhip_sh_stage1 as a PERSISTENT grid: at most waves_per_simd of the gather's waves on every SIMD, the workgroups walk
 * the queries themselves (0 = one workgroup per query, the default).  The gather kernel fits 4 waves per SIMD; with 3
 * one slot per SIMD is always free, so workgroups of other streams (the other batches' small kernels, RCCL) start at
 * once instead of waiting for the gather to drain.  Same results. */
void annhip_index_set_gather_slots(annhip_index *ix, int waves_per_simd);
/* Opt-in "fixed" query mode (default 0 = the reference's results, bit for bit).  With 1, annhip_query / annhip_query_on
 * on this index (whole index on one device) undo two accidents of the reference that cost most of its recall: a query
 * looks up the buckets of ITS OWN hash codes (the reference reads code[i*Q+x] from an array written as [x*T+i],
 * alg.c:489-499 vs compute.cl:223-231) and every slot of the candidate row competes (the reference orders only the
 * first 2^floor(log2 L) slots, alg.c:137-144).  Results: the k smallest distinct (squared distance, id) pairs among the
 * candidates, stage 2 likewise; (n, +inf) where fewer than k exist.  Not comparable with the reference -- checked
 * against a brute force over the same candidate sets (tests/test_gpu_fixed_mode.py).  precomp is unaffected. */
void annhip_index_set_fixed(annhip_index *ix, int fixed);
/* Point-shard an index that was built from all n rows: from now on this device owns rows [row_lo,row_hi) only
 * and reads them from shard_points_dev (device pointer to those rows, borrowed).  Tables and graph stay. */
void annhip_index_reshard(annhip_index *ix, const ftype *shard_points_dev, size_t row_lo, size_t row_hi);
/* Download the index into a save_t whose fields are malloc'd (free_save() releases it). */
void annhip_index_export(const annhip_index *ix, save_t *save);

/* ---- index files (the reference's save_t is memory-only) ------------------------------------ */
/* Write / read a save_t (format: approximatenn_amd/csrc/ann_saveio.cpp; checksummed; ids stored as 32 bit when
 * they fit).  Host-only.  Return 0 on success; on failure print to stderr and return -1 (read leaves *save zeroed).
 * A file is tied to the precision of the library that wrote it.  Read fills malloc'd fields (free_save()).        */
int annhip_save_write(const save_t *save, const char *path);
int annhip_save_read(const char *path, save_t *save);

/* ---- residency cache behind query_gpu()/precomp_gpu() ------------------------------------------------------------ */
/* The reference re-wraps points, graph and every table per call (alg.c:444-445,503-508); query_gpu() instead keeps up
 * to four indexes resident, keyed by (save, points) addresses plus a content fingerprint: par_maxes, row_means and
 * bases in full, 1 024 strided samples each of points, graph and every which_par[t].  In-place edits that miss the
 * samples are NOT seen: call annhip_cache_drop(save) (the bundled free_save() does) or annhip_cache_clear() after
 * editing, or set ANN_HIP_CACHE=strict (full content hash per call) / ANN_HIP_CACHE=off (upload per call).
 * annhip_cache_size() = resident indexes.  annhip_reload_env() re-reads the ANN_HIP_* switches (they are read once). */
void annhip_cache_clear(void);
/* Milliseconds one fingerprint of (save, points) takes on this host: strict != 0 = the full content hash that
 * ANN_HIP_CACHE=strict pays on every query() call (spread over the host thread pool, ANN_HIP_HOST_THREADS, default
 * min(cores, 16)), 0 = the sampled default.  Measurement only. */
double annhip_fingerprint_ms(const save_t *save, const ftype *points, int strict);
void annhip_cache_drop(const save_t *save);
size_t annhip_cache_size(void);
void annhip_reload_env(void);

/* ---- point rows sharded over several devices behind query_gpu()/precomp_gpu() (the reference is
 * single-device, gpu_comp.c:60-75) -------------------------------------------------------------------------------- */
/* ndev > 1: from now on precomp_gpu() builds across devices 0..ndev-1 of this process and query_gpu() keeps one resident
 * index per device -- rows [g*n/ndev,(g+1)*n/ndev) on device g, tables and graph replicated -- and answers by the owner
 * protocol (the annhip_sh_* sequence below; exchanges over RCCL: ncclCommInitAll, one group per exchange; librccl.so is
 * loaded on first use).  virtual_shards > 0 instead: that many shards on the CURRENT device with loop-back exchanges
 * (same host code and kernels; for single-GPU boxes and tests).  (0, 0) = one device, the default.  Results are
 * bit-identical in every mode.  The environment does the same without a code change: ANN_HIP_DEVICES=G,
 * ANN_HIP_VIRTUAL_SHARDS=G; a call to this function overrides it. */
void annhip_set_devices(int ndev, int virtual_shards);

/* ---- precomp on the device (alg.c:342-434) ----------------------------------------------------- */
/* Builds the index from ALL n rows on this device and keeps it resident.  Consumes libc random() in the
 * reference's order.  graph_dists_dev (device, ftype[n*k]) may be NULL. */
annhip_index *annhip_precomp_index(size_t n, size_t k, size_t d, const ftype *points,
                                   int points_on_device, int tries, size_t rots_before,
                                   size_t rot_len_before, size_t rots_after, size_t rot_len_after,
                                   ftype *graph_dists_dev);

/* precomp() in phases, for point-sharded hosts (one process per GPU; EVERY rank holds all n rows during the build and
 * draws the same rotations from the same random() seed).  Between the phases the host runs three collectives:
 *     h = annhip_precomp_begin(..., rank, world);  annhip_precomp_info(h, info);  annhip_precomp_init_merged(h, mi, md);
 *     for t in tries:  annhip_precomp_hash(h, t, lo, hi, codes_slice)     -> ALL-GATHER codes_all u32[n]
 *                      annhip_precomp_try(h, t, codes_all, mi, md)        (this rank's buckets: b mod world == rank)
 *     MIN ALL-REDUCE of mi (as int32) and md                               (every entry has exactly one writer)
 *     annhip_precomp_merge(h, mi, md);  annhip_precomp_graph(h, lo, hi, graph_slice, dists_slice)
 *                                                                          -> ALL-GATHER graph u32[n][k] (+ distances)
 *     ix = annhip_precomp_finish(h, graph_all);
 * mi u32[n][Wn], md ftype[n][Wn] with Wn = info[1]; info = {d_short, Wn, tries scored, tries, n, k}.
 * annhip_precomp_index() is exactly this sequence with world = 1.  Results are identical for any world size. */
typedef struct annhip_precomp annhip_precomp;
annhip_precomp *annhip_precomp_begin(size_t n, size_t k, size_t d, const ftype *points, int points_on_device, int tries,
                                     size_t rots_before, size_t rot_len_before, size_t rots_after, size_t rot_len_after,
                                     int rank, int world);
void annhip_precomp_info(const annhip_precomp *h, size_t out[6]);
void annhip_precomp_init_merged(annhip_precomp *h, uint32_t *merged_i_dev, ftype *merged_d_dev);
void annhip_precomp_hash(annhip_precomp *h, int t, size_t row_lo, size_t row_hi, uint32_t *codes_dev);
void annhip_precomp_try(annhip_precomp *h, int t, const uint32_t *codes_dev, uint32_t *merged_i_dev, ftype *merged_d_dev);
void annhip_precomp_merge(annhip_precomp *h, uint32_t *merged_i_dev, ftype *merged_d_dev);
void annhip_precomp_graph(annhip_precomp *h, size_t row_lo, size_t row_hi, uint32_t *graph_dev, ftype *graph_dists_dev);
annhip_index *annhip_precomp_finish(annhip_precomp *h, const uint32_t *graph_dev);

/* ---- whole query on one device (alg.c:458-519) ------------------------------------------------- */
/* y_dev: ftype[ycnt][d]; alias != 0: query x excludes point x (the y == points case, compute.cl:144-146).
 * mode 0: selection path with exact fallback; mode 1: exact path for every query.
 * ids_dev: size_t[ycnt][k], dists_dev: ftype[ycnt][k] (may be NULL).  Fully asynchronous on the index's
 * stream (the exact fallback is driven by a device-side count, no host read-back); returns -1 in that case, or
 * the number of exact-path queries when it is known on the host (mode 1, very large batches).  The running
 * count is in annhip_stats(). */
long annhip_query(annhip_index *ix, size_t ycnt, const ftype *y_dev, int alias, int mode,
                  size_t *ids_dev, ftype *dists_dev);

/* ---- overlapping independent batches ---------------------------------------------------------- */
/* annhip_query keeps its scratch in the index, so calls on one index are serial.  A caller that has several
 * independent batches gives each in-flight batch its own workspace and HIP stream: the small latency-bound stages of
 * batch i (hash, finalize, stage 2) then run underneath the HBM-bound stage-1 gather of batch i+1.
 * annhip_query_on = annhip_query with explicit workspace (NULL = the index's) and stream (a hipStream_t). */
typedef struct annhip_workspace annhip_workspace;
annhip_workspace *annhip_workspace_create(annhip_index *ix);
void annhip_workspace_destroy(annhip_workspace *ws);
long annhip_query_on(annhip_index *ix, annhip_workspace *ws, void *hip_stream, size_t ycnt, const ftype *y_dev,
                     int alias, int mode, size_t *ids_dev, ftype *dists_dev);

/* Query-sharded ("replica") hosts: every device holds ALL rows and the whole index, and answers a contiguous slice of every
 * batch.  Results depend on the whole batch (query x reads hash codes of other queries), so the codes of all
 * ycnt queries must be known everywhere: each device hashes its slice (annhip_sh_codes), ONE all-gather makes
 * codes_all_dev u32[ycnt*tries] ([q*tries+t]) -- the only exchange of the step -- and annhip_query_slice answers queries
 * [q_lo, q_lo+nq): y_slice_dev = their rows, ids_dev size_t[nq][k], dists_dev ftype[nq][k].  Same results as annhip_query
 * on the whole batch.  alias only with q_lo == 0.  SURVEY 8(e) names this split as the alternative to row sharding. */
long annhip_query_slice(annhip_index *ix, annhip_workspace *ws, void *hip_stream, size_t ycnt, size_t q_lo, size_t nq,
                        const ftype *y_slice_dev, const uint32_t *codes_all_dev, int alias, size_t *ids_dev, ftype *dists_dev);

/* Host-resident batches, pipelined: `lanes` batches may be in flight, each with pinned staging buffers, a workspace
 * and a HIP stream, so uploads, kernels and downloads of consecutive batches overlap (query_gpu serialises them and
 * synchronises on every call).  submit copies y_host (ftype[ycnt][d], ycnt <= max_ycnt) and returns a ticket >= 0, or
 * -1 when every lane is in flight (collect the oldest ticket first).  collect waits for that batch and copies
 * size_t[ycnt][k] ids and ftype[ycnt][k] squared distances (may be NULL) out; returns 0, or -1 for an unknown ticket.
 * alias != 0 as in annhip_query.  One host thread drives a stream object. */
typedef struct annhip_stream annhip_stream;
annhip_stream *annhip_stream_open(annhip_index *ix, size_t max_ycnt, int lanes);
long annhip_stream_submit(annhip_stream *st, size_t ycnt, const ftype *y_host, int alias);
int annhip_stream_collect(annhip_stream *st, long ticket, size_t *ids_host, ftype *dists_host);
void annhip_stream_close(annhip_stream *st);

/* ---- staged query, for point-sharded multi-GPU hosts (approximatenn_amd/sharded.py) ---------- */
/* One process per GPU; device g owns point rows [row_lo,row_hi) (annhip_index_create / annhip_index_reshard), tables
 * and graph are replicated.  Queries are dealt to OWNER devices in contiguous slices of qs = ceil(ycnt/G); the owner
 * merges its queries' candidates and runs their networks.  The host puts one collective between consecutive calls
 * (all-gather, all-to-all, all-gather, all-to-all, all-gather).  Every annhip_sh_* call is asynchronous on `hip_stream`
 * (a hipStream_t) and uses caller-provided buffers only, so several batches can be in flight on several streams.
 * "keys" are packed (squared distance bits, id) pairs of annhip_key_bytes() bytes each (8 for float, 16 for double). */
size_t annhip_key_bytes(void);
/* A HIP stream (hipStream_t) that may use every compute unit but `reserve` of them -- for the stage-1 gathers, so that
 * the small kernels and the RCCL collectives of the other in-flight batch always find free wave slots.  NULL if the
 * runtime refuses; release with annhip_stream_destroy. */
void *annhip_stream_create_reserving(int reserve);
void annhip_stream_destroy(void *hip_stream);
/* 0. hash codes of queries [q_lo,q_hi) of the batch: codes_slice_dev u32[(q-q_lo)*tries+t] (alg.c:462-492).  The
 *    all-gather of the slices is the [q*tries+t] array of the whole batch that stage 1 reads as [t*ycnt+q] (Q2).
 *    Only the codes stage 1 can read are computed (Q1). */
void annhip_sh_codes(annhip_index *ix, void *hip_stream, size_t ycnt, const ftype *y_dev, size_t q_lo, size_t q_hi,
                     uint32_t *codes_slice_dev);
/* 1. stage 1 of ALL ycnt queries over the rows this device owns: keys_dev key[ycnt][k+1] = its k+1 smallest distinct
 *    candidates per query, ascending, padded with (+inf, 0xFFFFFFFF); nvalid_dev u32[ycnt] = valid slots in the sorted
 *    prefix on ANY device (identical everywhere); nown_dev u32[ycnt] = rows gathered here.  Needs k <= P1. */
void annhip_sh_stage1(annhip_index *ix, void *hip_stream, size_t ycnt, const ftype *y_dev, int alias,
                      const uint32_t *codes_dev, void *keys_dev, uint32_t *nvalid_dev, uint32_t *nown_dev);
/* 2. owner of queries [q_lo, q_lo+qs): keys_in_dev key[ndev][qs][k+1] (what the all-to-all of step 1 delivers) ->
 *    the k globally best, top_id_dev u32[qs][k] / top_dist_dev ftype[qs][k], if the selection proof holds; otherwise
 *    top_id[.][0] = 0xFFFFFFFE ("flagged": exact ties between different ids, < k candidates).
 *    Slots of queries >= ycnt (ragged last slice) are filled with padding.  ndev <= 16. */
void annhip_sh_merge_finalize(annhip_index *ix, void *hip_stream, int ndev, size_t ycnt, size_t q_lo, size_t qs,
                              const void *keys_in_dev, const uint32_t *nvalid_dev, uint32_t *top_id_dev,
                              ftype *top_dist_dev);
/* 2b. exact stage 1 of the flagged queries, device-driven, around ONE fixed-size MIN all-reduce of rows_dist_dev
 *    (ftype[fcap][Lc1]):  begin = ascending list of the flagged queries, flist_dev u32[2+fcap] = {listed, total, list},
 *    derived from top_id_all_dev (identical on every device), + ids / owned distances of their first Lc1 slots;
 *    end = the reference's network on the reduced rows -> top_id_all_dev[x] (the flag disappears), top_dist_all_dev
 *    ftype[>=ycnt][k], and the owner's slices.  Flagged queries beyond fcap stay flagged (step 3 lists them). */
void annhip_sh_exact1_begin(annhip_index *ix, void *hip_stream, size_t ycnt, const ftype *y_dev, int alias,
                            const uint32_t *codes_dev, const uint32_t *top_id_all_dev, size_t fcap,
                            uint32_t *flist_dev, uint32_t *rows_id_dev, ftype *rows_dist_dev);
void annhip_sh_exact1_end(annhip_index *ix, void *hip_stream, size_t ycnt, size_t q_lo, size_t qs, size_t fcap,
                          const uint32_t *flist_dev, uint32_t *rows_id_dev, ftype *rows_dist_dev,
                          uint32_t *top_id_all_dev, ftype *top_dist_all_dev, uint32_t *top_id_dev,
                          ftype *top_dist_dev);
/* 3. every device, all queries: top_id_all_dev u32[>=ycnt][k] (the all-gather of step 2) -> dist_out_dev
 *    ftype[ycnt][Lc2-k] = distances of the stage-2 slots k..Lc2-1 this device owns, +inf elsewhere (supercharge,
 *    compute.cl:252-263 + compdists, alg.c:314-326).  Flagged queries are skipped and listed:
 *    flagged_dev u32[1+ycnt] = {count, query indices in no particular order}. */
void annhip_sh_stage2(annhip_index *ix, void *hip_stream, size_t ycnt, const ftype *y_dev, int alias,
                      const uint32_t *top_id_all_dev, ftype *dist_out_dev, uint32_t *flagged_dev);
/* 4. owner: dist_in_dev ftype[ndev][qs][Lc2-k] (the all-to-all of step 3), its own top_id/top_dist slices ->
 *    min over devices, the reference's network + rdups + network on the stage-2 row (alg.c:224-230,327), first k
 *    entries to out_id_dev u32[qs][k] / out_dist_dev ftype[qs][k] (flagged queries: 0xFFFFFFFE / +inf). */
void annhip_sh_final(annhip_index *ix, void *hip_stream, int ndev, size_t ycnt, size_t q_lo, size_t qs,
                     const uint32_t *top_id_dev, const ftype *top_dist_dev, const ftype *dist_in_dev,
                     uint32_t *out_id_dev, ftype *out_dist_dev);
/* Exact path, used for the flagged queries (on the index's stream, annhip_index_set_stream):
 * ids u32[nq][Lc1] and distances ftype[nq][Lc1] of the first Lc1 stage-1 slots of the queries listed in qidx_dev
 * (NULL = queries 0..nq-1); slots not owned here get +inf (MIN-reduce the distance rows across devices). */
void annhip_stage1_rows(annhip_index *ix, size_t ycnt, const ftype *y_dev, int alias,
                        const uint32_t *codes_dev, const uint32_t *qidx_dev, size_t nq,
                        uint32_t *ids_dev, ftype *dist_dev);
/* stage-2 rows of the listed queries: ids u32[nq][Lc2], distances ftype[nq][Lc2] from top_id_dev u32[ycnt][k] /
 * top_dist_dev ftype[ycnt][k] (rows indexed by query). */
void annhip_stage2_rows_list(annhip_index *ix, size_t ycnt, const ftype *y_dev, int alias, const uint32_t *qidx_dev,
                             size_t nq, const uint32_t *top_id_dev, const ftype *top_dist_dev, uint32_t *ids_dev,
                             ftype *dist_dev);
/* the reference's network+rdups+network on rows of reference length L (stage: 1 -> L1, 2 -> L2),
 * first k entries to out_id u32[.][k] / out_dist at row qidx_dev[i] (NULL = i). */
void annhip_exact_select(annhip_index *ix, int stage, size_t nq, uint32_t *ids_dev, ftype *dist_dev,
                         const uint32_t *qidx_dev, uint32_t *out_id_dev, ftype *out_dist_dev);

/* Test hook: sort_and_uniq (alg.c:224-230) on nq free-standing rows of reference length L (row stride
 * min(L, max(2^floor(log2 L), k) + 1)); with cand_d_dev / cand_i_dev [nq][k+1] (the k+1 smallest distinct keys of each
 * row's sorted prefix, ascending, padded with (+inf, 0xFFFFFFFF)) rows with a single run of tied distances are answered
 * by the tie path instead of the network; *resolved_dev (device counter, may be NULL) counts them.  derive != 0 with
 * cand_d_dev == NULL: the kernel derives those lists from the rows itself (what sharded hosts use).  Synchronous. */
void annhip_test_sort_rows(size_t L, size_t k, size_t nq, uint32_t *ids_dev, ftype *dist_dev, const ftype *cand_d_dev,
                           const uint32_t *cand_i_dev, uint32_t *out_id_dev, ftype *out_dist_dev,
                           unsigned long long *resolved_dev, int derive);

/* ---- content checksums (device memory in, 64-bit value out; synchronous) ------------------------------------------- */
/* annhip_checksum_dev: checksum of nbytes of device memory (4-byte aligned), independent of the launch geometry.
 * annhip_index_checksum: geometry, every bucket table, graph, means and projection rows of a resident index -- everything
 * a query reads except the point rows.  A multi-GPU host all-reduces both with MIN and MAX at start-up: ranks that do not
 * hold the same index and the same batch would return different answers (or hang in mismatched collectives). */
unsigned long long annhip_checksum_dev(const void *dev_ptr, size_t nbytes, void *hip_stream);
unsigned long long annhip_index_checksum(annhip_index *ix);

/* ---- recall scoring (counterpart of /root/reference/test_correctness.c:169-262) --------------- */
/* ranks_dev[q][j] (u64) = number of the n points strictly closer to query q than its j-th guessed neighbour, by one
 * tiled brute-force pass with the exact distance arithmetic of the query path.  points_dev holds ALL n rows;
 * guess_dev is size_t[ycnt][k] as query()/precomp() return it; self != 0 skips point q for query q (scoring the
 * graph precomp returns).  Synchronous. */
void annhip_recall_ranks(size_t n, size_t d, size_t k, const ftype *points_dev, size_t ycnt, const ftype *y_dev,
                         const size_t *guess_dev, int self, unsigned long long *ranks_dev);
/* The same with HOST pointers in and out (uploads, scores, downloads): for plain-C drivers (tests/harness/test_correctness.c). */
void annhip_recall_ranks_host(size_t n, size_t d, size_t k, const ftype *points, size_t ycnt, const ftype *y,
                              const size_t *guess, int self, unsigned long long *ranks_host);

/* ---- synthetic data of the reference's drivers ------------------------------------------------------ */
/* out[0..count) = iid N(0,1) by Box-Muller on the CALLER's libc random() stream, value for value what genRand /
 * rand_norm produce (/root/reference/time_results.c:10-13, randNorm.c:9-21), incl. the pending second value of a pair
 * that the reference keeps between calls (annhip_synth_reset() forgets it, as a fresh process would).  Host memory;
 * the draws are sequential, the libm part runs on several host threads. */
void annhip_synth_randnorm(size_t count, ftype *out);
void annhip_synth_reset(void);

/* ---- measurement ---------------------------------------------------------------------------------- */
/* profile = 1: bracket every stage-1 launch with HIP events on its stream, drop events at annhip_query's stage
 * boundaries and count the gathered rows (one more small kernel per step).  profile = 2: the stage-1 event pair ONLY
 * (two events per step; every event record costs ~5 us of stream time -- what bench.py keeps inside its timed region).
 * 0: off. */
void annhip_profile(annhip_index *ix, int profile);
/* out[0]=stage-1 launches, out[1]=their total ms (events; only while profiling), out[2]=rows gathered in stage 1
 * (owned valid slots; only with profile = 1), out[3]=rows gathered in stage-2/exact rows kernels (only with profile =
 * 1), out[4]=queries flagged for the exact path, out[5]=queries seen, out[6]=of the flagged ones: answered by the tie
 * path (ann_tie.h) instead of the literal network; counters accumulate since the last reset (reset != 0 clears them
 * after reading). */
void annhip_stats(annhip_index *ix, double out[8], int reset);
/* While profiling, annhip_query also drops HIP events at its stage boundaries; out[0..5] = accumulated ms of
 * hash codes, stage-1 kernel, finalize + exact fallback, stage-2 rows, stage-2 network, id widening. */
void annhip_stage_ms(annhip_index *ix, double out[6]);
/* The same through the host-pointer ABI: annhip_host_profile(1) makes the indexes resident behind query_gpu() /
 * precomp_gpu() record their stage-1 launches; annhip_host_stats() = annhip_stats() of the index resident for `save`
 * plus out[6] = P1, out[7] = L1 (returns 0, or -1 when no index is resident for it). */
void annhip_host_profile(int on);
int annhip_host_stats(const save_t *save, double out[8], int reset);
/* A sharded resident index (annhip_set_devices): annhip_host_shards() = the devices / virtual shards the index resident
 * for `save` is spread over (1 = one device, 0 = none resident); annhip_host_stats_shard() = annhip_host_stats() of one
 * shard (annhip_host_stats() itself then reports the slowest shard's launches / milliseconds and the rows summed). */
int annhip_host_shards(const save_t *save);
int annhip_host_stats_shard(const save_t *save, int shard, double out[8], int reset);

#ifdef __cplusplus
}
#endif
#endif
