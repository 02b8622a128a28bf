// ann_multi_host.h -- the point-sharded step driven from ONE process, behind the drop-in surface.
//
// BASELINE.json's north_star shards the point set across the GPUs of a node and asks for "C host code calling
// hand-written HIP kernels through a thin C-ABI".  The reference reaches its GPU backend only through
// query()/precomp() (/root/reference/algg.h:5-11, time_results.c:87-112), one context on one device
// (/root/reference/gpu_comp.c:60-75).  With ANN_HIP_DEVICES=G (or annhip_set_devices) query_gpu()/precomp_gpu() shard
// the point rows over G devices inside the calling process: one resident index per device (rows [g*n/G,(g+1)*n/G),
// tables and graph replicated), one HIP stream per device, and the owner protocol of DESIGN.md section 4 -- the same
// annhip_sh_* sequence approximatenn_amd/sharded.py issues under torch.distributed -- with the exchanges as RCCL calls
// (ncclCommInitAll + ncclGroupStart/End, <rccl/rccl.h>, the library loaded on demand).  ANN_HIP_VIRTUAL_SHARDS=G puts
// G shards on ONE device with loop-back exchanges (device-to-device copies on one stream): same host code, same
// kernels, testable on a single-GPU box.  Results are bit-identical to the single-device path whatever G.
//
// This file is included by ann_host.hip (it drives the static launchers' public wrappers and shares the residency cache).
#ifndef APPROXNN_HIP_ANN_MULTI_HOST_H
#define APPROXNN_HIP_ANN_MULTI_HOST_H
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <thread>
#include <type_traits>

// ----------------------------------------------------------------------------- RCCL, loaded on demand
struct RcclApi {
  void *lib = NULL;
  decltype(&ncclCommInitAll) CommInitAll = NULL;
  decltype(&ncclCommDestroy) CommDestroy = NULL;
  decltype(&ncclGroupStart) GroupStart = NULL;
  decltype(&ncclGroupEnd) GroupEnd = NULL;
  decltype(&ncclAllGather) AllGather = NULL;
  decltype(&ncclAllToAll) AllToAll = NULL;
  decltype(&ncclAllReduce) AllReduce = NULL;
  decltype(&ncclGetErrorString) GetErrorString = NULL;
  decltype(&ncclGetVersion) GetVersion = NULL;
};
static RcclApi g_rccl;
static std::vector<ncclComm_t> g_rccl_comms;  // one communicator per device 0..G-1, created once per process
#define RCCLCHECK(call)                                                                                      \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess) {                                                                                 \
      fprintf(stderr, "Error in RCCL: %s (%s:%d: %s)\n", g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", \
              __FILE__, __LINE__, #call);                                                                    \
      exit(1);                                                                                               \
    }                                                                                                        \
  } while (0)

static void rccl_load() {
  if (g_rccl.lib) return;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names)
    if ((g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!g_rccl.lib) die("ANN_HIP_DEVICES: cannot load librccl.so (the multi-device host exchanges candidates over RCCL)");
#define SYM(field, name)                                                    \
  do {                                                                      \
    g_rccl.field = (decltype(g_rccl.field))dlsym(g_rccl.lib, name);         \
    if (!g_rccl.field) die("librccl.so lacks " name);                       \
  } while (0)
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(AllGather, "ncclAllGather");
  SYM(AllToAll, "ncclAllToAll");
  SYM(AllReduce, "ncclAllReduce");
  SYM(GetErrorString, "ncclGetErrorString");
  SYM(GetVersion, "ncclGetVersion");
#undef SYM
}

static void rccl_comms(int G) {
  rccl_load();
  if ((int)g_rccl_comms.size() == G) return;
  for (ncclComm_t c : g_rccl_comms) (void)g_rccl.CommDestroy(c);
  g_rccl_comms.assign(G, (ncclComm_t)NULL);
  std::vector<int> devs(G);
  for (int g = 0; g < G; g++) devs[g] = g;
  RCCLCHECK(g_rccl.CommInitAll(g_rccl_comms.data(), G, devs.data()));
}

// ----------------------------------------------------------------------------- configuration
struct MultiCfg {
  int G = 0;          // 0 = single device (the default)
  bool virt = false;  // G shards on the current device, loop-back exchanges
  bool force_rccl = false;
};
static MultiCfg g_multi_cfg;
static bool g_multi_cfg_set = false;  // annhip_set_devices() overrides the environment

static MultiCfg multi_cfg() {
  if (g_multi_cfg_set) return g_multi_cfg;
  MultiCfg c;
  const int v = env_int("ANN_HIP_VIRTUAL_SHARDS", 0), r = env_int("ANN_HIP_DEVICES", 0);
  if (v > 0) c.G = v, c.virt = true;
  else if (r > 1 || (r == 1 && getenv("ANN_HIP_FORCE_RCCL"))) c.G = r;  // one "device set" of 1 rehearses the RCCL calls
  c.force_rccl = getenv("ANN_HIP_FORCE_RCCL") != NULL;
  if (c.G > 16) die("at most 16 shards (annhip_sh_merge_finalize)");
  return c;
}

extern "C" void annhip_set_devices(int ndev, int virtual_shards) {
  MultiCfg c;
  if (virtual_shards > 0) c.G = virtual_shards, c.virt = true;
  else if (ndev > 1) c.G = ndev;
  if (c.G > 16) die("at most 16 shards (annhip_sh_merge_finalize)");
  g_multi_cfg = c, g_multi_cfg_set = true;
}

// ----------------------------------------------------------------------------- small kernels of the loop-back exchange
struct PtrList {
  void *p[16];
};
template <typename T>
__global__ void multi_min_kernel(size_t count, int G, PtrList bufs) {  // bufs.p[0][i] = min over g of bufs.p[g][i]
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    T m = reinterpret_cast<const T *>(bufs.p[0])[i];
    for (int g = 1; g < G; g++) {
      const T v = reinterpret_cast<const T *>(bufs.p[g])[i];
      m = v < m ? v : m;
    }
    reinterpret_cast<T *>(bufs.p[0])[i] = m;
  }
}

// ----------------------------------------------------------------------------- the sharded index
struct MultiShard {
  int dev = 0;
  annhip_index *ix = NULL;
  hipStream_t s = NULL;
  FT *rows = NULL;  // this shard's point rows when this host owns them (after a sharded precomp on real devices)
  size_t lo = 0, hi = 0;
  DevBuf y, codes_slice, codes_all, keys, keys_in, nvalid, nown, top_i, top_d, top_all, top_d_all, s2, s2_in, flagged, flist,
      xrows_i, xrows_d, out_i, out_d, fl, r1i, r1d, r2i, r2d, full_i, full_d;
};

struct annhip_multi {
  int G = 0;
  bool virt = false;
  size_t n = 0, k = 0, d = 0;
  std::vector<MultiShard> sh;
  FT *shared_points = NULL;  // virtual shards: the one full copy of the points every shard's index borrows from
  PinBuf y_pin, out_pin;
  double exact_queries = 0, queries = 0, calls = 0;
};

struct DevScope {  // current device of the calling thread := the shard's, for the lifetime of the scope
  int prev = 0;
  explicit DevScope(int dev) {
    HIPCHECK(hipGetDevice(&prev));
    if (prev != dev) HIPCHECK(hipSetDevice(dev));
  }
  ~DevScope() { (void)hipSetDevice(prev); }
};

// fn(g) for every shard: on real devices in one host thread per device (the build phases and uploads are synchronous
// calls: run back to back they would use one GPU at a time), on virtual shards serially
template <typename F>
static void multi_each(annhip_multi *M, F fn, bool parallel) {
  if (M->virt || !parallel || M->G == 1) {
    for (int g = 0; g < M->G; g++) {
      DevScope ds(M->sh[g].dev);
      fn(g);
    }
    return;
  }
  std::vector<std::thread> th;
  for (int g = 0; g < M->G; g++)
    th.emplace_back([&, g] {
      HIPCHECK(hipSetDevice(M->sh[g].dev));
      fn(g);
    });
  for (auto &t : th) t.join();
}

// ---- exchanges.  send/recv are per-shard device pointers.  Real devices: one RCCL group over the G communicators, each
// call on its device's stream.  Virtual shards: device-to-device copies on the one stream all shards share.
static void multi_all_gather(annhip_multi *M, void *const *send, void *const *recv, size_t bytes) {
  if (!bytes) return;
  if (M->virt) {
    for (int g = 0; g < M->G; g++)
      for (int p = 0; p < M->G; p++)
        HIPCHECK(hipMemcpyAsync((char *)recv[g] + (size_t)p * bytes, send[p], bytes, hipMemcpyDeviceToDevice, M->sh[0].s));
    return;
  }
  RCCLCHECK(g_rccl.GroupStart());
  for (int g = 0; g < M->G; g++) RCCLCHECK(g_rccl.AllGather(send[g], recv[g], bytes, ncclUint8, g_rccl_comms[g], M->sh[g].s));
  RCCLCHECK(g_rccl.GroupEnd());
}

// send[g] = [G][bytes]: piece p goes to shard p; recv[g] = [G][bytes]: piece p came from shard p
static void multi_all_to_all(annhip_multi *M, void *const *send, void *const *recv, size_t bytes) {
  if (!bytes) return;
  if (M->virt) {
    for (int g = 0; g < M->G; g++)
      for (int p = 0; p < M->G; p++)
        HIPCHECK(hipMemcpyAsync((char *)recv[g] + (size_t)p * bytes, (const char *)send[p] + (size_t)g * bytes, bytes,
                                hipMemcpyDeviceToDevice, M->sh[0].s));
    return;
  }
  RCCLCHECK(g_rccl.GroupStart());
  for (int g = 0; g < M->G; g++) RCCLCHECK(g_rccl.AllToAll(send[g], recv[g], bytes, ncclUint8, g_rccl_comms[g], M->sh[g].s));
  RCCLCHECK(g_rccl.GroupEnd());
}

// buf[g][i] := min over the shards, in place, T = FT or int32 (stream: the shards' own; precomp passes the null stream)
template <typename T>
static void multi_all_min(annhip_multi *M, void *const *buf, size_t count, bool null_stream = false) {
  if (!count) return;
  if (M->virt) {
    hipStream_t s = null_stream ? (hipStream_t)0 : M->sh[0].s;
    PtrList pl;
    for (int g = 0; g < M->G; g++) pl.p[g] = buf[g];
    multi_min_kernel<T><<<grid_for(count, 256, 4096), 256, 0, s>>>(count, M->G, pl);
    HIPCHECK(hipGetLastError());
    for (int g = 1; g < M->G; g++) HIPCHECK(hipMemcpyAsync(buf[g], buf[0], sizeof(T) * count, hipMemcpyDeviceToDevice, s));
    return;
  }
  const ncclDataType_t ty = std::is_same<T, float>::value ? ncclFloat32 : std::is_same<T, double>::value ? ncclFloat64 : ncclInt32;
  RCCLCHECK(g_rccl.GroupStart());
  for (int g = 0; g < M->G; g++)
    RCCLCHECK(g_rccl.AllReduce(buf[g], buf[g], count, ty, ncclMin, g_rccl_comms[g], null_stream ? (hipStream_t)0 : M->sh[g].s));
  RCCLCHECK(g_rccl.GroupEnd());
}

static annhip_multi *multi_new(const MultiCfg &cfg, size_t n, size_t k, size_t d) {
  gpu_init();
  (void)env();  // loaded before any worker thread reads it
  annhip_multi *M = new annhip_multi();
  M->G = cfg.G, M->virt = cfg.virt, M->n = n, M->k = k, M->d = d;
  int cur = 0, ndev = 0;
  HIPCHECK(hipGetDevice(&cur));
  HIPCHECK(hipGetDeviceCount(&ndev));
  if (!cfg.virt && cfg.G > ndev) {
    fprintf(stderr, "approxnn_hip: ANN_HIP_DEVICES=%d but this process sees %d GPU(s)\n", cfg.G, ndev);
    exit(1);
  }
  if (!cfg.virt) rccl_comms(cfg.G);
  M->sh.resize(cfg.G);
  for (int g = 0; g < cfg.G; g++) {
    MultiShard &S = M->sh[g];
    S.dev = cfg.virt ? cur : g;
    S.lo = n * (size_t)g / cfg.G, S.hi = n * (size_t)(g + 1) / cfg.G;
    DevScope ds(S.dev);
    if (cfg.virt && g > 0)
      S.s = M->sh[0].s;  // one stream: the loop-back copies need no events
    else
      HIPCHECK(hipStreamCreateWithFlags(&S.s, hipStreamNonBlocking));
  }
  return M;
}

static void multi_destroy(annhip_multi *M) {
  if (!M) return;
  for (int g = 0; g < M->G; g++) {
    MultiShard &S = M->sh[g];
    DevScope ds(S.dev);
    HIPCHECK(hipDeviceSynchronize());
    annhip_index_destroy(S.ix);
    DevBuf *bufs[] = {&S.y, &S.codes_slice, &S.codes_all, &S.keys, &S.keys_in, &S.nvalid, &S.nown, &S.top_i, &S.top_d, &S.top_all,
                      &S.top_d_all, &S.s2, &S.s2_in, &S.flagged, &S.flist, &S.xrows_i, &S.xrows_d, &S.out_i, &S.out_d, &S.fl,
                      &S.r1i, &S.r1d, &S.r2i, &S.r2d, &S.full_i, &S.full_d};
    for (DevBuf *b : bufs) b->release();
    if (S.rows) HIPCHECK(hipFree(S.rows));
    if (!(M->virt && g > 0)) (void)hipStreamDestroy(S.s);
  }
  if (M->shared_points) {
    DevScope ds(M->sh[0].dev);
    HIPCHECK(hipFree(M->shared_points));
  }
  M->y_pin.release(), M->out_pin.release();
  delete M;
}

// A sharded resident index from a save_t in host memory (built anywhere: the reference's CPU path, a single-GPU run, an
// index file): every device receives its row slice and a copy of the tables and the graph.
static annhip_multi *multi_create(const MultiCfg &cfg, const save_t *save, const ftype *points) {
  annhip_multi *M = multi_new(cfg, save->n, save->k, save->d_long);
  multi_each(M, [&](int g) {
    MultiShard &S = M->sh[g];
    S.ix = annhip_index_create(save, points + S.lo * save->d_long, 0, S.lo, S.hi);
    S.ix->stream = S.s;
    S.ix->profile = g_host_profile ? 1 : 0;
  }, true);
  return M;
}

// ----------------------------------------------------------------------------- query: one batch, synchronous
#define MULTI_FCAP 32  // flagged queries the device-driven exact path takes per step (sharded.py: fcap)

// overlap: called once everything of the step is enqueued and before the host waits for it (query_gpu verifies the
// residency fingerprint there)
static void multi_query(annhip_multi *M, size_t Q, const ftype *y_host, int alias, size_t *ids_out, ftype *dists_out,
                        const std::function<void()> *overlap = NULL) {
  const int G = M->G;
  const size_t k = M->k, d = M->d, qs = (Q + G - 1) / G, Qp = qs * G;
  annhip_index *ix0 = M->sh[0].ix;
  const size_t T = ix0->T, K1 = k + 1, Lc1 = ix0->Lc1, Lc2 = ix0->Lc2, W2 = Lc2 - k, fcap = MULTI_FCAP;
  const bool exact_all = env().exact || (u32)k > ix0->P1;
  std::vector<void *> a(G), b(G);
  // ---- the batch: pinned bounce buffer, then to every device (virtual shards share one copy)
  char *y_pin = (char *)M->y_pin.need(sizeof(FT) * Q * d);
  HostPool::get().copy(y_pin, y_host, sizeof(FT) * Q * d);
  for (int g = 0; g < G; g++) {
    MultiShard &S = M->sh[g];
    DevScope ds(S.dev);
    if (M->virt && g > 0) continue;
    S.y.need(sizeof(FT) * Q * d);
    HIPCHECK(hipMemcpyAsync(S.y.p, y_pin, sizeof(FT) * Q * d, hipMemcpyHostToDevice, S.s));
  }
  auto yof = [&](int g) { return reinterpret_cast<const ftype *>(M->virt ? M->sh[0].y.p : M->sh[g].y.p); };
  // ---- 0. hash codes of the owned query slice; all-gather
  for (int g = 0; g < G; g++) {
    MultiShard &S = M->sh[g];
    DevScope ds(S.dev);
    S.codes_slice.need(sizeof(u32) * qs * T), S.codes_all.need(sizeof(u32) * Qp * T);
    S.keys.need(sizeof(Key) * Qp * K1), S.keys_in.need(sizeof(Key) * Qp * K1);
    S.nvalid.need(sizeof(u32) * Q), S.nown.need(sizeof(u32) * Q);
    S.top_i.need(sizeof(u32) * qs * k), S.top_d.need(sizeof(FT) * qs * k);
    S.top_all.need(sizeof(u32) * Qp * k), S.top_d_all.need(sizeof(FT) * Qp * k);
    S.s2.need(sizeof(FT) * Qp * W2), S.s2_in.need(sizeof(FT) * Qp * W2);
    S.flagged.need(sizeof(u32) * (Q + 1)), S.flist.need(sizeof(u32) * (2 + fcap));
    S.xrows_i.need(sizeof(u32) * fcap * Lc1), S.xrows_d.need(sizeof(FT) * fcap * Lc1);
    S.out_i.need(sizeof(u32) * qs * k), S.out_d.need(sizeof(FT) * qs * k);
    annhip_sh_codes(S.ix, S.s, Q, yof(g), g * qs, g * qs + qs, (u32 *)S.codes_slice.p);
  }
  for (int g = 0; g < G; g++) a[g] = M->sh[g].codes_slice.p, b[g] = M->sh[g].codes_all.p;
  multi_all_gather(M, a.data(), b.data(), sizeof(u32) * qs * T);
  if (!exact_all) {
    // ---- 1. stage 1 of ALL queries over the owned rows; the k+1 best keys travel to each query's owner
    for (int g = 0; g < G; g++) {
      MultiShard &S = M->sh[g];
      DevScope ds(S.dev);
      annhip_sh_stage1(S.ix, S.s, Q, yof(g), alias, (const u32 *)S.codes_all.p, S.keys.p, (u32 *)S.nvalid.p, (u32 *)S.nown.p);
    }
    for (int g = 0; g < G; g++) a[g] = M->sh[g].keys.p, b[g] = M->sh[g].keys_in.p;
    multi_all_to_all(M, a.data(), b.data(), sizeof(Key) * qs * K1);
    // ---- 2. owner: merge + selection proof; all-gather of the top-k ids
    for (int g = 0; g < G; g++) {
      MultiShard &S = M->sh[g];
      DevScope ds(S.dev);
      annhip_sh_merge_finalize(S.ix, S.s, G, Q, g * qs, qs, S.keys_in.p, (const u32 *)S.nvalid.p, (u32 *)S.top_i.p,
                               reinterpret_cast<ftype *>(S.top_d.p));
    }
    for (int g = 0; g < G; g++) a[g] = M->sh[g].top_i.p, b[g] = M->sh[g].top_all.p;
    multi_all_gather(M, a.data(), b.data(), sizeof(u32) * qs * k);
  } else {
    for (int g = 0; g < G; g++) {  // every query takes the exact path
      MultiShard &S = M->sh[g];
      DevScope ds(S.dev);
      fill_u32_kernel<<<grid_for(Qp * k, 256, 4096), 256, 0, S.s>>>(Qp * k, ANN_ID_FLAG, (u32 *)S.top_all.p);
      fill_u32_kernel<<<grid_for(qs * k, 256, 4096), 256, 0, S.s>>>(qs * k, ANN_ID_FLAG, (u32 *)S.top_i.p);
      HIPCHECK(hipGetLastError());
    }
  }
  // ---- 2b. flagged queries: exact stage 1 on the device around ONE fixed-size MIN all-reduce
  for (int g = 0; g < G; g++) {
    MultiShard &S = M->sh[g];
    DevScope ds(S.dev);
    annhip_sh_exact1_begin(S.ix, S.s, Q, yof(g), alias, (const u32 *)S.codes_all.p, (const u32 *)S.top_all.p, fcap,
                           (u32 *)S.flist.p, (u32 *)S.xrows_i.p, reinterpret_cast<ftype *>(S.xrows_d.p));
  }
  for (int g = 0; g < G; g++) a[g] = M->sh[g].xrows_d.p;
  multi_all_min<FT>(M, a.data(), fcap * Lc1);
  for (int g = 0; g < G; g++) {
    MultiShard &S = M->sh[g];
    DevScope ds(S.dev);
    annhip_sh_exact1_end(S.ix, S.s, Q, g * qs, qs, fcap, (const u32 *)S.flist.p, (u32 *)S.xrows_i.p,
                         reinterpret_cast<ftype *>(S.xrows_d.p), (u32 *)S.top_all.p, reinterpret_cast<ftype *>(S.top_d_all.p),
                         (u32 *)S.top_i.p, reinterpret_cast<ftype *>(S.top_d.p));
    // ---- 3. distances of the neighbour-of-neighbour slots this shard owns
    annhip_sh_stage2(S.ix, S.s, Q, yof(g), alias, (const u32 *)S.top_all.p, reinterpret_cast<ftype *>(S.s2.p), (u32 *)S.flagged.p);
  }
  for (int g = 0; g < G; g++) a[g] = M->sh[g].s2.p, b[g] = M->sh[g].s2_in.p;
  multi_all_to_all(M, a.data(), b.data(), sizeof(FT) * qs * W2);
  // ---- 4. owner: min over the partial rows + the reference's network; the owners' slices go straight to the host
  const size_t ib = (sizeof(u32) * Qp * k + 15) & ~(size_t)15, db = (sizeof(FT) * Qp * k + 15) & ~(size_t)15;
  char *out_pin = (char *)M->out_pin.need(ib + db + 64);
  u32 *ids32 = (u32 *)out_pin;
  FT *dd = (FT *)(out_pin + ib);
  u32 *head = (u32 *)(out_pin + ib + db);  // {flagged beyond fcap, flagged in total}
  for (int g = 0; g < G; g++) {
    MultiShard &S = M->sh[g];
    DevScope ds(S.dev);
    annhip_sh_final(S.ix, S.s, G, Q, g * qs, qs, (const u32 *)S.top_i.p, reinterpret_cast<const ftype *>(S.top_d.p),
                    reinterpret_cast<const ftype *>(S.s2_in.p), (u32 *)S.out_i.p, reinterpret_cast<ftype *>(S.out_d.p));
    HIPCHECK(hipMemcpyAsync(ids32 + g * qs * k, S.out_i.p, sizeof(u32) * qs * k, hipMemcpyDeviceToHost, S.s));
    HIPCHECK(hipMemcpyAsync(dd + g * qs * k, S.out_d.p, sizeof(FT) * qs * k, hipMemcpyDeviceToHost, S.s));
    if (g == 0) {
      HIPCHECK(hipMemcpyAsync(head, S.flagged.p, sizeof(u32), hipMemcpyDeviceToHost, S.s));
      HIPCHECK(hipMemcpyAsync(head + 1, (u32 *)S.flist.p + 1, sizeof(u32), hipMemcpyDeviceToHost, S.s));
    }
  }
  if (overlap) (*overlap)();
  for (int g = 0; g < G; g++) {
    if (M->virt && g > 0) break;
    DevScope ds(M->sh[g].dev);
    HIPCHECK(hipStreamSynchronize(M->sh[g].s));
  }
  const size_t nf = head[0];
  M->exact_queries += head[1], M->queries += (double)Q, M->calls += 1;
  if (nf) {
    // ---- repair (rare: duplicate-heavy data, k beyond the sorted prefix): the flagged queries the device-driven path
    // had no room for.  Every shard computes its part of their full rows, MIN all-reduce, the literal network; then
    // their stage-2 rows the same way; shard 0 produces the final rows and the host patches them in.
    std::vector<u32> fl(nf);
    {
      DevScope ds(M->sh[0].dev);
      HIPCHECK(hipMemcpy(fl.data(), (u32 *)M->sh[0].flagged.p + 1, sizeof(u32) * nf, hipMemcpyDeviceToHost));
    }
    std::sort(fl.begin(), fl.end());  // appended by atomics: the same order on every shard
    for (int g = 0; g < G; g++) {
      MultiShard &S = M->sh[g];
      DevScope ds(S.dev);
      S.fl.need(sizeof(u32) * nf);
      S.r1i.need(sizeof(u32) * nf * Lc1), S.r1d.need(sizeof(FT) * nf * Lc1);
      S.r2i.need(sizeof(u32) * nf * Lc2), S.r2d.need(sizeof(FT) * nf * Lc2);
      HIPCHECK(hipMemcpyAsync(S.fl.p, fl.data(), sizeof(u32) * nf, hipMemcpyHostToDevice, S.s));
      annhip_stage1_rows(S.ix, Q, yof(g), alias, (const u32 *)S.codes_all.p, (const u32 *)S.fl.p, nf, (u32 *)S.r1i.p,
                         reinterpret_cast<ftype *>(S.r1d.p));
    }
    for (int g = 0; g < G; g++) a[g] = M->sh[g].r1d.p;
    multi_all_min<FT>(M, a.data(), nf * Lc1);
    for (int g = 0; g < G; g++) {
      MultiShard &S = M->sh[g];
      DevScope ds(S.dev);
      annhip_exact_select(S.ix, 1, nf, (u32 *)S.r1i.p, reinterpret_cast<ftype *>(S.r1d.p), (const u32 *)S.fl.p,
                          (u32 *)S.top_all.p, reinterpret_cast<ftype *>(S.top_d_all.p));
      annhip_stage2_rows_list(S.ix, Q, yof(g), alias, (const u32 *)S.fl.p, nf, (const u32 *)S.top_all.p,
                              reinterpret_cast<const ftype *>(S.top_d_all.p), (u32 *)S.r2i.p, reinterpret_cast<ftype *>(S.r2d.p));
    }
    for (int g = 0; g < G; g++) a[g] = M->sh[g].r2d.p;
    multi_all_min<FT>(M, a.data(), nf * Lc2);
    {
      MultiShard &S = M->sh[0];
      DevScope ds(S.dev);
      S.full_i.need(sizeof(u32) * Qp * k), S.full_d.need(sizeof(FT) * Qp * k);
      annhip_exact_select(S.ix, 2, nf, (u32 *)S.r2i.p, reinterpret_cast<ftype *>(S.r2d.p), (const u32 *)S.fl.p, (u32 *)S.full_i.p,
                          reinterpret_cast<ftype *>(S.full_d.p));
      std::vector<u32> fi(Qp * k);
      std::vector<FT> fd(Qp * k);
      HIPCHECK(hipMemcpyAsync(fi.data(), S.full_i.p, sizeof(u32) * Qp * k, hipMemcpyDeviceToHost, S.s));
      HIPCHECK(hipMemcpyAsync(fd.data(), S.full_d.p, sizeof(FT) * Qp * k, hipMemcpyDeviceToHost, S.s));
      HIPCHECK(hipStreamSynchronize(S.s));
      for (u32 x : fl)
        for (size_t j = 0; j < k; j++) ids32[x * k + j] = fi[x * k + j], dd[x * k + j] = fd[x * k + j];
    }
    for (int g = 1; g < G && !M->virt; g++) {
      DevScope ds(M->sh[g].dev);
      HIPCHECK(hipStreamSynchronize(M->sh[g].s));
    }
  }
  for (size_t i = 0; i < Q * k; i++) ids_out[i] = ids32[i];  // u32 bit patterns -> the ABI's size_t ids
  if (dists_out) memcpy(dists_out, dd, sizeof(FT) * Q * k);
}

// ----------------------------------------------------------------------------- precomp across the shards
// The phases of annhip_precomp_* with the three exchanges between them (sharded.py:
// precomp_sharded).  Every device holds ALL rows during the build; afterwards each keeps its row slice.
// Returns the sharded resident index; graph ids (size_t[n][k], malloc) and their squared distances go to the caller.
static annhip_multi *multi_precomp(const MultiCfg &cfg, size_t n, size_t k, size_t d, const ftype *points, int tries,
                                   size_t rots_before, size_t rot_len_before, size_t rots_after, size_t rot_len_after,
                                   size_t **graph_out, ftype **dists_out) {
  const int G = cfg.G;
  // The one use of the caller's random() stream: shard 0's handle draws the transforms (before its first HIP call,
  // precomp_begin_impl), the other shards reuse them.  It also uploads the points to the current device.
  annhip_multi *M = NULL;
  std::vector<annhip_precomp *> h(G, (annhip_precomp *)NULL);
  h[0] = precomp_begin_impl(n, k, d, points, 0, tries, rots_before, rot_len_before, rots_after, rot_len_after, 0, G, NULL);
  int cur_dev = 0;
  RandGuard keep_callers_stream;
  HIPCHECK(hipGetDevice(&cur_dev));
  M = multi_new(cfg, n, k, d);
  if (M->sh[0].dev != cur_dev) die("multi-device precomp must start on device 0 (unset ANN_HIP_DEVICE)");
  const FT *full0 = h[0]->ix->d_points;  // shard 0's copy of all rows (owned by its index)
  multi_each(M, [&](int g) {
    if (g == 0) return;
    if (M->virt)  // same device: borrow shard 0's copy
      h[g] = precomp_begin_impl(n, k, d, reinterpret_cast<const ftype *>(full0), 1, tries, rots_before, rot_len_before,
                                rots_after, rot_len_after, g, G, &h[0]->hx);
    else
      h[g] = precomp_begin_impl(n, k, d, points, 0, tries, rots_before, rot_len_before, rots_after, rot_len_after, g, G,
                                &h[0]->hx);
  }, true);
  size_t info[6];
  annhip_precomp_info(h[0], info);
  const size_t Wn = info[1], rows_per = (n + G - 1) / G;
  std::vector<void *> mi(G), md(G), cs(G), ca(G);
  multi_each(M, [&](int g) {
    mi[g] = dev_alloc<u32>(n * Wn), md[g] = dev_alloc<FT>(n * Wn);
    cs[g] = dev_alloc<u32>(rows_per), ca[g] = dev_alloc<u32>(rows_per * G);
    HIPCHECK(hipMemset(cs[g], 0, sizeof(u32) * rows_per));
    if (G > 1) annhip_precomp_init_merged(h[g], (u32 *)mi[g], reinterpret_cast<ftype *>(md[g]));
    HIPCHECK(hipDeviceSynchronize());
  }, true);
  auto rlo = [&](int g) { return std::min(n, (size_t)g * rows_per); };
  auto rhi = [&](int g) { return std::min(n, rlo(g) + rows_per); };
  // the build phases run on the null stream of each device and are host-synchronous: the exchanges use it too
  std::vector<hipStream_t> keep(G);
  for (int g = 0; g < G; g++) keep[g] = M->sh[g].s, M->sh[g].s = (hipStream_t)0;
  auto sync_all = [&] {
    for (int g = 0; g < G; g++) {
      DevScope ds(M->sh[g].dev);
      HIPCHECK(hipDeviceSynchronize());
    }
  };
  for (int t = 0; t < tries; t++) {
    multi_each(M, [&](int g) { annhip_precomp_hash(h[g], t, rlo(g), rhi(g), (u32 *)cs[g]); }, true);
    multi_all_gather(M, cs.data(), ca.data(), sizeof(u32) * rows_per);
    sync_all();
    multi_each(M, [&](int g) { annhip_precomp_try(h[g], t, (const u32 *)ca[g], (u32 *)mi[g], reinterpret_cast<ftype *>(md[g])); }, true);
  }
  if (G > 1) {  // every entry has exactly one writer (bucket b: shard b mod G); the others hold INT32_MAX / +inf
    multi_all_min<int>(M, mi.data(), n * Wn, true);
    multi_all_min<FT>(M, md.data(), n * Wn, true);
    sync_all();
  }
  std::vector<void *> gs(G), gds(G), ga(G), gda(G);
  multi_each(M, [&](int g) {
    annhip_precomp_merge(h[g], (u32 *)mi[g], reinterpret_cast<ftype *>(md[g]));
    HIPCHECK(hipFree(mi[g]));
    HIPCHECK(hipFree(md[g]));
    HIPCHECK(hipFree(cs[g]));
    HIPCHECK(hipFree(ca[g]));
    gs[g] = dev_alloc<u32>(rows_per * k), gds[g] = dev_alloc<FT>(rows_per * k);
    ga[g] = dev_alloc<u32>(rows_per * G * k), gda[g] = dev_alloc<FT>(rows_per * G * k);
    HIPCHECK(hipMemset(gs[g], 0, sizeof(u32) * rows_per * k));
    HIPCHECK(hipMemset(gds[g], 0, sizeof(FT) * rows_per * k));
    annhip_precomp_graph(h[g], rlo(g), rhi(g), (u32 *)gs[g], reinterpret_cast<ftype *>(gds[g]));
  }, true);
  multi_all_gather(M, gs.data(), ga.data(), sizeof(u32) * rows_per * k);
  multi_all_gather(M, gds.data(), gda.data(), sizeof(FT) * rows_per * k);
  sync_all();
  for (int g = 0; g < G; g++) M->sh[g].s = keep[g];
  // results for the caller: from shard 0
  {
    DevScope ds(M->sh[0].dev);
    std::vector<u32> g32(n * k);
    HIPCHECK(hipMemcpy(g32.data(), ga[0], sizeof(u32) * n * k, hipMemcpyDeviceToHost));
    *graph_out = (size_t *)malloc(sizeof(size_t) * n * k);
    for (size_t i = 0; i < n * k; i++) (*graph_out)[i] = g32[i];
    if (dists_out) {
      *dists_out = (ftype *)malloc(sizeof(ftype) * n * k);
      HIPCHECK(hipMemcpy(*dists_out, gda[0], sizeof(FT) * n * k, hipMemcpyDeviceToHost));
    }
  }
  // the resident shards: finish every handle (complete index), then keep the own row slice only
  multi_each(M, [&](int g) {
    MultiShard &S = M->sh[g];
    S.ix = annhip_precomp_finish(h[g], (const u32 *)ga[g]);
    HIPCHECK(hipFree(gs[g]));
    HIPCHECK(hipFree(gds[g]));
    HIPCHECK(hipFree(ga[g]));
    HIPCHECK(hipFree(gda[g]));
    S.ix->stream = S.s;
    S.ix->profile = g_host_profile ? 1 : 0;
  }, true);
  if (M->virt) {  // all shards borrow shard 0's full copy, which this host takes over
    annhip_index *i0 = M->sh[0].ix;
    M->shared_points = i0->d_points;
    i0->own_points = false;
    for (int g = 0; g < G; g++) {
      MultiShard &S = M->sh[g];
      annhip_index_reshard(S.ix, reinterpret_cast<const ftype *>(M->shared_points + S.lo * d), S.lo, S.hi);
    }
  } else {
    multi_each(M, [&](int g) {
      MultiShard &S = M->sh[g];
      S.rows = dev_alloc<FT>((S.hi - S.lo) * d);
      HIPCHECK(hipMemcpy(S.rows, S.ix->d_points + S.lo * d, sizeof(FT) * (S.hi - S.lo) * d, hipMemcpyDeviceToDevice));
      annhip_index_reshard(S.ix, reinterpret_cast<const ftype *>(S.rows), S.lo, S.hi);  // frees the full copy
    }, true);
  }
  return M;
}
#endif
