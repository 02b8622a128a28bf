// tools/readbw.hip -- HBM read ceilings of the MI355X for the access patterns of the query path.
//   hipcc --offload-arch=gfx950 -O3 -o readbw tools/readbw.hip && ./readbw
// (1) grid-stride float4 read-reduce over 5 GiB, cached vs non-temporal loads, several grid sizes;
// (2) pure random 512-byte-row gather with the stage-1 lane layout (8 lanes x 4 x 16 B per row, 8 rows per wave pass),
//     ~14M rows like one cfg3 launch.  Measured round 1: stream 6.2 TB/s cached / 6.95 TB/s nt; gather 6.3 / 6.8 TB/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4 __attribute__((ext_vector_type(4)));
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void rd(const v4 *p, size_t n4, float *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  v4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    v4 t[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) t[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += t[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
// random 512-byte row gather: 8 lanes per row, 4 x 16 B each (the stage-1 access pattern), rows from a hash
template <bool NT>
__global__ __launch_bounds__(256) void gather(const v4 *p, size_t nrows, size_t per_wave, float *out) {
  const int lane = threadIdx.x & 63, g = lane >> 3, q = lane & 7;
  size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  v4 acc = {0, 0, 0, 0};
  unsigned long long h = wave * 0x9E3779B97F4A7C15ull + 12345;
  for (size_t it = 0; it < per_wave; it++) {
    h = h * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long hh = h + g * 0xD1B54A32D192ED03ull;
    hh ^= hh >> 29; hh *= 0xBF58476D1CE4E5B9ull; hh ^= hh >> 32;
    size_t row = hh % nrows;
    const v4 *rp = p + row * 32 + q;
#pragma unroll
    for (int c = 0; c < 4; c++) acc += NT ? __builtin_nontemporal_load(rp + c * 8) : rp[c * 8];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
// random 320-byte rows (the reference drivers' default d = 80 in float): 5 lanes per row, 4 x 16 B each, lane p takes chunks
// p, p+5, p+10, p+15 (the static oc = 5 layout of stage 1), 12 rows per wave pass; every row touches exactly 3 128-byte lines
template <bool NT>
__global__ __launch_bounds__(256) void gather320(const v4 *p, size_t nrows, size_t per_wave, float *out) {
  const int lane = threadIdx.x & 63, g = lane / 5, q = lane % 5;
  size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  v4 acc = {0, 0, 0, 0};
  unsigned long long h = wave * 0x9E3779B97F4A7C15ull + 12345;
  for (size_t it = 0; it < per_wave; it++) {
    h = h * 6364136223846793005ull + 1442695040888963407ull;
    unsigned long long hh = h + g * 0xD1B54A32D192ED03ull;
    hh ^= hh >> 29; hh *= 0xBF58476D1CE4E5B9ull; hh ^= hh >> 32;
    size_t row = hh % nrows;
    const v4 *rp = p + row * 20 + q;
    if (lane < 60) {
#pragma unroll
      for (int c = 0; c < 4; c++) acc += NT ? __builtin_nontemporal_load(rp + c * 5) : rp[c * 5];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = 1;
}
int main() {
  size_t bytes = (size_t)5120 << 20;  // 5 GiB, like the cfg3 point matrix
  v4 *p; float *out;
  hipMalloc(&p, bytes); hipMalloc(&out, 4); hipMemset(p, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  size_t n4 = bytes / 16;
  auto time = [&](auto launch, const char *name, double gb) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 5; r++) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3 / 1e3);
  };
  for (int blocks : {2048, 4096, 8192, 16384}) {
    char nm[96];
    snprintf(nm, 96, "stream read cached  u4 grid=%d", blocks); time([&] { rd<false, 4><<<blocks, 256>>>(p, n4, out); }, nm, bytes / 1e6);
    snprintf(nm, 96, "stream read nt      u4 grid=%d", blocks); time([&] { rd<true, 4><<<blocks, 256>>>(p, n4, out); }, nm, bytes / 1e6);
  }
  time([&] { rd<true, 8><<<4096, 256>>>(p, n4, out); }, "stream read nt      u8 grid=4096", bytes / 1e6);
  size_t nrows = bytes / 512;
  for (int wpc : {8, 16, 24, 32}) {
    int blocks = 256 * wpc / 4; size_t waves = (size_t)blocks * 4, per_wave = 14000000 / 8 / waves + 1;  // ~14M rows like cfg3
    double gb = (double)waves * per_wave * 8 * 512 / 1e6;
    char nm[96];
    snprintf(nm, 96, "random 512B rows cached  %2d waves/CU", wpc); time([&] { gather<false><<<blocks, 256>>>(p, nrows, per_wave, out); }, nm, gb);
    snprintf(nm, 96, "random 512B rows nt      %2d waves/CU", wpc); time([&] { gather<true><<<blocks, 256>>>(p, nrows, per_wave, out); }, nm, gb);
  }
  size_t nrows320 = bytes / 320;
  for (int wpc : {8, 16, 24, 32}) {
    int blocks = 256 * wpc / 4; size_t waves = (size_t)blocks * 4, per_wave = 14000000 / 12 / waves + 1;
    double gb = (double)waves * per_wave * 12 * 320 / 1e6;   // USEFUL bytes; 384 B (3 lines) are fetched per row
    char nm[96];
    snprintf(nm, 96, "random 320B rows cached  %2d waves/CU", wpc); time([&] { gather320<false><<<blocks, 256>>>(p, nrows320, per_wave, out); }, nm, gb);
    snprintf(nm, 96, "random 320B rows nt      %2d waves/CU", wpc); time([&] { gather320<true><<<blocks, 256>>>(p, nrows320, per_wave, out); }, nm, gb);
  }
  return 0;
}
