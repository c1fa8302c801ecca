// Per-CU store rate micro-benchmark: every workgroup (512 threads, 8 waves) writes one 256 x 256 fp32-sized tile (256 KB) of a
// large row-major matrix, in the shapes a GEMM epilogue can use.  Prints us per launch for a few grid sizes.
//   hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: planes, 8 B per lane per plane (16 lanes = one 128-B row segment), two stores per 4 elements   [the epilogue today]
// MODE 1: planes, 16 B per lane per plane (8 lanes = one 128-B row segment), two stores per 8 elements
// MODE 2: fp32, 16 B per lane (16 lanes = one 256-B row segment), one store per 4 elements
// NT: non-temporal
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void store_kernel(char* out, int ld_elems, int tiles_n, size_t lo_off_bytes) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int wr = wave >> 2, wc = wave & 3;                    // wave tile 128 rows x 64 columns
  const int m0 = tm * 256 + wr * 128, n0 = tn * 256 + wc * 64;
  if (MODE == 0) {
    const int row0 = lane >> 4, col = (lane & 15) * 4;
#pragma unroll 8
    for (int r = 0; r < 128; r += 4) {
      char* p = out + ((size_t)(m0 + r + row0) * ld_elems + n0 + col) * 2;
      u32x2 v = {(uint32_t)r, (uint32_t)lane};
      if (NT) { __builtin_nontemporal_store(v, (u32x2*)p); __builtin_nontemporal_store(v, (u32x2*)(p + lo_off_bytes)); }
      else { *(u32x2*)p = v; *(u32x2*)(p + lo_off_bytes) = v; }
    }
  } else if (MODE == 1) {
    const int row0 = lane >> 3, col = (lane & 7) * 8;
#pragma unroll 8
    for (int r = 0; r < 128; r += 8) {
      char* p = out + ((size_t)(m0 + r + row0) * ld_elems + n0 + col) * 2;
      u32x4 v = {(uint32_t)r, (uint32_t)lane, 1u, 2u};
      if (NT) { __builtin_nontemporal_store(v, (u32x4*)p); __builtin_nontemporal_store(v, (u32x4*)(p + lo_off_bytes)); }
      else { *(u32x4*)p = v; *(u32x4*)(p + lo_off_bytes) = v; }
    }
  } else {
    const int row0 = lane >> 4, col = (lane & 15) * 4;
#pragma unroll 8
    for (int r = 0; r < 128; r += 4) {
      char* p = out + ((size_t)(m0 + r + row0) * ld_elems + n0 + col) * 4;
      u32x4 v = {(uint32_t)r, (uint32_t)lane, 1u, 2u};
      if (NT) __builtin_nontemporal_store(v, (u32x4*)p);
      else *(u32x4*)p = v;
    }
  }
}

template <int MODE, bool NT>
static void run(const char* name, char* buf, int tiles_m, int tiles_n, int ld, size_t lo_off) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = tiles_m * tiles_n;
  for (int i = 0; i < 3; ++i) store_kernel<MODE, NT><<<grid, 512>>>(buf, ld, tiles_n, lo_off);
  hipDeviceSynchronize();
  hipEventRecord(a);
  const int reps = 50;
  for (int i = 0; i < reps; ++i) store_kernel<MODE, NT><<<grid, 512>>>(buf, ld, tiles_n, lo_off);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double us = ms / reps * 1e3;
  printf("  %-34s tiles %5d: %8.2f us per launch, %7.1f GB/s per CU-tile slot, %6.2f TB/s total\n", name, grid, us,
         262144.0 / (us * 1e-6) / 1e9 / (grid < 256 ? 1 : grid / 256.0), (double)grid * 262144.0 / (us * 1e-6) / 1e12);
}

int main() {
  const int N = 3072, tiles_n = N / 256;
  const int max_tm = 394;
  const size_t elems = (size_t)max_tm * 256 * N;
  char* buf;
  if (hipMalloc(&buf, elems * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, elems * 4);
  const size_t lo_off = elems * 2;
  for (int tm : {1, 5, 21, 42, 394}) {
    printf("tile rows %d\n", tm);
    run<0, true>("planes 8 B/lane nt (today)", buf, tm, tiles_n, N, lo_off);
    run<0, false>("planes 8 B/lane", buf, tm, tiles_n, N, lo_off);
    run<1, true>("planes 16 B/lane nt", buf, tm, tiles_n, N, lo_off);
    run<1, false>("planes 16 B/lane", buf, tm, tiles_n, N, lo_off);
    run<2, true>("fp32 16 B/lane nt", buf, tm, tiles_n, N, lo_off);
    run<2, false>("fp32 16 B/lane", buf, tm, tiles_n, N, lo_off);
  }
  return 0;
}
