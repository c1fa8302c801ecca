// Probe for "HBM-bound work resident beside the 256 x 256 GEMM workgroups" (DESIGN.md 9, round-4 item): a read-modify-write stream
// over three fp32 arrays (the p / m / v traffic of the fused out_layer.fc1 update: 24 B per element) in a kernel small enough to
// share a CU with one 8-wave GEMM workgroup (128 KiB of LDS, 2 x 240 of the 512 registers per SIMD lane): no LDS, <= 32 VGPRs,
// 4-wave workgroups.  Built as a tiny shared library and driven by tools/dbg/coresident_probe.py:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/dbg/micro/libcoresident_probe.so tools/dbg/micro/coresident_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));

__attribute__((amdgpu_num_vgpr(32))) __global__ __launch_bounds__(256) void rmw_stream_kernel(f4* __restrict__ p, f4* __restrict__ m,
                                                                                              f4* __restrict__ v, size_t n4, float g) {
  const uint32_t stride = gridDim.x * 256u, n = (uint32_t)n4;       // (n4 < 2^32: one 32-bit element index, uniform bases)
#pragma unroll 1
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
    f4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(m + i), c = __builtin_nontemporal_load(v + i);
    b = b * 0.9f + g * 0.1f;                              // an AdamW-like amount of arithmetic per element
    c = c * 0.999f + g * g * 0.001f;
    f4 d;
    for (int k = 0; k < 4; ++k) d[k] = b[k] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(c[k]) + 1e-6f);   // (approximate forms: a probe)
    a = a - 1e-3f * d - 1e-5f * a;
    __builtin_nontemporal_store(a, p + i);
    __builtin_nontemporal_store(b, m + i);
    __builtin_nontemporal_store(c, v + i);
  }
}

extern "C" int probe_rmw_stream(void* p, void* m, void* v, uint64_t n, int blocks, void* stream) {
  hipLaunchKernelGGL(rmw_stream_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f4*)p, (f4*)m, (f4*)v, (size_t)(n / 4), 0.01f);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
