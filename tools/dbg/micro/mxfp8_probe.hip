// Which (lane, byte) of the 32-byte A / B operands of v_mfma_scale_f32_16x16x128_f8f6f4 is which (row / column, k), and which block
// does a lane's scale byte apply to?  One wave, random e4m3 operands and E8M0 scales, checked against candidate layouts on the host.
//   hipcc --offload-arch=gfx950 -O2 -o mxfp8_probe mxfp8_probe.hip && ./mxfp8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, v4f* c) {
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  c[threadIdx.x] = acc;
}

static float e4m3(uint8_t v) {          // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

int run(int scale_mode);
int main() { for (int m = 0; m < 6; ++m) run(m); return 0; }
int run(int scale_mode) {
  printf("== mode %d (0: unit scales, 1: A scales vary, 2: both vary, 3: B scales vary, 4: both vary + gaussian data, 5: unit scales + gaussian data)\n", scale_mode);
  const bool gauss = scale_mode >= 4;
  if (scale_mode == 4) scale_mode = 2;
  if (scale_mode == 5) scale_mode = 0;
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  std::vector<int> SA(64), SB(64);
  srand(3);
  auto enc = [](float x) {        // nearest e4m3 by search (probe only)
    uint8_t best = 0; float bd = 1e30f;
    for (int v = 0; v < 256; ++v) { if ((v & 0x7F) == 0x7F) continue; const float d = std::fabs(e4m3((uint8_t)v) - x); if (d < bd) { bd = d; best = (uint8_t)v; } }
    return best; };
  auto gs = []() { float s = 0; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return (s - 6.0f) * 64.0f; };
  for (auto& v : A) { if (gauss) v = enc(gs()); else do v = rand() & 255; while ((v & 0x7F) == 0x7F); }
  for (auto& v : B) { if (gauss) v = enc(gs()); else do v = rand() & 255; while ((v & 0x7F) == 0x7F); }
  for (int l = 0; l < 64; ++l) { SA[l] = (scale_mode == 1 || scale_mode == 2) ? 124 + rand() % 7 : 127; SB[l] = scale_mode >= 2 ? 124 + rand() % 7 : 127; }   // byte 0 of the scale register
  void *da, *db, *dsa, *dsb, *dc;
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 1024);
  hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dsa, SA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, SB.data(), 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>((const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb, (v4f*)dc);
  std::vector<float> C(256);
  hipMemcpy(C.data(), dc, 1024, hipMemcpyDeviceToHost);
  // candidate k maps of byte j of lane-quarter q = l >> 4
  auto kmap = [](int hyp, int q, int j) { return hyp == 0 ? 32 * q + j : (hyp == 1 ? 16 * q + (j & 15) + 64 * (j >> 4) : 8 * q + (j & 7) + 32 * (j >> 3)); };
  for (int hyp = 0; hyp < 3; ++hyp) {
    // logical matrices under this hypothesis: Am[row][k], scale of (row, block k / 32) = 2^(SA[lane holding that block] - 127)
    double Am[16][128], Bm[16][128];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        const int k = kmap(hyp, l >> 4, j);
        // the scale of (row, 32-element block b) is byte 0 of the scale register of lane row + 16 b
        Am[l & 15][k] = (double)e4m3(A[l * 32 + j]) * std::ldexp(1.0, SA[(l & 15) + 16 * (k >> 5)] - 127);
        Bm[l & 15][k] = (double)e4m3(B[l * 32 + j]) * std::ldexp(1.0, SB[(l & 15) + 16 * (k >> 5)] - 127);
      }
    double worst = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += Am[row][k] * Bm[col][k];
        worst = std::fmax(worst, std::fabs(ref - C[l * 4 + r]) / (1.0 + std::fabs(ref)));
      }
    printf("hypothesis %d (k of byte j in lane quarter q: %s): worst relative error %.3e\n", hyp,
           hyp == 0 ? "32 q + j" : (hyp == 1 ? "16 q + (j & 15) + 64 (j >> 4)" : "8 q + (j & 7) + 32 (j >> 3)"), worst);
  }
  printf("C[0..3] of lane 0: %g %g %g %g\n", C[0], C[1], C[2], C[3]);
  return 0;
}
