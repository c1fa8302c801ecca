// MX-FP8 (OCP e4m3fn elements, one E8M0 power-of-two scale per 32 consecutive K elements) products on gfx950's block-scaled matrix
// instruction v_mfma_scale_f32_16x16x128_f8f6f4: the "fp8 MFMA" of BASELINE.json configs[4] (ViT-L/14 swap).  NOT a parity path:
// an e4m3 element keeps 3 mantissa bits, so one product is ~3 % away from the fp32 result -- north_star's 1e-3 bar belongs to
// the split-bf16 path, which stays the default everywhere.  This file is the inference-only fast mode.
//
//   lr2_quant_mxfp8 : fp32 [R, K] -> e4m3 bytes [R, K] + E8M0 scale bytes [R, K / 32]            (OCP MX v1.0 section 6.3)
//   lr2_gemm_mxfp8  : C[M, N] = (A_q . B_q^T) (+ bias) (GELU) (+ residual), A_q [M, K], B_q [N, K] as produced above
//
// Operand layout of the instruction, measured (tools/dbg/micro/mxfp8_probe.hip -- the ISA manual is not in this image): lane l
// holds 32 bytes of row (A) / column (B) l & 15; bytes 0-15 are k = 16 q .. 16 q + 15, bytes 16-31 are k = 64 + 16 q .. + 15 with
// q = l >> 4; the scale of (row, 32-element block b) is byte `opsel` of the scale register of lane row + 16 b.  So a lane's
// operand is two 16-byte pieces of a row-major row, and its scale is byte (l >> 4) of the row's 4 scale bytes for this K step.
// The adder tree of the instruction is not an fp32 sum: against an exact sum of the (exact) products the result is off by up to
// ~1e-3 of the largest term (same probe), which is below the element format's own error.
#include "fp8_common.h"
#include "lr2ppo_hip.h"

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

// ---- quantiser: one wave per 4 rows x 512 columns chunk; a lane owns 8 consecutive elements, 4 lanes one 32-element block ----
__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const float* __restrict__ x, uint8_t* __restrict__ q, uint8_t* __restrict__ s,
                                                          int R, int K, int ldx) {
  const int k8 = K / 8;                                  // 8-element groups per row
  const size_t total = (size_t)R * k8;
  // (total and every lane's first index are multiples of 4 lanes' worth: a block's four lanes enter and leave the loop together)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const bool ok = true;
    const int r = (int)(i / k8), g = (int)(i % k8);
    const float4 a = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + 8 * g);
    const float4 b = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + 8 * g + 4);
    float amax = fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                       fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
    amax = group4_max(amax);         // the block's 32 elements = 4 consecutive lanes (k8 is a multiple of 4)
    // shared exponent: floor(log2(amax)) - emax(e4m3 = 8), as a biased E8M0 byte; amax = 0 (or denormal): the smallest scale
    const int ex = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127;          // floor(log2(amax)) of a normal float
    int e = ex - 8;
    if (amax < 1.17549435e-38f) e = -127;
    if (e < -127) e = -127;
    if (e > 127) e = 127;
    const uint32_t ef = (uint32_t)(127 - e);
    const float inv = __uint_as_float(ef ? ef << 23 : 0x00400000u);       // 2^-e (e = 127: the denormal 2^-127)
    auto sat = [](float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); };
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(a.x * inv), sat(a.y * inv), w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(a.z * inv), sat(a.w * inv), w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(b.x * inv), sat(b.y * inv), w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(sat(b.z * inv), sat(b.w * inv), w1, true);
    if (ok) {
      *reinterpret_cast<int2*>(q + (size_t)r * K + 8 * g) = make_int2(w0, w1);
      if ((g & 3) == 0) s[(size_t)r * (K / 32) + (g >> 2)] = (uint8_t)(e + 127);
    }
  }
}

// ---- product: 4 waves (2 x 2) per workgroup, tile 128 x 128, wave tile 64 x 64 = 4 x 4 instruction tiles, K step 128 ----
// Fragments come straight from global memory (a lane's operand is 2 x 16 contiguous bytes of a row; the four lanes of a row read
// one 128-byte line between them): the L1 / L2 hit rate does the staging an LDS ring would do.  The next K step's fragments are
// requested before this step's 16 instructions.
struct Frags {
  v8i_t a[4], b[4];
  int sa[4], sb[4];
};

__device__ __forceinline__ void load_frags(const Mx8Params& p, Frags& f, const uint8_t* const (&arow)[4], const uint8_t* const (&brow)[4],
                                           const uint8_t* const (&asrow)[4], const uint8_t* const (&bsrow)[4], int k0, int q) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const v4i_t a0 = *reinterpret_cast<const v4i_t*>(arow[i] + k0 + 16 * q);
    const v4i_t a1 = *reinterpret_cast<const v4i_t*>(arow[i] + k0 + 64 + 16 * q);
    f.a[i] = v8i_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const v4i_t b0 = *reinterpret_cast<const v4i_t*>(brow[i] + k0 + 16 * q);
    const v4i_t b1 = *reinterpret_cast<const v4i_t*>(brow[i] + k0 + 64 + 16 * q);
    f.b[i] = v8i_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    f.sa[i] = (int)asrow[i][(k0 >> 5) + q];
    f.sb[i] = (int)bsrow[i][(k0 >> 5) + q];
  }
}

__global__ __launch_bounds__(256) void gemm_mxfp8_kernel(Mx8Params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = p.N / 128;
  // XCD-aware order is not needed for correctness; keep neighbouring tiles of one row on one XCD's L2 by walking N fastest
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int m0 = tm * 128 + wr * 64, n0 = tn * 128 + wc * 64;
  const int r16 = lane & 15, q = lane >> 4;
  const uint8_t* arow[4];
  const uint8_t* brow[4];
  const uint8_t* asrow[4];
  const uint8_t* bsrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + 16 * i + r16;
    if (m >= p.M) m = p.M - 1;                             // ragged M: clamped reads, masked stores
    const int n = n0 + 16 * i + r16;
    arow[i] = p.aq + (size_t)m * p.K;
    brow[i] = p.bq + (size_t)n * p.K;
    asrow[i] = p.as + (size_t)m * (p.K / 32);
    bsrow[i] = p.bs + (size_t)n * (p.K / 32);
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  Frags cur, nxt;
  load_frags(p, cur, arow, brow, asrow, bsrow, 0, q);
  for (int k0 = 0; k0 < p.K; k0 += 128) {
    if (k0 + 128 < p.K) load_frags(p, nxt, arow, brow, asrow, bsrow, k0 + 128, q);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cur.a[i], cur.b[j], acc[i][j], 0, 0, 0, cur.sa[i], 0, cur.sb[j]);
    if (k0 + 128 < p.K) cur = nxt;
  }
  // epilogue: acc[i][j][r] = C[m0 + 16 i + 4 q + r][n0 + 16 j + r16]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 16 * j + r16;
    const float bj = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * i + 4 * q + r;
        if (m < p.M) {
          float v = acc[i][j][r] + bj;
          if (p.act == 1) v = gelu_erf(v);
          if (p.resid) v += p.resid[(size_t)m * p.ld_resid + n];
          p.out[(size_t)m * p.ld_out + n] = v;
        }
      }
  }
}

// ---- the same product with the operand tiles staged through LDS (two stages of 128 rows x 128 bytes per operand = 64 KiB) ----
// A 16-byte chunk c of tile row r lives at r * 128 + ((c ^ (r & 7)) << 4): the 16 lanes of a fragment read (rows r .. r + 15, one
// chunk column) then hit 8 different 16-byte bank groups twice instead of one group 16 times.  Each thread moves four chunks of A
// and four of B per K step: requested from global memory before the step's matrix instructions, written to the other stage after.
__device__ __forceinline__ int mx_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

__global__ __launch_bounds__(256, 2) void gemm_mxfp8_lds_kernel(Mx8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                     // [2][128 * 128]
  char* sB = smem + 2 * 16384;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_n = p.N / 128;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  const int r16 = lane & 15, q = lane >> 4;
  // this thread's four chunks of a stage: rows (tid >> 3) + 32 it, chunk tid & 7
  const int cr = tid >> 3, cc = tid & 7;
  const uint8_t* ga[4];
  const uint8_t* gb[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    int m = tm * 128 + cr + 32 * it;
    if (m >= p.M) m = p.M - 1;
    ga[it] = p.aq + (size_t)m * p.K + 16 * cc;
    gb[it] = p.bq + (size_t)(tn * 128 + cr + 32 * it) * p.K + 16 * cc;
  }
  const uint8_t* asrow[4];
  const uint8_t* bsrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = tm * 128 + wr * 64 + 16 * i + r16;
    if (m >= p.M) m = p.M - 1;
    asrow[i] = p.as + (size_t)m * (p.K / 32) + q;
    bsrow[i] = p.bs + (size_t)(tn * 128 + wc * 64 + 16 * i + r16) * (p.K / 32) + q;
  }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  v4i_t ra[4], rb[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    ra[it] = *reinterpret_cast<const v4i_t*>(ga[it]);
    rb[it] = *reinterpret_cast<const v4i_t*>(gb[it]);
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    *reinterpret_cast<v4i_t*>(sA + mx_off(cr + 32 * it, cc)) = ra[it];
    *reinterpret_cast<v4i_t*>(sB + mx_off(cr + 32 * it, cc)) = rb[it];
  }
  __syncthreads();
  int st = 0;
  for (int k0 = 0; k0 < p.K; k0 += 128) {
    const bool more = k0 + 128 < p.K;
    if (more) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        ra[it] = *reinterpret_cast<const v4i_t*>(ga[it] + k0 + 128);
        rb[it] = *reinterpret_cast<const v4i_t*>(gb[it] + k0 + 128);
      }
    }
    int sa[4], sb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sa[i] = (int)asrow[i][k0 >> 5];
      sb[i] = (int)bsrow[i][k0 >> 5];
    }
    const char* cA = sA + st * 16384;
    const char* cB = sB + st * 16384;
    v8i_t fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wc * 64 + 16 * j + r16;
      const v4i_t b0 = *reinterpret_cast<const v4i_t*>(cB + mx_off(row, q));
      const v4i_t b1 = *reinterpret_cast<const v4i_t*>(cB + mx_off(row, 4 + q));
      fb[j] = v8i_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wr * 64 + 16 * i + r16;
      const v4i_t a0 = *reinterpret_cast<const v4i_t*>(cA + mx_off(row, q));
      const v4i_t a1 = *reinterpret_cast<const v4i_t*>(cA + mx_off(row, 4 + q));
      const v8i_t fa = v8i_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb[j], acc[i][j], 0, 0, 0, sa[i], 0, sb[j]);
    }
    if (more) {
      char* nA = sA + (st ^ 1) * 16384;
      char* nB = sB + (st ^ 1) * 16384;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        *reinterpret_cast<v4i_t*>(nA + mx_off(cr + 32 * it, cc)) = ra[it];
        *reinterpret_cast<v4i_t*>(nB + mx_off(cr + 32 * it, cc)) = rb[it];
      }
    }
    __syncthreads();
    st ^= 1;
  }
  // epilogue: the wave's 64 x 64 results through a wave-private LDS slab, 32 rows at a time (the operand stages are dead: the loop
  // ended with a barrier), so that a row leaves as 256 contiguous bytes instead of 16 four-byte pieces
  const int m0 = tm * 128 + wr * 64, n0 = tn * 128 + wc * 64;
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * 68);
  const int orow = lane >> 4, ocol = (lane & 15) * 4;
  const int n = n0 + ocol;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(16 * ii + 4 * q + r) * 68 + 16 * j + r16] = acc[2 * half + ii][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int lr = 4 * pass + orow;
      const int m = m0 + 32 * half + lr;
      if (m < p.M) {
        float4 v = *reinterpret_cast<const float4*>(slab + lr * 68 + ocol);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (p.act == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        if (p.resid) {
          const float4 rr = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ld_resid + n);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.out) *reinterpret_cast<float4*>(p.out + (size_t)m * p.ld_out + n) = v;
        if (p.out_hi) { if (p.out_lo_off) store_planes4(p.out_hi + (size_t)m * p.ld_planes + n, p.out_lo_off, v); else store_bf16x4(p.out_hi + (size_t)m * p.ld_planes + n, v); }
      }
      if (p.out_q) {
        // a row's 32-column MX block = 8 consecutive lanes x 4 columns (the wave tile's 64 columns are two blocks per row):
        // quantise here, as lr2_quant_mxfp8 would the stored row (rows past M compute on garbage and store nothing)
        float4 v = *reinterpret_cast<const float4*>(slab + lr * 68 + ocol);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (p.act == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        if (p.resid && m < p.M) {
          const float4 rr = *reinterpret_cast<const float4*>(p.resid + (size_t)m * p.ld_resid + n);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        amax = group8_max(amax);
        int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - 8;
        if (amax < 1.17549435e-38f) e = -127;
        if (e < -127) e = -127;
        if (e > 127) e = 127;
        const uint32_t ef = (uint32_t)(127 - e);
        const float inv = __uint_as_float(ef ? ef << 23 : 0x00400000u);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.x * inv, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.y * inv, -448.f, 448.f), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.z * inv, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.w * inv, -448.f, 448.f), w, true);
        if (m < p.M) {
          *reinterpret_cast<int*>(p.out_q + (size_t)m * p.N + n) = w;
          if ((lane & 7) == 0) p.out_s[(size_t)m * (p.N / 32) + (n >> 5)] = (uint8_t)(e + 127);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int lr2_quant_mxfp8(const void* x, int ldx, void* q, void* scales, int rows, int K, void* stream) {
  if (!x || !q || !scales || rows <= 0 || K <= 0) return LR2_ERR_ARG;
  if (K % 32 || ldx < K || ldx % 4) return LR2_ERR_SHAPE;
  const size_t total = (size_t)rows * (K / 8);
  size_t blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  LR2_LAUNCH(quant_mxfp8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (uint8_t*)q, (uint8_t*)scales,
             rows, K, ldx);
  return lr2_launch_status(__func__);
}

extern "C" int lr2_gemm_mxfp8(const void* a_q, const void* a_scales, const void* b_q, const void* b_scales, void* out, int ld_out,
                              const void* bias, const void* resid, int ld_resid, int act, void* out_q, void* out_scales,
                              void* out_hi, uint64_t out_lo_off, int ld_planes, int M, int N, int K, void* stream) {
  if (!a_q || !a_scales || !b_q || !b_scales || (!out && !out_q && !out_hi) || M <= 0 || N <= 0 || K <= 0 || (act != 0 && act != 1))
    return LR2_ERR_ARG;
  if (out_hi && (ld_planes < N || ld_planes % 4 || out_lo_off % 4)) return LR2_ERR_SHAPE;
  if ((out_q != nullptr) != (out_scales != nullptr)) return LR2_ERR_ARG;
  if (N % 128 || K % 128 || (out && (ld_out < N || ld_out % 4)) || (resid && (ld_resid < N || ld_resid % 4))) return LR2_ERR_SHAPE;
  Mx8Params p{(const uint8_t*)a_q, (const uint8_t*)a_scales, (const uint8_t*)b_q, (const uint8_t*)b_scales, (float*)out,
              (const float*)bias, (const float*)resid, M, N, K, ld_out, ld_resid, act, (uint8_t*)out_q, (uint8_t*)out_scales,
              (bf16_t*)out_hi, (size_t)out_lo_off, ld_planes};
  // Large products (the encoders' token products at M >= 1e4 rows): the 256 x 256 LDS-DMA ring of gemm256_mx.hip when its
  // one-workgroup-per-CU rounds are well filled; LR2_FP8_256=0 keeps everything on the 128 x 128 kernel (A/B).
  {
    static int env256 = -1;
    if (env256 < 0) {
      const char* e2 = getenv("LR2_FP8_256");
      env256 = e2 ? atoi(e2) : 1;
    }
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256), rounds = (t256 + 255) / 256;
    const uint64_t lim = 0xFFFFFD00ull;
    if (env256 && t256 >= 256 && t256 * 100 >= 80 * rounds * 256 && (uint64_t)M * K <= lim && (uint64_t)N * K <= lim)
      return launch_gemm256_mx(p, (hipStream_t)stream);
  }
  const int tiles = ((M + 127) / 128) * (N / 128);
  const char* e = getenv("LR2_FP8_LDS");          // 0: fragments straight from global memory (the first version; A/B)
  if (e && atoi(e) == 0) {
    if (out_q || out_hi || !out) return LR2_ERR_ARG;          // the first version writes fp32 only
    LR2_LAUNCH(gemm_mxfp8_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, p);
    return lr2_launch_status(__func__);
  }
  static bool attr = false;
  if (!attr) {
    if (lr2_allow_dynamic_lds(gemm_mxfp8_lds_kernel, 65536, "gemm_mxfp8")) return LR2_ERR_LAUNCH;
    attr = true;
  }
  LR2_LAUNCH(gemm_mxfp8_lds_kernel, dim3(tiles), dim3(256), 65536, (hipStream_t)stream, p);
  return lr2_launch_status(__func__);
}
