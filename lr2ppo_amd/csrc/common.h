// Shared device helpers for the lr2ppo gfx950 kernels (CDNA4: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits in HBM / LDS

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> packed bf16 pair (low half = a), one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  f32x2_t v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// erf with |error| <= 6e-7 (Abramowitz & Stegun 7.1.26, |eps| <= 1.5e-7 in exact arithmetic, plus fp32 rounding of a
// 5-term Horner + one v_rcp_f32 + one v_exp_f32): 12 branch-free instructions.  libm's erff is two exec-masked polynomial
// paths (~45 instructions per element when lanes diverge, which they do): in a GEMM epilogue that is not overlapped with
// matrix work that was 10 us of a 256 x 256 tile (measured on the FFN GELU GEMM of the encoders).
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
  const float r = __builtin_fmaf(-(p * t), e, 1.0f);
  return __builtin_copysignf(r, x);
}
// (three roundings, no contraction: the GEMM epilogue has more than one code path for the same element -- gemm_common.h's
// fast path for interior tiles -- and the compiler must not fuse differently in each)
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)
  const float h = 0.5f * x;
  const float c = 1.0f + erf_fast(x * 0.70710678118654752440f);
  return h * c;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
#pragma clang fp contract(off)
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return __builtin_fmaf(x, pdf, cdf);
}

// exp(x) as one multiply + v_exp_f32 (2^x, ~1 ulp): relative error <= 1e-7 + 6e-8 * |x| * log2(e) -- 1e-6 at x = -10, far
// inside the fp32-parity budget of a softmax.  libm's expf is ~12 instructions per element; in the attention kernels the
// softmax's vector work, not the matrix work, was the critical path (measured: 17 % MFMA busy).
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// Counter-based dropout mask: keep(idx) is a pure function of (seed, site, flat element index).
// The reference uses torch's Philox stream (nn.Dropout in finetune/xit.py:34,40,108), which cannot be
// reproduced across backends; oracle/lr2ppo_oracle.py::dropout_keep_mask restates THIS function.
//
// Round 3: one 32-bit hash serves TWO consecutive elements (16 bits each): h = lowbias32((idx >> 1) ^ k32), element idx keeps
// iff the 16-bit field (idx & 1) of h is >= thr16 = floor(p * 65536).  The 64-bit splitmix of rounds 1-2 cost ~45 VALU slots per
// element (two 64-bit multiplies at quarter rate); in the encoders' attention kernels -- one mask bit per probability, 2.4e8 per
// layer, recomputed in the forward and in both backward kernels -- that was 35-40 % of the kernel time (rocprofv3: forward 440 us
// in eval mode, 711 us in train mode).  Now ~10 slots per element where a lane owns aligned groups of 4 (dropout_keep4).
// The keep probability is 1 - thr16 / 65536 against the scale 1 / (1 - p): a relative bias below 1.7e-5.  Element indices are
// taken modulo 2^33 (every masked tensor of this library is far smaller).
__device__ __forceinline__ uint64_t dropout_key(uint64_t seed, uint32_t site) {
  return (((uint64_t)site) << 40) ^ (seed * 0x9E3779B97F4A7C15ull);
}
__device__ __forceinline__ uint32_t dropout_hash(uint64_t key, uint64_t pair) {
  uint32_t x = (uint32_t)pair ^ ((uint32_t)key ^ (uint32_t)(key >> 32));
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool dropout_keep(uint64_t key, uint64_t idx, uint32_t thr16) {
  const uint32_t h = dropout_hash(key, idx >> 1);
  return ((idx & 1) ? (h >> 16) : (h & 0xffffu)) >= thr16;
}
// the 4 elements idx4 .. idx4 + 3 (idx4 a multiple of 4): two hashes
__device__ __forceinline__ void dropout_keep4(uint64_t key, uint64_t idx4, uint32_t thr16, bool (&keep)[4]) {
  const uint32_t h0 = dropout_hash(key, idx4 >> 1), h1 = dropout_hash(key, (idx4 >> 1) + 1);
  keep[0] = (h0 & 0xffffu) >= thr16;
  keep[1] = (h0 >> 16) >= thr16;
  keep[2] = (h1 & 0xffffu) >= thr16;
  keep[3] = (h1 >> 16) >= thr16;
}
__device__ __forceinline__ float4 dropout_apply4(uint64_t key, uint64_t idx4, uint32_t thr16, float scale, float4 v) {
#pragma clang fp contract(off)
  bool k[4];
  dropout_keep4(key, idx4, thr16, k);
  return make_float4(k[0] ? v.x * scale : 0.f, k[1] ? v.y * scale : 0.f, k[2] ? v.z * scale : 0.f, k[3] ? v.w * scale : 0.f);
}
static inline uint32_t dropout_threshold(float p) {
  double t = (double)p * 65536.0;
  if (t > 65535.0) t = 65535.0;
  if (t < 0) t = 0;
  return (uint32_t)t;
}

// max over aligned groups of 4 / 8 consecutive lanes, result in every lane of the group, on the VALU's DPP path (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror).  __shfl_xor compiles to ds_bpermute_b32: an LDS round trip (~130 cycles) and a dozen index
// instructions per step -- three per 4-column group in the MX-FP8 epilogues made them 18 us of a 35-us K = 1024 tile (round 4).
__device__ __forceinline__ float group4_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  return v;
}
__device__ __forceinline__ float group8_max(float v) {
  v = group4_max(v);
  return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// fp32 x 4 -> packed bf16 hi and lo pairs (x = hi + lo, hi = bf16(x), lo = bf16(x - hi)): the operand format of the
// split-bf16 GEMM.  Element order in memory = argument order.
// (no contraction: lo must be the residual of the ROUNDED fp32 value -- a multiply in front of the call fused into the subtraction
// would give planes that differ from the fp32 tensor stored beside them, and differently in each code path)
__device__ __forceinline__ void split4(float4 v, u32x2_t& hv, u32x2_t& lv) {
#pragma clang fp contract(off)
  const uint32_t h01 = cvt_pk_bf16(v.x, v.y), h23 = cvt_pk_bf16(v.z, v.w);
  const uint32_t l01 = cvt_pk_bf16(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
  const uint32_t l23 = cvt_pk_bf16(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
  hv = u32x2_t{h01, h23};
  lv = u32x2_t{l01, l23};
}
// store 4 consecutive elements of a planes tensor (hi plane at p, lo plane lo_off elements later)
// four fp32 -> four bf16 (round to nearest even), one 8-byte store: a SINGLE bf16 plane (the MX-FP8 mode's attention operands)
__device__ __forceinline__ void store_bf16x4(bf16_t* p, float4 v) {
  *reinterpret_cast<u32x2_t*>(p) = u32x2_t{cvt_pk_bf16(v.x, v.y), cvt_pk_bf16(v.z, v.w)};
}
__device__ __forceinline__ void store_planes4(bf16_t* p, size_t lo_off, float4 v) {
  u32x2_t hv, lv;
  split4(v, hv, lv);
  *reinterpret_cast<u32x2_t*>(p) = hv;
  *reinterpret_cast<u32x2_t*>(p + lo_off) = lv;
}

// ---------------------------------------------------------------------------------------------
// Fused GEMM epilogue description (shared by the MFMA kernel and the split-K reducer); see gemm.hip::epilogue_vec4.
// ---------------------------------------------------------------------------------------------
struct Epilogue {
  const float* bias;
  const float* resid;
  const float* aux_z;
  float* out;
  float* out_z;
  bf16_t* out_hi;        // planes output (hi plane); lo plane at out_hi + lo_off
  size_t lo_off;
  int ld_planes;
  int ld_resid, ld_aux, ld_out, ld_z;
  // one 32-bit word for the four switches (the kernels run at 100+ SGPRs: every scalar the epilogue keeps live counts)
  uint32_t act : 4;          // 0 none, 1 GELU(erf), 2 multiply by GELU'(aux_z)
  uint32_t accumulate : 1;
  uint32_t stream_nt : 1;    // optimizer state p / m / v: non-temporal loads and stores (each byte is touched once per launch)
  uint32_t seed_dev : 1;     // dev_scalar is the device-resident dropout seed (graph capture, see dropout_key_of)
  uint32_t lr_dev : 1;       // dev_scalar is the device-resident learning rate of the fused AdamW step
  uint32_t store_nt : 1;     // result stores (out / out_z / planes) non-temporal: large outputs that the next kernel streams from HBM anyway
  float alpha;
  float drop_scale;      // 1/(1-p) or 0 when dropout is off
  uint32_t drop_thr;     // bits 0-15: threshold; bits 16-31: the site id when the seed is device-resident
  uint64_t drop_key;     // the mask key -- or, with drop_seed_dev set, the by-value part of the SEED (the key is formed in the kernel)
  float* adam_p;         // fused AdamW step on the result (weight-gradient GEMMs); NULL = store the result
  float* adam_m;
  float* adam_v;
  float adam_lr, adam_b1, adam_b2, adam_ob1, adam_ob2, adam_eps, adam_wd;
  float* colsum_partial; // gemm256 TN: [splits * tiles_n][M] partial column sums of A (bias gradient), or NULL
  const void* dev_scalar;  // ONE pointer for the two device-resident scalars (a launch has dropout or the fused optimizer step, never
                           // both): seed_dev -> uint64 seed (seed = *dev_scalar + drop_key, site = drop_thr >> 16); lr_dev -> float rate
};
// the mask key of an epilogue / kernel: precomputed on the host, or formed here when the seed lives in device memory
// The device seed is read on the SCALAR path (s_load: one SGPR pair for the wave, the 64-bit multiply of the key on the SALU).  A
// plain `*seed_dev` between the epilogue's global stores is compiled to a vector load per use (the compiler cannot prove the
// word is not clobbered) and the key to VALU work; the same read of the learning rate cost the 8-wave kernel 28 spilled VGPRs.
// Not volatile: identical reads merge.
__device__ __forceinline__ uint64_t scalar_load_u64(const uint64_t* p) {
  uint64_t v;
  asm("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
  return v;
}
__device__ __forceinline__ float scalar_load_f32(const float* p) {
  float v;
  asm("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p));
  return v;
}
__device__ __forceinline__ uint64_t dropout_key_of(const Epilogue& e) {
  return e.seed_dev ? dropout_key(e.drop_key + scalar_load_u64((const uint64_t*)e.dev_scalar), e.drop_thr >> 16) : e.drop_key;
}

// One AdamW element update, TencentPretrain semantics (correct_bias=False; eps outside the sqrt; decay after the
// update, on the updated weight).  Shared by adamw_kernel and the GEMM's fused epilogue so both give the same bits.
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr, float b1, float b2,
                                            float ob1, float ob2, float eps, float wd) {
  // every rounding pinned (no compiler-chosen contraction): the two call sites must agree bit for bit
#pragma clang fp contract(off)
  m = __builtin_fmaf(g, ob1, m * b1);
  v = __builtin_fmaf(g * g, ob2, v * b2);
  const float upd = m / (sqrtf(v) + eps);
  p = __builtin_fmaf(-lr, upd, p);
  if (wd > 0.f) p = __builtin_fmaf(p, -(lr * wd), p);
}

// hipGetLastError() reports the calling thread's most recent error from ANY runtime call, including benign ones
// made by the host framework (hipErrorNotReady from event queries).  Clear the slot before a launch so that the
// check after it reports this launch only.
#define LR2_LAUNCH(...)        \
  do {                         \
    (void)hipGetLastError();   \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

// 0 when the preceding launch was accepted, LR2_ERR_LAUNCH (-3) otherwise (the HIP error string goes to stderr).
static inline int lr2_launch_status(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  fprintf(stderr, "lr2ppo_hip: %s: %s\n", what, hipGetErrorString(e));
  return -3;
}
// Raise a kernel's dynamic-LDS limit.  hipFuncGetAttributes first: it forces the lazily loaded code object in,
// without which hipFuncSetAttribute can fail when this is the first kernel the process touches.
template <typename K>
static inline int lr2_allow_dynamic_lds(K kern, size_t bytes, const char* what) {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void*)kern);
  const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  (void)hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "lr2ppo_hip: %s: hipFuncSetAttribute(%zu B LDS): %s\n", what, bytes, hipGetErrorString(e));
    return -3;
  }
  return 0;
}
