// Encoder self-attention for the MX-FP8 mode (BASELINE.json configs[4] "fp8 MFMA"; FeatureExtractor(precision="mxfp8")): Q, K, V as ONE
// bf16 plane each (what lr2_gemm_mxfp8 writes with out_lo_off = 0), single-pass bf16 products on the matrix cores, fp32 softmax, and
// the result handed on as MX-FP8 (e4m3fn bytes + one E8M0 scale per 32 columns of a row): the A operand of the output projection,
// with no fp32 round trip and no separate quantise pass.  NOT the parity path -- selfattn.hip's split-bf16 (3-pass) kernels stay the
// default everywhere; here an operand keeps 8 mantissa bits, far more than the e4m3 elements around it.
//
//   replaces: MultiHeadedAttention's scores / softmax / context (tencentpretrain/layers/multi_headed_attn.py:60-74) in inference,
//   key mask -10000 * (seg <= 0) added after the 1 / sqrt(64) scale as upstream; no dropout (inference mode).
//
// Structure (selfattn.hip's one-block forward, with half the bytes and a third of the matrix work): one workgroup of 8 waves per
// (sequence, head); K and V of the head -- L <= 288 keys, one plane each -- resident in LDS (K rows XOR-swizzled for ds_read_b128
// fragments, V for ds_read_b64_tr_b16); each wave walks over 16-query sub-tiles: S^T = K Q^T (2 MFMAs per 16 keys), softmax in the
// log2 domain across the 4 lanes that share a query, P as bf16 straight from the accumulators, O = P V (4 MFMAs per 32 keys),
// 1 / sum on the 16 outputs, rows through the wave's LDS slab -> fp32 and / or MX-FP8.
#include <stdlib.h>

#include "common.h"
#include "lr2ppo_hip.h"

namespace {

constexpr int HD = 64;
constexpr int ROW_B = HD * 2;
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int k_off(int r, int u) { return r * ROW_B + ((u ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int r, int u) { return r * ROW_B + ((u ^ (((r >> 1) & 3) << 1)) << 4); }

__device__ __forceinline__ bf16x8_t tr_pair(const char* plane, int row_a, int row_b, int u, int half8) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(plane + v_off(row_a, u) + half8));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(plane + v_off(row_b, u) + half8));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void self_attn_bf16_mx_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                    const bf16_t* __restrict__ V, int ld,
                                                                    const int64_t* __restrict__ seg, float* __restrict__ Of,
                                                                    uint8_t* __restrict__ Oq, uint8_t* __restrict__ Os, int ld_o,
                                                                    int heads, int L, float scale) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + PLANE;
  float* sMask = reinterpret_cast<float*>(smem + 2 * PLANE);      // [LP], pre-multiplied by log2(e)
  float* sOut = sMask + LP;                                      // [NW waves][16][HD + 4]
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;
  const int qn = lane & 15, g = lane >> 4;
  const int n_sub = (L + 15) >> 4;

  auto load_q = [&](int sub_, bf16x8_t (&f)[2]) {
    const int q_row_ = sub_ * 16 + qn;
    const bool ok = sub_ < n_sub && q_row_ < L;
    const size_t o = (row0 + (ok ? q_row_ : 0)) * (size_t)ld + col0 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t a = {0, 0, 0, 0};
      if (ok) a = *reinterpret_cast<const u32x4_t*>(Q + o + 32 * ks);
      f[ks] = __builtin_bit_cast(bf16x8_t, a);
    }
  };
  const int sub_first = blockIdx.x * NW + wave, sub_step = gridDim.x * NW;
  bf16x8_t q[2], q_next[2];
  load_q(sub_first, q_next);

  // ---- stage K, V and the key mask: every request first, then the LDS writes ----
  {
    constexpr int TRIPS = (LP * 8 + 64 * NW - 1) / (64 * NW);
    u32x4_t kk[TRIPS], vv[TRIPS];
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = tid + it * 64 * NW;
      const int r = i >> 3, u = i & 7;
      kk[it] = u32x4_t{0, 0, 0, 0};
      vv[it] = kk[it];
      if (i < LP * 8 && r < L) {
        const size_t o = (row0 + r) * (size_t)ld + col0 + u * 8;
        kk[it] = *reinterpret_cast<const u32x4_t*>(K + o);
        vv[it] = *reinterpret_cast<const u32x4_t*>(V + o);
      }
    }
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = tid + it * 64 * NW;
      const int r = i >> 3, u = i & 7;
      if (i < LP * 8) {
        *reinterpret_cast<u32x4_t*>(sK + k_off(r, u)) = kk[it];
        *reinterpret_cast<u32x4_t*>(sV + v_off(r, u)) = vv[it];
      }
    }
  }
  for (int j = tid; j < LP; j += 64 * NW) sMask[j] = j < L ? ((seg[row0 + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
  __syncthreads();

  float* slab = sOut + wave * 16 * (HD + 4);
  const float scale2 = scale * LOG2E;
  for (int sub = sub_first; sub < n_sub; sub += sub_step) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) q[ks] = q_next[ks];
    load_q(sub + sub_step, q_next);
    // ---- S^T tiles: s[t][r] = S[query qn][key 16 t + 4 g + r] ----
    f32x4_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sK + k_off(16 * t + qn, g + 4 * ks));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, q[ks], acc, 0, 0, 0);
      }
      s[t] = acc;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4 mk = *reinterpret_cast<const float4*>(sMask + 16 * t + 4 * g);
      s[t][0] = __builtin_fmaf(s[t][0], scale2, mk.x);
      s[t][1] = __builtin_fmaf(s[t][1], scale2, mk.y);
      s[t][2] = __builtin_fmaf(s[t][2], scale2, mk.z);
      s[t][3] = __builtin_fmaf(s[t][3], scale2, mk.w);
      mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - mx);
        sum += s[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    // ---- O = (P~ V) / sum: P~ fragments straight from the accumulators (the contraction index is permuted the same way on both
    // operands: lane (tq, tp) supplies V rows base + tq of a 4-row group, as in selfattn.hip) ----
    f32x4_t o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
      const u32x4_t pw = {cvt_pk_bf16(s[2 * u][0], s[2 * u][1]), cvt_pk_bf16(s[2 * u][2], s[2 * u][3]),
                          cvt_pk_bf16(s[2 * u + 1][0], s[2 * u + 1][1]), cvt_pk_bf16(s[2 * u + 1][2], s[2 * u + 1][3])};
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
      const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bf16x8_t vf = tr_pair(sV, ra, rb, 2 * n + (tp >> 1), 8 * (tp & 1));
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[n], 0, 0, 0);
      }
    }
    // ---- o[n][r] = O[query 4 g + r][hd 16 n + (l & 15)] -> slab -> row-contiguous fp32 / MX-FP8 ----
    float inv_q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) inv_q[r] = __shfl(inv, 4 * g + r, 64);
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * (HD + 4) + 16 * n + qn] = o[n][r] * inv_q[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 4 + (lane >> 4), c = (lane & 15) * 4;
      const int qr = sub * 16 + r;
      const bool ok = qr < L;
      const float4 v = *reinterpret_cast<const float4*>(slab + r * (HD + 4) + c);
      const size_t row = row0 + (ok ? qr : 0);
      if (Of && ok) *reinterpret_cast<float4*>(Of + row * (size_t)ld_o + col0 + c) = v;
      if (Oq) {
        // the row's 32-column MX block = 8 consecutive lanes x 4 columns (the head's 64 columns are two blocks): as lr2_quant_mxfp8
        float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        amax = group8_max(amax);
        int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - 8;
        if (amax < 1.17549435e-38f) e = -127;
        if (e < -127) e = -127;
        if (e > 127) e = 127;
        const uint32_t ef = (uint32_t)(127 - e);
        const float sc = __uint_as_float(ef ? ef << 23 : 0x00400000u);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.x * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.y * sc, -448.f, 448.f), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.z * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.w * sc, -448.f, 448.f), w, true);
        if (ok) {
          *reinterpret_cast<int*>(Oq + row * (size_t)ld_o + col0 + c) = w;
          if ((lane & 7) == 0) Os[row * (size_t)(ld_o / 32) + ((col0 + c) >> 5)] = (uint8_t)(e + 127);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- persistent form (round 4): one workgroup per CU walks over the (sequence, head) pairs ----
// selfattn.hip's persistent forward with one plane per operand: a pair is phase A (S = Q K^T + softmax of ALL the wave's sub-tiles: K,
// mask; the probabilities stay in registers as bf16 fragments) and phase B (O = P V: V); the two mover waves load V of the pair during
// A and K + mask of the next pair during B by LDS-DMA and are the only ones that wait for memory.  12 waves (<= 168 VGPRs): compute
// wave w owns sub-tiles w and w + 10 (L <= 288: 18 sub-tiles at most).  Same arithmetic, same bits as the one-pair kernel above.
constexpr int PM_WAVES = 12, PM_MOVERS = 2, PM_COMPUTE = PM_WAVES - PM_MOVERS;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mx_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  void* q = (void*)(((uint64_t)hi << 32) | (uint64_t)lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// rows 8j .. 8j + 7 (j = j0, j0 + jstep, ...) of one head's K or V plane: LDS-DMA writes lane-linearly, the swizzle goes onto the source
template <int NT, bool IS_V>
__device__ __forceinline__ void dma_plane_rows(const __amdgpu_buffer_rsrc_t& src, char* dst, int lane, int j0, int jstep,
                                               uint32_t pair_off, uint32_t row_bytes, int L) {
  constexpr int LP = 16 * NT;
  const int rl = lane >> 3, sl = lane & 7;
  for (int j = j0; j < LP / 8; j += jstep) {
    const int r = 8 * j + rl;
    const int u = IS_V ? (sl ^ (((r >> 1) & 3) << 1)) : (sl ^ ((r >> 1) & 7));
    const uint32_t v = r < L ? pair_off + (uint32_t)r * row_bytes + (uint32_t)u * 16u : 0xFFFFFF00u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, LDS_PTR(dst + j * 1024), 16, v, 0, 0, 0);
  }
}
__device__ __forceinline__ void mx_phase_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int mx_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ uint32_t mx_lds_addr(const void* p) {
  uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
  asm volatile("" : "+v"(a));
  return a;
}

// phase A of one 16-query sub-tile: S^T = K Q^T, softmax in the log2 domain -> un-normalised probabilities as bf16 fragments + 1 / sum
template <int NT>
__device__ __forceinline__ void mx_phase_a(const char* sK, const float* sMask, const bf16x8_t (&q)[2], int lane, float scale2,
                                           bf16x8_t (&pf)[NT / 2], float& inv) {
  const int qn = lane & 15, g = lane >> 4;
  const uint32_t kb[2] = {mx_lds_addr(sK + k_off(qn, g)), mx_lds_addr(sK + k_off(qn, g + 4))};
  f32x4_t s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t kf = *(__attribute__((address_space(3))) const bf16x8_t*)(uintptr_t)(kb[ks] + 2048 * t);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, q[ks], acc, 0, 0, 0);
    }
    s[t] = acc;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 mk = *reinterpret_cast<const float4*>(sMask + 16 * t + 4 * g);
    s[t][0] = __builtin_fmaf(s[t][0], scale2, mk.x);
    s[t][1] = __builtin_fmaf(s[t][1], scale2, mk.y);
    s[t][2] = __builtin_fmaf(s[t][2], scale2, mk.z);
    s[t][3] = __builtin_fmaf(s[t][3], scale2, mk.w);
    mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - mx);
      sum += s[t][r];
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  inv = 1.0f / sum;
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    const u32x4_t pw = {cvt_pk_bf16(s[2 * u][0], s[2 * u][1]), cvt_pk_bf16(s[2 * u][2], s[2 * u][3]),
                        cvt_pk_bf16(s[2 * u + 1][0], s[2 * u + 1][1]), cvt_pk_bf16(s[2 * u + 1][2], s[2 * u + 1][3])};
    pf[u] = __builtin_bit_cast(bf16x8_t, pw);
  }
}

// phase B: O = (P~ V) / sum, rows through the wave's slab -> fp32 and / or MX-FP8 (the one-pair kernel's code)
template <int NT>
__device__ __forceinline__ void mx_phase_b(const char* sV, float* slab, const bf16x8_t (&pf)[NT / 2], float inv, int sub, int lane, int L,
                                           size_t row0, int col0, float* __restrict__ Of, uint8_t* __restrict__ Oq,
                                           uint8_t* __restrict__ Os, int ld_o) {
  const int qn = lane & 15, g = lane >> 4;
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  uint32_t vb[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) vb[n] = mx_lds_addr(sV + v_off(4 * g + tq, 2 * n + (tp >> 1)) + 8 * (tp & 1));
  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const uint32_t a = vb[n] + 4096 * u;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)a);
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)(a + 2048));
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      const s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[u], __builtin_bit_cast(bf16x8_t, vv), o[n], 0, 0, 0);
    }
  }
  float inv_q[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inv_q[r] = __shfl(inv, 4 * g + r, 64);
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(4 * g + r) * (HD + 4) + 16 * n + qn] = o[n][r] * inv_q[r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 4 + (lane >> 4), c = (lane & 15) * 4;
    const int qr = sub * 16 + r;
    const bool ok = qr < L;
    const float4 v = *reinterpret_cast<const float4*>(slab + r * (HD + 4) + c);
    // wave-uniform 64-bit bases + 32-bit lane offsets: the stores take the scalar-base form
    const uint32_t lrow = (uint32_t)(ok ? qr : 0);
    const size_t ubase = row0 * (size_t)ld_o + col0, sbase = row0 * (size_t)(ld_o / 32) + (col0 >> 5);
    if (Of && ok) *reinterpret_cast<float4*>(Of + ubase + (lrow * (uint32_t)ld_o + (uint32_t)c)) = v;
    if (Oq) {
      float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
      amax = group8_max(amax);
      int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - 8;
      if (amax < 1.17549435e-38f) e = -127;
      if (e < -127) e = -127;
      if (e > 127) e = 127;
      const uint32_t ef = (uint32_t)(127 - e);
      const float sc = __uint_as_float(ef ? ef << 23 : 0x00400000u);
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.x * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.y * sc, -448.f, 448.f), w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.z * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.w * sc, -448.f, 448.f), w, true);
      if (ok) {
        *reinterpret_cast<int*>(Oq + ubase + (lrow * (uint32_t)ld_o + (uint32_t)c)) = w;
        if ((lane & 7) == 0) Os[sbase + (lrow * (uint32_t)(ld_o / 32) + (uint32_t)(c >> 5))] = (uint8_t)(e + 127);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int NT>
__global__ __launch_bounds__(64 * PM_WAVES) void self_attn_bf16_mx_persist_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                                  const bf16_t* __restrict__ V, int ld,
                                                                                  const int64_t* __restrict__ seg, float* __restrict__ Of,
                                                                                  uint8_t* __restrict__ Oq, uint8_t* __restrict__ Os,
                                                                                  int ld_o, int heads, int L, float scale, int n_pairs,
                                                                                  uint32_t kv_bytes) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  constexpr int MK = (LP + 64 * PM_MOVERS - 1) / (64 * PM_MOVERS);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + PLANE;
  float* sMask = reinterpret_cast<float*>(smem + 2 * PLANE);      // [2][LP]: pair number it reads half it & 1
  float* sOut = sMask + 2 * LP;                                  // [PM_COMPUTE waves][16][HD + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_sub = (L + 15) >> 4;
  const uint32_t row_bytes = (uint32_t)ld * 2u;
  int p = blockIdx.x;
  if (p >= n_pairs) return;
  int b = __builtin_amdgcn_readfirstlane(p / heads), h = p - b * heads;      // (the quotient comes out of vector instructions)
  size_t row0 = (size_t)b * L;
  int col0 = h * HD;

  if (wave >= PM_COMPUTE) {
    // ---- movers ----
    const __amdgpu_buffer_rsrc_t k_src = mx_rsrc(K, kv_bytes), v_src = mx_rsrc(V, kv_bytes);
    const int j0 = wave - PM_COMPUTE, mtid = tid - 64 * PM_COMPUTE;
    dma_plane_rows<NT, false>(k_src, sK, lane, j0, PM_MOVERS, (uint32_t)((row0 * ld + col0) * 2), row_bytes, L);
    for (int j = mtid; j < LP; j += 64 * PM_MOVERS) sMask[j] = j < L ? ((seg[row0 + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mx_phase_barrier();
    for (int it = 0;; ++it) {
      dma_plane_rows<NT, true>(v_src, sV, lane, j0, PM_MOVERS, (uint32_t)((row0 * ld + col0) * 2), row_bytes, L);
      const int pn = p + gridDim.x;
      const bool more = pn < n_pairs;
      const int bn = __builtin_amdgcn_readfirstlane(pn / heads), hn = pn - bn * heads;
      const size_t row0n = (size_t)bn * L;
      float mk[MK];
      if (more) {
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PM_MOVERS;
          mk[i] = j < L ? ((seg[row0n + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // V of this pair has landed
      mx_phase_barrier();
      if (!more) break;
      dma_plane_rows<NT, false>(k_src, sK, lane, j0, PM_MOVERS, (uint32_t)((row0n * ld + hn * HD) * 2), row_bytes, L);
      float* mnext = sMask + ((it + 1) & 1) * LP;
#pragma unroll
      for (int i = 0; i < MK; ++i) {
        const int j = mtid + i * 64 * PM_MOVERS;
        if (j < LP) mnext[j] = mk[i];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // K of the next pair has landed
      mx_phase_barrier();
      p = pn; row0 = row0n; col0 = hn * HD;
    }
    return;
  }

  // ---- compute waves: sub-tiles `wave` and `wave + PM_COMPUTE` of every pair ----
  const int sub0 = wave, sub1 = wave + PM_COMPUTE;
  const bool has0 = sub0 < n_sub, has1 = sub1 < n_sub;
  float* slab = sOut + wave * 16 * (HD + 4);
  const float scale2 = scale * LOG2E;
  auto load_q = [&](size_t row0_, int col0_, int sub_, bf16x8_t (&f)[2]) {
    const int lane_ = mx_opaque(lane);
    const int q_row_ = sub_ * 16 + (lane_ & 15);
    const bool ok = sub_ < n_sub && q_row_ < L;
    const bf16_t* ub = Q + row0_ * (size_t)ld + col0_;
    const uint32_t o = (uint32_t)(ok ? q_row_ : 0) * (uint32_t)ld + 8u * (uint32_t)(lane_ >> 4);
    // unconditional loads (rows past the end read row 0 and are zeroed by a select): a branch here drags the uniform address
    // arithmetic into the divergent block, i.e. onto vector registers
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const u32x4_t a = *reinterpret_cast<const u32x4_t*>(ub + o + 32 * ks);
      f[ks] = __builtin_bit_cast(bf16x8_t, (u32x4_t{ok ? a[0] : 0u, ok ? a[1] : 0u, ok ? a[2] : 0u, ok ? a[3] : 0u}));
    }
  };
  bf16x8_t q0[2];
  load_q(row0, col0, sub0, q0);
  mx_phase_barrier();
  for (int it = 0;; ++it) {
    const float* mask = sMask + (it & 1) * LP;
    bf16x8_t p0[NT / 2], p1[NT / 2];
    float inv0 = 0.f, inv1 = 0.f;
    bf16x8_t q1[2];
    load_q(row0, col0, sub1, q1);                  // travels under the first sub-tile's phase A
    if (has0) mx_phase_a<NT>(sK, mask, q0, mx_opaque(lane), scale2, p0, inv0);
    if (has1) mx_phase_a<NT>(sK, mask, q1, mx_opaque(lane), scale2, p1, inv1);
    mx_phase_barrier();
    const int pn = p + gridDim.x;
    const bool more = pn < n_pairs;
    const int bn = __builtin_amdgcn_readfirstlane(pn / heads), hn = pn - bn * heads;
    const size_t row0n = (size_t)bn * L;
    if (has0) mx_phase_b<NT>(sV, slab, p0, inv0, sub0, mx_opaque(lane), L, row0, col0, Of, Oq, Os, ld_o);
    if (more) load_q(row0n, hn * HD, sub0, q0);    // the next pair's first sub-tile: under the second sub-tile's P V
    if (has1) mx_phase_b<NT>(sV, slab, p1, inv1, sub1, mx_opaque(lane), L, row0, col0, Of, Oq, Os, ld_o);
    if (!more) break;
    mx_phase_barrier();
    p = pn; row0 = row0n; col0 = hn * HD;
  }
}

static int mx_cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

template <int NT>
int launch_persist(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, const int64_t* seg, float* of, uint8_t* oq, uint8_t* os,
                   int ld_o, int batch, int heads, int L, float scale, uint32_t kv_bytes, hipStream_t stream) {
  constexpr int LP = 16 * NT;
  const size_t lds = (size_t)2 * LP * ROW_B + (size_t)2 * LP * 4 + (size_t)PM_COMPUTE * 16 * (HD + 4) * 4;
  static bool done = false;
  if (!done) {
    if (lr2_allow_dynamic_lds(self_attn_bf16_mx_persist_kernel<NT>, lds, "self_attn_fwd_bf16(persistent)")) return LR2_ERR_LAUNCH;
    done = true;
  }
  const int n_pairs = batch * heads;
  const int grid = n_pairs < mx_cu_count() ? n_pairs : mx_cu_count();
  LR2_LAUNCH((self_attn_bf16_mx_persist_kernel<NT>), dim3(grid), dim3(64 * PM_WAVES), lds, stream, q, k, v, ld, seg, of, oq, os, ld_o,
             heads, L, scale, n_pairs, kv_bytes);
  return lr2_launch_status("lr2_self_attn_fwd_bf16(persistent)");
}

template <int NT>
int launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, const int64_t* seg, float* of, uint8_t* oq, uint8_t* os, int ld_o,
           int batch, int heads, int L, float scale, hipStream_t stream) {
  {
    // the persistent form: at least one pair per CU, 32-bit byte offsets (LR2_ATTN_PERSIST=0: the A/B switch of selfattn.hip)
    static const bool on = !(getenv("LR2_ATTN_PERSIST") && atoi(getenv("LR2_ATTN_PERSIST")) == 0);
    const uint64_t span = ((uint64_t)batch * L - 1) * (uint64_t)ld * 2u + (uint64_t)heads * HD * 2u;
    if (on && batch * heads >= mx_cu_count() && (L + 15) / 16 <= 2 * PM_COMPUTE && span < 0xFFFFFF00ull)
      return launch_persist<NT>(q, k, v, ld, seg, of, oq, os, ld_o, batch, heads, L, scale, (uint32_t)span, stream);
  }
  constexpr int LP = 16 * NT, NW = 8;
  const size_t lds = (size_t)2 * LP * ROW_B + (size_t)LP * 4 + (size_t)NW * 16 * (HD + 4) * 4;
  static bool done = false;
  if (!done) {
    if (lr2_allow_dynamic_lds(self_attn_bf16_mx_kernel<NT, NW>, lds, "self_attn_fwd_bf16")) return LR2_ERR_LAUNCH;
    done = true;
  }
  LR2_LAUNCH((self_attn_bf16_mx_kernel<NT, NW>), dim3(1, heads, batch), dim3(64 * NW), lds, stream, q, k, v, ld, seg, of, oq, os, ld_o,
             heads, L, scale);
  return lr2_launch_status("lr2_self_attn_fwd_bf16");
}

}  // namespace

extern "C" int lr2_self_attn_fwd_bf16(const void* q, const void* k, const void* v, int ld, const int64_t* seg, void* o_f32, void* o_q,
                                      void* o_scales, int ld_o, int batch, int heads, int L, int head_dim, float scale, void* stream) {
  if (!q || !k || !v || !seg || (!o_f32 && !o_q) || batch <= 0 || heads <= 0 || L <= 0) return LR2_ERR_ARG;
  if ((o_q != nullptr) != (o_scales != nullptr)) return LR2_ERR_ARG;
  if (head_dim != HD || L > 288 || (ld % 8) || ld_o < heads * HD || (ld_o % 32)) return LR2_ERR_SHAPE;
  const bf16_t *qq = (const bf16_t*)q, *kk = (const bf16_t*)k, *vv = (const bf16_t*)v;
  hipStream_t s = (hipStream_t)stream;
#define GO(NT) return launch<NT>(qq, kk, vv, ld, seg, (float*)o_f32, (uint8_t*)o_q, (uint8_t*)o_scales, ld_o, batch, heads, L, scale, s)
  if (L <= 64) GO(4);
  if (L <= 128) GO(8);
  if (L <= 224) GO(14);
  GO(18);
#undef GO
}
