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

template <int NT>
int launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ld, const int64_t* seg, float* of, uint8_t* oq, uint8_t* os, int ld_o,
           int batch, int heads, int L, float scale, hipStream_t stream) {
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
