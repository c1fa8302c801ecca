// Encoder self-attention on the matrix cores (TencentPretrain MultiHeadedAttention core, head_dim 64, L <= 256).
//
//   S = Q K^T * scale + (seg[key] > 0 ? 0 : -10000);  P = softmax(S);  O = P V        (fp32 semantics)
//   replaces: tencentpretrain/layers/multi_headed_attn.py:61-74 + the mask of encoders/transformer_encoder.py:62-68
//
// Q, K, V arrive as bf16 hi/lo planes (the QKV GEMM's epilogue writes them), both products run as split-bf16 x3 on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation, the softmax is fp32.  One workgroup = one (sequence, head, 64 query
// rows); its 4 waves own 16 query rows each.  K and V of the head live in LDS as bf16 planes (4 x LP x 128 B).
//
// The score tile is computed TRANSPOSED, S^T = K Q^T: in the 16x16 accumulator layout a lane then holds, for ONE query
// (column l & 15), the keys 4*(l >> 4) + r of every 16-key tile -- exactly the shape of an MFMA A operand row.  Two
// adjacent key tiles give a lane 8 probabilities of its query: they are used directly as the A fragment of P V with the
// contraction index permuted (slot (g, j) <-> key 4g + j for j < 4, 16 + 4g + j - 4 otherwise); the V fragments are read
// with the same permutation by two ds_read_b64_tr_b16 (rows 4g .. 4g+3 and 16 + 4g .. 16 + 4g + 3 of the key block).
// P never touches LDS and no shuffle is needed between the two GEMMs.
#include "common.h"
#include "lr2ppo_hip.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int ROW_B = HD * 2;   // bytes of one K / V row in one LDS plane

// K plane: 16-B unit u of row r at u ^ ((r >> 1) & 7): conflict-free ds_read_b128 fragment reads (as in gemm.hip).
__device__ __forceinline__ int k_off(int r, int u) { return r * ROW_B + ((u ^ ((r >> 1) & 7)) << 4); }
// V plane: 32-B chunk c of row r at c ^ ((r >> 1) & 3): the 8 rows one half-wave touches in a transposed read land on
// 8 different 32-B slots of the 256-B bank row.
__device__ __forceinline__ int v_off(int r, int u) { return r * ROW_B + ((u ^ (((r >> 1) & 3) << 1)) << 4); }

__device__ __forceinline__ bf16x8_t tr_pair(const char* plane, int row_a, int row_b, int u, int half8) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(plane + v_off(row_a, u) + half8));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(plane + v_off(row_b, u) + half8));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int NT>   // NT = key tiles of 16 (even); LP = 16 * NT padded keys
__global__ __launch_bounds__(256) void self_attn_mfma_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                             const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
                                                             const int64_t* __restrict__ seg, float* __restrict__ O,
                                                             bf16_t* __restrict__ Oh, size_t o_lo_off, int ld_o, int heads,
                                                             int L, float scale) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                    // [hi | lo]
  char* sV = smem + 2 * PLANE;        // [hi | lo]
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);          // [LP]
  float* sOut = sMask + LP;                                          // [4 waves][16][HD + 4]
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;

  // ---- stage K, V (both planes) and the additive key mask ----
  for (int i = tid; i < LP * 8; i += 256) {
    const int r = i >> 3, u = i & 7;
    u32x4_t kh = {0, 0, 0, 0}, kl = kh, vh = kh, vl = kh;
    if (r < L) {
      const size_t o = (row0 + r) * (size_t)ld + col0 + u * 8;
      kh = *reinterpret_cast<const u32x4_t*>(Kh + o);
      kl = *reinterpret_cast<const u32x4_t*>(Kh + o + lo_off);
      vh = *reinterpret_cast<const u32x4_t*>(Vh + o);
      vl = *reinterpret_cast<const u32x4_t*>(Vh + o + lo_off);
    }
    *reinterpret_cast<u32x4_t*>(sK + k_off(r, u)) = kh;
    *reinterpret_cast<u32x4_t*>(sK + PLANE + k_off(r, u)) = kl;
    *reinterpret_cast<u32x4_t*>(sV + v_off(r, u)) = vh;
    *reinterpret_cast<u32x4_t*>(sV + PLANE + v_off(r, u)) = vl;
  }
  for (int j = tid; j < LP; j += 256) sMask[j] = j < L ? ((seg[row0 + j] > 0) ? 0.f : -10000.0f) : -INFINITY;

  // ---- this wave's 16 query rows as B fragments of S^T = K Q^T (lane: query l & 15, hd 8*(l >> 4) + 32*ks ..) ----
  const int qn = lane & 15, g = lane >> 4;
  const int q_row = qt * 64 + wave * 16 + qn;
  bf16x8_t qh[2], ql[2];
  {
    const bool ok = q_row < L;
    const size_t o = (row0 + (ok ? q_row : 0)) * (size_t)ld + col0 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t a = {0, 0, 0, 0}, c = a;
      if (ok) {
        a = *reinterpret_cast<const u32x4_t*>(Qh + o + 32 * ks);
        c = *reinterpret_cast<const u32x4_t*>(Qh + o + 32 * ks + lo_off);
      }
      qh[ks] = __builtin_bit_cast(bf16x8_t, a);
      ql[ks] = __builtin_bit_cast(bf16x8_t, c);
    }
  }
  __syncthreads();

  // ---- S^T tiles: acc[t][r] = S[query qn][key 16t + 4g + r] ----
  f32x4_t s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r = 16 * t + qn;          // A fragment: key row 16t + (l & 15), hd 8g + 32ks ..
      const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(sK + k_off(r, g + 4 * ks));
      const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(sK + PLANE + k_off(r, g + 4 * ks));
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[ks], acc, 0, 0, 0);
    }
    s[t] = acc;
  }

  // ---- softmax over the keys of query qn: in-lane over (t, r), across the 4 lanes l, l^16, l^32, l^48 ----
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 mk = *reinterpret_cast<const float4*>(sMask + 16 * t + 4 * g);
    s[t][0] = s[t][0] * scale + mk.x;
    s[t][1] = s[t][1] * scale + mk.y;
    s[t][2] = s[t][2] * scale + mk.z;
    s[t][3] = s[t][3] * scale + mk.w;
    mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] = expf(s[t][r] - mx);     // padded keys: exp(-inf) = 0
      sum += s[t][r];
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;

  // ---- O = P V over 32-key blocks; P fragments straight from the accumulators (permuted contraction index) ----
  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    float p[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = s[2 * u][r] * inv;
      p[4 + r] = s[2 * u + 1][r] * inv;
    }
    const uint32_t h01 = cvt_pk_bf16(p[0], p[1]), h23 = cvt_pk_bf16(p[2], p[3]);
    const uint32_t h45 = cvt_pk_bf16(p[4], p[5]), h67 = cvt_pk_bf16(p[6], p[7]);
    const uint32_t l01 = cvt_pk_bf16(p[0] - __uint_as_float(h01 << 16), p[1] - __uint_as_float(h01 & 0xffff0000u));
    const uint32_t l23 = cvt_pk_bf16(p[2] - __uint_as_float(h23 << 16), p[3] - __uint_as_float(h23 & 0xffff0000u));
    const uint32_t l45 = cvt_pk_bf16(p[4] - __uint_as_float(h45 << 16), p[5] - __uint_as_float(h45 & 0xffff0000u));
    const uint32_t l67 = cvt_pk_bf16(p[6] - __uint_as_float(h67 << 16), p[7] - __uint_as_float(h67 & 0xffff0000u));
    const bf16x8_t ph = __builtin_bit_cast(bf16x8_t, (u32x4_t{h01, h23, h45, h67}));
    const bf16x8_t pl = __builtin_bit_cast(bf16x8_t, (u32x4_t{l01, l23, l45, l67}));
    // transposed V reads: lane (tq, tp) of a 16-lane group supplies row base + tq, hd 16n + 4tp .. +3
    const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int unit = 2 * n + (tp >> 1), half8 = 8 * (tp & 1);
      const bf16x8_t vh = tr_pair(sV, ra, rb, unit, half8);
      const bf16x8_t vl = tr_pair(sV + PLANE, ra, rb, unit, half8);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh, o[n], 0, 0, 0);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl, o[n], 0, 0, 0);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh, o[n], 0, 0, 0);
    }
  }

  // ---- o[n][r] = O[query 4g + r][hd 16n + (l & 15)] -> LDS slab -> 16-B row-contiguous stores ----
  float* slab = sOut + wave * 16 * (HD + 4);
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(4 * g + r) * (HD + 4) + 16 * n + qn] = o[n][r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 4 + (lane >> 4), c = (lane & 15) * 4;
    const int qr = qt * 64 + wave * 16 + r;
    if (qr < L) {
      const float4 v = *reinterpret_cast<const float4*>(slab + r * (HD + 4) + c);
      const size_t off = (row0 + qr) * (size_t)ld_o + col0 + c;
      if (O) *reinterpret_cast<float4*>(O + off) = v;
      if (Oh) store_planes4(Oh + off, o_lo_off, v);
    }
  }
}

template <int NT>
int launch_self_attn(const bf16_t* q, const bf16_t* k, const bf16_t* v, size_t lo_off, int ld, const int64_t* seg, float* o,
                     bf16_t* oh, size_t o_lo_off, int ld_o, int batch, int heads, int L, float scale, hipStream_t stream) {
  constexpr int LP = 16 * NT;
  const size_t lds = (size_t)4 * LP * ROW_B + (size_t)LP * 4 + (size_t)4 * 16 * (HD + 4) * 4;
  auto kern = self_attn_mfma_kernel<NT>;
  static bool attr_set = false;
  if (!attr_set) {
    if (lr2_allow_dynamic_lds(kern, lds, "self_attn_fwd")) return LR2_ERR_LAUNCH;
    attr_set = true;
  }
  LR2_LAUNCH(kern, dim3((L + 63) / 64, heads, batch), dim3(256), lds, stream, q, k, v, lo_off, ld, seg, o, oh, o_lo_off, ld_o,
             heads, L, scale);
  return lr2_launch_status("lr2_self_attn_fwd");
}

template __global__ void self_attn_mfma_kernel<4>(const bf16_t*, const bf16_t*, const bf16_t*, size_t, int, const int64_t*,
                                                  float*, bf16_t*, size_t, int, int, int, float);
template __global__ void self_attn_mfma_kernel<8>(const bf16_t*, const bf16_t*, const bf16_t*, size_t, int, const int64_t*,
                                                  float*, bf16_t*, size_t, int, int, int, float);
template __global__ void self_attn_mfma_kernel<14>(const bf16_t*, const bf16_t*, const bf16_t*, size_t, int, const int64_t*,
                                                   float*, bf16_t*, size_t, int, int, int, float);
template __global__ void self_attn_mfma_kernel<16>(const bf16_t*, const bf16_t*, const bf16_t*, size_t, int, const int64_t*,
                                                   float*, bf16_t*, size_t, int, int, int, float);

}  // namespace

extern "C" int lr2_self_attn_fwd(const void* q_hi, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld,
                                 const int64_t* seg, void* o, void* o_hi, uint64_t o_lo_off, int ld_o, int batch, int heads,
                                 int L, int head_dim, float scale, void* stream) {
  if (!q_hi || !k_hi || !v_hi || !seg || (!o && !o_hi) || batch <= 0 || heads <= 0) return LR2_ERR_ARG;
  if (head_dim != HD || L < 1 || L > 256 || ld % 8 || ld_o % 4 || lo_off % 8) return LR2_ERR_SHAPE;
  const bf16_t *q = (const bf16_t*)q_hi, *k = (const bf16_t*)k_hi, *v = (const bf16_t*)v_hi;
  hipStream_t s = (hipStream_t)stream;
  if (L <= 64) return launch_self_attn<4>(q, k, v, lo_off, ld, seg, (float*)o, (bf16_t*)o_hi, o_lo_off, ld_o, batch, heads, L, scale, s);
  if (L <= 128) return launch_self_attn<8>(q, k, v, lo_off, ld, seg, (float*)o, (bf16_t*)o_hi, o_lo_off, ld_o, batch, heads, L, scale, s);
  if (L <= 224) return launch_self_attn<14>(q, k, v, lo_off, ld, seg, (float*)o, (bf16_t*)o_hi, o_lo_off, ld_o, batch, heads, L, scale, s);
  return launch_self_attn<16>(q, k, v, lo_off, ld, seg, (float*)o, (bf16_t*)o_hi, o_lo_off, ld_o, batch, heads, L, scale, s);
}
