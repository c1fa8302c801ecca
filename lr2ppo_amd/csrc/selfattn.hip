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
#include <stdlib.h>

#include "common.h"
#include "lr2ppo_hip.h"

namespace {

#ifndef LR2_SA_ABLATE
#define LR2_SA_ABLATE 0         // diagnostics (tools/dbg/attn_ablate.py builds variants of this file): bit 1 no K / V global loads, 2 no
#endif                          // sub-tile work, 4 no softmax arithmetic, 8 no lo split of P, 16 no P V product, 32 no S product, 64 no store,
                                // 128 no fragment loads of the persistent backward's compute waves (bits 1, 64, 128 apply to the backward)
constexpr int HD = 64;          // head dim
constexpr int ROW_B = HD * 2;   // bytes of one K / V row in one LDS plane
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// Dropout on the attention probabilities (multi_headed_attn.py:72): element (b, h, q, key) of the [B, H, L, L] tensor
// is kept iff dropout_keep(key, flat index, thr); thr == 0 switches it off.
struct DropP {
  uint64_t key;
  uint32_t thr;
  float inv_keep;   // 1 / (1 - p)
};
__device__ __forceinline__ float drop_mul(const DropP& d, uint64_t idx) {
  return dropout_keep(d.key, idx, d.thr) ? d.inv_keep : 0.0f;
}
// The mask of probability (b, h, q, key) is element ((b * heads + h) * L + q) * mask_pitch(L) + key of the mask stream: rows are
// pitched to a multiple of 4 so that a lane's 4 consecutive keys 4j .. 4j + 3 are one aligned group = two hashes (dropout_keep4).
__device__ __forceinline__ uint64_t mask_pitch(int L) { return (uint64_t)((L + 3) & ~3); }
__device__ __forceinline__ f32x4_t drop_mul4v(const DropP& d, uint64_t idx4, f32x4_t x) {
  bool k[4];
  dropout_keep4(d.key, idx4, d.thr, k);
  return f32x4_t{k[0] ? x[0] * d.inv_keep : 0.0f, k[1] ? x[1] * d.inv_keep : 0.0f, k[2] ? x[2] * d.inv_keep : 0.0f,
                 k[3] ? x[3] * d.inv_keep : 0.0f};
}
// x[0..3] *= mask / keep of the aligned group starting at idx4
__device__ __forceinline__ void drop_mul4(const DropP& d, uint64_t idx4, float& x0, float& x1, float& x2, float& x3) {
  bool k[4];
  dropout_keep4(d.key, idx4, d.thr, k);
  x0 = k[0] ? x0 * d.inv_keep : 0.0f;
  x1 = k[1] ? x1 * d.inv_keep : 0.0f;
  x2 = k[2] ? x2 * d.inv_keep : 0.0f;
  x3 = k[3] ? x3 * d.inv_keep : 0.0f;
}

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

// LDS fragment addresses as (per-lane base register) + (compile-time offset): the XOR swizzles above depend on the row only through
// bits that the tile index does not touch, so ONE base per (k-step) for K and one per head-column group for V serve every tile; the
// bases are made opaque to the optimiser (else it re-derives one address per read -- 70 live registers where 6 do).
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
  asm volatile("" : "+v"(a));
  return a;
}
// A copy of a lane-varying value the optimiser cannot see through: what is derived from it inside a loop is RE-derived every
// trip (a few integer instructions) instead of being hoisted and kept live -- or spilled -- across the whole persistent loop.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ bf16x8_t lds_ld16(uint32_t a) {
  return *(__attribute__((address_space(3))) const bf16x8_t*)(uintptr_t)a;
}
__device__ __forceinline__ bf16x8_t lds_tr_pair(uint32_t a, uint32_t b) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)a);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)b);
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// One 16-query sub-tile against the K / V planes resident in LDS, in two phases that touch different planes:
//   phase A (K, mask):  S^T = K Q^T (MFMA), fp32 softmax in the log2 domain -> the un-normalised probabilities as bf16 hi / lo
//                       fragments of the P V product + 1 / sum;
//   phase B (V):        O = (P~ V) / sum (MFMA), rows stored through the wave's LDS slab.
// qh / ql: the sub-tile's query fragments.
template <int NT, bool DROP = true>
__device__ __forceinline__ void attn_phase_a(const char* sK, const float* sMask, const bf16x8_t (&qh)[2], const bf16x8_t (&ql)[2],
                                             int sub, int lane, int L, int b, int h, int heads, float scale, float* __restrict__ lse,
                                             const DropP& dr, bf16x8_t (&ph)[NT / 2], bf16x8_t (&pl)[NT / 2], float& inv) {
  constexpr int PLANE = 16 * NT * ROW_B;
  const int qn = lane & 15, g = lane >> 4;
  const int q_row = sub * 16 + qn;
  // ---- S^T tiles: acc[t][r] = S[query qn][key 16t + 4g + r] ----
  const uint32_t kb[2] = {lds_addr(sK + k_off(qn, g)), lds_addr(sK + k_off(qn, g + 4))};
  f32x4_t s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    if (LR2_SA_ABLATE & 32) acc = f32x4_t{__builtin_bit_cast(float, (int)qh[0][0]), 0.f, (float)t, 0.f};
#pragma unroll
    for (int ks = 0; ks < ((LR2_SA_ABLATE & 32) ? 0 : 2); ++ks) {
      // A fragment: key row 16t + (l & 15), hd 8g + 32ks ..: k_off(16t + qn, g + 4ks) = 2048 t + k_off(qn, g + 4ks)
      const bf16x8_t kh = lds_ld16(kb[ks] + 2048 * t);
      const bf16x8_t kl = lds_ld16(kb[ks] + 2048 * t + PLANE);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[ks], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[ks], acc, 0, 0, 0);
    }
    s[t] = acc;
  }

  // ---- softmax over the keys of query qn: in-lane over (t, r), across the 4 lanes l, l^16, l^32, l^48 ----
  float mx = -INFINITY;
  const float scale2 = scale * LOG2E;
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
  if (!(LR2_SA_ABLATE & 4)) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] = __builtin_amdgcn_exp2f(s[t][r] - mx);     // padded keys: 2^(-inf) = 0
      sum += s[t][r];
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  } else sum = 1.0f + mx * 1e-30f;
  inv = 1.0f / sum;
  if (lse && g == 0 && q_row < L) lse[((size_t)b * heads + h) * L + q_row] = mx * LN2 + logf(sum);
  const uint64_t drow = (((uint64_t)b * heads + h) * L + (uint64_t)(q_row < L ? q_row : 0)) * mask_pitch(L);

  // ---- P~ = the un-normalised exponentials in (0, 1], as the A fragments of P V: straight from the accumulators (permuted
  // contraction index: slot (g, j) <-> key 4g + j for j < 4, 16 + 4g + j - 4 otherwise, of each 32-key block) ----
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    float p[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = s[2 * u][r];
      p[4 + r] = s[2 * u + 1][r];
    }
    if (DROP && dr.thr) {
      drop_mul4(dr, drow + 32 * u + 4 * g, p[0], p[1], p[2], p[3]);
      drop_mul4(dr, drow + 32 * u + 16 + 4 * g, p[4], p[5], p[6], p[7]);
    }
    const uint32_t h01 = cvt_pk_bf16(p[0], p[1]), h23 = cvt_pk_bf16(p[2], p[3]);
    const uint32_t h45 = cvt_pk_bf16(p[4], p[5]), h67 = cvt_pk_bf16(p[6], p[7]);
    const uint32_t l01 = cvt_pk_bf16(p[0] - __uint_as_float(h01 << 16), p[1] - __uint_as_float(h01 & 0xffff0000u));
    const uint32_t l23 = cvt_pk_bf16(p[2] - __uint_as_float(h23 << 16), p[3] - __uint_as_float(h23 & 0xffff0000u));
    const uint32_t l45 = cvt_pk_bf16(p[4] - __uint_as_float(h45 << 16), p[5] - __uint_as_float(h45 & 0xffff0000u));
    const uint32_t l67 = cvt_pk_bf16(p[6] - __uint_as_float(h67 << 16), p[7] - __uint_as_float(h67 & 0xffff0000u));
    ph[u] = __builtin_bit_cast(bf16x8_t, (u32x4_t{h01, h23, h45, h67}));
    pl[u] = (LR2_SA_ABLATE & 8) ? ph[u] : __builtin_bit_cast(bf16x8_t, (u32x4_t{l01, l23, l45, l67}));
  }
}

// HALF_SLAB: the wave's LDS slab holds 16 rows x 32 columns (the 16-wave persistent kernel: 13-14 slabs beside the K / V planes);
// the output then leaves in two halves of 32 head columns.  Same values either way.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// after_pv(): called between the last P V product and the output's way through the slab (the persistent kernel requests the next
// pair's query fragments there: the probability registers are dead by then).
template <int NT, bool HALF_SLAB = false, typename Hook = NoHook>
__device__ __forceinline__ void attn_phase_b(const char* sV, float* slab, const bf16x8_t (&ph)[NT / 2], const bf16x8_t (&pl)[NT / 2],
                                             float inv, int sub, int lane, int L, size_t row0, int col0, float* __restrict__ O,
                                             bf16_t* __restrict__ Oh, size_t o_lo_off, int ld_o, Hook after_pv = Hook()) {
  constexpr int PLANE = 16 * NT * ROW_B;
  const int qn = lane & 15, g = lane >> 4;
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  // transposed V reads: lane (tq, tp) of a 16-lane group supplies row 32u + 4g + tq (and that row + 16), hd 16n + 4tp .. +3:
  // v_off(32u + 4g + tq (+ 16), unit) = 4096 u (+ 2048) + v_off(4g + tq, unit)
  uint32_t vb[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) vb[n] = lds_addr(sV + v_off(4 * g + tq, 2 * n + (tp >> 1)) + 8 * (tp & 1));
  if constexpr (HALF_SLAB) {
    // 32 head columns at a time: 8 accumulator registers + the V fragments of two column groups beside the probabilities (a
    // 128-register wave); each half leaves through the 16 x 32 slab as soon as it is complete
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4_t o[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int u = 0; u < NT / 2; ++u) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const uint32_t a = vb[2 * half + n] + 4096 * u;
          const bf16x8_t vh = lds_tr_pair(a, a + 2048);
          const bf16x8_t vl = lds_tr_pair(a + PLANE, a + 2048 + PLANE);
          o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[u], vh, o[n], 0, 0, 0);
          o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[u], vl, o[n], 0, 0, 0);
          o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[u], vh, o[n], 0, 0, 0);
        }
      }
      if (half == 1) after_pv();
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(4 * g + r) * (32 + 4) + 16 * n + qn] = o[n][r] * __shfl(inv, 4 * g + r, 64);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int r = pass * 8 + (lane >> 3), c = (lane & 7) * 4;
        const int qr = sub * 16 + r;
        if (qr < L && (!(LR2_SA_ABLATE & 64) || slab[r] == 12345.f)) {
          const float4 v = *reinterpret_cast<const float4*>(slab + r * (32 + 4) + c);
          // uniform 64-bit base + 32-bit lane offset: the stores take the scalar-base form (no 64-bit address registers per lane)
          const size_t ubase = row0 * (size_t)ld_o + col0 + 32 * half;
          const uint32_t loff = (uint32_t)qr * (uint32_t)ld_o + (uint32_t)c;
          if (O) *reinterpret_cast<float4*>(O + ubase + loff) = v;
          if (Oh) store_planes4(Oh + ubase + loff, o_lo_off, v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
  // ---- O = (P~ V) / sum over 32-key blocks; the 1 / sum goes onto the 16 output values instead of the NT * 4 probabilities
  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    if (LR2_SA_ABLATE & 16) {
      o[u & 3][0] += __builtin_bit_cast(float, (int)ph[u][0]) + __builtin_bit_cast(float, (int)pl[u][1]);
      continue;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const bf16x8_t vh = lds_tr_pair(vb[n] + 4096 * u, vb[n] + 4096 * u + 2048);
      const bf16x8_t vl = lds_tr_pair(vb[n] + 4096 * u + PLANE, vb[n] + 4096 * u + 2048 + PLANE);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl[u], vh, o[n], 0, 0, 0);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[u], vl, o[n], 0, 0, 0);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph[u], vh, o[n], 0, 0, 0);
    }
  }

  after_pv();
  // ---- o[n][r] = O[query 4g + r][hd 16n + (l & 15)] -> LDS slab -> 16-B row-contiguous stores ----
  float inv_q[4];                         // 1 / sum of query 4g + r (lane 4g + r holds it: its own query is l & 15)
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
    if (qr < L && (!(LR2_SA_ABLATE & 64) || slab[r] == 12345.f)) {
      const float4 v = *reinterpret_cast<const float4*>(slab + r * (HD + 4) + c);
      const size_t off = (row0 + qr) * (size_t)ld_o + col0 + c;
      if (O) *reinterpret_cast<float4*>(O + off) = v;
      if (Oh) store_planes4(Oh + off, o_lo_off, v);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int NT>
__device__ __forceinline__ void attn_subtile_fwd(const char* sK, const char* sV, const float* sMask, float* slab,
                                                 const bf16x8_t (&qh)[2], const bf16x8_t (&ql)[2], int sub, int lane, int L, int b,
                                                 int h, int heads, size_t row0, int col0, float scale, float* __restrict__ lse,
                                                 const DropP& dr, float* __restrict__ O, bf16_t* __restrict__ Oh, size_t o_lo_off,
                                                 int ld_o) {
  bf16x8_t ph[NT / 2], pl[NT / 2];
  float inv;
  attn_phase_a<NT>(sK, sMask, qh, ql, sub, lane, L, b, h, heads, scale, lse, dr, ph, pl, inv);
  attn_phase_b<NT>(sV, slab, ph, pl, inv, sub, lane, L, row0, col0, O, Oh, o_lo_off, ld_o);
}

template <int NT, int NW>   // NT = key tiles of 16 (even), LP = 16 * NT padded keys; NW = waves per workgroup
__global__ __launch_bounds__(64 * NW) void self_attn_mfma_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                             const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
                                                             const int64_t* __restrict__ seg, float* __restrict__ O,
                                                             bf16_t* __restrict__ Oh, size_t o_lo_off, int ld_o, int heads,
                                                             int L, float scale, float* __restrict__ lse, DropP dr) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                    // [hi | lo]
  char* sV = smem + 2 * PLANE;        // [hi | lo]
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);          // [LP]
  float* sOut = sMask + LP;                                          // [NW waves][16][HD + 4]
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;

  const int qn = lane & 15, g = lane >> 4;
  const int n_sub = (L + 15) >> 4;
  // Query fragments of a 16-row sub-tile: B operand of S^T = K Q^T (lane: query l & 15, hd 8*(l >> 4) + 32*ks ..).
  // Requested one sub-tile ahead -- the first one before K / V are staged -- so that the HBM latency of the 64 x 4 x 16 B
  // never sits between two MFMA phases.
  auto load_q = [&](int sub_, bf16x8_t (&fh)[2], bf16x8_t (&fl)[2]) {
    const int q_row_ = sub_ * 16 + qn;
    const bool ok = sub_ < n_sub && q_row_ < L;
    const size_t o = (row0 + (ok ? q_row_ : 0)) * (size_t)ld + col0 + 8 * g;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t a = {0, 0, 0, 0}, c = a;
      if (ok) {
        a = *reinterpret_cast<const u32x4_t*>(Qh + o + 32 * ks);
        c = *reinterpret_cast<const u32x4_t*>(Qh + o + 32 * ks + lo_off);
      }
      fh[ks] = __builtin_bit_cast(bf16x8_t, a);
      fl[ks] = __builtin_bit_cast(bf16x8_t, c);
    }
  };
  const int sub_first = blockIdx.x * NW + wave, sub_step = gridDim.x * NW;
  bf16x8_t qh[2], ql[2], qh_next[2], ql_next[2];
  load_q(sub_first, qh_next, ql_next);

  // ---- stage K, V (both planes) and the additive key mask: every request first, then the LDS writes (one HBM latency
  // for the whole 4 x LP x 128 B instead of one per loop trip) ----
  {
    constexpr int TRIPS = (LP * 8 + 64 * NW - 1) / (64 * NW);
    u32x4_t kh[TRIPS], kl[TRIPS], vh[TRIPS], vl[TRIPS];
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = tid + it * 64 * NW;
      const int r = i >> 3, u = i & 7;
      kh[it] = u32x4_t{0, 0, 0, 0};
      kl[it] = kh[it]; vh[it] = kh[it]; vl[it] = kh[it];
      if (i < LP * 8 && r < L && !(LR2_SA_ABLATE & 1)) {
        const size_t o = (row0 + r) * (size_t)ld + col0 + u * 8;
        kh[it] = *reinterpret_cast<const u32x4_t*>(Kh + o);
        kl[it] = *reinterpret_cast<const u32x4_t*>(Kh + o + lo_off);
        vh[it] = *reinterpret_cast<const u32x4_t*>(Vh + o);
        vl[it] = *reinterpret_cast<const u32x4_t*>(Vh + o + lo_off);
      }
    }
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = tid + it * 64 * NW;
      const int r = i >> 3, u = i & 7;
      if (i < LP * 8) {
        *reinterpret_cast<u32x4_t*>(sK + k_off(r, u)) = kh[it];
        *reinterpret_cast<u32x4_t*>(sK + PLANE + k_off(r, u)) = kl[it];
        *reinterpret_cast<u32x4_t*>(sV + v_off(r, u)) = vh[it];
        *reinterpret_cast<u32x4_t*>(sV + PLANE + v_off(r, u)) = vl[it];
      }
    }
  }
  // additive key mask, pre-multiplied by log2(e): the softmax below works on t = s * scale * log2(e) + mask * log2(e)
  // (2^(t - max t) = e^(s' - max s')), one fma + one v_exp_f32 per score instead of fma, multiply and v_exp_f32
  for (int j = tid; j < LP; j += 64 * NW) sMask[j] = j < L ? ((seg[row0 + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;

  __syncthreads();
  // K / V stay resident; each wave walks over 16-query sub-tiles (blockIdx.x strides them when the grid splits the queries)
  for (int sub = sub_first; sub < ((LR2_SA_ABLATE & 2) ? 0 : n_sub); sub += sub_step) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) { qh[ks] = qh_next[ks]; ql[ks] = ql_next[ks]; }
  load_q(sub + sub_step, qh_next, ql_next);     // next sub-tile's queries travel while this one is computed

  attn_subtile_fwd<NT>(sK, sV, sMask, sOut + wave * 16 * (HD + 4), qh, ql, sub, lane, L, b, h, heads, row0, col0, scale, lse, dr,
                       O, Oh, o_lo_off, ld_o);
  }  // sub-tile loop
}

// ---- persistent forward: one workgroup per CU walks over (sequence, head) pairs, K / V travel by LDS-DMA under the compute ----
// The one-pair kernel above spends a quarter of its time waiting for its K / V planes (112 KiB per workgroup, one workgroup per CU:
// nothing else runs meanwhile), every workgroup pays its launch and the drain of its last stores, and its phases (staging, S,
// softmax, P V, stores) barely overlap: two waves per SIMD in the same phase (tools/dbg/attn_ablate.py).  Here
//   * a workgroup is 16 waves at <= 128 VGPRs: wave w < n_sub owns the 16-query sub-tile w of every pair (n_sub <= 14: L <= 224), waves
//     14 and 15 only move data -- three to four waves per SIMD in different places instead of two in the same one;
//   * a pair is two phases that touch different planes -- A: S = Q K^T + softmax (K, mask), the probabilities kept as bf16 fragments in
//     registers; B: O = P V (V) -- so that during phase A of pair i the V planes of pair i are loaded (V is free since the end of pair
//     i - 1) and during phase B of pair i the K planes, the key mask and the query fragments of pair i + 1 (K is free after phase A);
//   * the two mover waves issue every LDS-DMA piece and are the only ones that wait for memory (s_waitcnt vmcnt(0) before the phase
//     barrier); the compute waves' stores stay in flight across pairs.
// Two barriers per pair: after A (every K read retired -> K free; V landed) and after B (every V read retired; K, mask landed).
// Same arithmetic, same bits as self_attn_mfma_kernel (both call attn_phase_a / attn_phase_b).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sa_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  void* q = (void*)(((uint64_t)hi << 32) | (uint64_t)lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Rows 8j .. 8j + 7 (j = j0, j0 + jstep, ...) of one head's K or V, both planes, into the LDS image: LDS-DMA writes lane-linearly
// (lane l -> byte 16 l of the 1-KiB piece = row l >> 3, slot l & 7), so the swizzle of k_off / v_off is applied to the SOURCE unit;
// rows >= L are out-of-range requests (the descriptor's range check writes zeros).
template <int NT, bool IS_V>
__device__ __forceinline__ void dma_head_rows(const __amdgpu_buffer_rsrc_t& hi, const __amdgpu_buffer_rsrc_t& lo, char* dst, int lane,
                                              int j0, int jstep, uint32_t pair_off, uint32_t row_bytes, int L) {
  constexpr int LP = 16 * NT, PLANE = LP * ROW_B;
  const int rl = lane >> 3, sl = lane & 7;
  for (int j = j0; j < ((LR2_SA_ABLATE & 1) ? 0 : LP / 8); j += jstep) {
    const int r = 8 * j + rl;
    const int u = IS_V ? (sl ^ (((r >> 1) & 3) << 1)) : (sl ^ ((r >> 1) & 7));
    const uint32_t v = r < L ? pair_off + (uint32_t)r * row_bytes + (uint32_t)u * 16u : 0xFFFFFF00u;
    char* d = dst + j * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(hi, LDS_PTR(d), 16, v, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(lo, LDS_PTR(d + PLANE), 16, v, 0, 0, 0);
  }
}

// every LDS access of this wave retired, then meet the workgroup (no vmcnt wait: stores and DMA stay in flight)
__device__ __forceinline__ void phase_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int PS_WAVES = 16, PS_MOVERS = 2, PS_MAX_SUB = PS_WAVES - PS_MOVERS;

template <int NT, bool DROP>
__global__ __launch_bounds__(64 * PS_WAVES) void self_attn_persist_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                                          const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
                                                                          const int64_t* __restrict__ seg, float* __restrict__ O,
                                                                          bf16_t* __restrict__ Oh, size_t o_lo_off, int ld_o, int heads,
                                                                          int L, float scale, float* __restrict__ lse, DropP dr,
                                                                          int n_pairs, uint32_t kv_bytes) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  constexpr int MK = (LP + 64 * PS_MOVERS - 1) / (64 * PS_MOVERS);     // mask values per mover thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                    // [hi | lo]
  char* sV = smem + 2 * PLANE;        // [hi | lo]
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);          // [2][LP]: pair number it reads half it & 1
  float* sOut = sMask + 2 * LP;                                      // [PS_MAX_SUB waves][16][32 + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool mover = wave >= PS_MAX_SUB;
  const int mtid = tid - 64 * PS_MAX_SUB;                             // thread number among the movers
  const int qn = lane & 15, g = lane >> 4;
  const int n_sub = (L + 15) >> 4;
  const bool computes = wave < n_sub;
  float* slab = sOut + (mover ? 0 : wave) * 16 * (32 + 4);
  const uint32_t row_bytes = (uint32_t)ld * 2u;

  int p = blockIdx.x;
  if (p >= n_pairs) return;
  int b = p / heads, h = p - b * heads;
  size_t row0 = (size_t)b * L;
  int col0 = h * HD;

  if (mover) {
    // ---- the two mover waves ----
    const __amdgpu_buffer_rsrc_t k_hi = sa_rsrc(Kh, kv_bytes), k_lo = sa_rsrc(Kh + lo_off, kv_bytes);
    const __amdgpu_buffer_rsrc_t v_hi = sa_rsrc(Vh, kv_bytes), v_lo = sa_rsrc(Vh + lo_off, kv_bytes);
    const int j0 = wave - PS_MAX_SUB;
    dma_head_rows<NT, false>(k_hi, k_lo, sK, lane, j0, PS_MOVERS, (uint32_t)((row0 * ld + col0) * 2), row_bytes, L);
    // additive key mask, pre-multiplied by log2(e) (see self_attn_mfma_kernel)
    for (int j = mtid; j < LP; j += 64 * PS_MOVERS) sMask[j] = j < L ? ((seg[row0 + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    for (int it = 0;; ++it) {
      // phase A of pair p: its V planes travel
      dma_head_rows<NT, true>(v_hi, v_lo, sV, lane, j0, PS_MOVERS, (uint32_t)((row0 * ld + col0) * 2), row_bytes, L);
      const int pn = p + gridDim.x;
      const bool more = pn < n_pairs;
      const int bn = pn / heads, hn = pn - bn * heads;
      const size_t row0n = (size_t)bn * L;
      float mk[MK];
      if (more) {
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          mk[i] = j < L ? ((seg[row0n + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // V of this pair has landed
      phase_barrier();
      if (!more) break;
      // phase B of pair p: K and the mask of the next pair travel
      dma_head_rows<NT, false>(k_hi, k_lo, sK, lane, j0, PS_MOVERS, (uint32_t)((row0n * ld + hn * HD) * 2), row_bytes, L);
      float* mnext = sMask + ((it + 1) & 1) * LP;
#pragma unroll
      for (int i = 0; i < MK; ++i) {
        const int j = mtid + i * 64 * PS_MOVERS;
        if (j < LP) mnext[j] = mk[i];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // K of the next pair has landed
      phase_barrier();
      p = pn; row0 = row0n; col0 = hn * HD;
    }
    return;
  }

  if (!computes) {
    // ---- waves n_sub .. 13: nothing to compute, they only keep the barriers company ----
    phase_barrier();
    for (;;) {
      phase_barrier();
      p += gridDim.x;
      if (p >= n_pairs) break;
      phase_barrier();
    }
    return;
  }

  // ---- compute waves ----
  const int sub = wave;
  // query fragments of this wave's sub-tile of the pair whose first row is row0_ (B operand of S^T = K Q^T)
  auto load_q = [&](size_t row0_, int col0_, bf16x8_t (&fh)[2], bf16x8_t (&fl)[2]) {
    const int lane_ = opaque(lane);
    const int q_row_ = sub * 16 + (lane_ & 15);
    const bool ok = q_row_ < L;
    const bf16_t* ub = Qh + row0_ * (size_t)ld + col0_;                      // uniform
    const uint32_t o = (uint32_t)(ok ? q_row_ : 0) * (uint32_t)ld + 8u * (uint32_t)(lane_ >> 4);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4_t a = {0, 0, 0, 0}, c = a;
      if (ok) {
        a = *reinterpret_cast<const u32x4_t*>(ub + o + 32 * ks);
        c = *reinterpret_cast<const u32x4_t*>(ub + lo_off + o + 32 * ks);
      }
      fh[ks] = __builtin_bit_cast(bf16x8_t, a);
      fl[ks] = __builtin_bit_cast(bf16x8_t, c);
    }
  };
  bf16x8_t qh[2], ql[2];
  load_q(row0, col0, qh, ql);
  phase_barrier();
  for (int it = 0;; ++it) {
    const float* mask = sMask + (it & 1) * LP;
    bf16x8_t ph[NT / 2], pl[NT / 2];
    float inv = 0.f;
    if (!(LR2_SA_ABLATE & 2)) attn_phase_a<NT, DROP>(sK, mask, qh, ql, sub, opaque(lane), L, b, h, heads, scale, lse, dr, ph, pl, inv);
    phase_barrier();
    const int pn = p + gridDim.x;
    const bool more = pn < n_pairs;
    const int bn = pn / heads, hn = pn - bn * heads;
    const size_t row0n = (size_t)bn * L;
    // the next pair's queries are requested once the probability registers are dead; they travel under the output's stores and the
    // wait at the barrier
    auto next_q = [&]() { if (more) load_q(row0n, hn * HD, qh, ql); };
    if (!(LR2_SA_ABLATE & 2)) attn_phase_b<NT, true>(sV, slab, ph, pl, inv, sub, opaque(lane), L, row0, col0, O, Oh, o_lo_off, ld_o, next_q);
    if (!more) break;
    phase_barrier();
    p = pn; b = bn; h = hn; row0 = row0n; col0 = hn * HD;
  }
}

// ---- forward for sequences longer than one LDS-resident key block (L > 256: ViT-L/14's 257 tokens, RoBERTa's 514) ----
// Same arithmetic and fragment layout as self_attn_mfma_kernel, with the keys walked in blocks of LP = 16 * NT: K / V of
// ONE block live in LDS; every wave keeps, for each of its (up to SLOTS) 16-query sub-tiles, the running row maximum m,
// the running sum l and the un-normalised output accumulator across the blocks (online softmax):
//     m' = max(m, max_j s_j);  a = exp(m - m');  l = a l + sum_j exp(s_j - m');  O = a O + exp(s - m') V_block
// and normalises by l at the end.  Probability dropout multiplies exp(s - m') by mask / keep before the P V product; l is
// the sum WITHOUT the mask, so O / l equals dropout(softmax(S)) V exactly as in the one-block kernel.
// Padding keys (index >= L) carry -inf; every block holds at least one real key (real keys masked by seg get -10000,
// as upstream), so m' is finite from the first block on.
template <int SLOTS>
struct AttnState {
  f32x4_t o[SLOTS][4];
  float m[SLOTS], l[SLOTS];
};

template <int NT, int NW, int SLOTS, int J>
__device__ __forceinline__ void blocked_subtiles(AttnState<SLOTS>& st, const bf16_t* __restrict__ Qh, size_t lo_off, int ld,
                                                 size_t row0, int col0, const char* sK, const char* sV, const float* sMask,
                                                 int sub_first, int sub_step, int n_sub, int L, int k0, float scale, int heads,
                                                 int b, int h, const DropP& dr, int lane) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  const int qn = lane & 15, g = lane >> 4;
  const int sub = sub_first + J * sub_step;
  if (sub < n_sub) {            // wave-uniform
    const int q_row = sub * 16 + qn;
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
    f32x4_t s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r = 16 * t + qn;
        const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(sK + k_off(r, g + 4 * ks));
        const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(sK + PLANE + k_off(r, g + 4 * ks));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[ks], acc, 0, 0, 0);
      }
      s[t] = acc;
    }
    float mx = st.m[J];
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
    const float alpha = exp_fast(st.m[J] - mx);     // first block: exp(-inf) = 0
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s[t][r] = exp_fast(s[t][r] - mx);
        sum += s[t][r];
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    st.m[J] = mx;
    st.l[J] = st.l[J] * alpha + sum;
    // the output accumulator holds O[query 4g + r][..] in register r, the statistics belong to query (l & 15): fetch the
    // rescale factor of query 4g + r from the lane that owns it
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a_r = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
      for (int n = 0; n < 4; ++n) st.o[J][n][r] *= a_r;
    }
    const uint64_t drow = (((uint64_t)b * heads + h) * L + (uint64_t)(q_row < L ? q_row : 0)) * mask_pitch(L) + (uint64_t)k0;
    const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
    for (int u = 0; u < NT / 2; ++u) {
      float p[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[r] = s[2 * u][r];
        p[4 + r] = s[2 * u + 1][r];
      }
      if (dr.thr) {
        drop_mul4(dr, drow + 32 * u + 4 * g, p[0], p[1], p[2], p[3]);
        drop_mul4(dr, drow + 32 * u + 16 + 4 * g, p[4], p[5], p[6], p[7]);
      }
      const uint32_t h01 = cvt_pk_bf16(p[0], p[1]), h23 = cvt_pk_bf16(p[2], p[3]);
      const uint32_t h45 = cvt_pk_bf16(p[4], p[5]), h67 = cvt_pk_bf16(p[6], p[7]);
      const uint32_t l01 = cvt_pk_bf16(p[0] - __uint_as_float(h01 << 16), p[1] - __uint_as_float(h01 & 0xffff0000u));
      const uint32_t l23 = cvt_pk_bf16(p[2] - __uint_as_float(h23 << 16), p[3] - __uint_as_float(h23 & 0xffff0000u));
      const uint32_t l45 = cvt_pk_bf16(p[4] - __uint_as_float(h45 << 16), p[5] - __uint_as_float(h45 & 0xffff0000u));
      const uint32_t l67 = cvt_pk_bf16(p[6] - __uint_as_float(h67 << 16), p[7] - __uint_as_float(h67 & 0xffff0000u));
      const bf16x8_t ph = __builtin_bit_cast(bf16x8_t, (u32x4_t{h01, h23, h45, h67}));
      const bf16x8_t pl = __builtin_bit_cast(bf16x8_t, (u32x4_t{l01, l23, l45, l67}));
      const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int unit = 2 * n + (tp >> 1), half8 = 8 * (tp & 1);
        const bf16x8_t vh = tr_pair(sV, ra, rb, unit, half8);
        const bf16x8_t vl = tr_pair(sV + PLANE, ra, rb, unit, half8);
        st.o[J][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh, st.o[J][n], 0, 0, 0);
        st.o[J][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl, st.o[J][n], 0, 0, 0);
        st.o[J][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh, st.o[J][n], 0, 0, 0);
      }
    }
  }
  if constexpr (J + 1 < SLOTS)
    blocked_subtiles<NT, NW, SLOTS, J + 1>(st, Qh, lo_off, ld, row0, col0, sK, sV, sMask, sub_first, sub_step, n_sub, L, k0, scale,
                                           heads, b, h, dr, lane);
}

template <int SLOTS, int J>
__device__ __forceinline__ void blocked_finish(AttnState<SLOTS>& st, float* slab, int sub_first, int sub_step, int n_sub, int L,
                                               size_t row0, int col0, float* __restrict__ O, bf16_t* __restrict__ Oh,
                                               size_t o_lo_off, int ld_o, float* __restrict__ lse, int heads, int b, int h, int lane) {
  const int qn = lane & 15, g = lane >> 4;
  const int sub = sub_first + J * sub_step;
  if (sub < n_sub) {
    const float inv = 1.0f / st.l[J];
    const int q_row = sub * 16 + qn;
    if (lse && g == 0 && q_row < L) lse[((size_t)b * heads + h) * L + q_row] = st.m[J] + logf(st.l[J]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float inv_r = __shfl(inv, 4 * g + r, 64);
#pragma unroll
      for (int n = 0; n < 4; ++n) slab[(4 * g + r) * (HD + 4) + 16 * n + qn] = st.o[J][n][r] * inv_r;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 4 + (lane >> 4), c = (lane & 15) * 4;
      const int qr = sub * 16 + r;
      if (qr < L) {
        const float4 v = *reinterpret_cast<const float4*>(slab + r * (HD + 4) + c);
        const size_t off = (row0 + qr) * (size_t)ld_o + col0 + c;
        if (O) *reinterpret_cast<float4*>(O + off) = v;
        if (Oh) store_planes4(Oh + off, o_lo_off, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (J + 1 < SLOTS)
    blocked_finish<SLOTS, J + 1>(st, slab, sub_first, sub_step, n_sub, L, row0, col0, O, Oh, o_lo_off, ld_o, lse, heads, b, h, lane);
}

template <int NT, int NW, int SLOTS>
__global__ __launch_bounds__(64 * NW) void self_attn_blocked_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                                const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
                                                                const int64_t* __restrict__ seg, float* __restrict__ O,
                                                                bf16_t* __restrict__ Oh, size_t o_lo_off, int ld_o, int heads,
                                                                int L, float scale, float* __restrict__ lse, DropP dr,
                                                                int n_blocks) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 2 * PLANE;
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);
  float* sOut = sMask + LP;
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;
  const int n_sub = (L + 15) >> 4;
  const int sub_first = blockIdx.x * NW + wave, sub_step = gridDim.x * NW;   // host: sub_first + SLOTS * sub_step >= n_sub
  AttnState<SLOTS> st;
#pragma unroll
  for (int j = 0; j < SLOTS; ++j) {
    st.m[j] = -INFINITY;
    st.l[j] = 0.f;
#pragma unroll
    for (int n = 0; n < 4; ++n) st.o[j][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  for (int blk = 0; blk < n_blocks; ++blk) {
    const int k0 = blk * LP;
    if (blk) __syncthreads();                       // every wave is done reading the previous block
    constexpr int TRIPS = (LP * 8 + 64 * NW - 1) / (64 * NW);
    u32x4_t kh[TRIPS], kl[TRIPS], vh[TRIPS], vl[TRIPS];
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = tid + it * 64 * NW;
      const int r = i >> 3, u = i & 7;
      kh[it] = u32x4_t{0, 0, 0, 0};
      kl[it] = kh[it]; vh[it] = kh[it]; vl[it] = kh[it];
      if (i < LP * 8 && k0 + r < L) {
        const size_t o = (row0 + k0 + r) * (size_t)ld + col0 + u * 8;
        kh[it] = *reinterpret_cast<const u32x4_t*>(Kh + o);
        kl[it] = *reinterpret_cast<const u32x4_t*>(Kh + o + lo_off);
        vh[it] = *reinterpret_cast<const u32x4_t*>(Vh + o);
        vl[it] = *reinterpret_cast<const u32x4_t*>(Vh + o + lo_off);
      }
    }
#pragma unroll
    for (int it = 0; it < TRIPS; ++it) {
      const int i = tid + it * 64 * NW;
      const int r = i >> 3, u = i & 7;
      if (i < LP * 8) {
        *reinterpret_cast<u32x4_t*>(sK + k_off(r, u)) = kh[it];
        *reinterpret_cast<u32x4_t*>(sK + PLANE + k_off(r, u)) = kl[it];
        *reinterpret_cast<u32x4_t*>(sV + v_off(r, u)) = vh[it];
        *reinterpret_cast<u32x4_t*>(sV + PLANE + v_off(r, u)) = vl[it];
      }
    }
    for (int j = tid; j < LP; j += 64 * NW) sMask[j] = (k0 + j < L) ? ((seg[row0 + k0 + j] > 0) ? 0.f : -10000.0f) : -INFINITY;
    __syncthreads();
    blocked_subtiles<NT, NW, SLOTS, 0>(st, Qh, lo_off, ld, row0, col0, sK, sV, sMask, sub_first, sub_step, n_sub, L, k0, scale, heads,
                                       b, h, dr, lane);
  }
  blocked_finish<SLOTS, 0>(st, sOut + wave * 16 * (HD + 4), sub_first, sub_step, n_sub, L, row0, col0, O, Oh, o_lo_off, ld_o, lse,
                           heads, b, h, lane);
}

// fp32 x 8 -> A fragment pair (hi, lo) of the split product
__device__ __forceinline__ void split8(const float (&p)[8], bf16x8_t& hi, bf16x8_t& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = cvt_pk_bf16(p[2 * i], p[2 * i + 1]);
    l[i] = cvt_pk_bf16(p[2 * i] - __uint_as_float(h[i] << 16), p[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
  }
  hi = __builtin_bit_cast(bf16x8_t, (u32x4_t{h[0], h[1], h[2], h[3]}));
  lo = __builtin_bit_cast(bf16x8_t, (u32x4_t{l[0], l[1], l[2], l[3]}));
}

__device__ __forceinline__ bf16x8_t tr_pair_k(const char* plane, int row_a, int row_b, int u, int half8) {
  // transposed read from a plane kept in the K layout (k_off): correct, 2-4 way bank conflicts accepted
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(plane + k_off(row_a, u) + half8));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(plane + k_off(row_b, u) + half8));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// 16 rows x 64 columns of one planes matrix as MFMA fragments (lane: row l & 15, columns 8*(l >> 4) + 32*ks ..)
__device__ __forceinline__ void load_frags(const bf16_t* hi_plane, size_t lo_off, size_t elem_off, bool ok, bf16x8_t (&fh)[2],
                                           bf16x8_t (&fl)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    u32x4_t a = {0, 0, 0, 0}, c = a;
    if (ok) {
      a = *reinterpret_cast<const u32x4_t*>(hi_plane + elem_off + 32 * ks);
      c = *reinterpret_cast<const u32x4_t*>(hi_plane + elem_off + 32 * ks + lo_off);
    }
    fh[ks] = __builtin_bit_cast(bf16x8_t, a);
    fl[ks] = __builtin_bit_cast(bf16x8_t, c);
  }
}

// the same with a wave-uniform base pointer + a 32-bit lane offset (scalar-base loads: no 64-bit address registers per lane)
__device__ __forceinline__ void load_frags_u(const bf16_t* ubase, size_t lo_off, uint32_t lane_off, bool ok, bf16x8_t (&fh)[2],
                                             bf16x8_t (&fl)[2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    u32x4_t a = {0, 0, 0, 0}, c = a;
    if (ok) {
      a = *reinterpret_cast<const u32x4_t*>(ubase + lane_off + 32 * ks);
      c = *reinterpret_cast<const u32x4_t*>(ubase + lo_off + lane_off + 32 * ks);
    }
    fh[ks] = __builtin_bit_cast(bf16x8_t, a);
    fl[ks] = __builtin_bit_cast(bf16x8_t, c);
  }
}

__device__ __forceinline__ f32x4_t mfma3(bf16x8_t ah, bf16x8_t al, bf16x8_t bh, bf16x8_t bl, f32x4_t acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// 16 x 64 accumulator tile (o[n][r] = X[row 4g + r][col 16n + (l & 15)]) -> planes rows via the wave's LDS slab
__device__ __forceinline__ void store_tile_planes(const f32x4_t (&o)[4], float* slab, int lane, int row_first, int rows_valid,
                                                  bf16_t* dst_hi, size_t lo_off, size_t row_stride, size_t base) {
  const int qn = lane & 15, g = lane >> 4;
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(4 * g + r) * (HD + 4) + 16 * n + qn] = o[n][r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int r = pass * 4 + (lane >> 4), c = (lane & 15) * 4;
    if (row_first + r < rows_valid) {
      const float4 v = *reinterpret_cast<const float4*>(slab + r * (HD + 4) + c);
      store_planes4(dst_hi + base + (size_t)(row_first + r) * row_stride + c, lo_off, v);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// ---- backward, part 1: dQ (+ the per-query statistics part 2 needs) -------------------------------------------------
// Same decomposition as the forward: workgroup = (sequence, head, 64 queries), K and V of the head in LDS, everything
// in the transposed layout (lane = one query, 4 keys per 16-key tile):
//   S^T = K Q^T, P = softmax;  dPd^T = V dO^T;  dP = dPd o M;  D = sum_k dP P;  dS = P (dP - D) * scale;  dQ = dS K
// (M = dropout keep / (1 - p)).  dQ goes to columns [h*64, h*64+64) of the dQKV planes matrix.
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void self_attn_bwd_dq_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                               const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
                                                               const bf16_t* __restrict__ dOh, size_t do_lo_off, int ld_do,
                                                               const int64_t* __restrict__ seg, bf16_t* __restrict__ dQh,
                                                               size_t dq_lo_off, int ld_dq, float* __restrict__ lse,
                                                               float* __restrict__ dsum, int heads, int L, float scale,
                                                               DropP dr) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 2 * PLANE;        // K layout too: V is an A operand here (rows = keys, contraction over hd)
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);
  float* sOut = sMask + LP;
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;
  for (int i = tid; i < LP * 8; i += 64 * NW) {
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
    *reinterpret_cast<u32x4_t*>(sV + k_off(r, u)) = vh;
    *reinterpret_cast<u32x4_t*>(sV + PLANE + k_off(r, u)) = vl;
  }
  for (int j = tid; j < LP; j += 64 * NW) sMask[j] = j < L ? ((seg[row0 + j] > 0) ? 0.f : -10000.0f) : -INFINITY;
  const int qn = lane & 15, g = lane >> 4;
  __syncthreads();
  const int n_sub = (L + 15) >> 4;
  for (int sub = blockIdx.x * NW + wave; sub < n_sub; sub += gridDim.x * NW) {
  const int q_row = sub * 16 + qn;
  const bool q_ok = q_row < L;
  bf16x8_t qh[2], ql[2], gh[2], gl[2];
  load_frags(Qh, lo_off, (row0 + (q_ok ? q_row : 0)) * (size_t)ld + col0 + 8 * g, q_ok, qh, ql);
  load_frags(dOh, do_lo_off, (row0 + (q_ok ? q_row : 0)) * (size_t)ld_do + col0 + 8 * g, q_ok, gh, gl);

  f32x4_t s[NT], dp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    f32x4_t a = {0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r = 16 * t + qn;
      const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(sK + k_off(r, g + 4 * ks));
      const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(sK + PLANE + k_off(r, g + 4 * ks));
      const bf16x8_t vh = *reinterpret_cast<const bf16x8_t*>(sV + k_off(r, g + 4 * ks));
      const bf16x8_t vl = *reinterpret_cast<const bf16x8_t*>(sV + PLANE + k_off(r, g + 4 * ks));
      a = mfma3(kh, kl, qh[ks], ql[ks], a);
      d = mfma3(vh, vl, gh[ks], gl[ks], d);
    }
    s[t] = a;
    dp[t] = d;
  }
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
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] = exp_fast(s[t][r] - mx);
      sum += s[t][r];
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const uint64_t drow = (((uint64_t)b * heads + h) * L + (uint64_t)(q_ok ? q_row : 0)) * mask_pitch(L);
  float dd = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (dr.thr) dp[t] = drop_mul4v(dr, drow + 16 * t + 4 * g, dp[t]);    // dP = dPd o M
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] *= inv;                                                     // P
      dd += dp[t][r] * s[t][r];
    }
  }
  dd += __shfl_xor(dd, 16, 64);
  dd += __shfl_xor(dd, 32, 64);
  if (g == 0 && q_ok) {
    const size_t si = ((size_t)b * heads + h) * L + q_row;
    lse[si] = mx + logf(sum);
    dsum[si] = dd;
  }
  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
#pragma unroll
  for (int u = 0; u < NT / 2; ++u) {
    float e[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[r] = s[2 * u][r] * (dp[2 * u][r] - dd) * scale;
      e[4 + r] = s[2 * u + 1][r] * (dp[2 * u + 1][r] - dd) * scale;
    }
    bf16x8_t eh, el;
    split8(e, eh, el);
    const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int unit = 2 * n + (tp >> 1), half8 = 8 * (tp & 1);
      o[n] = mfma3(eh, el, tr_pair_k(sK, ra, rb, unit, half8), tr_pair_k(sK + PLANE, ra, rb, unit, half8), o[n]);
    }
  }
  store_tile_planes(o, sOut + wave * 16 * (HD + 4), lane, sub * 16, L, dQh, dq_lo_off, (size_t)ld_dq,
                    row0 * (size_t)ld_dq + col0);
  }  // sub-tile loop
}

// ---- backward, part 2: dK, dV ------------------------------------------------------------------------------------------
// Workgroup = (sequence, head, 64 keys); Q and dO of the head in LDS; each wave owns 16 keys (K, V fragments in
// registers) and walks over the queries in the NON-transposed layout (lane = one key, 4 queries per 16-query tile):
//   S = Q K^T, P = exp(S - lse[q]);  dPd = dO V^T;  Pd = P o M, dP = dPd o M, dS = P (dP - D[q]) * scale
//   dV = Pd^T dO,  dK = dS^T Q       (contraction over queries: P / dS tiles reused as A fragments, Q / dO transposed reads)
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void self_attn_bwd_dkv_kernel(const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh,
                                                                const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
                                                                const bf16_t* __restrict__ dOh, size_t do_lo_off, int ld_do,
                                                                const int64_t* __restrict__ seg, bf16_t* __restrict__ dKh,
                                                                bf16_t* __restrict__ dVh, size_t dkv_lo_off, int ld_dkv,
                                                                const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                int heads, int L, float scale, DropP dr) {
  constexpr int LP = 16 * NT;
  constexpr int PLANE = LP * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;                    // K layout: fragment reads (rows = queries) + transposed reads
  char* sG = smem + 2 * PLANE;        // dO
  float* sLse = reinterpret_cast<float*>(smem + 4 * PLANE);   // [LP]
  float* sD = sLse + LP;                                      // [LP]
  float* sOut = sD + LP;
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;
  for (int i = tid; i < LP * 8; i += 64 * NW) {
    const int r = i >> 3, u = i & 7;
    u32x4_t a = {0, 0, 0, 0}, c = a, d = a, e = a;
    if (r < L) {
      const size_t o = (row0 + r) * (size_t)ld + col0 + u * 8;
      const size_t og = (row0 + r) * (size_t)ld_do + col0 + u * 8;
      a = *reinterpret_cast<const u32x4_t*>(Qh + o);
      c = *reinterpret_cast<const u32x4_t*>(Qh + o + lo_off);
      d = *reinterpret_cast<const u32x4_t*>(dOh + og);
      e = *reinterpret_cast<const u32x4_t*>(dOh + og + do_lo_off);
    }
    *reinterpret_cast<u32x4_t*>(sQ + k_off(r, u)) = a;
    *reinterpret_cast<u32x4_t*>(sQ + PLANE + k_off(r, u)) = c;
    *reinterpret_cast<u32x4_t*>(sG + k_off(r, u)) = d;
    *reinterpret_cast<u32x4_t*>(sG + PLANE + k_off(r, u)) = e;
  }
  for (int j = tid; j < LP; j += 64 * NW) {
    const size_t si = ((size_t)b * heads + h) * L + j;
    sLse[j] = j < L ? lse[si] : INFINITY;     // padded queries: P = exp(-inf) = 0
    sD[j] = j < L ? dsum[si] : 0.f;
  }
  const int kn = lane & 15, g = lane >> 4;
  __syncthreads();
  const int n_sub = (L + 15) >> 4;
  for (int sub = blockIdx.x * NW + wave; sub < n_sub; sub += gridDim.x * NW) {
  const int key = sub * 16 + kn;
  const bool k_ok = key < L;
  bf16x8_t kh[2], kl[2], vh[2], vl[2];
  load_frags(Kh, lo_off, (row0 + (k_ok ? key : 0)) * (size_t)ld + col0 + 8 * g, k_ok, kh, kl);
  load_frags(Vh, lo_off, (row0 + (k_ok ? key : 0)) * (size_t)ld + col0 + 8 * g, k_ok, vh, vl);
  const float kmask = k_ok ? ((seg[row0 + key] > 0) ? 0.f : -10000.0f) : -INFINITY;

  f32x4_t dv[4], dk[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) dv[n] = dk[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const uint64_t dbase = ((uint64_t)b * heads + h) * (uint64_t)L;
#pragma unroll 1
  for (int u = 0; u < NT / 2; ++u) {
    float pd[8], ds[8];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int t = 2 * u + half;
      f32x4_t a = {0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int r = 16 * t + kn;         // A fragments: query row 16t + (l & 15)
        const bf16x8_t qh = *reinterpret_cast<const bf16x8_t*>(sQ + k_off(r, g + 4 * ks));
        const bf16x8_t ql = *reinterpret_cast<const bf16x8_t*>(sQ + PLANE + k_off(r, g + 4 * ks));
        const bf16x8_t gh = *reinterpret_cast<const bf16x8_t*>(sG + k_off(r, g + 4 * ks));
        const bf16x8_t gl = *reinterpret_cast<const bf16x8_t*>(sG + PLANE + k_off(r, g + 4 * ks));
        a = mfma3(qh, ql, kh[ks], kl[ks], a);      // S[query 16t + 4g + r][key kn]
        d = mfma3(gh, gl, vh[ks], vl[ks], d);      // dPd
      }
      const float4 ls = *reinterpret_cast<const float4*>(sLse + 16 * t + 4 * g);
      const float4 dd = *reinterpret_cast<const float4*>(sD + 16 * t + 4 * g);
      const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp_fast(a[r] * scale + kmask - lsv[r]);
        float m = 1.0f;
        if (dr.thr) {
          const int q = 16 * t + 4 * g + r;
          m = drop_mul(dr, (dbase + (uint64_t)(q < L ? q : 0)) * mask_pitch(L) + (uint64_t)(k_ok ? key : 0));
        }
        pd[4 * half + r] = p * m;
        ds[4 * half + r] = p * (d[r] * m - ddv[r]) * scale;
      }
    }
    bf16x8_t ph, pl, eh, el;
    split8(pd, ph, pl);
    split8(ds, eh, el);
    const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int unit = 2 * n + (tp >> 1), half8 = 8 * (tp & 1);
      dv[n] = mfma3(ph, pl, tr_pair_k(sG, ra, rb, unit, half8), tr_pair_k(sG + PLANE, ra, rb, unit, half8), dv[n]);
      dk[n] = mfma3(eh, el, tr_pair_k(sQ, ra, rb, unit, half8), tr_pair_k(sQ + PLANE, ra, rb, unit, half8), dk[n]);
    }
  }
  float* slab = sOut + wave * 16 * (HD + 4);
  store_tile_planes(dk, slab, lane, sub * 16, L, dKh, dkv_lo_off, (size_t)ld_dkv, row0 * (size_t)ld_dkv + col0);
  store_tile_planes(dv, slab, lane, sub * 16, L, dVh, dkv_lo_off, (size_t)ld_dkv, row0 * (size_t)ld_dkv + col0);
  }  // sub-tile loop
}

// ---- persistent backward (round 4): the forward's log-sum-exp and output are inputs, nothing is recomputed twice ----
// With lse[q] from the forward, P = exp(S - lse) needs no row maximum / row sum, and D[q] = sum_k dP P = sum_d dO[q, d] O[q, d] needs
// no pass over the keys: both kernels STREAM over 32-row blocks of the resident operand -- one S / dP tile pair in registers at a time
// (the one-pair kernels above hold 14 S tiles and 14 dP tiles of a sub-tile) -- which fits 16-wave workgroups at <= 128 VGPRs:
//   * wave w < n_sub owns sub-tile w (16 queries in the dQ kernel, 16 keys in the dK / dV kernel) of every pair; waves 14, 15 move data;
//   * the resident planes (K, V / Q, dO: both in the d_off layout) are refilled IN HALVES while the other half is being used: a row
//     block is dead once every wave has passed it, so after block H1 - 1 (barrier "mid") the movers load rows [0, 32 H1) of the NEXT
//     pair and after the last block (barrier "end") the rest; a mover waits for its pieces (s_waitcnt vmcnt(0)) before the NEXT barrier,
//     i.e. half a pair later; no compute wave waits for memory except for its own 16-row fragments, requested before the stores.
// Arithmetic: dQ kernel  S^T = K Q^T, dPd^T = V dO^T, P = exp(S scale + mask - lse), dS = P (dPd o M - D) scale, dQ = dS K
//             dKV kernel S = Q K^T, dPd = dO V^T, Pd = P o M, dS as above, dV = Pd^T dO, dK = dS^T Q          (M = keep / (1 - p))
// replaces: autograd of tencentpretrain/layers/multi_headed_attn.py:61-74 (as the one-pair kernels do).
// LDS image of a plane that is read BOTH as row fragments (ds_read_b128: 16 rows x one 16-B unit) and transposed (ds_read_b64_tr_b16:
// 8 rows x 32 B per half-wave): unit u of row r at u ^ x(r), x = 2 ((r >> 1) & 3) + ((r >> 3) & 1).  x is a bijection of the 8 row
// pairs of a 16-row tile (fragment reads: 16 distinct 16-B slots = all 64 banks once) and x >> 1 takes 4 distinct values on the 4 row
// pairs of each 8-row group (transposed reads: 8 distinct 32-B bank groups); k_off's x = (r >> 1) & 7 gives the second only two ways.
__device__ __forceinline__ int d_swz(int r) { return 2 * ((r >> 1) & 3) + ((r >> 3) & 1); }
__device__ __forceinline__ int d_off(int r, int u) { return r * ROW_B + ((u ^ d_swz(r)) << 4); }

template <int NT>
__device__ __forceinline__ void dma_rows_k(const __amdgpu_buffer_rsrc_t& hi, const __amdgpu_buffer_rsrc_t& lo, char* dst, int lane,
                                           int j_begin, int j_end, int jstep, uint32_t pair_off, uint32_t row_bytes, int L) {
  constexpr int LP = 16 * NT, PLANE = LP * ROW_B;
  const int rl = lane >> 3, sl = lane & 7;
  for (int j = j_begin; j < ((LR2_SA_ABLATE & 1) ? 0 : j_end); j += jstep) {
    const int r = 8 * j + rl;
    const int u = sl ^ d_swz(r);
    const uint32_t v = r < L ? pair_off + (uint32_t)r * row_bytes + (uint32_t)u * 16u : 0xFFFFFF00u;
    char* d = dst + j * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(hi, LDS_PTR(d), 16, v, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(lo, LDS_PTR(d + PLANE), 16, v, 0, 0, 0);
  }
}

// eight bf16 (one fragment) -> fp32 sum of hi + lo products with another fragment pair: sum_i (ah + al)_i (bh + bl)_i
__device__ __forceinline__ float frag_dot(bf16x8_t ah, bf16x8_t al, bf16x8_t bh, bf16x8_t bl) {
  const u32x4_t a = __builtin_bit_cast(u32x4_t, ah), c = __builtin_bit_cast(u32x4_t, al);
  const u32x4_t b = __builtin_bit_cast(u32x4_t, bh), d = __builtin_bit_cast(u32x4_t, bl);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = __uint_as_float(a[i] << 16) + __uint_as_float(c[i] << 16);
    const float x1 = __uint_as_float(a[i] & 0xffff0000u) + __uint_as_float(c[i] & 0xffff0000u);
    const float y0 = __uint_as_float(b[i] << 16) + __uint_as_float(d[i] << 16);
    const float y1 = __uint_as_float(b[i] & 0xffff0000u) + __uint_as_float(d[i] & 0xffff0000u);
    acc = __builtin_fmaf(x0, y0, acc);
    acc = __builtin_fmaf(x1, y1, acc);
  }
  return acc;
}

// 16 x 64 accumulator tile -> planes rows through a 16 x 32 slab, 32 columns at a time
__device__ __forceinline__ void store_tile_planes_half(const f32x4_t (&o)[4], float* slab, int lane, int row_first, int rows_valid,
                                                       bf16_t* dst_hi, size_t lo_off, size_t row_stride, size_t base) {
  const int qn = lane & 15, g = lane >> 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(4 * g + r) * (32 + 4) + 16 * n + qn] = o[2 * half + n][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = pass * 8 + (lane >> 3), c = (lane & 7) * 4;
      if (row_first + r < rows_valid && (!(LR2_SA_ABLATE & 64) || slab[r] == 12345.f)) {
        const float4 v = *reinterpret_cast<const float4*>(slab + r * (32 + 4) + c);
        store_planes4(dst_hi + base + 32 * half + ((uint32_t)(row_first + r) * (uint32_t)row_stride + (uint32_t)c), lo_off, v);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

struct BwdArgs {
  const bf16_t *q, *k, *v;      // hi planes (lo plane lo_off elements behind), row stride ld
  size_t lo_off;
  int ld;
  const bf16_t* go;             // dO hi plane
  size_t do_lo_off;
  int ld_do;
  const bf16_t* o;              // forward output hi plane
  size_t o_lo_off;
  int ld_o;
  const int64_t* seg;
  bf16_t *dq, *dk, *dv;
  size_t d_lo_off;
  int ld_d;
  const float* lse;             // [batch, heads, L] from the forward
  float* dsum;                  // [batch, heads, L]: written by the dQ kernel, read by the dK / dV kernel
  int heads, L, n_pairs;
  float scale;
  DropP dr;
  uint32_t qkv_bytes, do_bytes; // descriptor spans from q / k / v and from go
};

// rows [0, 32 * H1) are the first half of the resident planes
template <int NT>
struct Halves {
  static constexpr int NB = NT / 2, H1 = (NB + 1) / 2, J_MID = 4 * H1, J_END = 2 * NT;
};

template <int NT, bool DROP>
__global__ __launch_bounds__(64 * PS_WAVES) void self_attn_bwd_dq_persist_kernel(BwdArgs A) {
  constexpr int LP = 16 * NT, PLANE = LP * ROW_B, NB = Halves<NT>::NB, H1 = Halves<NT>::H1;
  constexpr int MK = (LP + 64 * PS_MOVERS - 1) / (64 * PS_MOVERS);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 2 * PLANE;        // same layout: V is an A operand here (rows = keys, contraction over hd)
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);          // [LP], pre-multiplied by log2(e)
  float* sOut = sMask + LP;                                          // [PS_MAX_SUB waves][16][32 + 4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int heads = A.heads, L = A.L, n_pairs = A.n_pairs;
  const int n_sub = (L + 15) >> 4;
  int p = blockIdx.x;
  if (p >= n_pairs) return;

  if (wave >= PS_MAX_SUB) {
    // ---- movers ----
    const __amdgpu_buffer_rsrc_t k_hi = sa_rsrc(A.k, A.qkv_bytes), k_lo = sa_rsrc(A.k + A.lo_off, A.qkv_bytes);
    const __amdgpu_buffer_rsrc_t v_hi = sa_rsrc(A.v, A.qkv_bytes), v_lo = sa_rsrc(A.v + A.lo_off, A.qkv_bytes);
    const int j0 = wave - PS_MAX_SUB, mtid = tid - 64 * PS_MAX_SUB;
    const uint32_t row_bytes = (uint32_t)A.ld * 2u;
    auto pair_off = [&](int pp) { const int b = pp / heads, h = pp - b * heads; return (uint32_t)(((size_t)b * L * A.ld + h * HD) * 2); };
    auto mask_of = [&](int pp, int j) -> float {
      const int b = pp / heads;
      return j < L ? ((A.seg[(size_t)b * L + j] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
    };
    dma_rows_k<NT>(k_hi, k_lo, sK, lane, j0, Halves<NT>::J_END, PS_MOVERS, pair_off(p), row_bytes, L);
    dma_rows_k<NT>(v_hi, v_lo, sV, lane, j0, Halves<NT>::J_END, PS_MOVERS, pair_off(p), row_bytes, L);
    for (int j = mtid; j < LP; j += 64 * PS_MOVERS) sMask[j] = mask_of(p, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    for (;;) {
      const int pn = p + gridDim.x;
      const bool more = pn < n_pairs;
      phase_barrier();                                   // mid: rows [0, 32 H1) of this pair are dead
      if (more) {
        dma_rows_k<NT>(k_hi, k_lo, sK, lane, j0, Halves<NT>::J_MID, PS_MOVERS, pair_off(pn), row_bytes, L);
        dma_rows_k<NT>(v_hi, v_lo, sV, lane, j0, Halves<NT>::J_MID, PS_MOVERS, pair_off(pn), row_bytes, L);
        float mk[MK];
#pragma unroll
        for (int i = 0; i < MK; ++i) mk[i] = mask_of(pn, mtid + i * 64 * PS_MOVERS);
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          if (j < 32 * H1) sMask[j] = mk[i];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      phase_barrier();                                   // end: the rest is dead
      if (!more) break;
      dma_rows_k<NT>(k_hi, k_lo, sK, lane, Halves<NT>::J_MID + j0, Halves<NT>::J_END, PS_MOVERS, pair_off(pn), row_bytes, L);
      dma_rows_k<NT>(v_hi, v_lo, sV, lane, Halves<NT>::J_MID + j0, Halves<NT>::J_END, PS_MOVERS, pair_off(pn), row_bytes, L);
      {
        float mk[MK];
#pragma unroll
        for (int i = 0; i < MK; ++i) mk[i] = mask_of(pn, mtid + i * 64 * PS_MOVERS);
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          if (j >= 32 * H1 && j < LP) sMask[j] = mk[i];
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      p = pn;
    }
    return;
  }
  if (wave >= n_sub) {
    // ---- nothing to compute: keep the barriers company ----
    phase_barrier();
    for (;;) {
      phase_barrier();
      phase_barrier();
      p += gridDim.x;
      if (p >= n_pairs) break;
    }
    return;
  }

  // ---- compute waves: sub-tile `wave` = 16 queries of every pair ----
  const int sub = wave;
  const int qn = lane & 15, g = lane >> 4;
  const int q_row = sub * 16 + qn;
  const bool q_ok = q_row < L;
  float* slab = sOut + wave * 16 * (32 + 4);
  const float scale = A.scale, scale2 = A.scale * LOG2E;
  const uint32_t kb[2] = {lds_addr(sK + d_off(qn, g)), lds_addr(sK + d_off(qn, g + 4))};
  const uint32_t vbk[2] = {lds_addr(sV + d_off(qn, g)), lds_addr(sV + d_off(qn, g + 4))};
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  uint32_t ktb[4];                                        // transposed reads of K: rows 32u + 4g + tq (+ 16)
#pragma unroll
  for (int n = 0; n < 4; ++n) ktb[n] = lds_addr(sK + d_off(4 * g + tq, 2 * n + (tp >> 1)) + 8 * (tp & 1));

  bf16x8_t qh[2], ql[2], gh[2], gl[2];
  float dD = 0.f, lse2 = 0.f;
  // this wave's rows of pair pp: Q and dO fragments, D = sum_d dO O, lse
  auto fetch = [&](int pp) {
    const int b = pp / heads, h = pp - b * heads;
    const size_t row0_ = (size_t)b * L;
    const uint32_t lr = (uint32_t)(q_ok ? q_row : 0);
    bf16x8_t oh[2], ol[2];
    const bool ld_ok = q_ok && !(LR2_SA_ABLATE & 128);
    load_frags_u(A.q + row0_ * (size_t)A.ld + h * HD, A.lo_off, lr * (uint32_t)A.ld + 8u * g, ld_ok, qh, ql);
    load_frags_u(A.go + row0_ * (size_t)A.ld_do + h * HD, A.do_lo_off, lr * (uint32_t)A.ld_do + 8u * g, ld_ok, gh, gl);
    load_frags_u(A.o + row0_ * (size_t)A.ld_o + h * HD, A.o_lo_off, lr * (uint32_t)A.ld_o + 8u * g, ld_ok, oh, ol);
    const size_t si0 = (size_t)pp * L;
    lse2 = q_ok ? A.lse[si0 + lr] * LOG2E : 0.f;
    float d = frag_dot(gh[0], gl[0], oh[0], ol[0]) + frag_dot(gh[1], gl[1], oh[1], ol[1]);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    dD = d;
    if (g == 0 && q_ok) A.dsum[si0 + lr] = d;
  };
  fetch(p);
  phase_barrier();
  for (;;) {
    const int b = p / heads, h = p - b * heads;
    const size_t row0 = (size_t)b * L;
    const uint64_t drow = (((uint64_t)b * heads + h) * L + (uint64_t)(q_ok ? q_row : 0)) * mask_pitch(L);
    f32x4_t o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      if (u == H1) phase_barrier();                      // mid
      float e[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * u + half;
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          a = mfma3(lds_ld16(kb[ks] + 2048 * t), lds_ld16(kb[ks] + 2048 * t + PLANE), qh[ks], ql[ks], a);
          d = mfma3(lds_ld16(vbk[ks] + 2048 * t), lds_ld16(vbk[ks] + 2048 * t + PLANE), gh[ks], gl[ks], d);
        }
        const float4 mk = *reinterpret_cast<const float4*>(sMask + 16 * t + 4 * g);
        const float mkv[4] = {mk.x, mk.y, mk.z, mk.w};
        if (DROP && A.dr.thr) d = drop_mul4v(A.dr, drow + 16 * t + 4 * g, d);      // dP = dPd o M
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], scale2, mkv[r] - lse2));
          e[4 * half + r] = pr * (d[r] - dD) * scale;
        }
      }
      bf16x8_t eh, el;
      split8(e, eh, el);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const uint32_t ad = ktb[n] + 4096 * u;
        o[n] = mfma3(eh, el, lds_tr_pair(ad, ad + 2048), lds_tr_pair(ad + PLANE, ad + 2048 + PLANE), o[n]);
      }
    }
    const int pn = p + gridDim.x;
    const bool more = pn < n_pairs;
    if (more) fetch(pn);                                 // the next pair's rows travel under the stores and the barrier
    store_tile_planes_half(o, slab, lane, sub * 16, L, A.dq, A.d_lo_off, (size_t)A.ld_d, row0 * (size_t)A.ld_d + h * HD);
    phase_barrier();                                     // end
    if (!more) break;
    p = pn;
  }
}

template <int NT, bool DROP>
__global__ __launch_bounds__(64 * PS_WAVES) void self_attn_bwd_dkv_persist_kernel(BwdArgs A) {
  constexpr int LP = 16 * NT, PLANE = LP * ROW_B, NB = Halves<NT>::NB, H1 = Halves<NT>::H1;
  constexpr int MK = (LP + 64 * PS_MOVERS - 1) / (64 * PS_MOVERS);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;                    // d_off layout: fragment reads (rows = queries) + transposed reads
  char* sG = smem + 2 * PLANE;        // dO
  float* sLse = reinterpret_cast<float*>(smem + 4 * PLANE);   // [LP] lse * log2(e); +inf for padded queries (P = 0)
  float* sD = sLse + LP;                                      // [LP]
  float* sOut = sD + LP;                                      // [PS_MAX_SUB waves][16][32 + 4]
  // DROP: the pair's dropout decisions, one byte per (sub-tile, 32-query block, lane) = the 8 (query, key) elements that lane owns in
  // that block, written by the mover waves one pair ahead (two buffers): the hashes cost the compute waves nothing -- neither issue
  // slots nor the registers their temporaries would need beside the fragments
  constexpr int KEEP_BYTES = PS_MAX_SUB * NB * 64;
  uint8_t* sKeep = reinterpret_cast<uint8_t*>(sOut + PS_MAX_SUB * 16 * (32 + 4));      // [2][KEEP_BYTES]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int heads = A.heads, L = A.L, n_pairs = A.n_pairs;
  const int n_sub = (L + 15) >> 4;
  int p = blockIdx.x;
  if (p >= n_pairs) return;

  if (wave >= PS_MAX_SUB) {
    // ---- movers ----
    const __amdgpu_buffer_rsrc_t q_hi = sa_rsrc(A.q, A.qkv_bytes), q_lo = sa_rsrc(A.q + A.lo_off, A.qkv_bytes);
    const __amdgpu_buffer_rsrc_t g_hi = sa_rsrc(A.go, A.do_bytes), g_lo = sa_rsrc(A.go + A.do_lo_off, A.do_bytes);
    const int j0 = wave - PS_MAX_SUB, mtid = tid - 64 * PS_MAX_SUB;
    const uint32_t q_row_bytes = (uint32_t)A.ld * 2u, g_row_bytes = (uint32_t)A.ld_do * 2u;
    auto q_off = [&](int pp) { const int b = pp / heads, h = pp - b * heads; return (uint32_t)(((size_t)b * L * A.ld + h * HD) * 2); };
    auto g_off = [&](int pp) { const int b = pp / heads, h = pp - b * heads; return (uint32_t)(((size_t)b * L * A.ld_do + h * HD) * 2); };
    // element index of (query q, key) of pair pp = pp L pitch + q pitch + key: pp L pitch is a multiple of 4 and the hash reads the low
    // 32 bits of (index >> 1) only, so everything per lane is 32-bit arithmetic (the same decisions as dropout_keep on the 64-bit index)
    auto write_keep = [&](int pp, uint8_t* dst) {
      const uint32_t pitch32 = (uint32_t)mask_pitch(L);
      const uint32_t pair_half32 = (uint32_t)(((uint64_t)pp * (uint64_t)L * mask_pitch(L)) >> 1);
      for (int i = mtid; i < n_sub * NB * 64; i += 64 * PS_MOVERS) {
        const int sub_ = i / (NB * 64), rem = i - sub_ * (NB * 64), u = rem >> 6, ln = rem & 63;
        const int key_ = sub_ * 16 + (ln & 15);
        const uint32_t col = (uint32_t)(key_ < L ? key_ : 0);
        uint32_t keep = 0;
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
          const int q = 32 * u + 16 * (e8 >> 2) + 4 * (ln >> 4) + (e8 & 3);
          const uint32_t x = (uint32_t)(q < L ? q : 0) * pitch32 + col;
          const uint32_t hsh = dropout_hash(A.dr.key, (uint64_t)(pair_half32 + (x >> 1)));
          keep |= (uint32_t)(((x & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= A.dr.thr) << e8;
        }
        dst[i] = (uint8_t)keep;
      }
    };
    dma_rows_k<NT>(q_hi, q_lo, sQ, lane, j0, Halves<NT>::J_END, PS_MOVERS, q_off(p), q_row_bytes, L);
    dma_rows_k<NT>(g_hi, g_lo, sG, lane, j0, Halves<NT>::J_END, PS_MOVERS, g_off(p), g_row_bytes, L);
    if (DROP && A.dr.thr) write_keep(p, sKeep);
    for (int j = mtid; j < LP; j += 64 * PS_MOVERS) {
      const size_t si = (size_t)p * L + j;
      sLse[j] = j < L ? A.lse[si] * LOG2E : INFINITY;
      sD[j] = j < L ? A.dsum[si] : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_barrier();
    for (int it = 0;; ++it) {
      const int pn = p + gridDim.x;
      const bool more = pn < n_pairs;
      phase_barrier();                                   // mid
      if (more) {
        dma_rows_k<NT>(q_hi, q_lo, sQ, lane, j0, Halves<NT>::J_MID, PS_MOVERS, q_off(pn), q_row_bytes, L);
        dma_rows_k<NT>(g_hi, g_lo, sG, lane, j0, Halves<NT>::J_MID, PS_MOVERS, g_off(pn), g_row_bytes, L);
        if (DROP && A.dr.thr) write_keep(pn, sKeep + ((it + 1) & 1) * KEEP_BYTES);     // that buffer's readers finished a pair ago
        float l2[MK], dd[MK];
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          const size_t si = (size_t)pn * L + (j < L ? j : 0);
          l2[i] = j < L ? A.lse[si] * LOG2E : INFINITY;
          dd[i] = j < L ? A.dsum[si] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          if (j < 32 * H1) { sLse[j] = l2[i]; sD[j] = dd[i]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      phase_barrier();                                   // end
      if (!more) break;
      dma_rows_k<NT>(q_hi, q_lo, sQ, lane, Halves<NT>::J_MID + j0, Halves<NT>::J_END, PS_MOVERS, q_off(pn), q_row_bytes, L);
      dma_rows_k<NT>(g_hi, g_lo, sG, lane, Halves<NT>::J_MID + j0, Halves<NT>::J_END, PS_MOVERS, g_off(pn), g_row_bytes, L);
      {
        float l2[MK], dd[MK];
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          const size_t si = (size_t)pn * L + (j < L ? j : 0);
          l2[i] = j < L ? A.lse[si] * LOG2E : INFINITY;
          dd[i] = j < L ? A.dsum[si] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MK; ++i) {
          const int j = mtid + i * 64 * PS_MOVERS;
          if (j >= 32 * H1 && j < LP) { sLse[j] = l2[i]; sD[j] = dd[i]; }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      p = pn;
    }
    return;
  }
  if (wave >= n_sub) {
    phase_barrier();
    for (;;) {
      phase_barrier();
      phase_barrier();
      p += gridDim.x;
      if (p >= n_pairs) break;
    }
    return;
  }

  // ---- compute waves: sub-tile `wave` = 16 keys of every pair ----
  const int sub = wave;
  const int kn = lane & 15, g = lane >> 4;
  const int key = sub * 16 + kn;
  const bool k_ok = key < L;
  float* slab = sOut + wave * 16 * (32 + 4);
  const float scale = A.scale, scale2 = A.scale * LOG2E;
  // fragment / transposed-read bases into sQ; the same row of sG is 2 * PLANE bytes further (added to the block's scalar offset)
  const uint32_t qb[2] = {lds_addr(sQ + d_off(kn, g)), lds_addr(sQ + d_off(kn, g + 4))};
  const uint32_t sq0 = lds_addr(sQ);
  bf16x8_t kh[2], kl[2], vh[2], vl[2];
  float kmask2 = 0.f;
  auto fetch = [&](int pp) {
    const int b = pp / heads, h = pp - b * heads;
    const size_t row0_ = (size_t)b * L;
    const int lane_ = opaque(lane);                       // addresses re-derived per pair, not kept live across the block loops
    const int key_ = sub * 16 + (lane_ & 15);
    const bool ok_ = key_ < L;
    const uint32_t lr = (uint32_t)(ok_ ? key_ : 0), lo8 = 8u * (uint32_t)(lane_ >> 4);
    const bool ld_ok = ok_ && !(LR2_SA_ABLATE & 128);
    load_frags_u(A.k + row0_ * (size_t)A.ld + h * HD, A.lo_off, lr * (uint32_t)A.ld + lo8, ld_ok, kh, kl);
    load_frags_u(A.v + row0_ * (size_t)A.ld + h * HD, A.lo_off, lr * (uint32_t)A.ld + lo8, ld_ok, vh, vl);
    kmask2 = ok_ ? ((A.seg[row0_ + lr] > 0) ? 0.f : -10000.0f * LOG2E) : -INFINITY;
  };
  fetch(p);
  phase_barrier();
  for (int it = 0;; ++it) {
    const int b = p / heads, h = p - b * heads;
    const size_t row0 = (size_t)b * L;
    const uint8_t* keep_buf = sKeep + (it & 1) * KEEP_BYTES;
    f32x4_t dv[4], dk[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) dv[n] = dk[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // one 32-query block: rolled loops (an unrolled pair lets the scheduler hoist seven blocks' worth of hashes and fragments)
    auto block = [&](int u) {
      const uint32_t uo = 4096u * (uint32_t)u, ug = uo + 2u * PLANE;          // block offsets into sQ / sG (uniform)
      // the block's 8 dropout decisions: one byte the mover waves left in LDS
      uint32_t keep = 0xffu;
      if (DROP && A.dr.thr) keep = keep_buf[(sub * NB + u) * 64 + opaque(lane)];
      float pd[8], ds[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        f32x4_t a = {0.f, 0.f, 0.f, 0.f}, d = a;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const uint32_t aq = qb[ks] + uo + 2048 * half, ag = qb[ks] + ug + 2048 * half;
          a = mfma3(lds_ld16(aq), lds_ld16(aq + PLANE), kh[ks], kl[ks], a);   // S[query 32u + 16 half + 4g + r][key kn]
          d = mfma3(lds_ld16(ag), lds_ld16(ag + PLANE), vh[ks], vl[ks], d);   // dPd
        }
        const float4 ls = *reinterpret_cast<const float4*>(sLse + 32 * u + 16 * half + 4 * g);
        const float4 dd = *reinterpret_cast<const float4*>(sD + 32 * u + 16 * half + 4 * g);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(a[r], scale2, kmask2 - lsv[r]));
          const float m = (DROP && A.dr.thr) ? (((keep >> (4 * half + r)) & 1u) ? A.dr.inv_keep : 0.0f) : 1.0f;
          pd[4 * half + r] = pr * m;
          ds[4 * half + r] = pr * (d[r] * m - ddv[r]) * scale;
        }
      }
      // transposed-read bases, re-derived per block from the lane number (a dozen integer instructions against four registers held
      // live across both loops): row 4g + tq of the block, 16-B unit 2n + (tp >> 1), 8-B half tp & 1
      uint32_t qtb[4];
      {
        const int lane_ = opaque(lane);
        const int tq = (lane_ >> 2) & 3, tp = lane_ & 3, rr = 4 * (lane_ >> 4) + tq;
#pragma unroll
        for (int n = 0; n < 4; ++n) qtb[n] = sq0 + (uint32_t)(d_off(rr, 2 * n + (tp >> 1)) + 8 * (tp & 1));
      }
      {
        bf16x8_t ph, pl;
        split8(pd, ph, pl);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const uint32_t ag = qtb[n] + ug;
          dv[n] = mfma3(ph, pl, lds_tr_pair(ag, ag + 2048), lds_tr_pair(ag + PLANE, ag + 2048 + PLANE), dv[n]);
        }
      }
      {
        bf16x8_t eh, el;
        split8(ds, eh, el);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const uint32_t aq = qtb[n] + uo;
          dk[n] = mfma3(eh, el, lds_tr_pair(aq, aq + 2048), lds_tr_pair(aq + PLANE, aq + 2048 + PLANE), dk[n]);
        }
      }
    };
#pragma unroll 1
    for (int u = 0; u < H1; ++u) block(u);
    phase_barrier();                                     // mid
#pragma unroll 1
    for (int u = H1; u < NB; ++u) block(u);
    const int pn = p + gridDim.x;
    const bool more = pn < n_pairs;
    if (more) fetch(pn);
    const size_t base = row0 * (size_t)A.ld_d + h * HD;
    store_tile_planes_half(dk, slab, opaque(lane), sub * 16, L, A.dk, A.d_lo_off, (size_t)A.ld_d, base);
    store_tile_planes_half(dv, slab, opaque(lane), sub * 16, L, A.dv, A.d_lo_off, (size_t)A.ld_d, base);
    phase_barrier();                                     // end
    if (!more) break;
    p = pn;
  }
}

// ---- backward for sequences beyond one LDS-resident block (L > 256): the same two kernels with a block loop ---------------
// dQ: the keys are walked in blocks of LPB = 16*NT, TWICE.  Sweep 1 keeps, per query, the running maximum m, the sum
// l = sum_k exp(s_k - m) and a = sum_k exp(s_k - m) dP_k (rescaled like the forward's accumulator when m grows), which give
// lse = m + log l and D = sum_k P_k dP_k = a / l without a second statistic pass; sweep 2 recomputes S and dP per block, forms
// dS = P (dP - D) * scale and accumulates dQ = dS K.  Five matrix products per (query, key) pair instead of three.
template <int NT>
__global__ __launch_bounds__(256) void self_attn_bwd_dq_blocked_kernel(
    const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh, const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
    const bf16_t* __restrict__ dOh, size_t do_lo_off, int ld_do, const int64_t* __restrict__ seg, bf16_t* __restrict__ dQh,
    size_t dq_lo_off, int ld_dq, float* __restrict__ lse, float* __restrict__ dsum, int heads, int L, float scale, DropP dr) {
  constexpr int LPB = 16 * NT;
  constexpr int PLANE = LPB * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + 2 * PLANE;        // K layout (V is an A operand here)
  float* sMask = reinterpret_cast<float*>(smem + 4 * PLANE);
  float* sOut = sMask + LPB;
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;
  const int qn = lane & 15, g = lane >> 4;
  const int sub = blockIdx.x * 4 + wave;             // one 16-query sub-tile per wave (waves past the end idle through the barriers)
  const int q_row = sub * 16 + qn;
  const bool q_ok = q_row < L;
  bf16x8_t qh[2], ql[2], gh[2], gl[2];
  load_frags(Qh, lo_off, (row0 + (q_ok ? q_row : 0)) * (size_t)ld + col0 + 8 * g, q_ok, qh, ql);
  load_frags(dOh, do_lo_off, (row0 + (q_ok ? q_row : 0)) * (size_t)ld_do + col0 + 8 * g, q_ok, gh, gl);
  const uint64_t drow = (((uint64_t)b * heads + h) * L + (uint64_t)(q_ok ? q_row : 0)) * mask_pitch(L);

  float m = -INFINITY, l = 0.f, a = 0.f, lse_q = 0.f, dd = 0.f;
  f32x4_t o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;

#pragma unroll 1
  for (int sweep = 0; sweep < 2; ++sweep) {
#pragma unroll 1
    for (int k0 = 0; k0 < L; k0 += LPB) {
      __syncthreads();                               // the previous block's readers are done
      for (int i = tid; i < LPB * 8; i += 256) {
        const int r = i >> 3, u = i & 7;
        u32x4_t kh = {0, 0, 0, 0}, kl = kh, vh = kh, vl = kh;
        if (k0 + r < L) {
          const size_t off = (row0 + k0 + r) * (size_t)ld + col0 + u * 8;
          kh = *reinterpret_cast<const u32x4_t*>(Kh + off);
          kl = *reinterpret_cast<const u32x4_t*>(Kh + off + lo_off);
          vh = *reinterpret_cast<const u32x4_t*>(Vh + off);
          vl = *reinterpret_cast<const u32x4_t*>(Vh + off + lo_off);
        }
        *reinterpret_cast<u32x4_t*>(sK + k_off(r, u)) = kh;
        *reinterpret_cast<u32x4_t*>(sK + PLANE + k_off(r, u)) = kl;
        *reinterpret_cast<u32x4_t*>(sV + k_off(r, u)) = vh;
        *reinterpret_cast<u32x4_t*>(sV + PLANE + k_off(r, u)) = vl;
      }
      for (int j = tid; j < LPB; j += 256) sMask[j] = (k0 + j < L) ? ((seg[row0 + k0 + j] > 0) ? 0.f : -10000.0f) : -INFINITY;
      __syncthreads();

      f32x4_t s[NT], dp[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, d = sa;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int r = 16 * t + qn;
          const bf16x8_t kh = *reinterpret_cast<const bf16x8_t*>(sK + k_off(r, g + 4 * ks));
          const bf16x8_t kl = *reinterpret_cast<const bf16x8_t*>(sK + PLANE + k_off(r, g + 4 * ks));
          const bf16x8_t vh = *reinterpret_cast<const bf16x8_t*>(sV + k_off(r, g + 4 * ks));
          const bf16x8_t vl = *reinterpret_cast<const bf16x8_t*>(sV + PLANE + k_off(r, g + 4 * ks));
          sa = mfma3(kh, kl, qh[ks], ql[ks], sa);
          d = mfma3(vh, vl, gh[ks], gl[ks], d);
        }
        const float4 mk = *reinterpret_cast<const float4*>(sMask + 16 * t + 4 * g);
        sa[0] = sa[0] * scale + mk.x;
        sa[1] = sa[1] * scale + mk.y;
        sa[2] = sa[2] * scale + mk.z;
        sa[3] = sa[3] * scale + mk.w;
        // dP = dPd o M (keys past L: P is 0 there, whatever the mask says)
        if (dr.thr) d = drop_mul4v(dr, drow + (uint64_t)(k0 + 16 * t + 4 * g), d);
        s[t] = sa;
        dp[t] = d;
      }
      if (sweep == 0) {
        float bm = -INFINITY;
#pragma unroll
        for (int t = 0; t < NT; ++t) bm = fmaxf(fmaxf(bm, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float m_new = fmaxf(m, bm);              // finite: every block holds at least one real key
        const float corr = exp_fast(m - m_new);        // 0 on the first block (m = -inf)
        l *= corr;
        a *= corr;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = exp_fast(s[t][r] - m_new);
            l += e;
            a = __builtin_fmaf(e, dp[t][r], a);
          }
        m = m_new;
      } else {
#pragma unroll
        for (int u = 0; u < NT / 2; ++u) {
          float e[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            e[r] = exp_fast(s[2 * u][r] - lse_q) * (dp[2 * u][r] - dd) * scale;
            e[4 + r] = exp_fast(s[2 * u + 1][r] - lse_q) * (dp[2 * u + 1][r] - dd) * scale;
          }
          bf16x8_t eh, el;
          split8(e, eh, el);
          const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const int unit = 2 * n + (tp >> 1), half8 = 8 * (tp & 1);
            o[n] = mfma3(eh, el, tr_pair_k(sK, ra, rb, unit, half8), tr_pair_k(sK + PLANE, ra, rb, unit, half8), o[n]);
          }
        }
      }
    }
    if (sweep == 0) {                                  // per-query statistics (lanes of one query hold disjoint key subsets)
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      lse_q = m + logf(l);
      dd = a / l;
      if (g == 0 && q_ok) {
        const size_t si = ((size_t)b * heads + h) * L + q_row;
        lse[si] = lse_q;
        dsum[si] = dd;
      }
    }
  }
  if (sub * 16 < L)
    store_tile_planes(o, sOut + wave * 16 * (HD + 4), lane, sub * 16, L, dQh, dq_lo_off, (size_t)ld_dq, row0 * (size_t)ld_dq + col0);
}

// dK, dV: each wave keeps its 16 keys' K / V fragments and accumulators while the queries (Q, dO, lse, D) pass through LDS in
// blocks of LPB.
template <int NT>
__global__ __launch_bounds__(256) void self_attn_bwd_dkv_blocked_kernel(
    const bf16_t* __restrict__ Qh, const bf16_t* __restrict__ Kh, const bf16_t* __restrict__ Vh, size_t lo_off, int ld,
    const bf16_t* __restrict__ dOh, size_t do_lo_off, int ld_do, const int64_t* __restrict__ seg, bf16_t* __restrict__ dKh,
    bf16_t* __restrict__ dVh, size_t dkv_lo_off, int ld_dkv, const float* __restrict__ lse, const float* __restrict__ dsum,
    int heads, int L, float scale, DropP dr) {
  constexpr int LPB = 16 * NT;
  constexpr int PLANE = LPB * ROW_B;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;
  char* sG = smem + 2 * PLANE;
  float* sLse = reinterpret_cast<float*>(smem + 4 * PLANE);
  float* sD = sLse + LPB;
  float* sOut = sD + LPB;
  const int h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row0 = (size_t)b * L;
  const int col0 = h * HD;
  const int kn = lane & 15, g = lane >> 4;
  const int sub = blockIdx.x * 4 + wave;
  const int key = sub * 16 + kn;
  const bool k_ok = key < L;
  bf16x8_t kh[2], kl[2], vh[2], vl[2];
  load_frags(Kh, lo_off, (row0 + (k_ok ? key : 0)) * (size_t)ld + col0 + 8 * g, k_ok, kh, kl);
  load_frags(Vh, lo_off, (row0 + (k_ok ? key : 0)) * (size_t)ld + col0 + 8 * g, k_ok, vh, vl);
  const float kmask = k_ok ? ((seg[row0 + key] > 0) ? 0.f : -10000.0f) : -INFINITY;
  f32x4_t dv[4], dk[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) dv[n] = dk[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
  const uint64_t dbase = ((uint64_t)b * heads + h) * (uint64_t)L;
#pragma unroll 1
  for (int q0 = 0; q0 < L; q0 += LPB) {
    __syncthreads();
    for (int i = tid; i < LPB * 8; i += 256) {
      const int r = i >> 3, u = i & 7;
      u32x4_t qa = {0, 0, 0, 0}, qc = qa, ga = qa, gc = qa;
      if (q0 + r < L) {
        const size_t off = (row0 + q0 + r) * (size_t)ld + col0 + u * 8;
        const size_t og = (row0 + q0 + r) * (size_t)ld_do + col0 + u * 8;
        qa = *reinterpret_cast<const u32x4_t*>(Qh + off);
        qc = *reinterpret_cast<const u32x4_t*>(Qh + off + lo_off);
        ga = *reinterpret_cast<const u32x4_t*>(dOh + og);
        gc = *reinterpret_cast<const u32x4_t*>(dOh + og + do_lo_off);
      }
      *reinterpret_cast<u32x4_t*>(sQ + k_off(r, u)) = qa;
      *reinterpret_cast<u32x4_t*>(sQ + PLANE + k_off(r, u)) = qc;
      *reinterpret_cast<u32x4_t*>(sG + k_off(r, u)) = ga;
      *reinterpret_cast<u32x4_t*>(sG + PLANE + k_off(r, u)) = gc;
    }
    for (int j = tid; j < LPB; j += 256) {
      const size_t si = ((size_t)b * heads + h) * L + q0 + j;
      sLse[j] = (q0 + j < L) ? lse[si] : INFINITY;     // padded queries: P = exp(-inf) = 0
      sD[j] = (q0 + j < L) ? dsum[si] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int u = 0; u < NT / 2; ++u) {
      float pd[8], ds[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = 2 * u + half;
        f32x4_t sa = {0.f, 0.f, 0.f, 0.f}, d = sa;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int r = 16 * t + kn;
          const bf16x8_t qh = *reinterpret_cast<const bf16x8_t*>(sQ + k_off(r, g + 4 * ks));
          const bf16x8_t ql = *reinterpret_cast<const bf16x8_t*>(sQ + PLANE + k_off(r, g + 4 * ks));
          const bf16x8_t gh = *reinterpret_cast<const bf16x8_t*>(sG + k_off(r, g + 4 * ks));
          const bf16x8_t gl = *reinterpret_cast<const bf16x8_t*>(sG + PLANE + k_off(r, g + 4 * ks));
          sa = mfma3(qh, ql, kh[ks], kl[ks], sa);
          d = mfma3(gh, gl, vh[ks], vl[ks], d);
        }
        const float4 ls = *reinterpret_cast<const float4*>(sLse + 16 * t + 4 * g);
        const float4 dd = *reinterpret_cast<const float4*>(sD + 16 * t + 4 * g);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, ddv[4] = {dd.x, dd.y, dd.z, dd.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = exp_fast(sa[r] * scale + kmask - lsv[r]);
          float mm = 1.0f;
          if (dr.thr) {
            const int q = q0 + 16 * t + 4 * g + r;
            mm = drop_mul(dr, (dbase + (uint64_t)(q < L ? q : 0)) * mask_pitch(L) + (uint64_t)(k_ok ? key : 0));
          }
          pd[4 * half + r] = p * mm;
          ds[4 * half + r] = p * (d[r] * mm - ddv[r]) * scale;
        }
      }
      bf16x8_t ph, pl, eh, el;
      split8(pd, ph, pl);
      split8(ds, eh, el);
      const int ra = 32 * u + 4 * g + tq, rb = ra + 16;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int unit = 2 * n + (tp >> 1), half8 = 8 * (tp & 1);
        dv[n] = mfma3(ph, pl, tr_pair_k(sG, ra, rb, unit, half8), tr_pair_k(sG + PLANE, ra, rb, unit, half8), dv[n]);
        dk[n] = mfma3(eh, el, tr_pair_k(sQ, ra, rb, unit, half8), tr_pair_k(sQ + PLANE, ra, rb, unit, half8), dk[n]);
      }
    }
  }
  if (sub * 16 < L) {
    float* slab = sOut + wave * 16 * (HD + 4);
    store_tile_planes(dk, slab, lane, sub * 16, L, dKh, dkv_lo_off, (size_t)ld_dkv, row0 * (size_t)ld_dkv + col0);
    store_tile_planes(dv, slab, lane, sub * 16, L, dVh, dkv_lo_off, (size_t)ld_dkv, row0 * (size_t)ld_dkv + col0);
  }
}

static DropP make_drop(float p, uint64_t seed, uint32_t site) {
  DropP d{0, 0, 1.0f};
  if (p > 0.f) {
    d.thr = dropout_threshold(p);
    d.inv_keep = 1.0f / (1.0f - p);
    d.key = (((uint64_t)site) << 40) ^ (seed * 0x9E3779B97F4A7C15ull);
  }
  return d;
}

template <typename Kern>
int allow_lds_once(Kern kern, size_t lds, bool& done, const char* what) {
  if (!done) {
    if (lr2_allow_dynamic_lds(kern, lds, what)) return LR2_ERR_LAUNCH;
    done = true;
  }
  return 0;
}

// grid.x: how many workgroups share one (sequence, head).  One is best (K/V or Q/dO are staged once) as long as the grid
// still fills the chip; small batches split the sub-tiles over up to 4 workgroups.
static int attn_chunks(int batch, int heads, int L) {
  const int n_sub = (L + 15) / 16, max_chunks = (n_sub + 3) / 4;
  int c = (512 + batch * heads - 1) / (batch * heads);
  if (c < 1) c = 1;
  return c > max_chunks ? max_chunks : c;
}

struct AttnArgs {
  const bf16_t *q, *k, *v;
  size_t lo_off;
  int ld;
  const int64_t* seg;
  int batch, heads, L;
  float scale;
  DropP dr;
  hipStream_t stream;
};

// Forward: 8 waves per workgroup (2 per SIMD) hide the LDS-read latency of the dependent tile chains; the 256-key
// variant keeps 4 (its K/V planes + 8 output slabs would not fit the 160 KiB of LDS).
static bool attn_persist_enabled();
static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
  }
  return n;
}

// The persistent form: at least one pair per CU, every operand offset a 32-bit byte count (LR2_ATTN_PERSIST=0: an A/B switch, read
// once per process).
template <int NT>
int launch_fwd_persist(const AttnArgs& a, float* o, bf16_t* oh, size_t o_lo_off, int ld_o, float* lse, uint32_t kv_bytes) {
  constexpr int LP = 16 * NT;
  const size_t lds = (size_t)4 * LP * ROW_B + (size_t)2 * LP * 4 + (size_t)PS_MAX_SUB * 16 * (32 + 4) * 4;
  static bool done = false;
  static bool done_drop = false;
  if (allow_lds_once(self_attn_persist_kernel<NT, false>, lds, done, "self_attn_fwd(persistent)")) return LR2_ERR_LAUNCH;
  if (allow_lds_once(self_attn_persist_kernel<NT, true>, lds, done_drop, "self_attn_fwd(persistent, dropout)")) return LR2_ERR_LAUNCH;
  const int n_pairs = a.batch * a.heads;
  const int grid = n_pairs < cu_count() ? n_pairs : cu_count();
  if (a.dr.thr)
    LR2_LAUNCH((self_attn_persist_kernel<NT, true>), dim3(grid), dim3(64 * PS_WAVES), lds, a.stream, a.q, a.k, a.v, a.lo_off, a.ld, a.seg, o,
               oh, o_lo_off, ld_o, a.heads, a.L, a.scale, lse, a.dr, n_pairs, kv_bytes);
  else
    LR2_LAUNCH((self_attn_persist_kernel<NT, false>), dim3(grid), dim3(64 * PS_WAVES), lds, a.stream, a.q, a.k, a.v, a.lo_off, a.ld, a.seg, o,
               oh, o_lo_off, ld_o, a.heads, a.L, a.scale, lse, a.dr, n_pairs, kv_bytes);
  return lr2_launch_status("lr2_self_attn_fwd(persistent)");
}

template <int NT>
int launch_fwd(const AttnArgs& a, float* o, bf16_t* oh, size_t o_lo_off, int ld_o, float* lse) {
  constexpr int LP = 16 * NT;
  constexpr int NW = NT <= 14 ? 8 : 4;
  if constexpr (NT <= 14) {
    const bool persist_on = attn_persist_enabled();
    // bytes a K / V descriptor spans from its first element: the last row's head columns end (rows - 1) * ld + heads * 64 elements on
    const uint64_t span = ((uint64_t)a.batch * a.L - 1) * (uint64_t)a.ld * 2u + (uint64_t)a.heads * HD * 2u;
    if (persist_on && a.batch * a.heads >= cu_count() && (a.L + 15) / 16 <= PS_MAX_SUB && span < 0xFFFFFF00ull)
      return launch_fwd_persist<NT>(a, o, oh, o_lo_off, ld_o, lse, (uint32_t)span);
  }
  const size_t lds = (size_t)4 * LP * ROW_B + (size_t)LP * 4 + (size_t)NW * 16 * (HD + 4) * 4;
  static bool done = false;
  if (allow_lds_once(self_attn_mfma_kernel<NT, NW>, lds, done, "self_attn_fwd")) return LR2_ERR_LAUNCH;
  const int n_sub = (a.L + 15) / 16, max_chunks = (n_sub + NW - 1) / NW;
  int chunks = attn_chunks(a.batch, a.heads, a.L);
  if (chunks > max_chunks) chunks = max_chunks;
  LR2_LAUNCH((self_attn_mfma_kernel<NT, NW>), dim3(chunks, a.heads, a.batch), dim3(64 * NW), lds, a.stream, a.q, a.k, a.v,
             a.lo_off, a.ld, a.seg, o, oh, o_lo_off, ld_o, a.heads, a.L, a.scale, lse, a.dr);
  return lr2_launch_status("lr2_self_attn_fwd");
}

// L > 256: key blocks of 16 * NT keys, nb = ceil(L / 224) blocks of equal (rounded) size; SLOTS sub-tiles of 16 queries per
// wave, the query range split over gridDim.x workgroups when a sequence has more than 8 * SLOTS sub-tiles.
template <int NT, int SLOTS>
int launch_fwd_blocked(const AttnArgs& a, float* o, bf16_t* oh, size_t o_lo_off, int ld_o, float* lse, int n_blocks) {
  constexpr int LP = 16 * NT, NW = 8;
  const size_t lds = (size_t)4 * LP * ROW_B + (size_t)LP * 4 + (size_t)NW * 16 * (HD + 4) * 4;
  static bool done = false;
  if (allow_lds_once(self_attn_blocked_kernel<NT, NW, SLOTS>, lds, done, "self_attn_fwd(blocked)")) return LR2_ERR_LAUNCH;
  const int n_sub = (a.L + 15) / 16;
  const int chunks = (n_sub + NW * SLOTS - 1) / (NW * SLOTS);
  LR2_LAUNCH((self_attn_blocked_kernel<NT, NW, SLOTS>), dim3(chunks, a.heads, a.batch), dim3(64 * NW), lds, a.stream, a.q, a.k, a.v,
             a.lo_off, a.ld, a.seg, o, oh, o_lo_off, ld_o, a.heads, a.L, a.scale, lse, a.dr, n_blocks);
  return lr2_launch_status("lr2_self_attn_fwd(blocked)");
}

static int fwd_blocked_dispatch(const AttnArgs& a, float* o, bf16_t* oh, size_t o_lo_off, int ld_o, float* lse) {
  const int nb = (a.L + 223) / 224;
  const int tiles = (((a.L + nb - 1) / nb) + 15) / 16;       // key tiles per block
  const int n_sub = (a.L + 15) / 16;
  const bool few = n_sub <= 8 * 3;                           // 3 sub-tile slots per wave are enough (L <= 384); else 4 (+ query chunks)
#define BLK(NT)                                                                                        \
  return few ? launch_fwd_blocked<NT, 3>(a, o, oh, o_lo_off, ld_o, lse, (a.L + 16 * NT - 1) / (16 * NT)) \
             : launch_fwd_blocked<NT, 4>(a, o, oh, o_lo_off, ld_o, lse, (a.L + 16 * NT - 1) / (16 * NT));
  if (tiles <= 10) { BLK(10) }
  if (tiles <= 12) { BLK(12) }
  // 14 key tiles: 3 query sub-tiles per wave always (the 4-slot variant needs 11 VGPRs more than the 256 of two waves per SIMD)
  return launch_fwd_blocked<14, 3>(a, o, oh, o_lo_off, ld_o, lse, (a.L + 16 * 14 - 1) / (16 * 14));
#undef BLK
}

// The backward's persistent form needs the forward's output and log-sum-exp (o != nullptr: lse is then an INPUT), at least one pair per
// CU, 32-bit byte offsets into every operand.  (LR2_ATTN_PERSIST=0: the A/B switch of the forward applies here too.)
static bool attn_persist_enabled() {
  static const bool on = !(getenv("LR2_ATTN_PERSIST") && atoi(getenv("LR2_ATTN_PERSIST")) == 0);
  return on;
}
static bool bwd_persist_ok(int batch, int heads, int L, int ld, int ld_do, bool has_o) {
  const uint64_t span = ((uint64_t)batch * L - 1) * (uint64_t)ld * 2u + (uint64_t)heads * HD * 2u;
  const uint64_t span_do = ((uint64_t)batch * L - 1) * (uint64_t)ld_do * 2u + (uint64_t)heads * HD * 2u;
  return attn_persist_enabled() && has_o && L <= 16 * PS_MAX_SUB && batch * heads >= cu_count() && span < 0xFFFFFF00ull &&
         span_do < 0xFFFFFF00ull;
}

template <int NT>
int launch_bwd_persist(const AttnArgs& a, const bf16_t* go, size_t do_lo_off, int ld_do, const bf16_t* o, size_t o_lo_off, int ld_o,
                       bf16_t* dq, bf16_t* dk, bf16_t* dv, size_t d_lo_off, int ld_d, const float* lse, float* dsum) {
  constexpr int LP = 16 * NT;
  const size_t lds1 = (size_t)4 * LP * ROW_B + (size_t)LP * 4 + (size_t)PS_MAX_SUB * 16 * (32 + 4) * 4;
  const size_t lds2 = (size_t)4 * LP * ROW_B + (size_t)LP * 8 + (size_t)PS_MAX_SUB * 16 * (32 + 4) * 4;
  const size_t lds2_drop = lds2 + (size_t)2 * PS_MAX_SUB * (NT / 2) * 64;       // + the dropout decision bytes of two pairs
  static bool d1 = false, d2 = false, d3 = false, d4 = false;
  if (allow_lds_once(self_attn_bwd_dq_persist_kernel<NT, false>, lds1, d1, "self_attn_bwd_dq(persistent)")) return LR2_ERR_LAUNCH;
  if (allow_lds_once(self_attn_bwd_dq_persist_kernel<NT, true>, lds1, d2, "self_attn_bwd_dq(persistent, dropout)")) return LR2_ERR_LAUNCH;
  if (allow_lds_once(self_attn_bwd_dkv_persist_kernel<NT, false>, lds2, d3, "self_attn_bwd_dkv(persistent)")) return LR2_ERR_LAUNCH;
  if (allow_lds_once(self_attn_bwd_dkv_persist_kernel<NT, true>, lds2_drop, d4, "self_attn_bwd_dkv(persistent, dropout)")) return LR2_ERR_LAUNCH;
  BwdArgs A{};
  A.q = a.q; A.k = a.k; A.v = a.v; A.lo_off = a.lo_off; A.ld = a.ld;
  A.go = go; A.do_lo_off = do_lo_off; A.ld_do = ld_do;
  A.o = o; A.o_lo_off = o_lo_off; A.ld_o = ld_o;
  A.seg = a.seg; A.dq = dq; A.dk = dk; A.dv = dv; A.d_lo_off = d_lo_off; A.ld_d = ld_d;
  A.lse = lse; A.dsum = dsum; A.heads = a.heads; A.L = a.L; A.n_pairs = a.batch * a.heads; A.scale = a.scale; A.dr = a.dr;
  A.qkv_bytes = (uint32_t)(((uint64_t)a.batch * a.L - 1) * (uint64_t)a.ld * 2u + (uint64_t)a.heads * HD * 2u);
  A.do_bytes = (uint32_t)(((uint64_t)a.batch * a.L - 1) * (uint64_t)ld_do * 2u + (uint64_t)a.heads * HD * 2u);
  const int grid = A.n_pairs < cu_count() ? A.n_pairs : cu_count();
  if (a.dr.thr) {
    LR2_LAUNCH((self_attn_bwd_dq_persist_kernel<NT, true>), dim3(grid), dim3(64 * PS_WAVES), lds1, a.stream, A);
    if (lr2_launch_status("lr2_self_attn_bwd(dq, persistent)")) return LR2_ERR_LAUNCH;
    LR2_LAUNCH((self_attn_bwd_dkv_persist_kernel<NT, true>), dim3(grid), dim3(64 * PS_WAVES), lds2_drop, a.stream, A);
  } else {
    LR2_LAUNCH((self_attn_bwd_dq_persist_kernel<NT, false>), dim3(grid), dim3(64 * PS_WAVES), lds1, a.stream, A);
    if (lr2_launch_status("lr2_self_attn_bwd(dq, persistent)")) return LR2_ERR_LAUNCH;
    LR2_LAUNCH((self_attn_bwd_dkv_persist_kernel<NT, false>), dim3(grid), dim3(64 * PS_WAVES), lds2, a.stream, A);
  }
  return lr2_launch_status("lr2_self_attn_bwd(dkv, persistent)");
}

template <int NT>
int launch_bwd(const AttnArgs& a, const bf16_t* go, size_t do_lo_off, int ld_do, bf16_t* dq, bf16_t* dk, bf16_t* dv,
               size_t d_lo_off, int ld_d, float* lse, float* dsum) {
  constexpr int LP = 16 * NT;
  constexpr int NW = NT <= 14 ? 8 : 4;          // two waves per SIMD where the K / V (Q / dO) planes + 8 output slabs fit the LDS
  const size_t lds1 = (size_t)4 * LP * ROW_B + (size_t)LP * 4 + (size_t)NW * 16 * (HD + 4) * 4;
  const size_t lds2 = (size_t)4 * LP * ROW_B + (size_t)LP * 8 + (size_t)NW * 16 * (HD + 4) * 4;
  static bool done1 = false, done2 = false;
  if (allow_lds_once(self_attn_bwd_dq_kernel<NT, NW>, lds1, done1, "self_attn_bwd_dq")) return LR2_ERR_LAUNCH;
  if (allow_lds_once(self_attn_bwd_dkv_kernel<NT, NW>, lds2, done2, "self_attn_bwd_dkv")) return LR2_ERR_LAUNCH;
  const int n_sub = (a.L + 15) / 16, max_chunks = (n_sub + NW - 1) / NW;
  int chunks = attn_chunks(a.batch, a.heads, a.L);
  if (chunks > max_chunks) chunks = max_chunks;
  const dim3 grid(chunks, a.heads, a.batch);
  LR2_LAUNCH((self_attn_bwd_dq_kernel<NT, NW>), grid, dim3(64 * NW), lds1, a.stream, a.q, a.k, a.v, a.lo_off, a.ld, go, do_lo_off,
             ld_do, a.seg, dq, d_lo_off, ld_d, lse, dsum, a.heads, a.L, a.scale, a.dr);
  if (lr2_launch_status("lr2_self_attn_bwd(dq)")) return LR2_ERR_LAUNCH;
  LR2_LAUNCH((self_attn_bwd_dkv_kernel<NT, NW>), grid, dim3(64 * NW), lds2, a.stream, a.q, a.k, a.v, a.lo_off, a.ld, go, do_lo_off,
             ld_do, a.seg, dk, dv, d_lo_off, ld_d, (const float*)lse, (const float*)dsum, a.heads, a.L, a.scale, a.dr);
  return lr2_launch_status("lr2_self_attn_bwd(dkv)");
}

// L > 256: query / key blocks of 128 rows through 64 KiB of LDS (two workgroups per CU)
int launch_bwd_blocked(const AttnArgs& a, const bf16_t* go, size_t do_lo_off, int ld_do, bf16_t* dq, bf16_t* dk, bf16_t* dv,
                       size_t d_lo_off, int ld_d, float* lse, float* dsum) {
  constexpr int NT = 8, LPB = 16 * NT;
  const size_t lds1 = (size_t)4 * LPB * ROW_B + (size_t)LPB * 4 + (size_t)4 * 16 * (HD + 4) * 4;
  const size_t lds2 = (size_t)4 * LPB * ROW_B + (size_t)LPB * 8 + (size_t)4 * 16 * (HD + 4) * 4;
  static bool done1 = false, done2 = false;
  if (allow_lds_once(self_attn_bwd_dq_blocked_kernel<NT>, lds1, done1, "self_attn_bwd_dq_blocked")) return LR2_ERR_LAUNCH;
  if (allow_lds_once(self_attn_bwd_dkv_blocked_kernel<NT>, lds2, done2, "self_attn_bwd_dkv_blocked")) return LR2_ERR_LAUNCH;
  const int n_sub = (a.L + 15) / 16;
  const dim3 grid((n_sub + 3) / 4, a.heads, a.batch);
  LR2_LAUNCH(self_attn_bwd_dq_blocked_kernel<NT>, grid, dim3(256), lds1, a.stream, a.q, a.k, a.v, a.lo_off, a.ld, go, do_lo_off,
             ld_do, a.seg, dq, d_lo_off, ld_d, lse, dsum, a.heads, a.L, a.scale, a.dr);
  if (lr2_launch_status("lr2_self_attn_bwd(dq, blocked)")) return LR2_ERR_LAUNCH;
  LR2_LAUNCH(self_attn_bwd_dkv_blocked_kernel<NT>, grid, dim3(256), lds2, a.stream, a.q, a.k, a.v, a.lo_off, a.ld, go, do_lo_off,
             ld_do, a.seg, dk, dv, d_lo_off, ld_d, (const float*)lse, (const float*)dsum, a.heads, a.L, a.scale, a.dr);
  return lr2_launch_status("lr2_self_attn_bwd(dkv, blocked)");
}

#define LR2_SA_INST_FWD(NT)                                                                                                    \
  template __global__ void self_attn_mfma_kernel<NT, (NT <= 14 ? 8 : 4)>(const bf16_t*, const bf16_t*, const bf16_t*, size_t,  \
                                                                         int, const int64_t*, float*, bf16_t*, size_t, int,   \
                                                                         int, int, float, float*, DropP);
LR2_SA_INST_FWD(4)
LR2_SA_INST_FWD(8)
LR2_SA_INST_FWD(14)
#define LR2_SA_INST(NT)                                                                                                        \
  template __global__ void self_attn_bwd_dq_kernel<NT, (NT <= 14 ? 8 : 4)>(const bf16_t*, const bf16_t*, const bf16_t*, size_t, int, const bf16_t*, \
                                                       size_t, int, const int64_t*, bf16_t*, size_t, int, float*, float*, int,  \
                                                       int, float, DropP);                                                     \
  template __global__ void self_attn_bwd_dkv_kernel<NT, (NT <= 14 ? 8 : 4)>(const bf16_t*, const bf16_t*, const bf16_t*, size_t, int,               \
                                                        const bf16_t*, size_t, int, const int64_t*, bf16_t*, bf16_t*, size_t,   \
                                                        int, const float*, const float*, int, int, float, DropP);
LR2_SA_INST(4)
LR2_SA_INST(8)
LR2_SA_INST(14)
LR2_SA_INST(16)

// ---------------------------------------------------------------------------------------------------------------------
// Attention for the FIRST query of every sequence only (the [CLS] row): what pooling 'first' (utils/misc.py:23-35 upstream)
// keeps of the last encoder layer.  One workgroup per (sequence, head): scores of the one query against all L keys,
// fp32 softmax with the same additive key mask and the same exp as the full kernels, then P V.  K / V rows are read as
// whole 128-B hi and lo rows (bf16 planes, x = hi + lo).  ~0.2 % of the full layer's attention work: a vector kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FT_THREADS = 256;
constexpr int FT_MAXL = 4096;

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__global__ __launch_bounds__(FT_THREADS) void first_token_attn_kernel(const float* __restrict__ q, int ld_q,
                                                                       const bf16_t* __restrict__ k_hi,
                                                                       const bf16_t* __restrict__ v_hi, size_t lo_off, int ld,
                                                                       const int64_t* __restrict__ seg, float* __restrict__ o,
                                                                       int ld_o, int heads, int L, float scale) {
  // Round 4: 8 threads per key row (16 B of the hi plane + 16 B of the lo plane each: the 128-byte row segment of a head is one
  // coalesced request), 32 rows per sweep of the workgroup, for the scores AND for P V -- the first version read a row per thread
  // (64 cache lines per wave instruction) and V two bytes per lane: 245 us at 512 x 12 x 197 = 2.5 TB/s.
  __shared__ float sq[HD];
  __shared__ float sp[FT_MAXL];
  __shared__ float red[FT_THREADS / 64];
  __shared__ float so[FT_THREADS / 8][HD + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rr = tid >> 3, c8 = (tid & 7) * 8;            // row within a sweep, first of this thread's 8 head columns
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const size_t row0 = (size_t)b * L;
  if (tid < HD) sq[tid] = q[(size_t)b * ld_q + h * HD + tid];
  __syncthreads();
  float qv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qv[i] = sq[c8 + i];
  float mx = -INFINITY;
  for (int j0 = 0; j0 < L; j0 += FT_THREADS / 8) {
    const int j = j0 + rr;
    float acc = 0.f;
    if (j < L) {
      const bf16_t* kr = k_hi + (row0 + j) * (size_t)ld + h * HD + c8;
      const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(kr);
      const u32x4_t lv = *reinterpret_cast<const u32x4_t*>(kr + lo_off);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        acc = __builtin_fmaf(qv[2 * w], bf_lo(hv[w]) + bf_lo(lv[w]), acc);
        acc = __builtin_fmaf(qv[2 * w + 1], bf_hi(hv[w]) + bf_hi(lv[w]), acc);
      }
    }
    // the row's 8 partial sums sit in 8 consecutive lanes
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (j < L) {
      const float sc = acc * scale + ((seg[row0 + j] > 0) ? 0.f : -10000.0f);
      if ((tid & 7) == 0) sp[j] = sc;
      mx = fmaxf(mx, sc);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < L; j += FT_THREADS) {
    const float e = exp_fast(sp[j] - mx);
    sp[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  // O = P V: thread (rr, c8) sums its 8 columns over the rows rr, rr + 32, ...; the 32 row groups are combined through LDS
  float ov[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ov[i] = 0.f;
  for (int j = rr; j < L; j += FT_THREADS / 8) {
    const bf16_t* vr = v_hi + (row0 + j) * (size_t)ld + h * HD + c8;
    const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(vr);
    const u32x4_t lv = *reinterpret_cast<const u32x4_t*>(vr + lo_off);
    const float pj = sp[j];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      ov[2 * w] = __builtin_fmaf(pj, bf_lo(hv[w]) + bf_lo(lv[w]), ov[2 * w]);
      ov[2 * w + 1] = __builtin_fmaf(pj, bf_hi(hv[w]) + bf_hi(lv[w]), ov[2 * w + 1]);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) so[rr][c8 + i] = ov[i];
  __syncthreads();
  if (tid < HD) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < FT_THREADS / 8; ++r) t += so[r][tid];
    o[(size_t)b * ld_o + h * HD + tid] = t / sum;
  }
}

}  // namespace

#define LR2_SA_DISPATCH(L, CALL)  \
  if ((L) <= 64) return CALL(4);  \
  if ((L) <= 128) return CALL(8); \
  if ((L) <= 224) return CALL(14); \
  return CALL(16);

extern "C" int lr2_self_attn_fwd(const void* q_hi, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld,
                                 const int64_t* seg, void* o, void* o_hi, uint64_t o_lo_off, int ld_o, void* lse, float drop_p,
                                 uint64_t drop_seed, uint32_t drop_site, int batch, int heads, int L, int head_dim, float scale,
                                 void* stream) {
  if (!q_hi || !k_hi || !v_hi || !seg || (!o && !o_hi) || batch <= 0 || heads <= 0) return LR2_ERR_ARG;
  if (head_dim != HD || L < 1 || ld % 8 || ld_o % 4 || lo_off % 8 || o_lo_off % 4) return LR2_ERR_SHAPE;
  if (drop_p < 0.f || drop_p >= 1.f) return LR2_ERR_ARG;
  const AttnArgs a{(const bf16_t*)q_hi, (const bf16_t*)k_hi, (const bf16_t*)v_hi, (size_t)lo_off, ld, seg, batch, heads, L,
                   scale, make_drop(drop_p, drop_seed, drop_site), (hipStream_t)stream};
  // L > 224: key blocks with a running max / sum.  (A one-block kernel for 225 <= L <= 256 -- 16 key tiles, 4 waves -- needs
  // 512 VGPRs + 710 spilled: the two-block walk of the blocked kernel is the forward for those lengths.)
  if (L > 224) return fwd_blocked_dispatch(a, (float*)o, (bf16_t*)o_hi, (size_t)o_lo_off, ld_o, (float*)lse);
  if (L <= 64) return launch_fwd<4>(a, (float*)o, (bf16_t*)o_hi, (size_t)o_lo_off, ld_o, (float*)lse);
  if (L <= 128) return launch_fwd<8>(a, (float*)o, (bf16_t*)o_hi, (size_t)o_lo_off, ld_o, (float*)lse);
  return launch_fwd<14>(a, (float*)o, (bf16_t*)o_hi, (size_t)o_lo_off, ld_o, (float*)lse);
}

extern "C" int lr2_self_attn_bwd(const void* q_hi, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld,
                                 const void* do_hi, uint64_t do_lo_off, int ld_do, const int64_t* seg, void* dq_hi, void* dk_hi,
                                 void* dv_hi, uint64_t d_lo_off, int ld_d, const void* o_hi, uint64_t o_lo_off, int ld_o,
                                 void* lse_ws, void* dsum_ws, float drop_p, uint64_t drop_seed, uint32_t drop_site, int batch,
                                 int heads, int L, int head_dim, float scale, void* stream) {
  if (!q_hi || !k_hi || !v_hi || !do_hi || !seg || !dq_hi || !dk_hi || !dv_hi || !lse_ws || !dsum_ws || batch <= 0 || heads <= 0)
    return LR2_ERR_ARG;
  if (head_dim != HD || L < 1 || ld % 8 || ld_do % 8 || ld_d % 8 || lo_off % 8 || do_lo_off % 8 || d_lo_off % 8)
    return LR2_ERR_SHAPE;
  if (o_hi && (ld_o % 8 || o_lo_off % 8)) return LR2_ERR_SHAPE;
  if (drop_p < 0.f || drop_p >= 1.f) return LR2_ERR_ARG;
  const AttnArgs a{(const bf16_t*)q_hi, (const bf16_t*)k_hi, (const bf16_t*)v_hi, (size_t)lo_off, ld, seg, batch, heads, L,
                   scale, make_drop(drop_p, drop_seed, drop_site), (hipStream_t)stream};
  if (L > 256)
    return launch_bwd_blocked(a, (const bf16_t*)do_hi, (size_t)do_lo_off, ld_do, (bf16_t*)dq_hi, (bf16_t*)dk_hi, (bf16_t*)dv_hi,
                              (size_t)d_lo_off, ld_d, (float*)lse_ws, (float*)dsum_ws);
  if (bwd_persist_ok(batch, heads, L, ld, ld_do, o_hi != nullptr)) {
#define CALLP(NT)                                                                                                                      \
  launch_bwd_persist<NT>(a, (const bf16_t*)do_hi, (size_t)do_lo_off, ld_do, (const bf16_t*)o_hi, (size_t)o_lo_off, ld_o, (bf16_t*)dq_hi, \
                         (bf16_t*)dk_hi, (bf16_t*)dv_hi, (size_t)d_lo_off, ld_d, (const float*)lse_ws, (float*)dsum_ws)
    if (L <= 64) return CALLP(4);
    if (L <= 128) return CALLP(8);
    return CALLP(14);
#undef CALLP
  }
#define CALL(NT)                                                                                                           \
  launch_bwd<NT>(a, (const bf16_t*)do_hi, (size_t)do_lo_off, ld_do, (bf16_t*)dq_hi, (bf16_t*)dk_hi, (bf16_t*)dv_hi,        \
                 (size_t)d_lo_off, ld_d, (float*)lse_ws, (float*)dsum_ws)
  LR2_SA_DISPATCH(L, CALL)
#undef CALL
}

// Which form of the attention kernels a call of this shape runs (a pure function of the shape, the device's CU count and
// LR2_ATTN_PERSIST): *fwd_persistent / *bwd_persistent = 1 when lr2_self_attn_fwd / lr2_self_attn_bwd (the latter given o_hi) take the
// persistent kernels.  Tests assert on it; ld / ld_do as in the calls.
extern "C" int lr2_self_attn_plan(int batch, int heads, int L, int ld, int ld_do, int* fwd_persistent, int* bwd_persistent) {
  if (batch <= 0 || heads <= 0 || L < 1) return LR2_ERR_ARG;
  const uint64_t span = ((uint64_t)batch * L - 1) * (uint64_t)ld * 2u + (uint64_t)heads * HD * 2u;
  if (fwd_persistent)
    *fwd_persistent = attn_persist_enabled() && L <= 16 * PS_MAX_SUB && batch * heads >= cu_count() && span < 0xFFFFFF00ull;
  if (bwd_persistent) *bwd_persistent = bwd_persist_ok(batch, heads, L, ld, ld_do, true);
  return 0;
}

extern "C" int lr2_first_token_attn(const void* q, int ld_q, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld,
                                    const int64_t* seg, void* o, int ld_o, int batch, int heads, int L, int head_dim,
                                    float scale, void* stream) {
  if (!q || !k_hi || !v_hi || !seg || !o || batch <= 0 || heads <= 0) return LR2_ERR_ARG;
  if (head_dim != HD || L < 1 || L > FT_MAXL || ld % 8 || lo_off % 8) return LR2_ERR_SHAPE;
  LR2_LAUNCH(first_token_attn_kernel, dim3(batch * heads), dim3(FT_THREADS), 0, (hipStream_t)stream, (const float*)q, ld_q,
             (const bf16_t*)k_hi, (const bf16_t*)v_hi, (size_t)lo_off, ld, seg, (float*)o, ld_o, heads, L, scale);
  return lr2_launch_status(__func__);
}
