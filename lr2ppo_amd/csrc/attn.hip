// Attention cores for gfx950.
//  * XiT cross-attention (finetune/xit.py:125-148): Lq = 196 text tokens against Lk = 16 image tokens, 8 heads
//    of 96; softmax(Q K^T) with NO pre-scale, probabilities divided by sqrt(768) afterwards; the "causal" mask
//    of the reference is a no-op (xit.py:140) and is therefore not implemented.  K/V of one (sequence, head)
//    are 2 x 6 KiB and live in LDS; one lane owns one query row, the 16 scores stay in registers, softmax
//    needs no cross-lane traffic at all.
//  * TencentPretrain self-attention (layers/multi_headed_attn.py:61-74): L <= 256, head 64, additive
//    -10000 key mask from `seg` applied after the 1/sqrt(d) scale.
#include "common.h"
#include "lr2ppo_hip.h"

namespace {

constexpr int MAX_LK = 16;
constexpr int MAX_HD = 96;

// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void xattn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, float* __restrict__ O, int o_planes,
                                                        size_t o_lo_off, int heads, int Lq, int Lk, int hd,
                                                        float post_scale) {
  __shared__ __attribute__((aligned(16))) float sK[MAX_LK][MAX_HD];
  __shared__ __attribute__((aligned(16))) float sV[MAX_LK][MAX_HD];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int E = heads * hd;
  const int t = threadIdx.x;
  for (int idx = t; idx < MAX_LK * hd; idx += 256) {
    const int j = idx / hd, d = idx % hd;
    float kv = 0.f, vv = 0.f;
    if (j < Lk) {
      const size_t o = ((size_t)b * Lk + j) * E + (size_t)h * hd + d;
      kv = K[o];
      vv = V[o];
    }
    sK[j][d] = kv;
    sV[j][d] = vv;
  }
  __syncthreads();
  if (t >= Lq) return;
  const float* q = Q + ((size_t)b * Lq + t) * E + (size_t)h * hd;
  float s[MAX_LK];
#pragma unroll
  for (int j = 0; j < MAX_LK; ++j) s[j] = 0.f;
  for (int d = 0; d < hd; d += 4) {
    const float4 q4 = *reinterpret_cast<const float4*>(q + d);
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) {
      const float4 k4 = *reinterpret_cast<const float4*>(&sK[j][d]);
      s[j] += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
    }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAX_LK; ++j) mx = (j < Lk) ? fmaxf(mx, s[j]) : mx;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAX_LK; ++j) {
    s[j] = (j < Lk) ? exp_fast(s[j] - mx) : 0.f;
    sum += s[j];
  }
  const float inv = post_scale / sum;
#pragma unroll
  for (int j = 0; j < MAX_LK; ++j) s[j] *= inv;
  const size_t oo = ((size_t)b * Lq + t) * E + (size_t)h * hd;
  for (int d = 0; d < hd; d += 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) {
      const float4 v4 = *reinterpret_cast<const float4*>(&sV[j][d]);
      a.x += s[j] * v4.x; a.y += s[j] * v4.y; a.z += s[j] * v4.z; a.w += s[j] * v4.w;
    }
    if (o_planes) store_planes4(reinterpret_cast<bf16_t*>(O) + oo + d, o_lo_off, a);
    else *reinterpret_cast<float4*>(O + oo + d) = a;
  }
}

// Backward.  O = sum_j (c p_j) V_j, p = softmax(S), S = Q K^T, c = post_scale:
//   dPs_j = c (dO . V_j);  dS_j = p_j (dPs_j - sum_j' p_j' dPs_j');  dQ = sum_j dS_j K_j
//   dK_j = sum_q dS_qj Q_q;  dV_j = sum_q c p_qj dO_q
__global__ __launch_bounds__(256) void xattn_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                        const float* __restrict__ V, const float* __restrict__ dO,
                                                        float* __restrict__ dQ, float* __restrict__ dK,
                                                        float* __restrict__ dV, int planes, size_t q_lo_off,
                                                        size_t kv_lo_off, int heads, int Lq, int Lk, int hd,
                                                        float post_scale) {
  extern __shared__ __attribute__((aligned(16))) float bwd_smem[];
  typedef float RowK[MAX_HD];
  typedef float RowP[MAX_LK];
  typedef float RowA[MAX_HD + 1];
  RowK* sK = reinterpret_cast<RowK*>(bwd_smem);                              // [MAX_LK][MAX_HD]
  RowK* sV = sK + MAX_LK;                                                    // [MAX_LK][MAX_HD]
  RowP* sP = reinterpret_cast<RowP*>(sV + MAX_LK);                           // [256][MAX_LK]  c * p
  RowP* sS = sP + 256;                                                       // [256][MAX_LK]  dS
  RowA (*sAcc)[2 * MAX_LK] = reinterpret_cast<RowA (*)[2 * MAX_LK]>(sS + 256);  // [2][2*MAX_LK][MAX_HD+1]
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int E = heads * hd;
  const int t = threadIdx.x;
  for (int idx = t; idx < MAX_LK * hd; idx += 256) {
    const int j = idx / hd, d = idx % hd;
    float kv = 0.f, vv = 0.f;
    if (j < Lk) {
      const size_t o = ((size_t)b * Lk + j) * E + (size_t)h * hd + d;
      kv = K[o];
      vv = V[o];
    }
    sK[j][d] = kv;
    sV[j][d] = vv;
  }
  __syncthreads();
  if (t < Lq) {
    const size_t ro = ((size_t)b * Lq + t) * E + (size_t)h * hd;
    float s[MAX_LK], dp[MAX_LK];
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) { s[j] = 0.f; dp[j] = 0.f; }
    for (int d = 0; d < hd; d += 4) {
      const float4 q4 = *reinterpret_cast<const float4*>(Q + ro + d);
      const float4 g4 = *reinterpret_cast<const float4*>(dO + ro + d);
#pragma unroll
      for (int j = 0; j < MAX_LK; ++j) {
        const float4 k4 = *reinterpret_cast<const float4*>(&sK[j][d]);
        const float4 v4 = *reinterpret_cast<const float4*>(&sV[j][d]);
        s[j] += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
        dp[j] += g4.x * v4.x + g4.y * v4.y + g4.z * v4.z + g4.w * v4.w;
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) mx = (j < Lk) ? fmaxf(mx, s[j]) : mx;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) {
      s[j] = (j < Lk) ? exp_fast(s[j] - mx) : 0.f;
      sum += s[j];
    }
    const float inv = 1.0f / sum;
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) {
      s[j] *= inv;                 // p_j
      dp[j] *= post_scale;         // dL/dp_j
      dot += s[j] * dp[j];
    }
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) {
      const float ds = s[j] * (dp[j] - dot);
      sS[t][j] = ds;
      sP[t][j] = s[j] * post_scale;
      dp[j] = ds;
    }
    for (int d = 0; d < hd; d += 4) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < MAX_LK; ++j) {
        const float4 k4 = *reinterpret_cast<const float4*>(&sK[j][d]);
        a.x += dp[j] * k4.x; a.y += dp[j] * k4.y; a.z += dp[j] * k4.z; a.w += dp[j] * k4.w;
      }
      if (planes) store_planes4(reinterpret_cast<bf16_t*>(dQ) + ro + d, q_lo_off, a);
      else *reinterpret_cast<float4*>(dQ + ro + d) = a;
    }
  }
  __syncthreads();
  // dK / dV: thread <-> (column d, half of the query rows); 16 + 16 accumulators per thread
  {
    const int d = t & 127, part = t >> 7;
    const int half = (Lq + 1) / 2;
    const int q0 = part * half;
    const int q1 = (q0 + half < Lq) ? q0 + half : Lq;
    float ak[MAX_LK], av[MAX_LK];
#pragma unroll
    for (int j = 0; j < MAX_LK; ++j) { ak[j] = 0.f; av[j] = 0.f; }
    if (d < hd) {
      for (int q = q0; q < q1; ++q) {
        const size_t ro = ((size_t)b * Lq + q) * E + (size_t)h * hd + d;
        const float x = Q[ro], y = dO[ro];
#pragma unroll
        for (int j4 = 0; j4 < MAX_LK; j4 += 4) {
          const float4 ds4 = *reinterpret_cast<const float4*>(&sS[q][j4]);
          const float4 p4 = *reinterpret_cast<const float4*>(&sP[q][j4]);
          ak[j4 + 0] += ds4.x * x; ak[j4 + 1] += ds4.y * x; ak[j4 + 2] += ds4.z * x; ak[j4 + 3] += ds4.w * x;
          av[j4 + 0] += p4.x * y; av[j4 + 1] += p4.y * y; av[j4 + 2] += p4.z * y; av[j4 + 3] += p4.w * y;
        }
      }
#pragma unroll
      for (int j = 0; j < MAX_LK; ++j) {
        sAcc[part][j][d] = ak[j];
        sAcc[part][MAX_LK + j][d] = av[j];
      }
    }
  }
  __syncthreads();
  for (int idx = t; idx < Lk * (hd / 4); idx += 256) {
    const int j = idx / (hd / 4), d = (idx % (hd / 4)) * 4;
    const size_t o = ((size_t)b * Lk + j) * E + (size_t)h * hd + d;
    float4 gk, gv;
    gk.x = sAcc[0][j][d + 0] + sAcc[1][j][d + 0]; gk.y = sAcc[0][j][d + 1] + sAcc[1][j][d + 1];
    gk.z = sAcc[0][j][d + 2] + sAcc[1][j][d + 2]; gk.w = sAcc[0][j][d + 3] + sAcc[1][j][d + 3];
    gv.x = sAcc[0][MAX_LK + j][d + 0] + sAcc[1][MAX_LK + j][d + 0]; gv.y = sAcc[0][MAX_LK + j][d + 1] + sAcc[1][MAX_LK + j][d + 1];
    gv.z = sAcc[0][MAX_LK + j][d + 2] + sAcc[1][MAX_LK + j][d + 2]; gv.w = sAcc[0][MAX_LK + j][d + 3] + sAcc[1][MAX_LK + j][d + 3];
    if (planes) {
      store_planes4(reinterpret_cast<bf16_t*>(dK) + o, kv_lo_off, gk);
      store_planes4(reinterpret_cast<bf16_t*>(dV) + o, kv_lo_off, gv);
    } else {
      *reinterpret_cast<float4*>(dK + o) = gk;
      *reinterpret_cast<float4*>(dV + o) = gv;
    }
  }
}

}  // namespace

extern "C" int lr2_xattn_fwd(const void* Q, const void* K, const void* V, void* O, int o_planes, uint64_t o_lo_off, int batch,
                             int heads, int Lq, int Lk, int head_dim, float post_scale, void* stream) {
  if (!Q || !K || !V || !O || batch <= 0 || heads <= 0) return LR2_ERR_ARG;
  if (Lq < 1 || Lq > 256 || Lk < 1 || Lk > MAX_LK || head_dim > MAX_HD || head_dim % 4 != 0) return LR2_ERR_SHAPE;
  LR2_LAUNCH(xattn_fwd_kernel, dim3(batch * heads), dim3(256), 0, (hipStream_t)stream, (const float*)Q,
                     (const float*)K, (const float*)V, (float*)O, o_planes, (size_t)o_lo_off, heads, Lq, Lk, head_dim, post_scale);
  return lr2_launch_status(__func__);
}

extern "C" int lr2_xattn_bwd(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK, void* dV,
                             int planes, uint64_t q_lo_off, uint64_t kv_lo_off, int batch, int heads, int Lq, int Lk,
                             int head_dim, float post_scale, void* stream) {
  if (!Q || !K || !V || !dO || !dQ || !dK || !dV || batch <= 0 || heads <= 0) return LR2_ERR_ARG;
  if (Lq < 1 || Lq > 256 || Lk < 1 || Lk > MAX_LK || head_dim > MAX_HD || head_dim % 4 != 0) return LR2_ERR_SHAPE;
  const size_t lds = sizeof(float) * ((size_t)2 * MAX_LK * MAX_HD + 2 * 256 * MAX_LK + 2 * 2 * MAX_LK * (MAX_HD + 1));
  static bool attr_set = false;
  if (!attr_set) {
    if (lr2_allow_dynamic_lds(xattn_bwd_kernel, lds, "xattn_bwd")) return LR2_ERR_LAUNCH;
    attr_set = true;
  }
  LR2_LAUNCH(xattn_bwd_kernel, dim3(batch * heads), dim3(256), lds, (hipStream_t)stream, (const float*)Q,
                     (const float*)K, (const float*)V, (const float*)dO, (float*)dQ, (float*)dK, (float*)dV, planes,
                     (size_t)q_lo_off, (size_t)kv_lo_off, heads, Lq, Lk, head_dim, post_scale);
  return lr2_launch_status(__func__);
}
