// LayerNorm forward/backward and column reductions for gfx950.
// One 64-lane wavefront per row: D = 768 is 3 float4 per lane; mean/variance by wave shuffles
// (no LDS), HBM-bound.  replaces nn.LayerNorm (finetune/xit.py:37,74,93-94) and the TencentPretrain
// LayerNorm (tencentpretrain/layers/layer_norm.py:5-21) plus their autograd backward.
#include "common.h"
#include "lr2ppo_hip.h"

namespace {

constexpr int MAXV = 4;  // float4 per lane -> D <= 1024

__device__ __forceinline__ size_t mapped_row_offset(int r, int group, uint64_t group_stride, int D) {
  return (size_t)(r / group) * group_stride + (size_t)(r % group) * D;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            bf16_t* __restrict__ out_hi, size_t out_lo_off,
                                                            float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int rows, int D, float eps,
                                                            int mode, int group, uint64_t group_stride,
                                                            uint8_t* __restrict__ out_q, uint8_t* __restrict__ out_s) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + (size_t)r * D;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = (i * 64 + lane) * 4;
    if (e < D) {
      v[i] = *reinterpret_cast<const float4*>(xr + e);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = (i * 64 + lane) * 4;
    if (e < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
  q = wave_sum(q);
  float rstd;
  if (mode == 0) rstd = 1.0f / sqrtf(q / (float)D + eps);
  else rstd = 1.0f / (sqrtf(q / (float)(D - 1)) + eps);
  if (lane == 0) {
    if (mean_out) mean_out[r] = mean;
    if (rstd_out) rstd_out[r] = rstd;
  }
  const size_t off = mapped_row_offset(r, group, group_stride, D);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = (i * 64 + lane) * 4;
    if (e < D) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + e);
      const float4 b = *reinterpret_cast<const float4*>(beta + e);
      float4 y;
      y.x = (v[i].x - mean) * rstd * g.x + b.x;
      y.y = (v[i].y - mean) * rstd * g.y + b.y;
      y.z = (v[i].z - mean) * rstd * g.z + b.z;
      y.w = (v[i].w - mean) * rstd * g.w + b.w;
      if (out) *reinterpret_cast<float4*>(out + off + e) = y;
      if (out_hi) store_planes4(out_hi + off + e, out_lo_off, y);
      if (out_q) {
        // MX-FP8 (csrc/fp8.hip's rule): a 32-column block = 8 consecutive lanes x 4 columns; D % 32 == 0, so a block's lanes are
        // all inside the row or all outside
        float amax = fmaxf(fmaxf(fabsf(y.x), fabsf(y.y)), fmaxf(fabsf(y.z), fabsf(y.w)));
        amax = group8_max(amax);
        int ex = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - 8;
        if (amax < 1.17549435e-38f) ex = -127;
        if (ex < -127) ex = -127;
        if (ex > 127) ex = 127;
        const uint32_t ef = (uint32_t)(127 - ex);
        const float inv = __uint_as_float(ef ? ef << 23 : 0x00400000u);
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(y.x * inv, -448.f, 448.f), __builtin_amdgcn_fmed3f(y.y * inv, -448.f, 448.f), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(y.z * inv, -448.f, 448.f), __builtin_amdgcn_fmed3f(y.w * inv, -448.f, 448.f), w, true);
        *reinterpret_cast<int*>(out_q + (size_t)r * D + e) = w;
        if ((lane & 7) == 0) out_s[(size_t)r * (D / 32) + (e >> 5)] = (uint8_t)(ex + 127);
      }
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * c2), g = dy*gamma
//   mode 0 (nn.LayerNorm, biased variance, eps inside the sqrt):      c2 = sum(g*xhat) / D
//   mode 1 (TencentPretrain LayerNorm, y = gamma (x - mean) / (std + eps) + beta with the UNBIASED std, layer_norm.py:16-21):
//           c2 = sum(g*xhat) / ((D - 1) * (1 - eps * rstd)),  rstd = 1 / (std + eps)      [std * rstd = 1 - eps * rstd]
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, int group, uint64_t group_stride,
                                                            const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ resid_grad, float* __restrict__ dx_f32,
                                                            bf16_t* __restrict__ dxm_hi, size_t dxm_lo_off, float drop_scale, uint32_t drop_thr,
                                                            uint64_t drop_key_host, const uint64_t* __restrict__ drop_seed_dev, uint64_t drop_seed,
                                                            uint32_t drop_site, float* __restrict__ partials, int rows, int D,
                                                            int mode, float eps) {
  const uint64_t drop_key = drop_seed_dev ? dropout_key(drop_seed + scalar_load_u64(drop_seed_dev), drop_site) : drop_key_host;
  __shared__ float red[4][2][MAXV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 dg[MAXV], db[MAXV], gam[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int e = (i * 64 + lane) * 4;
    gam[i] = (e < D) ? *reinterpret_cast<const float4*>(gamma + e) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const float mean = mean_in[r], rstd = rstd_in[r];
    const size_t doff = mapped_row_offset(r, group, group_stride, D);
    float4 xh[MAXV], g[MAXV], rgv[MAXV];
    float s1 = 0.f, s2 = 0.f;
    // the residual gradient is requested with x and dy (it is only added at the end: its latency hides behind the two wave sums)
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e = (i * 64 + lane) * 4;
      rgv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < D && resid_grad) rgv[i] = *reinterpret_cast<const float4*>(resid_grad + (size_t)r * D + e);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < D) {
        const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * D + e);
        const float4 dyv = *reinterpret_cast<const float4*>(dy + doff + e);
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        g[i] = make_float4(dyv.x * gam[i].x, dyv.y * gam[i].y, dyv.z * gam[i].z, dyv.w * gam[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        dg[i].x += dyv.x * xh[i].x; dg[i].y += dyv.y * xh[i].y; dg[i].z += dyv.z * xh[i].z; dg[i].w += dyv.w * xh[i].w;
        db[i].x += dyv.x; db[i].y += dyv.y; db[i].z += dyv.z; db[i].w += dyv.w;
      }
    }
    const float c1 = wave_sum(s1) / (float)D;
    const float c2 = wave_sum(s2) / (mode == 0 ? (float)D : (float)(D - 1) * (1.0f - eps * rstd));
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int e = (i * 64 + lane) * 4;
      if (e < D) {
        float4 d;
        const float tx = g[i].x - c1 - xh[i].x * c2, ty = g[i].y - c1 - xh[i].y * c2;
        const float tz = g[i].z - c1 - xh[i].z * c2, tw = g[i].w - c1 - xh[i].w * c2;
        const size_t o = (size_t)r * D + e;
        {
          // product and sum rounded separately, as in rounds 1-3 (there the residual gradient was loaded in a block of its own and the
          // compiler could not fuse the two; now that the load is issued early it would): moving a load must not move a bit
#pragma clang fp contract(off)
          d.x = rstd * tx; d.y = rstd * ty; d.z = rstd * tz; d.w = rstd * tw;
          if (resid_grad) {
            const float4 rg = rgv[i];
            d.x = d.x + rg.x; d.y = d.y + rg.y; d.z = d.z + rg.z; d.w = d.w + rg.w;
          }
        }
        if (dx_f32) *reinterpret_cast<float4*>(dx_f32 + o) = d;
        if (dxm_hi) {
          float4 m = d;
          if (drop_scale != 0.f) {
            m = dropout_apply4(drop_key, o, drop_thr, drop_scale, d);
          }
          store_planes4(dxm_hi + o, dxm_lo_off, m);
        }
      }
    }
  }
  // block reduce dgamma / dbeta over the 4 waves, then one partial row per block
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int e = (i * 64 + lane) * 4;
    red[wave][0][e + 0] = dg[i].x; red[wave][0][e + 1] = dg[i].y; red[wave][0][e + 2] = dg[i].z; red[wave][0][e + 3] = dg[i].w;
    red[wave][1][e + 0] = db[i].x; red[wave][1][e + 1] = db[i].y; red[wave][1][e + 2] = db[i].z; red[wave][1][e + 3] = db[i].w;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[w][0][c];
      b += red[w][1][c];
    }
    partials[(size_t)blockIdx.x * 2 * D + c] = a;
    partials[(size_t)blockIdx.x * 2 * D + D + c] = b;
  }
}

// 64 columns x 4 row slices per workgroup: coalesced 256-B reads, rows split 4 ways, LDS combine.
__global__ __launch_bounds__(256) void partials_finish_kernel(const float* __restrict__ partials, int nblocks, int cols,
                                                              int ld, float* __restrict__ out, int accumulate) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < cols) {
    // 8 independent loads in flight per thread (the loop is latency-bound: 32 dependent 1-us round trips otherwise);
    // the association is fixed, so the result does not depend on timing
    int b = slice;
    for (; b + 28 < nblocks; b += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = partials[(size_t)(b + 4 * u) * ld + c];
      s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    for (; b < nblocks; b += 4) s += partials[(size_t)b * ld + c];
  }
  red[slice][lane] = s;
  __syncthreads();
  if (slice == 0 && c < cols) {
    const float t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[c] = accumulate ? out[c] + t : t;
  }
}

// thread <-> 4 adjacent columns, block <-> 1024 columns x one row chunk; fp32 input or bf16 hi/lo planes
template <bool PLANES>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x_, size_t lo_off, int rows, int cols, int ld,
                                                     float* __restrict__ partials) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= cols) return;
  const int chunk = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * chunk;
  int r1 = r0 + chunk;
  if (r1 > rows) r1 = rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int r = r0; r < r1; ++r) {
    if (PLANES) {
      const bf16_t* p = (const bf16_t*)x_ + (size_t)r * ld + c;
      const u32x2_t h = *reinterpret_cast<const u32x2_t*>(p), l = *reinterpret_cast<const u32x2_t*>(p + lo_off);
      s.x += __uint_as_float(h[0] << 16) + __uint_as_float(l[0] << 16);
      s.y += __uint_as_float(h[0] & 0xffff0000u) + __uint_as_float(l[0] & 0xffff0000u);
      s.z += __uint_as_float(h[1] << 16) + __uint_as_float(l[1] << 16);
      s.w += __uint_as_float(h[1] & 0xffff0000u) + __uint_as_float(l[1] & 0xffff0000u);
    } else {
      const float4 v = *reinterpret_cast<const float4*>((const float*)x_ + (size_t)r * ld + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(partials + (size_t)blockIdx.y * cols + c) = s;
}

}  // namespace

extern "C" int lr2_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* out, void* out_hi,
                                 uint64_t out_lo_off, void* mean, void* rstd, int rows, int D, float eps, int mode,
                                 int group, uint64_t group_stride, void* stream) {
  if (!x || !gamma || !beta || (!out && !out_hi) || rows <= 0) return LR2_ERR_ARG;
  if (D % 4 != 0 || D > MAXV * 256 || D < 4) return LR2_ERR_SHAPE;
  if (group <= 0) { group = rows; group_stride = 0; }
  LR2_LAUNCH(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x,
             (const float*)gamma, (const float*)beta, (float*)out, (bf16_t*)out_hi, (size_t)out_lo_off, (float*)mean,
             (float*)rstd, rows, D, eps, mode, group, group_stride, (uint8_t*)nullptr, (uint8_t*)nullptr);
  return lr2_launch_status(__func__);
}

extern "C" int lr2_layernorm_fwd_mxfp8(const void* x, const void* gamma, const void* beta, void* out, void* out_q, void* out_scales,
                                       int rows, int D, float eps, int mode, void* stream) {
  if (!x || !gamma || !beta || !out_q || !out_scales || rows <= 0) return LR2_ERR_ARG;
  if (mode != 0 && mode != 1) return LR2_ERR_ARG;
  if (D % 32 != 0 || D > MAXV * 256 || D < 32) return LR2_ERR_SHAPE;
  LR2_LAUNCH(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x,
             (const float*)gamma, (const float*)beta, (float*)out, (bf16_t*)nullptr, (size_t)0, (float*)nullptr,
             (float*)nullptr, rows, D, eps, mode, rows, (uint64_t)0, (uint8_t*)out_q, (uint8_t*)out_scales);
  return lr2_launch_status(__func__);
}

extern "C" int lr2_layernorm_bwd(const void* dy, int group, uint64_t group_stride, const void* x, const void* gamma,
                                 const void* mean, const void* rstd, const void* resid_grad, void* dx, void* dxm_hi,
                                 uint64_t dxm_lo_off, float drop_p, uint64_t drop_seed, uint32_t drop_site, const void* drop_seed_dev,
                                 void* partials, int nblocks, int rows, int D, int mode, float eps, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !partials || (!dx && !dxm_hi) || rows <= 0 || nblocks <= 0)
    return LR2_ERR_ARG;
  if (mode != 0 && mode != 1) return LR2_ERR_ARG;
  if (D % 4 != 0 || D > MAXV * 256 || D < 4) return LR2_ERR_SHAPE;
  if (group <= 0) { group = rows; group_stride = 0; }
  float scale = 0.f;
  uint32_t thr = 0;
  uint64_t key = 0;
  if (drop_p > 0.f) {
    scale = 1.0f / (1.0f - drop_p);
    thr = dropout_threshold(drop_p);
    key = (((uint64_t)drop_site) << 40) ^ (drop_seed * 0x9E3779B97F4A7C15ull);
  }
  LR2_LAUNCH(layernorm_bwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const float*)dy, group, group_stride,
             (const float*)x, (const float*)gamma, (const float*)mean, (const float*)rstd, (const float*)resid_grad,
             (float*)dx, (bf16_t*)dxm_hi, (size_t)dxm_lo_off, scale, thr, key, (const uint64_t*)drop_seed_dev, drop_seed, drop_site,
             (float*)partials, rows, D, mode, eps);
  return lr2_launch_status(__func__);
}

extern "C" int lr2_colsum_partials_finish(const void* partials, int nblocks, int cols, int ld, void* out, int accumulate,
                                          void* stream) {
  if (!partials || !out || nblocks <= 0 || cols <= 0) return LR2_ERR_ARG;
  LR2_LAUNCH(partials_finish_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const float*)partials, nblocks, cols, ld, (float*)out, accumulate);
  return lr2_launch_status(__func__);
}

extern "C" int lr2_colsum(const void* x, int is_planes, uint64_t lo_off, int rows, int cols, int ld, void* partials,
                          int nblocks, void* out, void* stream) {
  if (!x || !partials || !out || rows <= 0 || cols <= 0 || nblocks <= 0) return LR2_ERR_ARG;
  if (cols % 4 != 0 || ld % 4 != 0) return LR2_ERR_SHAPE;
  if (nblocks > rows) nblocks = rows;
  dim3 grid((cols + 1023) / 1024, nblocks);
  if (is_planes)
    LR2_LAUNCH(colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (size_t)lo_off, rows, cols, ld, (float*)partials);
  else
    LR2_LAUNCH(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (size_t)0, rows, cols, ld, (float*)partials);
  if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  return lr2_colsum_partials_finish(partials, nblocks, cols, cols, out, 0, stream);
}
