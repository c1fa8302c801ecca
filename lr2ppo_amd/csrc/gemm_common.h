// Pieces shared by the GEMM kernels of this library (gemm.hip: the general NT / NN / TN kernel family; gemm256.hip: the
// 256 x 256 ping-pong kernel for large planes x planes NT products): launch parameters, the wave-uniform buffer descriptor,
// and the fused epilogue (bias, GELU (+ saved pre-activation), dropout, GELU', residual, accumulate, fp32 and / or bf16
// hi/lo planes output, fused AdamW).
#pragma once
#include "common.h"
#include "lr2ppo_hip.h"

namespace lr2gemm {

constexpr int NTHREADS = 256;

struct GemmParams {
  const void* A;
  const void* B;
  int M, N, K;
  int lda, ldb;                 // elements
  uint32_t a_bytes, b_bytes;    // bytes addressable from A / B (one plane for planes operands)
  uint32_t a_lo_off, b_lo_off;  // byte offset from the hi plane to the lo plane (planes operands)
  int k_tiles_per_split;        // in units of BK
  int tiles_m, tiles_n;         // output tile grid
  float* partial;               // split-K workspace [splits][M][N] or nullptr
  int dma_stages;               // LDS images per planes operand: 2 = double buffered (1 workgroup/CU at BM=128), 1 = single
  int waves8;                   // planes x planes, 128 x 128 tiles: 8-wave workgroups (wave tile 64 x 32)
  int ablate;                   // diagnostics only (LR2_GEMM_ABLATE): 2 no global loads, 4 no LDS fill
  Epilogue epi;
};

// Buffer descriptor built from provably wave-uniform words (else hipcc wraps every buffer op in a waterfall loop).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  void* q = (void*)(((uint64_t)hi << 32) | (uint64_t)lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// XCD-aware grouped rasterisation.  Tiles are first put in strip-major order (strips of 8 tiles along N, row-major
// inside a strip), so any 64 consecutive tiles form an 8 x 8 patch sharing 8 A panels and 8 B panels; that sequence is
// cut into 8 equal contiguous chunks, one per XCD (workgroups are dealt round-robin to the XCDs: blockIdx % 8 labels
// the XCD group, blockIdx / 8 is the dispatch order inside it).  Speed only -- the map is a bijection.
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int bid, int& tm, int& tn) {
  const int T = tiles_m * tiles_n;
  const int q = T >> 3, r = T & 7, xcd = bid & 7, local = bid >> 3;
  const int i = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
  const int SN = tiles_n < 8 ? tiles_n : 8;
  const int full = (tiles_n / SN) * tiles_m * SN;  // tiles inside full-width strips
  if (i < full) {
    const int strip = i / (tiles_m * SN), rem = i % (tiles_m * SN);
    tm = rem / SN;
    tn = strip * SN + rem % SN;
  } else {
    const int rw = tiles_n % SN, rem = i - full;  // last, narrower strip
    tm = rem / rw;
    tn = (tiles_n / SN) * SN + rem % rw;
  }
}

// ---- epilogue ------------------------------------------------------------------------------------
// Wave tile WM x WN, staged through a private LDS slab of 32 x (WN + 4) floats, 32 rows at a time.  After the
// transpose each lane owns 4 consecutive columns of one row, so every global access below is a 16-B vector and a
// row segment of WN*4 bytes is contiguous across 16 (WN = 64) or 8 (WN = 32) lanes.
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ void epilogue_vec4(const Epilogue& e, float4 v, int m, int n, int N) {
  v.x *= e.alpha; v.y *= e.alpha; v.z *= e.alpha; v.w *= e.alpha;
  if (e.bias) {
    const float4 b = ld4(e.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (e.act == 1) {
    if (e.out_z) st4(e.out_z + (size_t)m * e.ld_z + n, v);
    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
  }
  if (e.drop_scale != 0.0f) {
    const uint64_t idx = (uint64_t)m * (uint64_t)N + (uint64_t)n;
    v.x = dropout_keep(e.drop_key, idx + 0, e.drop_thr) ? v.x * e.drop_scale : 0.0f;
    v.y = dropout_keep(e.drop_key, idx + 1, e.drop_thr) ? v.y * e.drop_scale : 0.0f;
    v.z = dropout_keep(e.drop_key, idx + 2, e.drop_thr) ? v.z * e.drop_scale : 0.0f;
    v.w = dropout_keep(e.drop_key, idx + 3, e.drop_thr) ? v.w * e.drop_scale : 0.0f;
  }
  if (e.act == 2) {
    const float4 z = ld4(e.aux_z + (size_t)m * e.ld_aux + n);
    v.x *= gelu_erf_grad(z.x); v.y *= gelu_erf_grad(z.y); v.z *= gelu_erf_grad(z.z); v.w *= gelu_erf_grad(z.w);
  }
  if (e.resid) {
    const float4 r = ld4(e.resid + (size_t)m * e.ld_resid + n);
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  if (e.adam_p) {  // fused optimizer step (reached through the split-K reducer; the direct path prefetches, see below)
    const size_t off = (size_t)m * e.ld_out + n;
    float4 p = ld4(e.adam_p + off), mm = ld4(e.adam_m + off), vv = ld4(e.adam_v + off);
    adam_update(p.x, v.x, mm.x, vv.x, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    adam_update(p.y, v.y, mm.y, vv.y, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    adam_update(p.z, v.z, mm.z, vv.z, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    adam_update(p.w, v.w, mm.w, vv.w, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    st4(e.adam_p + off, p); st4(e.adam_m + off, mm); st4(e.adam_v + off, vv);
    return;
  }
  if (e.out) {
    float* p = e.out + (size_t)m * e.ld_out + n;
    if (e.accumulate) {
      const float4 o = ld4(p);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    st4(p, v);
  }
  if (e.out_hi) {  // bf16 hi/lo planes for the next GEMM (same bytes as the fp32 tensor they replace)
    u32x2_t hv, lv;
    split4(v, hv, lv);
    bf16_t* ph = e.out_hi + (size_t)m * e.ld_planes + n;
    *reinterpret_cast<u32x2_t*>(ph) = hv;
    *reinterpret_cast<u32x2_t*>(ph + e.lo_off) = lv;
  }
}

// One 32-row slab (accumulator tile rows 2*HALF and 2*HALF + 1) of a wave tile.  HALF is a template parameter: with a
// runtime loop the compiler does not always unroll (the 128-row wave tile of gemm256.hip) and then indexes `acc`
// dynamically, which sends the accumulators to scratch memory.
template <int WN, int MI, int NI, int HALF>
__device__ __forceinline__ void epilogue_wave_half(const GemmParams& g, f32x4_t (&acc)[MI][NI], float* slab, int mw, int nw,
                                                   int lane, float* partial) {
  constexpr int LDW = WN + 4;
  constexpr int LPR = WN / 4;    // lanes per row
  constexpr int RPP = 64 / LPR;  // rows per pass
  const int gq = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[(16 * ii + 4 * gq + r) * LDW + 16 * j + c16] = acc[2 * HALF + ii][j][r];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int pass = 0; pass < 32 / RPP; ++pass) {
    const int row = pass * RPP + lane / LPR, col = (lane % LPR) * 4;
    const float4 v = ld4(slab + row * LDW + col);
    const int m = mw + 32 * HALF + row, n = nw + col;
    if (m < g.M && n < g.N) {
      if (partial) st4(partial + (size_t)m * g.N + n, v);
      else epilogue_vec4(g.epi, v, m, n, g.N);
    }
  }
  __builtin_amdgcn_wave_barrier();
}

template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void epilogue_wave(const GemmParams& g, f32x4_t (&acc)[MI][NI], float* slab, int mw, int nw,
                                              int lane, float* partial) {
  static_assert(WM == 32 || WM == 64 || WM == 128, "wave tile rows");
  epilogue_wave_half<WN, MI, NI, 0>(g, acc, slab, mw, nw, lane, partial);
  if constexpr (WM >= 64) epilogue_wave_half<WN, MI, NI, 1>(g, acc, slab, mw, nw, lane, partial);
  if constexpr (WM >= 128) {
    epilogue_wave_half<WN, MI, NI, 2>(g, acc, slab, mw, nw, lane, partial);
    epilogue_wave_half<WN, MI, NI, 3>(g, acc, slab, mw, nw, lane, partial);
  }
}

// Fused AdamW epilogue: the weight, exp_avg and exp_avg_sq vectors of all 32 rows of a slab are requested BEFORE the
// accumulators are transposed through LDS (24 independent 16-B loads per lane in flight; with the loads issued one
// slab pass at a time the 12 GB p/m/v stream of out_layer.fc1 would be latency-bound at ~3 TB/s).
template <int WM, int WN, int MI, int NI>
__device__ __forceinline__ void epilogue_wave_adam(const GemmParams& g, f32x4_t (&acc)[MI][NI], float* slab, int mw,
                                                   int nw, int lane) {
  constexpr int LDW = WN + 4;
  constexpr int LPR = WN / 4;
  constexpr int RPP = 64 / LPR;
  constexpr int NP = 32 / RPP;
  const Epilogue& e = g.epi;
  const int gq = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int half = 0; half < WM / 32; ++half) {
    float4 p4[NP], m4[NP], v4[NP];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int m = mw + 32 * half + pass * RPP + lane / LPR, n = nw + (lane % LPR) * 4;
      if (m < g.M && n < g.N) {
        const size_t off = (size_t)m * e.ld_out + n;
        p4[pass] = ld4(e.adam_p + off);
        m4[pass] = ld4(e.adam_m + off);
        v4[pass] = ld4(e.adam_v + off);
      }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(16 * ii + 4 * gq + r) * LDW + 16 * j + c16] = acc[2 * half + ii][j][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int row = pass * RPP + lane / LPR, col = (lane % LPR) * 4;
      float4 v = ld4(slab + row * LDW + col);
      const int m = mw + 32 * half + row, n = nw + col;
      if (m < g.M && n < g.N) {
        v.x *= e.alpha; v.y *= e.alpha; v.z *= e.alpha; v.w *= e.alpha;
        float4 p = p4[pass], mm = m4[pass], vv = v4[pass];
        adam_update(p.x, v.x, mm.x, vv.x, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
        adam_update(p.y, v.y, mm.y, vv.y, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
        adam_update(p.z, v.z, mm.z, vv.z, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
        adam_update(p.w, v.w, mm.w, vv.w, e.adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
        const size_t off = (size_t)m * e.ld_out + n;
        st4(e.adam_p + off, p); st4(e.adam_m + off, mm); st4(e.adam_v + off, vv);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace lr2gemm
