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
  int splits;                   // split-K factor (grid = tiles_m * tiles_n * splits workgroups, 1-D)
  float* partial;               // split-K workspace [splits][M][N] or nullptr
  int dma_stages;               // LDS images per planes operand: 2 = double buffered (1 workgroup/CU at BM=128), 1 = single
  int waves8;                   // planes x planes, 128 x 128 tiles: 8-wave workgroups (wave tile 64 x 32)
  int ablate;                   // diagnostics only (LR2_GEMM_ABLATE): 2 no global loads, 4 no LDS fill
  int strip_n;                  // tiles per strip along N of the XCD-aware tile order (0 = 8; LR2_GEMM_STRIP: an A/B switch of the 256 x 256 NT kernel)
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
// position of workgroup `bid` in a sequence of T work units cut into 8 contiguous chunks, one per XCD group (bid % 8)
__device__ __forceinline__ int xcd_chunk_index(int T, int bid) {
  const int q = T >> 3, r = T & 7, xcd = bid & 7, local = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int bid, int& tm, int& tn, int strip = 8) {
  const int i = xcd_chunk_index(tiles_m * tiles_n, bid);
  const int SN = tiles_n < strip ? tiles_n : strip;
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
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  __builtin_nontemporal_store(f32x4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4_t*>(p));
}

// The epilogue is split into "request everything this element needs from HBM" and "combine + store", so that a caller can
// issue the loads of all its elements before the first store: on CDNA loads and stores share one in-order counter
// (vmcnt), so a load issued after a store can only be waited for together with that store.
// NS = register slots per element group: 3 serves any combination ({GELU' input, residual, accumulate target} or the
// fused optimizer's {p, m, v}); 1 serves kernels whose accumulators leave room for one request per element only
// (gemm256: the host routes a product there only when at most one of aux_z / resid / accumulate is set and no adam).
template <int NS>
struct EpiLoads {
  float4 s[NS];
};
template <int NS>
__device__ __forceinline__ void epilogue_load4(const Epilogue& e, EpiLoads<NS>& L, int m, int n) {
  constexpr int SR = NS == 3 ? 1 : 0, SO = NS == 3 ? 2 : 0;
  if constexpr (NS == 1) {
    // one request slot (the host sends a launch with two requests to an NS = 3 kernel): choose the source first and issue ONE
    // load.  Three conditional loads into the same registers made the compiler wait (vmcnt(0): the previous slab's stores
    // included) before each address computation, on top of the wait at the point of use.
    const float* src = nullptr;
    int ld = 0;
    if (e.act == 2) { src = e.aux_z; ld = e.ld_aux; }
    else if (e.resid) { src = e.resid; ld = e.ld_resid; }
    else if (e.out && e.accumulate) { src = e.out; ld = e.ld_out; }
    if (src) L.s[0] = ld4(src + (size_t)m * ld + n);
    return;
  }
  if (e.act == 2) L.s[0] = ld4(e.aux_z + (size_t)m * e.ld_aux + n);
  if (e.resid) L.s[SR] = ld4(e.resid + (size_t)m * e.ld_resid + n);
  if (NS == 3 && e.adam_p) {
    const size_t off = (size_t)m * e.ld_out + n;
    if (e.stream_nt) {
      L.s[0] = ld4_nt(e.adam_p + off);
      L.s[SR] = ld4_nt(e.adam_m + off);
      L.s[SO] = ld4_nt(e.adam_v + off);
    } else {
      L.s[0] = ld4(e.adam_p + off);
      L.s[SR] = ld4(e.adam_m + off);
      L.s[SO] = ld4(e.adam_v + off);
    }
  } else if (e.out && e.accumulate) {
    L.s[SO] = ld4(e.out + (size_t)m * e.ld_out + n);
  }
}
// alpha * v (+ bias): ONE fused multiply-add per element when there is a bias, written out so that every code path that forms
// the same element (general / fast path, reducer) rounds the same way
__device__ __forceinline__ float4 scale_bias4(const Epilogue& e, float4 v, float4 b) {
  if (e.bias)
    return make_float4(__builtin_fmaf(e.alpha, v.x, b.x), __builtin_fmaf(e.alpha, v.y, b.y), __builtin_fmaf(e.alpha, v.z, b.z),
                       __builtin_fmaf(e.alpha, v.w, b.w));
  return make_float4(e.alpha * v.x, e.alpha * v.y, e.alpha * v.z, e.alpha * v.w);
}
template <int NS>
__device__ __forceinline__ void epilogue_apply4(const Epilogue& e, float4 v, float4 b, const EpiLoads<NS>& L, int m, int n, int N) {
#pragma clang fp contract(off)      // one rounding per written operation, whatever path forms the element (see epilogue_fast)
  constexpr int SR = NS == 3 ? 1 : 0, SO = NS == 3 ? 2 : 0;
  v = scale_bias4(e, v, b);
  if (e.act == 1) {
    if (e.out_z) { if (e.store_nt) st4_nt(e.out_z + (size_t)m * e.ld_z + n, v); else st4(e.out_z + (size_t)m * e.ld_z + n, v); }
    v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w);
  }
  if (e.drop_scale != 0.0f) {
    const uint64_t idx = (uint64_t)m * (uint64_t)N + (uint64_t)n;
    v = dropout_apply4(dropout_key_of(e), idx, e.drop_thr & 0xFFFFu, e.drop_scale, v);   // idx = m * N + n, multiples of 4
  }
  if (e.act == 2) {
    const float4 z = L.s[0];
    v.x *= gelu_erf_grad(z.x); v.y *= gelu_erf_grad(z.y); v.z *= gelu_erf_grad(z.z); v.w *= gelu_erf_grad(z.w);
  }
  if (e.resid) { v.x += L.s[SR].x; v.y += L.s[SR].y; v.z += L.s[SR].z; v.w += L.s[SR].w; }
  if (NS == 3 && e.adam_p) {  // fused optimizer step
    const size_t off = (size_t)m * e.ld_out + n;
    float4 p = L.s[0], mm = L.s[SR], vv = L.s[SO];
const float adam_lr = e.lr_dev ? scalar_load_f32((const float*)e.dev_scalar) : e.adam_lr;
        adam_update(p.x, v.x, mm.x, vv.x, adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    adam_update(p.y, v.y, mm.y, vv.y, adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    adam_update(p.z, v.z, mm.z, vv.z, adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    adam_update(p.w, v.w, mm.w, vv.w, adam_lr, e.adam_b1, e.adam_b2, e.adam_ob1, e.adam_ob2, e.adam_eps, e.adam_wd);
    if (e.stream_nt) { st4_nt(e.adam_p + off, p); st4_nt(e.adam_m + off, mm); st4_nt(e.adam_v + off, vv); }
    else { st4(e.adam_p + off, p); st4(e.adam_m + off, mm); st4(e.adam_v + off, vv); }
    return;
  }
  if (e.out) {
    if (e.accumulate) { v.x += L.s[SO].x; v.y += L.s[SO].y; v.z += L.s[SO].z; v.w += L.s[SO].w; }
    if (e.store_nt) st4_nt(e.out + (size_t)m * e.ld_out + n, v); else st4(e.out + (size_t)m * e.ld_out + n, v);
  }
  if (e.out_hi) {  // bf16 hi/lo planes for the next GEMM (same bytes as the fp32 tensor they replace)
    u32x2_t hv, lv;
    split4(v, hv, lv);
    bf16_t* ph = e.out_hi + (size_t)m * e.ld_planes + n;
    if (e.store_nt) {
      __builtin_nontemporal_store(hv, reinterpret_cast<u32x2_t*>(ph));
      __builtin_nontemporal_store(lv, reinterpret_cast<u32x2_t*>(ph + e.lo_off));
    } else {
      *reinterpret_cast<u32x2_t*>(ph) = hv;
      *reinterpret_cast<u32x2_t*>(ph + e.lo_off) = lv;
    }
  }
}
// one element group, loads and stores together (the split-K reducer: a grid-stride loop with one group in flight per thread)
__device__ __forceinline__ void epilogue_vec4(const Epilogue& e, float4 v, int m, int n, int N) {
  EpiLoads<3> L;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e.bias) b = ld4(e.bias + n);
  epilogue_load4<3>(e, L, m, n);
  epilogue_apply4<3>(e, v, b, L, m, n, N);
}

// LDS accesses of the epilogue slab go through inline asm.  With LDS-DMA in the kernel hipcc cannot tell a pending
// LDS-DMA write from any other outstanding VMEM operation, so it puts `s_waitcnt vmcnt(0)` in front of every LDS access
// it can see -- in an epilogue that means "wait for every global store issued so far" before each slab read: the stores
// of a tile were serialised at one HBM write latency each (measured: 30 us per 256 x 256 tile, 30 % of a K = 768 tile's
// life).  A wave's LDS operations execute in order, so write -> read of the wave-private slab needs no barrier; the
// data of the reads is waited for with lgkmcnt(0) before the first use.
template <int IMM>
__device__ __forceinline__ void slab_write(uint32_t addr, float v) {
  asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "i"(IMM) : "memory");
}
template <int IMM>
__device__ __forceinline__ float4 slab_read4(uint32_t addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

template <int WN, int NI, int II, int J, int R>
__device__ __forceinline__ void slab_write_tile(uint32_t wbase, const f32x4_t& t) {
  slab_write<((16 * II + R) * (WN + 4) + 16 * J) * 4>(wbase, t[R]);
  if constexpr (R < 3) slab_write_tile<WN, NI, II, J, R + 1>(wbase, t);
}
template <int WN, int MI, int NI, int HALF, int II, int J>
__device__ __forceinline__ void slab_write_all(uint32_t wbase, f32x4_t (&acc)[MI][NI]) {
  slab_write_tile<WN, NI, II, J, 0>(wbase, acc[2 * HALF + II][J]);
  if constexpr (J + 1 < NI) slab_write_all<WN, MI, NI, HALF, II, J + 1>(wbase, acc);
  else if constexpr (II == 0) slab_write_all<WN, MI, NI, HALF, 1, 0>(wbase, acc);
}
template <int WN, int NP, int PASS>
__device__ __forceinline__ void slab_read_all(uint32_t rbase, float4 (&v)[NP]) {
  constexpr int RPP = 64 / (WN / 4);
  v[PASS] = slab_read4<PASS * RPP * (WN + 4) * 4>(rbase);
  if constexpr (PASS + 1 < NP) slab_read_all<WN, NP, PASS + 1>(rbase, v);
}

// One 32-row slab (accumulator tile rows 2*HALF and 2*HALF + 1) of a wave tile at a time.  HALF is a template
// parameter: with a runtime loop the compiler does not always unroll (the 128-row wave tile of gemm256.hip) and then
// indexes `acc` dynamically, which sends the accumulators to scratch memory.
// The HBM requests of slab h+1 (residual / GELU' input / accumulate target / optimizer state) are issued BEFORE the
// stores of slab h: loads and stores share the in-order vmcnt counter, so a load issued behind a store can only be
// waited for together with that store (one HBM write latency per slab otherwise).
template <int WN, int NS>
struct EpiSlab {
  static constexpr int NP = 32 / (64 / (WN / 4));
  EpiLoads<NS> L[NP];
};

template <int WN, int NS, int HALF>
__device__ __forceinline__ void epilogue_request(const GemmParams& g, EpiSlab<WN, NS>& S, int mw, int nw, int lane) {
  constexpr int LPR = WN / 4, RPP = 64 / LPR, NP = 32 / RPP;
  const int row0 = lane / LPR, n = nw + (lane % LPR) * 4;
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int m = mw + 32 * HALF + pass * RPP + row0;
    if (m < g.M && n < g.N) epilogue_load4<NS>(g.epi, S.L[pass], m, n);
  }
}

template <int WN, int MI, int NI, int HALF>
__device__ __forceinline__ void epilogue_to_slab(f32x4_t (&acc)[MI][NI], float* slab, int lane) {
  constexpr int LDW = WN + 4;
  const int gq = lane >> 4, c16 = lane & 15;
  slab_write_all<WN, MI, NI, HALF, 0, 0>(lds_addr(slab) + (uint32_t)((4 * gq * LDW + c16) * 4), acc);
}

// Fast paths for a wave tile that lies inside the matrix.  The general path (epilogue_apply4) decides everything per element
// group: range checks and a dozen wave-uniform switches cut each pass of a slab into many small basic blocks, and with two waves
// per SIMD in the epilogue nothing hides the latencies inside one (GELU alone is rcp + exp + a degree-4 polynomial per element).
// Here the form is decided ONCE per slab and the NP passes are one basic block: the compiler interleaves their dependency chains.
// Same operations in the same order per element as epilogue_apply4: same bits (tools/dbg/epi_ab.py compares two builds).
//   ACT 0 / 1 (GELU; Z: store the pre-activation) / 2 (multiply by GELU'(aux));  DROP: mask after the activation;
//   RESID: add the residual row;  OUT: fp32 result;  PL: bf16 hi / lo planes result;  NT: non-temporal stores.
template <int NP, int RPP, int NS, int ACT, bool Z, bool DROP, bool RESID, bool OUT, bool PL, bool NT>
__device__ __forceinline__ void epilogue_fast(const Epilogue& e, const float4 (&v)[NP], float4 b, const EpiLoads<NS> (&L)[NP],
                                              int m0, int n, int N) {
#pragma clang fp contract(off)
  constexpr int SR = NS == 3 ? 1 : 0;
  uint64_t key = 0;
  if constexpr (DROP) key = dropout_key_of(e);
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int m = m0 + pass * RPP;
    float4 x = scale_bias4(e, v[pass], b);
    if constexpr (ACT == 1) {
      if constexpr (Z) {
        if constexpr (NT) st4_nt(e.out_z + (size_t)m * e.ld_z + n, x);
        else st4(e.out_z + (size_t)m * e.ld_z + n, x);
      }
      x.x = gelu_erf(x.x); x.y = gelu_erf(x.y); x.z = gelu_erf(x.z); x.w = gelu_erf(x.w);
    }
    if constexpr (DROP) x = dropout_apply4(key, (uint64_t)m * (uint64_t)N + (uint64_t)n, e.drop_thr & 0xFFFFu, e.drop_scale, x);
    if constexpr (ACT == 2) {
      const float4 z = L[pass].s[0];
      x.x *= gelu_erf_grad(z.x); x.y *= gelu_erf_grad(z.y); x.z *= gelu_erf_grad(z.z); x.w *= gelu_erf_grad(z.w);
    }
    if constexpr (RESID) {
      const float4 r = L[pass].s[SR];
      x.x += r.x; x.y += r.y; x.z += r.z; x.w += r.w;
    }
    if constexpr (OUT) {
      if constexpr (NT) st4_nt(e.out + (size_t)m * e.ld_out + n, x);
      else st4(e.out + (size_t)m * e.ld_out + n, x);
    }
    if constexpr (PL) {
      u32x2_t hv, lv;
      split4(x, hv, lv);
      bf16_t* p = e.out_hi + (size_t)m * e.ld_planes + n;
      if constexpr (NT) {
        __builtin_nontemporal_store(hv, reinterpret_cast<u32x2_t*>(p));
        __builtin_nontemporal_store(lv, reinterpret_cast<u32x2_t*>(p + e.lo_off));
      } else {
        *reinterpret_cast<u32x2_t*>(p) = hv;
        *reinterpret_cast<u32x2_t*>(p + e.lo_off) = lv;
      }
    }
  }
}

// -> true when one of the instantiated forms took the slab.  WIDE 2: the full list (the 256 x 256 kernels, where the encoders'
// training products run); 1: the 8-wave kernels (the heads' M = 12 544 products); 0: the other planes x planes kernels of the general
// family, the inference forms only; -1: none (fifty kernel instantiations: compile time).
template <int NP, int RPP, int NS, int WIDE>
__device__ __forceinline__ bool epilogue_fast_dispatch(const Epilogue& e, const float4 (&v)[NP], float4 b,
                                                       const EpiLoads<NS> (&L)[NP], int m0, int n, int N) {
  if constexpr (WIDE < 0) return false;        // kernels with an fp32 operand: HBM-bound weight streams, the epilogue is not their cost
  if (e.adam_p || e.accumulate) return false;
  const int act = e.act;
  const bool z = e.out_z != nullptr, drop = e.drop_scale != 0.0f, resid = e.resid != nullptr, out = e.out != nullptr,
             pl = e.out_hi != nullptr;
  if (z && act != 1) return false;
#define LR2_FAST(ACT, Z, DROP, RESID, OUT, PL)                                                                         \
  if (act == ACT && z == Z && drop == DROP && resid == RESID && out == OUT && pl == PL) {                             \
    if (e.store_nt) epilogue_fast<NP, RPP, NS, ACT, Z, DROP, RESID, OUT, PL, true>(e, v, b, L, m0, n, N);             \
    else epilogue_fast<NP, RPP, NS, ACT, Z, DROP, RESID, OUT, PL, false>(e, v, b, L, m0, n, N);                        \
    return true;                                                                                                     \
  }
  LR2_FAST(0, false, false, false, false, true)     // bias -> planes (QKV)
  LR2_FAST(1, false, false, false, false, true)     // bias, GELU -> planes (FFN-1, inference)
  LR2_FAST(0, false, false, true, true, false)      // bias + residual -> fp32 (attention output, FFN-2)
  if constexpr (WIDE >= 1) {
    LR2_FAST(1, true, false, false, false, true)    // FFN-1, training: pre-activation kept
    LR2_FAST(2, false, false, false, false, true)   // FFN-2 input gradient: GELU' -> planes
    LR2_FAST(0, false, true, true, true, false)     // training: dropout, + residual -> fp32
    LR2_FAST(0, false, false, false, true, false)   // plain fp32 result (input gradients)
    LR2_FAST(0, false, false, true, true, true)     // + residual -> fp32 and planes
    LR2_FAST(0, false, true, false, false, true)    // dropout -> planes
  }
  if constexpr (WIDE >= 2) {     // (activation + dropout: 28 spilled VGPRs under the 8-wave kernels' 128-register cap)
    LR2_FAST(1, true, true, false, false, true)     // XiT FFN-1, training: pre-activation kept, GELU, dropout -> planes
    LR2_FAST(2, false, true, false, false, true)    // XiT FFN-2 input gradient: dropout mask, GELU' -> planes
  }
#undef LR2_FAST
  return false;
}

template <int WN, int HALF, int NS, int WIDE>
__device__ __forceinline__ void epilogue_from_slab(const GemmParams& g, float* slab, int mw, int nw, int lane, float* partial,
                                                   float4 bias4, const EpiSlab<WN, NS>& S) {
  constexpr int LDW = WN + 4;
  constexpr int LPR = WN / 4;    // lanes per row
  constexpr int RPP = 64 / LPR;  // rows per pass
  constexpr int NP = 32 / RPP;
  const int row0 = lane / LPR, col = (lane % LPR) * 4;
  const int n = nw + col;
  float4 v[NP];
  slab_read_all<WN, NP, 0>(lds_addr(slab) + (uint32_t)((row0 * LDW + col) * 4), v);   // one row segment of 4 columns per lane
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // (LR2_GEMM_ABLATE & 256 switches the fast forms off: tools/dbg/fuzz_epilogue.py compares the two paths bit for bit)
  if (!partial && !(g.ablate & 256) && mw + 32 * HALF + 32 <= g.M && nw + WN <= g.N) {   // wave-uniform: the slab lies inside the matrix
    if (epilogue_fast_dispatch<NP, RPP, NS, WIDE>(g.epi, v, bias4, S.L, mw + 32 * HALF + row0, n, g.N)) return;
  }
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int m = mw + 32 * HALF + pass * RPP + row0;
    if (m < g.M && n < g.N) {
      if (partial) st4(partial + (size_t)m * g.N + n, v[pass]);
      else epilogue_apply4<NS>(g.epi, v[pass], bias4, S.L[pass], m, n, g.N);
    }
  }
}

// PIPE: request slab h+1's HBM operands before slab h's stores (after slab h's accumulators have moved to LDS, so the
// register peak is acc - 32 + 3 x 32 for NS = 1).  Kernels that keep up to three requests per element (NS = 3) and run
// several workgroups per CU request per slab instead: their register budget decides their occupancy.
template <int WM, int WN, int MI, int NI, int NS, int HALF, bool PIPE, int WIDE>
__device__ __forceinline__ void epilogue_pipeline(const GemmParams& g, f32x4_t (&acc)[MI][NI], float* slab, int mw, int nw,
                                                  int lane, float* partial, float4 bias4, EpiSlab<WN, NS>& cur) {
  if constexpr (!PIPE) {
    if (!partial) epilogue_request<WN, NS, HALF>(g, cur, mw, nw, lane);
  }
  epilogue_to_slab<WN, MI, NI, HALF>(acc, slab, lane);
  EpiSlab<WN, NS> next;
  if constexpr (PIPE && HALF + 1 < WM / 32) {
    if (!partial) epilogue_request<WN, NS, HALF + 1>(g, next, mw, nw, lane);
  }
  epilogue_from_slab<WN, HALF, NS, WIDE>(g, slab, mw, nw, lane, partial, bias4, cur);
  if constexpr (HALF + 1 < WM / 32)
    epilogue_pipeline<WM, WN, MI, NI, NS, HALF + 1, PIPE, WIDE>(g, acc, slab, mw, nw, lane, partial, bias4, PIPE ? next : cur);
}

template <int WM, int WN, int MI, int NI, int NS = 3, int WIDE = (NS == 1 ? 2 : 0)>
__device__ __forceinline__ void epilogue_wave(const GemmParams& g, f32x4_t (&acc)[MI][NI], float* slab, int mw, int nw,
                                              int lane, float* partial) {
  static_assert(WM == 32 || WM == 64 || WM == 96 || WM == 128, "wave tile rows");
  constexpr bool PIPE = false;   // cross-slab prefetch (NS == 1) measured 256 VGPRs + spills in gemm256: per-slab requests only
  const int n = nw + (lane % (WN / 4)) * 4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  EpiSlab<WN, NS> first;
  if (!partial) {
    if (g.epi.bias && n < g.N) bias4 = ld4(g.epi.bias + n);
    if constexpr (PIPE) epilogue_request<WN, NS, 0>(g, first, mw, nw, lane);
  }
  // The wait for the bias load belongs HERE, once, in code every lane passes.  Without this use the first reads of bias4 sit
  // inside the slabs' range-checked (exec-masked) blocks, none of which dominates the next, so the compiler waits before each of
  // them -- and with loads and stores on one in-order counter the only wait it can write there is vmcnt(0): every slab waited for
  // the stores of the slab before it, one HBM write latency each (12 us of a 67-us K = 768 tile, tools/dbg/tile_contention.py).
  asm volatile("" ::"v"(bias4.x), "v"(bias4.y), "v"(bias4.z), "v"(bias4.w));
  epilogue_pipeline<WM, WN, MI, NI, NS, 0, PIPE, WIDE>(g, acc, slab, mw, nw, lane, partial, bias4, first);
}

// Fused AdamW epilogue: the weight, exp_avg and exp_avg_sq vectors of all 32 rows of a slab are requested BEFORE the
// accumulators are transposed through LDS (24 independent 16-B loads per lane in flight; with the loads issued one
// slab pass at a time the 12 GB p/m/v stream of out_layer.fc1 would be latency-bound at ~3 TB/s).
template <int WM, int WN, int MI, int NI, int WIDE = 0>
__device__ __forceinline__ void epilogue_wave_adam(const GemmParams& g, f32x4_t (&acc)[MI][NI], float* slab, int mw,
                                                   int nw, int lane) {
  epilogue_wave<WM, WN, MI, NI, 3, WIDE>(g, acc, slab, mw, nw, lane, nullptr);
}

}  // namespace lr2gemm
