// fp32-grade GEMM on the bf16 matrix cores of gfx950 with fused epilogues:  C[M,N] = op(A) . op(B)
//
//   TA == 0 : A is [M][K] row-major (K contiguous)          TA == 1 : A is [K][M] (M contiguous)
//   TB == 0 : B is [N][K] row-major (nn.Linear weight)      TB == 1 : B is [K][N] (N contiguous)
//
//   forward  y = x W^T          : TA=0 TB=0   (A = x [M,K],   B = W [N,K])
//   dgrad    dx = dy W          : TA=0 TB=1   (A = dy [M,N'], B = W [N',K'] read as [K=N'][N=K'])
//   wgrad    dW = dy^T x        : TA=1 TB=1   (A = dy [K=M'][M=N'], B = x [K=M'][N=K'])
//
// Replaces the cuBLAS sgemm calls reached through nn.Linear in the reference
// (finetune/ppo.py:164-170, finetune/xit.py:103-148, tencentpretrain/layers/*.py).
//
// Precision.  The reference is fp32 end to end and the parity bar is 1e-3 on logits; a single bf16 pass misses it
// (2.5e-3).  Every operand is therefore used as a split pair x = hi + lo (hi = bf16(x), lo = bf16(x - hi)).
// PASSES == 3 accumulates lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulators: fp32-grade
// results (~17 mantissa bits per operand) at up to 1/3 of the bf16 MFMA peak = 5x the chip's fp32-MFMA peak.
// PASSES == 1 keeps hi*hi only.
//
// Operand sources (per operand, template flags APL / BPL):
//   fp32  : fp32 matrix in HBM -> VGPR (16-B buffer loads) -> split in registers -> ds_write_b64 into LDS, single
//           buffered, tile t+1 in flight in registers while tile t is multiplied.  Used for the 2 GB out_layer.fc1
//           weight, which is streamed once per pass and must not be duplicated.
//   planes: the producer already wrote the operand as two bf16 planes [hi | lo] (same bytes as fp32).  Tiles go
//           HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B/lane, no VGPRs, no VALU).  Default: ONE LDS image per
//           workgroup and two workgroups per CU (measured 285-339 TFLOP/s fp32-equivalent on the head's shapes vs
//           239-275 with a double-buffered image and one workgroup per CU; LR2_GEMM_DMA_STAGES=2 selects the latter).  The measured profile of the all-fp32 form showed its phases (loads, split, LDS
//           write, fragment reads, MFMA) adding up serially with the matrix pipe ~30 % busy; with planes the split is
//           paid once by the producer's epilogue instead of once per consuming tile (24-98x), and the loop is
//           DMA + ds_read + MFMA only.
// Both sources zero-fill rows past the end of the matrix through the buffer descriptor's range check, so ragged M, N
// and (TN form) ragged contraction length need no host padding.  K-contiguous operands are read from LDS with
// ds_read_b128, contraction-strided operands keep their HBM orientation and are read with ds_read_b64_tr_b16
// (hardware transpose): nothing is ever transposed in HBM.  LDS images are XOR-swizzled (on the DMA *source* address
// for planes, on the ds_write address for fp32) -- 0 bank conflicts measured (profiles/).
//
// Epilogue: the accumulators of a wave are staged through LDS so that every global access is a full 16-B vector on
// 256-B contiguous row segments; bias, GELU (+ saved pre-activation), dropout, GELU', residual, accumulate, and the
// output either as fp32 or as bf16 hi/lo planes for the next GEMM.
#include <stdlib.h>

#include "gemm_common.h"

namespace lr2gemm {

// ---- LDS image helpers (units of 16 bytes = 8 bf16) -------------------------------------------
// K-contiguous tile: [R rows][BK/8 units].  BK = 64 (128-B rows): unit u of row r lives at unit u ^ ((r >> 1) & 7).
// BK = 32 (64-B rows, four rows per 256-B bank row): unit u lives at u ^ {0,3,2,1}[(r >> 2) & 3], which makes the four
// rows of a residue class mod 4 that one ds_read_b128 lane group touches land on four different 16-B slots.
template <int BK>
__device__ __forceinline__ int swz_kc(int r) {
  if (BK == 64) return (r >> 1) & 7;
  const int t = (r >> 2) & 3;
  return (4 - t) & 3;
}
// contraction-strided tile: [BK k-rows][UPR units]; XOR on 32-byte chunks (bit 0 of the unit untouched).
template <int UPR>
__device__ __forceinline__ int swz_tr(int k) {
  return ((((k & 3) | (((k >> 3) & 1) << 2))) << 1) & (UPR - 1);
}

// ---- operand source 1: fp32 in HBM, split in registers ---------------------------------------
template <int BR, int BK, bool TR, int PASSES>
struct RegStager {
  static constexpr int NV = BR * BK / 4 / NTHREADS;  // float4 per thread per tile
  static constexpr int UPR = TR ? BR / 8 : BK / 8;
  static constexpr int TILE_BYTES = BR * BK * 2;
  static constexpr int VPK = BK / 4;                 // float4 per row of a K-contiguous tile
  u32x4_t regs[NV];
  uint32_t voff, qstep, kstep;
  int tid;

  __device__ __forceinline__ void init(int tid_, int r0, int ld, int ktile0) {
    tid = tid_;
    if (!TR) {
      const int row = tid / VPK, kg = tid % VPK;
      voff = (uint32_t)((((uint64_t)(r0 + row)) * (uint64_t)ld + (uint64_t)kg * 4u) * 4u);
      qstep = (uint32_t)ld * 4u * (NTHREADS / VPK);
      kstep = BK * 4u;
    } else {
      constexpr int VPR = BR / 4;  // float4 per k-row
      const int k = tid / VPR, rg = tid % VPR;
      voff = (uint32_t)((((uint64_t)k) * (uint64_t)ld + (uint64_t)r0 + (uint64_t)rg * 4u) * 4u);
      qstep = (uint32_t)ld * 4u * (NTHREADS / VPR);
      kstep = (uint32_t)ld * 4u * BK;
    }
    voff += (uint32_t)ktile0 * kstep;
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rsrc) {
#pragma unroll
    for (int q = 0; q < NV; ++q) regs[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + q * qstep, 0, 0);
    voff += kstep;
  }
  __device__ __forceinline__ int lds_offset(int q) const {
    if (!TR) {
      const int row = q * (NTHREADS / VPK) + tid / VPK, kg = tid % VPK;
      return (row * UPR + ((kg >> 1) ^ swz_kc<BK>(row))) * 16 + (kg & 1) * 8;
    } else {
      constexpr int VPR = BR / 4;
      const int k = q * (NTHREADS / VPR) + tid / VPR, rg = tid % VPR;
      return (k * UPR + ((rg >> 1) ^ swz_tr<UPR>(k))) * 16 + (rg & 1) * 8;
    }
  }
  // hi pair = v_cvt_pk_bf16_f32(x0, x1) is already in LDS element order; its halves widened back to fp32 by a
  // shift / mask give the residuals whose packed conversion is the lo pair: 6 VALU per pair, no repacking.
  __device__ __forceinline__ void store(char* tile_hi) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const float x0 = __uint_as_float(regs[q][0]), x1 = __uint_as_float(regs[q][1]);
      const float x2 = __uint_as_float(regs[q][2]), x3 = __uint_as_float(regs[q][3]);
      const uint32_t h01 = cvt_pk_bf16(x0, x1), h23 = cvt_pk_bf16(x2, x3);
      const int off = lds_offset(q);
      u32x2_t hv = {h01, h23};
      *reinterpret_cast<u32x2_t*>(tile_hi + off) = hv;
      if (PASSES == 3) {
        const uint32_t l01 = cvt_pk_bf16(x0 - __uint_as_float(h01 << 16), x1 - __uint_as_float(h01 & 0xffff0000u));
        const uint32_t l23 = cvt_pk_bf16(x2 - __uint_as_float(h23 << 16), x3 - __uint_as_float(h23 & 0xffff0000u));
        u32x2_t lv = {l01, l23};
        *reinterpret_cast<u32x2_t*>(tile_hi + TILE_BYTES + off) = lv;
      }
    }
  }
};

// ---- operand source 2: bf16 hi/lo planes in HBM, LDS-DMA ---------------------------------------
// LDS-DMA writes lane-linearly (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE
// address; the LDS image is then identical to the one RegStager writes and read_frag() serves both.
template <int BR, int BK, bool TR, int PASSES, int NW = 4>
struct DmaStager {
  static constexpr int UNITS = BR * BK / 8;         // 16-byte units per plane tile (BR x BK bf16)
  static constexpr int PER_WAVE = UNITS / 64 / NW;  // DMA instructions per wave per plane
  static constexpr int UPR = TR ? BR / 8 : BK / 8;
  static constexpr int TILE_BYTES = BR * BK * 2;
  uint32_t voff[PER_WAVE];
  uint32_t kstep;
  __device__ __forceinline__ void init(int wave, int lane, int r0, int ld, int ktile0) {
    kstep = TR ? (uint32_t)ld * 2u * BK : 2u * BK;
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
      const int p = (wave * PER_WAVE + q) * 64 + lane;
      if (!TR) {
        const int row = p / UPR, u = (p % UPR) ^ swz_kc<BK>(row);
        voff[q] = (uint32_t)(((uint64_t)(r0 + row) * (uint64_t)ld) * 2u + (uint32_t)u * 16u);
      } else {
        const int k = p / UPR, u = (p % UPR) ^ swz_tr<UPR>(k);
        voff[q] = (uint32_t)(((uint64_t)k * (uint64_t)ld + (uint64_t)r0) * 2u + (uint32_t)u * 16u);
      }
      voff[q] += (uint32_t)ktile0 * kstep;
    }
  }
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc_hi, __amdgpu_buffer_rsrc_t rsrc_lo, char* tile_hi,
                                        int wave) {
#pragma unroll
    for (int q = 0; q < PER_WAVE; ++q) {
      char* dst = tile_hi + (wave * PER_WAVE + q) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_hi, LDS_PTR(dst), 16, voff[q], 0, 0, 0);
      if (PASSES == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_lo, LDS_PTR(dst + TILE_BYTES), 16, voff[q], 0, 0, 0);
      voff[q] += kstep;
    }
  }
};

// fragment for v_mfma_f32_16x16x32_bf16: lane l holds X[row l&15][k = 8*(l>>4) + j], j = 0..7
template <int BR, int BK, bool TR>
__device__ __forceinline__ bf16x8_t read_frag(const char* tile, int rbase, int ks, int lane) {
  if (!TR) {
    const int row = rbase + (lane & 15);
    const int u = (4 * ks + (lane >> 4)) ^ swz_kc<BK>(row);
    return *reinterpret_cast<const bf16x8_t*>(tile + (row * (BK / 8) + u) * 16);
  } else {
    constexpr int UPR = BR / 8;
    const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int g = lane >> 4;
    const int ka = 32 * ks + 8 * g + q, kb = ka + 4;
    const int u = (rbase >> 3) + (p >> 1);
    const char* pa = tile + (ka * UPR + (u ^ swz_tr<UPR>(ka))) * 16 + 8 * (p & 1);
    const char* pb = tile + (kb * UPR + (u ^ swz_tr<UPR>(kb))) * 16 + 8 * (p & 1);
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)pa);
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)pb);
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// Transposed fragment read issued through inline asm.  While an LDS-DMA is in flight hipcc puts s_waitcnt vmcnt(0) in
// front of every ds_read_b64_tr_b16 *builtin* (it cannot disambiguate the read from the DMA's LDS destination), which
// would serialise the copy of tile t+1 with the multiply of tile t.  An asm read is invisible to that pass; the caller
// owns the wait (s_waitcnt lgkmcnt + sched_barrier) before the first use.
template <int BR>
__device__ __forceinline__ bf16x8_t read_frag_tr_asm(const char* tile, int rbase, int ks, int lane) {
  constexpr int UPR = BR / 8;
  const int i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int g = lane >> 4;
  const int ka = 32 * ks + 8 * g + q, kb = ka + 4;
  const int u = (rbase >> 3) + (p >> 1);
  const char* pa = tile + (ka * UPR + (u ^ swz_tr<UPR>(ka))) * 16 + 8 * (p & 1);
  const char* pb = tile + (kb * UPR + (u ^ swz_tr<UPR>(kb))) * 16 + 8 * (p & 1);
  const uint32_t aa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)pa;
  const uint32_t ab = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)pb;
  u32x2_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(aa));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ab));
  u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// K-tile multiply for planes (LDS-DMA) operands with at least one contraction-strided operand: all fragments of the
// tile are requested up front (strided ones by asm, see above), the k-step-0 MFMAs start once the first half of the
// LDS queue has drained (LDS returns in order, so a counted lgkmcnt is exact), k-step 1 after the rest.
template <int BM, int BN, int BK, int MI, int NI, bool TA, bool TB, int PASSES>
__device__ __forceinline__ void compute_tile_preload(const char* a_hi, const char* b_hi, int wm0, int wn0, int lane,
                                                     f32x4_t (&acc)[MI][NI]) {
  constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2;
  constexpr int KS = BK / 32;
  bf16x8_t ah[KS][MI], al[KS][MI], bh[KS][NI], bl[KS][NI];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (TA) {
        ah[ks][i] = read_frag_tr_asm<BM>(a_hi, wm0 + 16 * i, ks, lane);
        if (PASSES == 3) al[ks][i] = read_frag_tr_asm<BM>(a_hi + A_TILE, wm0 + 16 * i, ks, lane);
      } else {
        ah[ks][i] = read_frag<BM, BK, false>(a_hi, wm0 + 16 * i, ks, lane);
        if (PASSES == 3) al[ks][i] = read_frag<BM, BK, false>(a_hi + A_TILE, wm0 + 16 * i, ks, lane);
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (TB) {
        bh[ks][j] = read_frag_tr_asm<BN>(b_hi, wn0 + 16 * j, ks, lane);
        if (PASSES == 3) bl[ks][j] = read_frag_tr_asm<BN>(b_hi + B_TILE, wn0 + 16 * j, ks, lane);
      } else {
        bh[ks][j] = read_frag<BN, BK, false>(b_hi, wn0 + 16 * j, ks, lane);
        if (PASSES == 3) bl[ks][j] = read_frag<BN, BK, false>(b_hi + B_TILE, wn0 + 16 * j, ks, lane);
      }
    }
    if (ks == 0 && KS == 2) __builtin_amdgcn_sched_barrier(0);   // keep the k-step-0 requests ahead of the k-step-1 ones
  }
  constexpr int NIMGS = PASSES == 3 ? 2 : 1;
  constexpr int PER_KS = (MI * (TA ? 2 : 1) + NI * (TB ? 2 : 1)) * NIMGS;   // LDS instructions per k-step
  // s_waitcnt immediate (gfx9 encoding): vmcnt = 63 and expcnt = 7 (no wait), lgkmcnt in bits [11:8]
  constexpr int WAIT_HALF = 0xC07F | (((KS == 2 && PER_KS <= 15) ? PER_KS : 0) << 8);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks == 0) __builtin_amdgcn_s_waitcnt(WAIT_HALF);   // the k-step-0 fragments have landed (LDS returns in order)
    else __builtin_amdgcn_s_waitcnt(0xC07F);              // everything has
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (PASSES == 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
    }
  }
}

// per (ks, j) group: R reads that prefetch the next group (+ R0 for the A fragments of the second k-step in group 0),
// then MF MFMAs
template <int GRP, int NGRP, int R0, int R, int MF>
__device__ __forceinline__ void pin_pipeline() {
  if constexpr (GRP < NGRP) {
    constexpr int reads = (GRP == 0 ? R0 : 0) + (GRP + 1 < NGRP ? R : 0);
    if constexpr (reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, reads, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);
    pin_pipeline<GRP + 1, NGRP, R0, R, MF>();
  }
}

// One K-tile of MFMA work for a wave.  Fragment reads run one MFMA group ahead of their use (register double
// buffering, pinned with sched_group_barrier: hipcc otherwise sinks every read to its first use and drains
// lgkmcnt(0) ten times per tile).
template <int BM, int BN, int BK, int MI, int NI, bool TA, bool TB, int PASSES>
__device__ __forceinline__ void compute_tile(const char* a_hi, const char* b_hi, int wm0, int wn0, int lane,
                                             f32x4_t (&acc)[MI][NI]) {
  constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2;
  constexpr int KS = BK / 32;
  bf16x8_t ah[KS][MI], al[KS][MI], bh[2], bl[2];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    ah[0][i] = read_frag<BM, BK, TA>(a_hi, wm0 + 16 * i, 0, lane);
    if (PASSES == 3) al[0][i] = read_frag<BM, BK, TA>(a_hi + A_TILE, wm0 + 16 * i, 0, lane);
  }
  bh[0] = read_frag<BN, BK, TB>(b_hi, wn0, 0, lane);
  if (PASSES == 3) bl[0] = read_frag<BN, BK, TB>(b_hi + B_TILE, wn0, 0, lane);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int cur = (ks * NI + j) & 1, nxt = cur ^ 1;
      const int nj = (j + 1 < NI) ? j + 1 : 0, nks = (j + 1 < NI) ? ks : ks + 1;
      if (nks < KS) {
        bh[nxt] = read_frag<BN, BK, TB>(b_hi, wn0 + 16 * nj, nks, lane);
        if (PASSES == 3) bl[nxt] = read_frag<BN, BK, TB>(b_hi + B_TILE, wn0 + 16 * nj, nks, lane);
      }
      if (j == 0 && ks == 0 && KS == 2) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ah[KS - 1][i] = read_frag<BM, BK, TA>(a_hi, wm0 + 16 * i, KS - 1, lane);
          if (PASSES == 3) al[KS - 1][i] = read_frag<BM, BK, TA>(a_hi + A_TILE, wm0 + 16 * i, KS - 1, lane);
        }
      }
      if (PASSES == 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks][i], bh[cur], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks][i], bl[cur], acc[i][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks][i], bh[cur], acc[i][j], 0, 0, 0);
    }
  }
  constexpr int NIMGS = PASSES == 3 ? 2 : 1;
  constexpr int RPF = TB ? 2 : 1, RPA = TA ? 2 : 1;  // LDS read instructions per fragment
  __builtin_amdgcn_sched_group_barrier(0x100, (MI * RPA + RPF) * NIMGS, 0);
  pin_pipeline<0, KS * NI, (KS == 2 ? MI * RPA * NIMGS : 0), RPF * NIMGS, MI * PASSES>();
}

// ---- the kernel ------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, int PASSES, bool APL, bool BPL, int NW = 4>
// 8-wave workgroups are meant to run two per CU = 4 waves per SIMD: cap the register allocation at 128 for them
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void gemm_kernel(GemmParams g) {
  static_assert(NW == 4 || (APL && BPL), "8-wave workgroups exist for planes x planes operands only");
  static_assert((BM / WM) * (BN / WN) == NW, "wave grid");
  constexpr int MI = WM / 16, NI = WN / 16;
  constexpr int WAVES_N = BN / WN;
  constexpr int NIMG = PASSES == 3 ? 2 : 1;
  constexpr int A_TILE = BM * BK * 2, B_TILE = BN * BK * 2;  // one bf16 image
  constexpr int A_STAGE = NIMG * A_TILE, B_STAGE = NIMG * B_TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool dbuf = g.dma_stages == 2;       // planes operands double buffered?
  char* lds_a = smem;
  char* lds_b = smem + ((APL && dbuf) ? 2 : 1) * A_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Work unit = (K split, output tile); the XCD-aware tile rasterisation inside each split.  One-row-of-tiles products with
  // split-K (out_layer.fc1 forward: M = 64, 24 column tiles x 32 splits) put the units in split-major order and cut that
  // into 8 contiguous chunks, one per XCD: the column tiles of a K split run on ONE XCD, so the thin activation slice they
  // all re-read ([64, K / splits] planes) is fetched into one L2 instead of eight (PMC: 2.48 GB of fabric reads for 2.04 GB
  // algorithmic before; 0.395 -> 0.383 ms).  Square-ish split-K products (the weight-gradient GEMMs) measured 10 % SLOWER
  // that way and keep the per-split rasterisation.
  int tm, tn, split;
  if (g.splits > 1 && g.tiles_m == 1) {
    const int tiles = g.tiles_m * g.tiles_n;
    const int i = xcd_chunk_index(tiles * g.splits, blockIdx.x);
    split = i / tiles;
    const int t = i - split * tiles;
    tm = t / g.tiles_n;
    tn = t - tm * g.tiles_n;
  } else {
    const int tiles = g.tiles_m * g.tiles_n;
    split = blockIdx.x / tiles;
    tile_coords(g.tiles_m, g.tiles_n, blockIdx.x - split * tiles, tm, tn);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int total_k_tiles = (g.K + BK - 1) / BK;
  const int kt_begin = split * g.k_tiles_per_split;
  int kt_end = kt_begin + g.k_tiles_per_split;
  if (kt_end > total_k_tiles) kt_end = total_k_tiles;
  const int nt = kt_end - kt_begin;

  __amdgpu_buffer_rsrc_t rsrc_a = uniform_rsrc(g.A, g.a_bytes);
  __amdgpu_buffer_rsrc_t rsrc_b = uniform_rsrc(g.B, g.b_bytes);
  __amdgpu_buffer_rsrc_t rsrc_a_lo = uniform_rsrc((const char*)g.A + (APL ? g.a_lo_off : 0), g.a_bytes);
  __amdgpu_buffer_rsrc_t rsrc_b_lo = uniform_rsrc((const char*)g.B + (BPL ? g.b_lo_off : 0), g.b_bytes);

  RegStager<BM, BK, TA, PASSES> ra;
  RegStager<BN, BK, TB, PASSES> rb;
  DmaStager<BM, BK, TA, PASSES, NW> da;
  DmaStager<BN, BK, TB, PASSES, NW> db;
  if (APL) da.init(wave, lane, m0, g.lda, kt_begin);
  else ra.init(tid, m0, g.lda, kt_begin);
  if (BPL) db.init(wave, lane, n0, g.ldb, kt_begin);
  else rb.init(tid, n0, g.ldb, kt_begin);

  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  if (nt > 0) {
    if (APL) da.issue(rsrc_a, rsrc_a_lo, lds_a, wave);
    else ra.load(rsrc_a);
    if (BPL) db.issue(rsrc_b, rsrc_b_lo, lds_b, wave);
    else rb.load(rsrc_b);
    if (!APL) ra.store(lds_a);
    if (!BPL) rb.store(lds_b);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const bool more = (t + 1 < nt);
    const int cur = t & 1;
    char* a_cur = lds_a + ((APL && dbuf) ? cur : 0) * A_STAGE;
    char* b_cur = lds_b + ((BPL && dbuf) ? cur : 0) * B_STAGE;
    if (more && !(g.ablate & 2)) {  // tile t+1: HBM -> LDS (planes) or HBM -> VGPR (fp32) while tile t is multiplied
      if (APL) { if (dbuf) da.issue(rsrc_a, rsrc_a_lo, lds_a + (cur ^ 1) * A_STAGE, wave); }
      else ra.load(rsrc_a);
      if (BPL) { if (dbuf) db.issue(rsrc_b, rsrc_b_lo, lds_b + (cur ^ 1) * B_STAGE, wave); }
      else rb.load(rsrc_b);
    }
    if constexpr ((APL || BPL) && (TA || TB))
      compute_tile_preload<BM, BN, BK, MI, NI, TA, TB, PASSES>(a_cur, b_cur, wm0, wn0, lane, acc);
    else
      compute_tile<BM, BN, BK, MI, NI, TA, TB, PASSES>(a_cur, b_cur, wm0, wn0, lane, acc);
    if (!APL || !BPL || !dbuf) {
      __syncthreads();  // every wave is done reading the single-buffered image(s)
      if (more && !(g.ablate & 4)) {
        if (!APL) ra.store(lds_a);
        else if (!dbuf) da.issue(rsrc_a, rsrc_a_lo, lds_a, wave);
        if (!BPL) rb.store(lds_b);
        else if (!dbuf) db.issue(rsrc_b, rsrc_b_lo, lds_b, wave);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA for tile t+1 has landed
    __syncthreads();                                   // ... and everyone's; tile t's buffers may be refilled
  }

  // ---- epilogue through LDS (operand images are dead after the last barrier) ----
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * (WN + 4));
  float* part = g.partial ? g.partial + (size_t)split * (size_t)g.M * (size_t)g.N : nullptr;
  if (g.epi.adam_p && !part) epilogue_wave_adam<WM, WN, MI, NI, ((APL && BPL) ? (NW == 8 ? 1 : 0) : -1)>(g, acc, slab, m0 + wm0, n0 + wn0, lane);
  else epilogue_wave<WM, WN, MI, NI, 3, ((APL && BPL) ? (NW == 8 ? 1 : 0) : -1)>(g, acc, slab, m0 + wm0, n0 + wn0, lane, part);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N,
                                                            Epilogue epi) {
  const size_t total4 = (size_t)M * (size_t)N / 4;
  const size_t total = (size_t)M * (size_t)N;
  for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < splits; ++z) {
      const float4 v = ld4(partial + (size_t)z * total + i4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const size_t idx = i4 * 4;
    epilogue_vec4(epi, s, (int)(idx / N), (int)(idx % N), N);
  }
}

// Explicit instantiation of every kernel the dispatcher can reach: hipcc (ROCm 7.2) emits the host launch stub for only
// a few of the implicit instantiations of this 9-parameter kernel template, leaving the others undefined at load time.
#define LR2_GEMM_INST(BM, BK, WN, TA, TB, P, APL, BPL) \
  template __global__ void gemm_kernel<BM, 128, BK, 64, WN, TA, TB, P, APL, BPL>(GemmParams);
#define LR2_GEMM_INST_SRC(BM, WN, TA, TB, P)        \
  LR2_GEMM_INST(BM, 64, WN, TA, TB, P, false, false) \
  LR2_GEMM_INST(BM, 64, WN, TA, TB, P, true, false)  \
  LR2_GEMM_INST(BM, 64, WN, TA, TB, P, true, true)   \
  LR2_GEMM_INST(BM, 32, WN, TA, TB, P, true, true)
#define LR2_GEMM_INST_FORM(BM, WN, P)           \
  LR2_GEMM_INST_SRC(BM, WN, false, false, P)     \
  LR2_GEMM_INST_SRC(BM, WN, false, true, P)      \
  LR2_GEMM_INST_SRC(BM, WN, true, true, P)
#define LR2_GEMM_INST_W8(BK, TA, TB) \
  template __global__ void gemm_kernel<128, 128, BK, 64, 32, TA, TB, 3, true, true, 8>(GemmParams);
LR2_GEMM_INST_W8(64, false, false)
LR2_GEMM_INST_W8(32, false, false)
LR2_GEMM_INST_W8(32, false, true)
LR2_GEMM_INST_FORM(128, 64, 1)
LR2_GEMM_INST_FORM(128, 64, 3)
LR2_GEMM_INST_FORM(64, 32, 1)
LR2_GEMM_INST_FORM(64, 32, 3)

template <int BM, int BN, int BK, int WM, int WN, bool TA, bool TB, int PASSES, bool APL, bool BPL>
int launch(const GemmParams& p_in, int splits, hipStream_t stream) {
  GemmParams p = p_in;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.splits = splits;
  if (splits <= 1) p.partial = nullptr;     // (a diagnostics build may pass a workspace for other purposes)
  dim3 grid(p.tiles_m * p.tiles_n * splits);
  constexpr int NIMG = PASSES == 3 ? 2 : 1;
  const size_t main_lds = (size_t)((APL && p.dma_stages == 2) ? 2 : 1) * NIMG * BM * BK * 2 +
                          (size_t)((BPL && p.dma_stages == 2) ? 2 : 1) * NIMG * BN * BK * 2;
  constexpr size_t epi_lds = (size_t)4 * 32 * (WN + 4) * 4;
  const size_t lds = main_lds > epi_lds ? main_lds : epi_lds;
  constexpr size_t max_lds = (size_t)2 * NIMG * BM * BK * 2 + (size_t)2 * NIMG * BN * BK * 2;
  // 8-wave workgroups: NT at either K depth, NN at 32-deep tiles (lr2_gemm never asks for them with a transposed A, and the
  // 64-deep NN variant does not fit 128 VGPRs without scratch: it is not built)
  if constexpr (APL && BPL && BM == 128 && PASSES == 3 && !TA && !(TB && BK == 64)) {
    if (p.waves8) {   // 8 waves per workgroup, wave tile 64 x 32
      auto k8 = gemm_kernel<128, 128, BK, 64, 32, TA, TB, 3, true, true, 8>;
      constexpr size_t epi8 = (size_t)8 * 32 * (32 + 4) * 4;
      const size_t lds8 = main_lds > epi8 ? main_lds : epi8;
      static bool attr_set8 = false;
      if (!attr_set8) {
        if (lr2_allow_dynamic_lds(k8, max_lds > epi8 ? max_lds : epi8, "gemm")) return LR2_ERR_LAUNCH;
        attr_set8 = true;
      }
      LR2_LAUNCH(k8, grid, dim3(512), lds8, stream, p);
      return lr2_launch_status(__func__);
    }
  }
  auto kern = gemm_kernel<BM, BN, BK, WM, WN, TA, TB, PASSES, APL, BPL>;
  static bool attr_set = false;
  if (!attr_set) {
    if (lr2_allow_dynamic_lds(kern, max_lds > epi_lds ? max_lds : epi_lds, "gemm")) return LR2_ERR_LAUNCH;
    attr_set = true;
  }
  LR2_LAUNCH(kern, grid, dim3(NTHREADS), lds, stream, p);
  return lr2_launch_status(__func__);
}

template <int BK, bool TA, bool TB, bool APL, bool BPL>
int dispatch(const GemmParams& p, int splits, int bm, int passes, hipStream_t stream) {
  if (passes == 1) {
    if (bm == 64) return launch<64, 128, BK, 64, 32, TA, TB, 1, APL, BPL>(p, splits, stream);
    return launch<128, 128, BK, 64, 64, TA, TB, 1, APL, BPL>(p, splits, stream);
  }
  if (bm == 64) return launch<64, 128, BK, 64, 32, TA, TB, 3, APL, BPL>(p, splits, stream);
  return launch<128, 128, BK, 64, 64, TA, TB, 3, APL, BPL>(p, splits, stream);
}

template <int BK, bool APL, bool BPL>
int dispatch_form(const GemmParams& p, int splits, int bm, int passes, int ta, int tb, hipStream_t s) {
  if (!ta && !tb) return dispatch<BK, false, false, APL, BPL>(p, splits, bm, passes, s);
  if (!ta && tb) return dispatch<BK, false, true, APL, BPL>(p, splits, bm, passes, s);
  if (ta && tb) return dispatch<BK, true, true, APL, BPL>(p, splits, bm, passes, s);
  return LR2_ERR_ARG;  // (1,0) is not a form the path needs
}

Epilogue to_device_epilogue(const lr2_epilogue* e) {
  Epilogue d{};
  d.bias = (const float*)e->bias;
  d.resid = (const float*)e->resid;
  d.aux_z = (const float*)e->aux_z;
  d.out = (float*)e->out;
  d.out_z = (float*)e->out_z;
  d.out_hi = (bf16_t*)e->out_hi;
  d.lo_off = (size_t)e->out_lo_off;
  d.ld_planes = e->ld_planes;
  d.ld_resid = e->ld_resid;
  d.ld_aux = e->ld_aux;
  d.ld_out = e->ld_out;
  d.ld_z = e->ld_z;
  d.act = (uint32_t)e->act & 15u;
  d.accumulate = e->accumulate ? 1u : 0u;
  d.alpha = e->alpha;
  if (e->drop_p > 0.f) {
    d.drop_scale = 1.0f / (1.0f - e->drop_p);
    d.drop_thr = dropout_threshold(e->drop_p);
    d.drop_key = (((uint64_t)e->drop_site) << 40) ^ (e->drop_seed * 0x9E3779B97F4A7C15ull);
    if (e->drop_seed_dev) {          // key formed in the kernel: carry the seed's by-value part and the site
      d.dev_scalar = e->drop_seed_dev;
      d.seed_dev = 1;
      d.drop_key = e->drop_seed;
      d.drop_thr |= e->drop_site << 16;
    }
  }
  if (e->adam_p) {  // same double -> float conversions as lr2_adamw_multi
    d.adam_p = (float*)e->adam_p;
    d.adam_m = (float*)e->adam_m;
    d.adam_v = (float*)e->adam_v;
    d.adam_lr = (float)e->adam_lr;
    if (e->adam_lr_dev) {
      d.dev_scalar = e->adam_lr_dev;
      d.lr_dev = 1;
    }
    d.adam_b1 = (float)e->adam_beta1;
    d.adam_b2 = (float)e->adam_beta2;
    d.adam_ob1 = (float)(1.0 - e->adam_beta1);
    d.adam_ob2 = (float)(1.0 - e->adam_beta2);
    d.adam_eps = (float)e->adam_eps;
    d.adam_wd = (float)e->adam_weight_decay;
  }
  return d;
}

int launch_gemm256_nt(const GemmParams& p, hipStream_t stream);   // gemm256.hip
int launch_gemm256_tn(const GemmParams& p, int splits, hipStream_t stream);

}  // namespace lr2gemm
using namespace lr2gemm;

static uint64_t g_launches[3] = {0, 0, 0};     // 256 NT, 256 TN, general family (host-side counters: lr2_gemm_launch_counts)
extern "C" int lr2_gemm_launch_counts(uint64_t counts[3]) {
  if (!counts) return LR2_ERR_ARG;
  for (int i = 0; i < 3; ++i) counts[i] = g_launches[i];
  return 0;
}

// How lr2_gemm schedules an eligible large NT product of planes (block_m = 256, no fused dropout): *rows_256 = the leading rows that
// go to the 256 x 256 kernel as WHOLE rounds of the chip (0: no row split -- one launch), *tail_block_m = the tile height of the
// general kernel that takes the remaining rows.  Pure host arithmetic (no device call): tests read the plan, lr2_gemm follows it.
extern "C" int lr2_gemm_row_split_plan(int M, int N, int K, int* rows_256, int* tail_block_m) {
  if (!rows_256 || !tail_block_m || M <= 0 || N <= 0 || K <= 0) return LR2_ERR_ARG;
  *rows_256 = 0;
  *tail_block_m = 128;
  static int rs_env = -1;
  if (rs_env < 0) {
    const char* e = getenv("LR2_GEMM_ROWSPLIT");
    rs_env = e ? atoi(e) : 1;
  }
  const int tn256 = (N + 255) / 256, tiles256 = ((M + 255) / 256) * tn256;
  const int full = tiles256 / 256, rem = tiles256 - full * 256;
  const int M1 = ((full * 256) / tn256) * 256;          // rows of the tile rows that fit `full` rounds
  // (up to 4 whole rounds: behind 18 rounds -- the encoders at 512 frames -- a thin last round is 2 % of the launch, not worth a seam)
  if (!rs_env || full < 1 || full > 4 || rem <= 0 || 2 * rem >= 256 || (K % 64) != 0 || M1 <= 0 || M1 >= M) return 0;
  *rows_256 = M1;
  // tail tiles: 64 rows when 128-row tiles would not fill the 512 resident slots, or would leave a thin last round
  const int M2 = M - M1, t128 = ((M2 + 127) / 128) * ((N + 127) / 128), last = t128 % 512;
  *tail_block_m = (t128 < 512 || (last > 0 && last <= 128)) ? 64 : 128;
  return 0;
}

extern "C" int lr2_gemm(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int trans_a, int trans_b,
                        uint64_t a_bytes, uint64_t b_bytes, int a_planes, uint64_t a_lo_off, int b_planes,
                        uint64_t b_lo_off, const lr2_epilogue* epi, void* splitk_ws, int splits, int block_m, int passes,
                        void* stream) {
  if (!A || !B || M <= 0 || N <= 0 || K <= 0 || !epi || (!epi->out && !epi->out_hi && !epi->adam_p)) return LR2_ERR_ARG;
  if (epi->adam_p && (!epi->adam_m || !epi->adam_v || epi->out || epi->out_hi || epi->ld_out % 4)) return LR2_ERR_ARG;
  if (passes != 1 && passes != 3) return LR2_ERR_ARG;
  if (epi->drop_seed_dev && epi->drop_site >= 65536u) return LR2_ERR_ARG;      // the site travels in 16 bits beside the threshold
  if (epi->adam_p && epi->drop_p > 0.f) return LR2_ERR_ARG;                    // the update consumes the result: nothing to mask
  if (epi->colsum && (!trans_a || !trans_b || !epi->colsum_ws || (M % 4))) return LR2_ERR_ARG;   // weight-gradient form only
  const bool want256 = block_m == 256;
  if (block_m != 64) block_m = 128;
  // K-contiguous operands need whole K tiles (a ragged K would read into the next row, not zeros); ragged M / N are
  // handled by the zero-filling range check on loads plus masked stores.
  static int bk_env = -1, ablate = -1, stages = -1, w8_env = 0;
  if (ablate < 0) {
    const char* e = getenv("LR2_GEMM_ABLATE");
    ablate = e ? atoi(e) : 0;
    const char* st = getenv("LR2_GEMM_DMA_STAGES");
    stages = st ? atoi(st) : 0;
    const char* bk = getenv("LR2_GEMM_BK");
    bk_env = bk ? (atoi(bk) == 64 ? 64 : 32) : 0;
    const char* w8 = getenv("LR2_GEMM_W8");
    w8_env = w8 ? atoi(w8) : 1;
  }
  // planes x planes with a contraction-strided B (NN, TN): 32-deep K tiles, two LDS stages (64 KB), two workgroups per
  // CU -- measured +10-14 % on the wgrad shapes; NT and anything with an fp32 operand: 64-deep tiles, one stage.
  static int g256_env = -1;
  if (g256_env < 0) {
    const char* e = getenv("LR2_GEMM_256");
    g256_env = e ? atoi(e) : 1;
  }
  // Large NT products of planes: the 256 x 256 ping-pong kernel (gemm256.hip, 32-deep K steps) when the caller asks for it
  // (block_m == 256) and the shape is eligible; otherwise the general kernel family.
  const bool use256 = want256 && g256_env && a_planes && b_planes && !trans_a && !trans_b && passes == 3 && splits <= 1 &&
                      (K % 32) == 0 && a_bytes <= 0xFFFFFD00ull && b_bytes <= 0xFFFFFD00ull && !epi->adam_p &&
                      ((epi->act == 2) + (epi->resid != nullptr) + (epi->accumulate != 0)) <= 1;   // one request slot per element
  // Long-contraction TN products of planes (weight gradients at >= 4096 token rows): the TN form of that kernel, tiles x K-splits
  // in one round of the chip, raw slabs + the reducer below.
  const bool use256tn = want256 && g256_env && a_planes && b_planes && trans_a && trans_b && passes == 3 &&
                        a_bytes <= 0xFFFFFD00ull && b_bytes <= 0xFFFFFD00ull && !epi->adam_p &&
                        ((epi->act == 2) + (epi->resid != nullptr) + (epi->accumulate != 0)) <= 1;
  const int BK = (use256 || use256tn) ? 32 : (a_planes && b_planes) ? (bk_env ? bk_env : (trans_b ? 32 : 64)) : 64;
  if ((!trans_a || !trans_b) && (K % BK != 0)) return LR2_ERR_SHAPE;
  const int a_align = a_planes ? 8 : 4, b_align = b_planes ? 8 : 4;  // 16-byte rows
  if ((lda % a_align) || (ldb % b_align) || (N % 4)) return LR2_ERR_SHAPE;
  if ((epi->out && epi->ld_out % 4) || (epi->out_z && epi->ld_z % 4) || (epi->resid && epi->ld_resid % 4) ||
      (epi->aux_z && epi->ld_aux % 4) || (epi->out_hi && epi->ld_planes % 4))
    return LR2_ERR_SHAPE;
  if (a_bytes >= (1ull << 32) || b_bytes >= (1ull << 32) || a_lo_off >= (1ull << 32) || b_lo_off >= (1ull << 32))
    return LR2_ERR_SHAPE;
  if (!a_planes && b_planes) return LR2_ERR_ARG;  // (fp32 A, planes B) is not a combination the path needs
  if (epi->accumulate && !epi->out) return LR2_ERR_ARG;
  if (splits < 1) splits = 1;
  const int total_k_tiles = (K + BK - 1) / BK;
  if (splits > total_k_tiles) splits = total_k_tiles;
  if (splits > 1 && !splitk_ws) return LR2_ERR_ARG;
  GemmParams p{};
  p.A = A;
  p.B = B;
  p.M = M;
  p.N = N;
  p.K = K;
  p.lda = lda;
  p.ldb = ldb;
  p.a_bytes = (uint32_t)a_bytes;
  p.b_bytes = (uint32_t)b_bytes;
  p.a_lo_off = (uint32_t)a_lo_off;
  p.b_lo_off = (uint32_t)b_lo_off;
  p.k_tiles_per_split = (total_k_tiles + splits - 1) / splits;
  splits = (total_k_tiles + p.k_tiles_per_split - 1) / p.k_tiles_per_split;
  p.partial = (splits > 1 || (ablate & 32)) ? (float*)splitk_ws : nullptr;
  p.epi = to_device_epilogue(epi);
  p.epi.stream_nt = (ablate & 64) ? 0 : 1;   // fused AdamW: +2-3 % on the 12-GB p / m / v stream (A/B with LR2_GEMM_ABLATE=64)
  p.ablate = ablate;
  // Results of 256 MiB and more (the encoders' token GEMMs at M >= 1e5 rows: larger than L2 + the 256-MiB MALL, streamed from HBM by
  // the next kernel whatever we do) are stored non-temporally: +1..4 % on the K = 768 shapes of the 256 x 256 kernel, A/B with
  // LR2_GEMM_ABLATE=128 (= off).  Smaller results keep the default policy: their consumer may still find them on chip.
  p.epi.store_nt = (!(ablate & 128) && (uint64_t)M * (uint64_t)N * 4ull >= (256ull << 20)) ? 1 : 0;
  // 128 x 128 planes tiles, NT / NN: 8-wave workgroups (two workgroups per CU = 4 waves per SIMD) overlap the MFMA issue,
  // the LDS-DMA issue and the fragment waits of different waves: +5..21 % over 4 waves (tools/gemm_bench.py); TN: equal.
  p.waves8 = (w8_env && !trans_a) ? 1 : 0;
  // 64-deep planes tiles: one image + 2 workgroups/CU measured faster than two images + 1 workgroup/CU
  p.dma_stages = stages ? (stages == 1 ? 1 : 2) : (BK == 32 ? 2 : 1);
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (use256) {
    // ROW SPLIT (round 4).  A 256 x 256 tile owns a whole CU (128 KiB of LDS), so a launch runs in rounds of 256 tiles and a last
    // round that is mostly empty costs a full tile time on a mostly idle chip (M = 12 544, N = 3072: 588 tiles = 2.3 rounds).  When
    // the last round would be less than half full, the rows that fill WHOLE rounds go to the 256 x 256 kernel and the remaining
    // rows to the 128- / 64-row kernels (2-3 workgroups per CU, short tiles): two launches, no partial slabs, no reduction, each
    // output row computed by exactly one of them.  Measured in one process (tools/dbg/rowsplit_ab.py): 12 544 x 3072 x 768
    // 199 (NN, 128-row tiles) / 184 (three rounds) -> 166 us; 25 088 x 768 x 3072 340 -> 279; 6272 x 3072 x 768 95 -> 91.
    // Stream-K over the same tiles was priced and not built: a partial 256 x 256 tile is a 256-KiB fp32 slab, ~2 per workgroup
    // (128 MB written and read back per launch), more than the idle part of the last round costs.
    // Not with a fused dropout mask: its element index is relative to the launch's first row.
    int M1 = 0, bm2 = 128;
    if (epi->drop_p <= 0.f && lr2_gemm_row_split_plan(M, N, K, &M1, &bm2) == 0 && M1 > 0) {
      GemmParams p1 = p;
      p1.M = M1;
      rc = launch_gemm256_nt(p1, s);
      if (rc) return rc;
      ++g_launches[0];
      ++g_launches[2];
      GemmParams p2 = p;
      const int M2 = M - M1;
      p2.M = M2;
      p2.A = (const char*)A + (size_t)M1 * (size_t)lda * 2u;              // planes: 2-byte elements, lo plane at the same offset
      p2.a_bytes = (uint32_t)(a_bytes - (uint64_t)M1 * (uint64_t)lda * 2u);
      Epilogue& e2 = p2.epi;
      if (e2.resid) e2.resid += (size_t)M1 * e2.ld_resid;
      if (e2.aux_z) e2.aux_z += (size_t)M1 * e2.ld_aux;
      if (e2.out) e2.out += (size_t)M1 * e2.ld_out;
      if (e2.out_z) e2.out_z += (size_t)M1 * e2.ld_z;
      if (e2.out_hi) e2.out_hi += (size_t)M1 * e2.ld_planes;
      p2.k_tiles_per_split = K / 64;
      p2.partial = nullptr;
      p2.dma_stages = stages ? (stages == 1 ? 1 : 2) : 1;
      return dispatch_form<64, true, true>(p2, 1, bm2, 3, 0, 0, s);
    }
    ++g_launches[0];
    return launch_gemm256_nt(p, s);
  }
  const bool fused_colsum = use256tn && epi->colsum;
  if (fused_colsum) p.epi.colsum_partial = (float*)epi->colsum_ws;
  ++g_launches[use256tn ? 1 : 2];
  if (use256tn) rc = launch_gemm256_tn(p, splits, s);
  else if (a_planes && b_planes && BK == 32) rc = dispatch_form<32, true, true>(p, splits, block_m, passes, trans_a, trans_b, s);
  else if (a_planes && b_planes) rc = dispatch_form<64, true, true>(p, splits, block_m, passes, trans_a, trans_b, s);
  else if (a_planes) rc = dispatch_form<64, true, false>(p, splits, block_m, passes, trans_a, trans_b, s);
  else rc = dispatch_form<64, false, false>(p, splits, block_m, passes, trans_a, trans_b, s);
  if (rc) return rc;
  if (splits > 1) {
    const size_t total4 = (size_t)M * N / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    LR2_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)splitk_ws, splits, M, N, p.epi);
    if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  }
  if (epi->colsum) {
    // bias gradient of the layer this weight gradient belongs to: the 256 x 256 kernel left splits * tiles_n partial rows; any
    // other path sums the columns of A in a pass of its own
    if (fused_colsum)
      return lr2_colsum_partials_finish(epi->colsum_ws, splits * ((N + 255) / 256), M, M, epi->colsum, 0, stream);
    return lr2_colsum(A, a_planes, a_planes ? a_lo_off / 2 : 0, K, M, lda, epi->colsum_ws, K < 128 ? K : 128, epi->colsum, stream);
  }
  return 0;
}
