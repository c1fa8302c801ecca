// 256 x 256 "ping-pong" GEMM for large NT products of bf16 hi/lo planes:  C[M,N] = A[M,K] . B[N,K]^T  (fp32-grade,
// three v_mfma_f32_16x16x32_bf16 products per tile pair: lo*hi + hi*lo + hi*hi, fp32 accumulate).
//
// Replaces, for the token GEMMs of the encoders and heads (M >= a few thousand rows), the cuBLAS sgemm calls reached
// through nn.Linear in the reference (tencentpretrain/layers/multi_headed_attn.py:55-76, position_ffn.py:12-15,
// finetune/ppo.py:164-170) -- same contract as gemm.hip's NT form, same fused epilogue (gemm_common.h).
//
// Structure (CDNA4, one workgroup of 8 waves per CU, 128 KiB of LDS):
//   * tile 256 x 256, 32-deep K steps; wave (wr, wc) of a 2 x 4 grid owns a 128 x 64 block = 8 x 4 accumulator tiles
//     (128 accumulator VGPRs).  A K step is 96 MFMAs per wave (32 tiles x 3 products) against 64 KiB of operands staged
//     for the whole workgroup: 1.5x the matrix work per staged byte of a plain-bf16 256 x 256 x 64 step.
//   * operands travel HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane) into a ring of 2 stages x 4 parts
//     (A rows of accumulator half 0 / 1, B columns of half 0 / 1; hi and lo plane of a part are adjacent).  A part is
//     refilled for K step t+2 as soon as its last reader of step t has passed, so 6-7 parts (12-14 KiB per wave) are in
//     flight at any time; waits are COUNTED (s_waitcnt vmcnt(12) / (6)), never vmcnt(0), and barriers are bare s_barrier
//     (a __syncthreads() would drain every in-flight DMA).  Past the last K step the refills become out-of-range
//     requests (the buffer descriptor's range check writes zeros) so the counts stay uniform.
//   * a K step is four phases, one accumulator quadrant each (A half x B half, 24 MFMAs): phase = LOAD section (issue 2
//     DMA pieces, ds_read_b128 the fragments this phase is missing: 12 / 4 / 8 / 4 reads) + MFMA section.  The two wave
//     groups (waves 0-3 and 4-7 = one wave per SIMD each) run ONE SECTION APART: while a group issues its 24 MFMAs its
//     SIMD partner loads, so every SIMD's matrix pipe always has exactly one wave feeding it and the LDS / DMA work of
//     the other hides underneath.  Two s_barrier per phase keep the groups in that lock step.
//   * LDS image of a part: [128 rows][4 units of 16 B], unit u of row r at u ^ swz(r) (four 64-B rows share a 256-B
//     bank row: conflict-free ds_read_b128); LDS-DMA writes lane-linearly, so the swizzle is applied to the SOURCE address.
//   * hazards: a fragment read happens at least one barrier after every wave's counted wait for that part (RAW); a part is
//     refilled one barrier after both groups' reads of it were retired by lgkmcnt(0) (WAR).
#include <stdlib.h>

#include "gemm_common.h"

namespace lr2gemm {
namespace g256 {

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int NWAVES = 8;
constexpr int PLANE = 128 * BK * 2;   // one plane of one part: 128 rows x 32 k x 2 B = 8 KiB
constexpr int PART = 2 * PLANE;       // hi + lo
constexpr int STAGE = 4 * PART;       // parts A0, B0, B1, A1
constexpr int LDS_BYTES = 2 * STAGE;  // 128 KiB
constexpr int SLOT_A0 = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_A1 = 3;
constexpr uint32_t OOB = 0xFFFFFF00u;  // voffset beyond any descriptor this library builds (operands are < 4 GiB - 512 B)

__device__ __forceinline__ int swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }

template <int IMM>
__device__ __forceinline__ bf16x8_t lds_read16(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
  return __builtin_bit_cast(bf16x8_t, v);
}

struct Ctx {
  __amdgpu_buffer_rsrc_t a_hi, a_lo, b_hi, b_lo;
  uint32_t voff_a[2], voff_b[2];   // per-lane source byte offsets of this wave's piece of part A(ah) / B(bh) at K step 0
  uint32_t rd_a[2], rd_b[2];       // per-lane LDS read bases for stage 0 / 1
  char* smem;
  int wave, nt;
};

// Two DMA pieces (hi + lo plane) of one part for K step `tile` into stage `stage`.
template <int SLOT, bool IS_A, int HALF>
__device__ __forceinline__ void issue_part(const Ctx& c, int tile, int stage) {
  const uint32_t base = IS_A ? c.voff_a[HALF] : c.voff_b[HALF];
  const uint32_t v = (tile < c.nt) ? base + (uint32_t)tile * (BK * 2) : OOB;
  char* dst = c.smem + stage * STAGE + SLOT * PART + c.wave * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? c.a_hi : c.b_hi, LDS_PTR(dst), 16, v, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? c.a_lo : c.b_lo, LDS_PTR(dst + PLANE), 16, v, 0, 0, 0);
}

template <int SLOT, int MIH>
__device__ __forceinline__ void read_a_half(uint32_t base, bf16x8_t (&hi)[MIH], bf16x8_t (&lo)[MIH]) {
  hi[0] = lds_read16<SLOT * PART + 0 * 1024>(base);
  hi[1] = lds_read16<SLOT * PART + 1 * 1024>(base);
  hi[2] = lds_read16<SLOT * PART + 2 * 1024>(base);
  if constexpr (MIH == 4) hi[3] = lds_read16<SLOT * PART + 3 * 1024>(base);
  lo[0] = lds_read16<SLOT * PART + PLANE + 0 * 1024>(base);
  lo[1] = lds_read16<SLOT * PART + PLANE + 1 * 1024>(base);
  lo[2] = lds_read16<SLOT * PART + PLANE + 2 * 1024>(base);
  if constexpr (MIH == 4) lo[3] = lds_read16<SLOT * PART + PLANE + 3 * 1024>(base);
}
template <int SLOT>
__device__ __forceinline__ void read_b_half(uint32_t base, bf16x8_t (&hi)[2], bf16x8_t (&lo)[2]) {
  hi[0] = lds_read16<SLOT * PART + 0 * 1024>(base);
  hi[1] = lds_read16<SLOT * PART + 1 * 1024>(base);
  lo[0] = lds_read16<SLOT * PART + PLANE + 0 * 1024>(base);
  lo[1] = lds_read16<SLOT * PART + PLANE + 1 * 1024>(base);
}

// End of a LOAD section: retire the DMA parts the NEXT load section reads (counted), retire this section's fragment
// reads, meet the other group.  s_waitcnt immediates (gfx9 encoding): vmcnt[3:0] in bits 3:0, vmcnt[5:4] in bits 15:14,
// expcnt 7 (no wait) in bits 6:4, lgkmcnt in bits 11:8.
template <int VM>
__device__ __forceinline__ void end_load_section() {
  constexpr int imm = (VM & 15) | ((VM >> 4) << 14) | (7 << 4) | (0 << 8);
  __builtin_amdgcn_s_waitcnt(imm);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// 24 MFMAs of one accumulator quadrant: products lo*hi, hi*lo, hi*hi, eight independent accumulators between two
// updates of the same one.
template <int AH, int BH, int MIH>
__device__ __forceinline__ void mfma_section(f32x4_t (&acc)[2 * MIH][4], const bf16x8_t (&ahi)[MIH], const bf16x8_t (&alo)[MIH],
                                             const bf16x8_t (&bhi)[2], const bf16x8_t (&blo)[2]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MIH; ++i)
      acc[AH * MIH + i][BH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(alo[i], bhi[j], acc[AH * MIH + i][BH * 2 + j], 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MIH; ++i)
      acc[AH * MIH + i][BH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[i], blo[j], acc[AH * MIH + i][BH * 2 + j], 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < MIH; ++i)
      acc[AH * MIH + i][BH * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahi[i], bhi[j], acc[AH * MIH + i][BH * 2 + j], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// One K step (tile t, compile-time stage S).  A part may be refilled once its last reader has passed (A0 after load
// section 0, B1 after 1, A1 after 2, B0 after 3).  Two placements of the refills (V, A/B-tested in one process with
// LR2_GEMM256_VARIANT):
//   V = 0: as early as possible, 2 pieces per section: 0: B0(t+1) [other stage]  1: A0(t+2)  2: B1(t+2)  3: A1(t+2)
//   V = 1: in the two light sections only (no section carries 12 fragment reads AND DMA issue):
//          1: B0(t+1), A0(t+2)   3: B1(t+2), A1(t+2)
// The counted waits leave exactly the parts issued after the one the NEXT section reads in flight (2 pieces per part).
template <int S, int V, int MIH>
__device__ __forceinline__ void k_step(const Ctx& c, int t, f32x4_t (&acc)[2 * MIH][4]) {
  bf16x8_t ahi[MIH], alo[MIH], bhi[2], blo[2];
  // phase 0: quadrant (A0, B0)
  if (V == 0) issue_part<SLOT_B0, false, 0>(c, t + 1, S ^ 1);
  read_a_half<SLOT_A0, MIH>(c.rd_a[S], ahi, alo);
  read_b_half<SLOT_B0>(c.rd_b[S], bhi, blo);
  end_load_section<V == 0 ? 12 : 10>();
  mfma_section<0, 0, MIH>(acc, ahi, alo, bhi, blo);
  // phase 1: (A0, B1)
  if (V == 1) issue_part<SLOT_B0, false, 0>(c, t + 1, S ^ 1);
  issue_part<SLOT_A0, true, 0>(c, t + 2, S);
  read_b_half<SLOT_B1>(c.rd_b[S], bhi, blo);
  end_load_section<12>();
  mfma_section<0, 1, MIH>(acc, ahi, alo, bhi, blo);
  // phase 2: (A1, B1)
  if (V == 0) issue_part<SLOT_B1, false, 1>(c, t + 2, S);
  read_a_half<SLOT_A1, MIH>(c.rd_a[S], ahi, alo);
  end_load_section<V == 0 ? 12 : 10>();
  mfma_section<1, 1, MIH>(acc, ahi, alo, bhi, blo);
  // phase 3: (A1, B0)
  if (V == 1) issue_part<SLOT_B1, false, 1>(c, t + 2, S);
  issue_part<SLOT_A1, true, 1>(c, t + 2, S);
  read_b_half<SLOT_B0>(c.rd_b[S], bhi, blo);
  end_load_section<6>();
  mfma_section<1, 0, MIH>(acc, ahi, alo, bhi, blo);
}

// MIH = accumulator tiles per half of a wave's rows: 4 -> the 256 x 256 tile; 3 -> a 192 x 256 tile (wave tile 96 x 64) for launches
// of less than one round of 256-row tiles that fit one round of 192-row tiles too (M = 12 544, N = 768: 147 -> 198 workgroups, each
// with three quarters of the work).  Same ring: an A part then holds 96 rows, the two waves whose 16-row pieces fall beyond them issue
// out-of-range requests (zeros into the unused quarter of the part), so every wave's counted waits stay as they are.
template <int V, int MIH = 4>
__global__ __launch_bounds__(512, 2) void gemm256_nt_kernel(GemmParams g) {
  constexpr int BMT = 64 * MIH, WMT = 32 * MIH;      // tile rows, wave-tile rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  int tm, tn;
  tile_coords(g.tiles_m, g.tiles_n, blockIdx.x, tm, tn, g.strip_n > 0 ? g.strip_n : 8);
  const int m0 = tm * BMT, n0 = tn * BN;

  Ctx c;
  c.smem = smem;
  c.wave = wave;
  c.nt = (g.ablate & 16) ? 0 : g.K / BK;      // diagnostics (LR2_GEMM_ABLATE): 16 = no main loop, 8 = no epilogue memory traffic
  c.a_hi = uniform_rsrc(g.A, g.a_bytes);
  c.a_lo = uniform_rsrc((const char*)g.A + g.a_lo_off, g.a_bytes);
  c.b_hi = uniform_rsrc(g.B, g.b_bytes);
  c.b_lo = uniform_rsrc((const char*)g.B + g.b_lo_off, g.b_bytes);
  {
    // this wave's 1-KiB piece of a part = local rows wave*16 .. +16; lane l fills unit (l & 3) of row (l >> 2), which holds
    // K chunk (l & 3) ^ swz(row)
    const int lr = wave * 16 + (lane >> 2);
    const uint32_t ku = (uint32_t)((lane & 3) ^ swz(lr)) * 16u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // part A(h): rows wr*WMT + h*(WMT/2) + [0, WMT/2) of both wr (part-local row lr = wr*(WMT/2) + that index)
      const int awr = lr / (WMT / 2), ain = lr - awr * (WMT / 2);
      const int arow = m0 + awr * WMT + h * (WMT / 2) + ain;
      const int bcol = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);    // part B(h): cols wc*64 + h*32 + [0, 32) of all wc
      const uint64_t oa = (uint64_t)arow * (uint64_t)g.lda * 2u + ku;
      const uint64_t ob = (uint64_t)bcol * (uint64_t)g.ldb * 2u + ku;
      c.voff_a[h] = (lr < WMT && oa < (uint64_t)OOB) ? (uint32_t)oa : OOB;
      c.voff_b[h] = ob < (uint64_t)OOB ? (uint32_t)ob : OOB;
    }
    const int r16 = lane & 15;
    const uint32_t lane_off = (uint32_t)(r16 * 64 + (((lane >> 4) ^ swz(r16)) * 16));
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      c.rd_a[s] = sm + s * STAGE + wr * (MIH * 1024) + lane_off;     // A part: local row wr*(16 MIH) + i*16 + r16
      c.rd_b[s] = sm + s * STAGE + wc * 2048 + lane_off;     // B part: local row wc*32 + j*16 + r16
    }
  }

  f32x4_t acc[2 * MIH][4];
#pragma unroll
  for (int i = 0; i < 2 * MIH; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: everything of K steps 0 and 1 except B0(1), in the steady-state issue order
  issue_part<SLOT_A0, true, 0>(c, 0, 0);
  issue_part<SLOT_B1, false, 1>(c, 0, 0);
  issue_part<SLOT_A1, true, 1>(c, 0, 0);
  issue_part<SLOT_B0, false, 0>(c, 0, 0);
  issue_part<SLOT_A0, true, 0>(c, 1, 1);
  issue_part<SLOT_B1, false, 1>(c, 1, 1);
  issue_part<SLOT_A1, true, 1>(c, 1, 1);
  end_load_section<6>();                        // A0(0), B1(0), A1(0), B0(0) have landed, everyone's
  if (wr == 1) {                                // waves 4-7 run one section behind waves 0-3 (wr is wave-uniform)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // diagnostics (LR2_GEMM_ABLATE & 32, never in a timed run): shader cycles and 100 MHz ticks of the main loop, per
  // workgroup, into the split-K workspace pointer (unused by this kernel) -> the clock the chip holds under this load
  uint64_t tc0 = 0, tr0 = 0;
  if (g.ablate & 32) {
    tc0 = __builtin_amdgcn_s_memtime();
    tr0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int t = 0; t < c.nt; t += 2) {
    k_step<0, V, MIH>(c, t, acc);
    if (t + 1 < c.nt) k_step<1, V, MIH>(c, t + 1, acc);
  }
  if ((g.ablate & 32) && g.partial && tid == 0) {
    const uint64_t tc1 = __builtin_amdgcn_s_memtime(), tr1 = __builtin_amdgcn_s_memrealtime();
    uint64_t* dbg = reinterpret_cast<uint64_t*>(g.partial) + 2 * (size_t)blockIdx.x;
    dbg[0] = tc1 - tc0;
    dbg[1] = tr1 - tr0;
  }
  if (wr == 0) {                                // same number of barriers for every wave
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail refills have landed (zeros): LDS is reusable
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * (64 + 4));
  if (g.ablate & 8) {
    GemmParams g2 = g;
    g2.M = 0;
    epilogue_wave<WMT, 64, 2 * MIH, 4, 1>(g2, acc, slab, m0 + wr * WMT, n0 + wc * 64, lane, nullptr);
    return;
  }
  epilogue_wave<WMT, 64, 2 * MIH, 4, 1>(g, acc, slab, m0 + wr * WMT, n0 + wc * 64, lane, nullptr);
}


}  // namespace g256

// ---------------------------------------------------------------------------------------------------------------------
// TN form of the same kernel:  C[M,N] = A^T . B  with A [K][lda] (M contiguous) and B [K][ldb] (N contiguous), both bf16 hi/lo
// planes -- the weight gradient dW = dY^T X of an nn.Linear at M_tokens = K >= 1e4 (encoder training, the stage-1 head at 20
// tags): the output is small (768 x 3072: 36 tiles), the contraction is long, so the grid is tiles x K-splits (one round of the
// chip) and the epilogue is a raw fp32 slab per split + the fixed-order reducer of gemm.hip.
//   replaces: autograd of nn.Linear's weight in tencentpretrain/layers/position_ffn.py:12-15, multi_headed_attn.py:55-76,
//   finetune/ppo.py:164-170 (cuBLAS sgemm with a transposed operand upstream).
// Same ring, same four-phase / two-group ping-pong, same counted waits as the NT kernel above.  What differs:
//   * a part's LDS image is [32 k-rows][16 units of 8 consecutive m (or n)]: the operands keep their HBM orientation, a DMA
//     piece of a wave is 4 k-rows x 256 B; unit u of k-row k sits at u ^ swz_tr(k) (XOR on 32-byte chunks, as gemm.hip's
//     contraction-strided image: conflict-free transposed reads);
//   * fragments are read with ds_read_b64_tr_b16 (two per fragment: k-rows 8g + q and 8g + q + 4): twice the LDS instructions
//     of the NT form for the same bytes;
//   * the ring is laid out [A half][stage][plane] / [B half][stage][plane] so that every fragment address is one of six per-lane
//     bases (the XOR swizzle does not commute with the tile offset) plus an immediate < 64 KiB.
namespace g256t {

using g256::BK;
using g256::BM;
using g256::BN;
using g256::OOB;
constexpr int PLANE = 32 * 256;        // one plane of one part: 32 k-rows x 128 m x 2 B = 8 KiB
constexpr int STAGE_STRIDE = 2 * PLANE;   // hi + lo of one stage of one part
constexpr int HALF_STRIDE = 2 * STAGE_STRIDE;
constexpr int REGION = 2 * HALF_STRIDE;   // A region, then B region: 64 KiB each
constexpr int LDS_BYTES = 2 * REGION;

__device__ __forceinline__ int swz_tr16(int k) { return ((((k & 3) | (((k >> 3) & 1) << 2))) << 1) & 15; }

template <int IMM>
__device__ __forceinline__ bf16x8_t lds_read_tr(uint32_t addr) {
  u32x2_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(IMM));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(IMM + 4 * 256));
  u32x4_t v = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bf16x8_t, v);
}

struct Ctx {
  __amdgpu_buffer_rsrc_t a_hi, a_lo, b_hi, b_lo;
  uint32_t voff_a[2], voff_b[2];   // per-lane source byte offsets of this wave's piece of part A(h) / B(h) at this split's K step 0
  uint32_t kstep_a, kstep_b;       // bytes per 32-row K step
  uint32_t rd_a[4], rd_b[2];       // per-lane LDS read bases of the wave's A tiles i = 0..3 / B tiles j = 0..1 (stage 0, half 0, hi)
  char* smem;
  int wave, nt;
};

template <bool IS_A, int HALF>
__device__ __forceinline__ void issue_part(const Ctx& c, int tile, int stage) {
  const uint32_t base = IS_A ? c.voff_a[HALF] : c.voff_b[HALF];
  const uint32_t step = IS_A ? c.kstep_a : c.kstep_b;
  const uint64_t o = (uint64_t)base + (uint64_t)(uint32_t)tile * (uint64_t)step;
  const uint32_t v = (tile < c.nt && base != OOB && o < (uint64_t)OOB) ? (uint32_t)o : OOB;
  char* dst = c.smem + (IS_A ? 0 : REGION) + HALF * HALF_STRIDE + stage * STAGE_STRIDE + c.wave * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? c.a_hi : c.b_hi, LDS_PTR(dst), 16, v, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? c.a_lo : c.b_lo, LDS_PTR(dst + PLANE), 16, v, 0, 0, 0);
}

template <int HALF, int S>
__device__ __forceinline__ void read_a_half(const Ctx& c, bf16x8_t (&hi)[4], bf16x8_t (&lo)[4]) {
  constexpr int O = HALF * HALF_STRIDE + S * STAGE_STRIDE;
  hi[0] = lds_read_tr<O>(c.rd_a[0]);
  hi[1] = lds_read_tr<O>(c.rd_a[1]);
  hi[2] = lds_read_tr<O>(c.rd_a[2]);
  hi[3] = lds_read_tr<O>(c.rd_a[3]);
  lo[0] = lds_read_tr<O + PLANE>(c.rd_a[0]);
  lo[1] = lds_read_tr<O + PLANE>(c.rd_a[1]);
  lo[2] = lds_read_tr<O + PLANE>(c.rd_a[2]);
  lo[3] = lds_read_tr<O + PLANE>(c.rd_a[3]);
}
template <int HALF, int S>
__device__ __forceinline__ void read_b_half(const Ctx& c, bf16x8_t (&hi)[2], bf16x8_t (&lo)[2]) {
  constexpr int O = HALF * HALF_STRIDE + S * STAGE_STRIDE;
  hi[0] = lds_read_tr<O>(c.rd_b[0]);
  hi[1] = lds_read_tr<O>(c.rd_b[1]);
  lo[0] = lds_read_tr<O + PLANE>(c.rd_b[0]);
  lo[1] = lds_read_tr<O + PLANE>(c.rd_b[1]);
}

// Column sums of A (the bias gradient that belongs to this weight gradient) from the fragments a wave holds anyway: lane l of a
// fragment carries A[k = 8 * (l >> 4) + j][m = tile row (l & 15)], j = 0..7, as packed bf16 pairs; hi and lo plane are added in
// fp32.  cs[i] = this lane's share for m-tile i; the four lane groups are combined once, after the main loop.
__device__ __forceinline__ float frag_sum(const bf16x8_t& f) {
  const u32x4_t w = __builtin_bit_cast(u32x4_t, f);
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < 4; ++d) s += __uint_as_float(w[d] << 16) + __uint_as_float(w[d] & 0xffff0000u);
  return s;
}
// The four waves of one wr hold the same A fragments: wave wc sums tile wc of each half (a quarter of the vector work each, in
// parallel on the four SIMDs).  wc is wave-uniform: a branch per case keeps the fragment index a compile-time constant.
template <int AH>
__device__ __forceinline__ void colsum_frags(float (&cs)[2], int wc, const bf16x8_t (&ahi)[4], const bf16x8_t (&alo)[4]) {
  if (wc == 0) cs[AH] += frag_sum(ahi[0]) + frag_sum(alo[0]);
  else if (wc == 1) cs[AH] += frag_sum(ahi[1]) + frag_sum(alo[1]);
  else if (wc == 2) cs[AH] += frag_sum(ahi[2]) + frag_sum(alo[2]);
  else cs[AH] += frag_sum(ahi[3]) + frag_sum(alo[3]);
}

// One K step; the section / refill / counted-wait schedule of g256::k_step<S, 0>.  do_cs (wave-uniform): this wave adds the K
// step's A fragments to its column sums (after the MFMAs of the phase were issued: the vector work runs under the matrix pipe).
template <int S>
__device__ __forceinline__ void k_step(const Ctx& c, int t, f32x4_t (&acc)[8][4], float (&cs)[2], int wc, bool do_cs) {
  using g256::end_load_section;
  using g256::mfma_section;
  bf16x8_t ahi[4], alo[4], bhi[2], blo[2];
  issue_part<false, 0>(c, t + 1, S ^ 1);          // phase 0: quadrant (A0, B0)
  read_a_half<0, S>(c, ahi, alo);
  read_b_half<0, S>(c, bhi, blo);
  end_load_section<12>();
  mfma_section<0, 0, 4>(acc, ahi, alo, bhi, blo);
  issue_part<true, 0>(c, t + 2, S);               // phase 1: (A0, B1)
  read_b_half<1, S>(c, bhi, blo);
  if (do_cs) colsum_frags<0>(cs, wc, ahi, alo);
  end_load_section<12>();
  mfma_section<0, 1, 4>(acc, ahi, alo, bhi, blo);
  issue_part<false, 1>(c, t + 2, S);              // phase 2: (A1, B1)
  read_a_half<1, S>(c, ahi, alo);
  end_load_section<12>();
  mfma_section<1, 1, 4>(acc, ahi, alo, bhi, blo);
  issue_part<true, 1>(c, t + 2, S);               // phase 3: (A1, B0)
  read_b_half<0, S>(c, bhi, blo);
  if (do_cs) colsum_frags<1>(cs, wc, ahi, alo);
  end_load_section<6>();
  mfma_section<1, 0, 4>(acc, ahi, alo, bhi, blo);
}

__global__ __launch_bounds__(512, 2) void gemm256_tn_kernel(GemmParams g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // work units in split-major order, cut into 8 contiguous chunks (one per XCD): the tiles of one K range -- which share its A
  // and B panels -- run on one XCD's L2
  const int tiles = g.tiles_m * g.tiles_n;
  const int unit = xcd_chunk_index(tiles * g.splits, blockIdx.x);
  const int split = unit / tiles, tile = unit - split * tiles;
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int total_steps = (g.K + BK - 1) / BK;
  const int t0 = split * g.k_tiles_per_split;

  Ctx c;
  c.smem = smem;
  c.wave = wave;
  c.nt = min(g.k_tiles_per_split, total_steps - t0);
  c.a_hi = uniform_rsrc(g.A, g.a_bytes);
  c.a_lo = uniform_rsrc((const char*)g.A + g.a_lo_off, g.a_bytes);
  c.b_hi = uniform_rsrc(g.B, g.b_bytes);
  c.b_lo = uniform_rsrc((const char*)g.B + g.b_lo_off, g.b_bytes);
  c.kstep_a = (uint32_t)g.lda * 2u * BK;
  c.kstep_b = (uint32_t)g.ldb * 2u * BK;
  {
    // this wave's 1-KiB piece of a part = k-rows wave*4 .. +4; lane l fills unit slot (l & 15) of k-row (l >> 4), which holds
    // source unit (l & 15) ^ swz_tr16(k-row) = 8 consecutive m (n) of the part
    const int kl = wave * 4 + (lane >> 4);
    const int pu = ((lane & 15) ^ swz_tr16(kl)) * 8;          // part-local index of the unit's first element
    const uint64_t krow = (uint64_t)t0 * BK + (uint64_t)kl;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int am = m0 + (pu >> 6) * 128 + h * 64 + (pu & 63);   // part A(h): rows wr*128 + h*64 + [0, 64) of both wr
      const int bn = n0 + (pu >> 5) * 64 + h * 32 + (pu & 31);    // part B(h): cols wc*64 + h*32 + [0, 32) of all wc
      const uint64_t oa = (krow * (uint64_t)g.lda + (uint64_t)am) * 2u;
      const uint64_t ob = (krow * (uint64_t)g.ldb + (uint64_t)bn) * 2u;
      // units past the row's end would read the next k-row's first columns: they only feed output rows / columns >= M / N, which
      // the epilogue masks -- but a unit that STRADDLES lda cannot exist (lda % 8 == 0)
      c.voff_a[h] = (am < g.lda && oa < (uint64_t)OOB) ? (uint32_t)oa : OOB;
      c.voff_b[h] = (bn < g.ldb && ob < (uint64_t)OOB) ? (uint32_t)ob : OOB;
    }
    // fragment bases (gemm.hip::read_frag, TR form): lane (g4, q, p) reads 8 B at k-row 8*g4 + q, unit (rbase >> 3) + (p >> 1)
    const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g4 = lane >> 4;
    const int ka = 8 * g4 + q;
    const int sx = swz_tr16(ka);                               // == swz_tr16(ka + 4)
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t rowb = (uint32_t)(ka * 256 + 8 * (pp & 1));
#pragma unroll
    for (int i = 0; i < 4; ++i) c.rd_a[i] = sm + rowb + (uint32_t)((((wr * 8 + 2 * i + (pp >> 1)) ^ sx)) * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) c.rd_b[j] = sm + REGION + rowb + (uint32_t)((((wc * 4 + 2 * j + (pp >> 1)) ^ sx)) * 16);
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: everything of K steps 0 and 1 except B0(1), in the steady-state issue order
  issue_part<true, 0>(c, 0, 0);
  issue_part<false, 1>(c, 0, 0);
  issue_part<true, 1>(c, 0, 0);
  issue_part<false, 0>(c, 0, 0);
  issue_part<true, 0>(c, 1, 1);
  issue_part<false, 1>(c, 1, 1);
  issue_part<true, 1>(c, 1, 1);
  g256::end_load_section<6>();
  if (wr == 1) {                                // waves 4-7 run one section behind waves 0-3
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  // Column sums of A (g.epi.colsum_partial): every K step of an A row panel is seen by the tiles_n workgroups of that panel's tile
  // row; global step tau is summed by the one with tn == tau % tiles_n, wave (wr, wc) taking m-tile wc of each half of its wr
  // rows.  Every workgroup writes its [256] slice of partial row (split * tiles_n + tn): nothing to zero beforehand.
  float cs[2] = {0.f, 0.f};
  const bool cs_on = g.epi.colsum_partial != nullptr;
  int cs_phase = (t0 + g.tiles_n - tn) % g.tiles_n;      // (global step - tn) mod tiles_n of local step 0
  for (int t = 0; t < c.nt; t += 2) {
    k_step<0>(c, t, acc, cs, wc, cs_on && cs_phase == 0);
    cs_phase = cs_phase + 1 == g.tiles_n ? 0 : cs_phase + 1;
    if (t + 1 < c.nt) k_step<1>(c, t + 1, acc, cs, wc, cs_on && cs_phase == 0);
    cs_phase = cs_phase + 1 == g.tiles_n ? 0 : cs_phase + 1;
  }
  if (wr == 0) {                                // same number of barriers for every wave
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail refills have landed: LDS is reusable
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  if (cs_on) {
    float* dst = g.epi.colsum_partial + (size_t)(split * g.tiles_n + tn) * (size_t)g.M;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v = cs[h];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m0 + wr * 128 + h * 64 + wc * 16 + (lane & 15);
      if (lane < 16 && m < g.M) dst[m] = v;
    }
  }
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * (64 + 4));
  float* partial = g.partial ? g.partial + (size_t)split * (size_t)g.M * (size_t)g.N : nullptr;
  epilogue_wave<128, 64, 8, 4, 1>(g, acc, slab, m0 + wr * 128, n0 + wc * 64, lane, partial);
}

}  // namespace g256t

// Host entry: planes x planes, TN, passes == 3, plain epilogue (alpha / accumulate-free store or split-K slabs); the caller runs the
// split-K reducer.  p.k_tiles_per_split is in units of 32 rows.
int launch_gemm256_tn(const GemmParams& p_in, int splits, hipStream_t stream) {
  using namespace g256t;
  GemmParams p = p_in;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.splits = splits;
  if (splits <= 1) p.partial = nullptr;
  static bool attr_set = false;
  if (!attr_set) {
    if (lr2_allow_dynamic_lds(gemm256_tn_kernel, LDS_BYTES, "gemm256_tn")) return LR2_ERR_LAUNCH;
    attr_set = true;
  }
  LR2_LAUNCH(gemm256_tn_kernel, dim3(p.tiles_m * p.tiles_n * splits), dim3(512), LDS_BYTES, stream, p);
  return lr2_launch_status(__func__);
}

// Host entry for gemm.hip's dispatcher.  Requirements (checked by the caller): planes x planes, NT, passes == 3,
// K % 32 == 0, no split-K, operand extents < 4 GiB - 512 B.
// rows per tile the NT launcher picks for a shape: 192 when one round of 256-row tiles would leave CUs idle that a round of 192-row
// tiles fills (LR2_GEMM_192=0: always 256; read once per process)
int gemm256_nt_tile_rows(int M, int N) {
  static const bool on = !(getenv("LR2_GEMM_192") && atoi(getenv("LR2_GEMM_192")) == 0);
  const int tn = (N + g256::BN - 1) / g256::BN;
  const int t256 = ((M + 255) / 256) * tn, t192 = ((M + 191) / 192) * tn;
  return (on && t256 < 256 && t192 <= 256 && t192 > t256) ? 192 : 256;
}

int launch_gemm256_nt(const GemmParams& p_in, hipStream_t stream) {
  using namespace g256;
  GemmParams p = p_in;
  const int bm = gemm256_nt_tile_rows(p.M, p.N);
  p.tiles_m = (p.M + bm - 1) / bm;
  p.tiles_n = (p.N + BN - 1) / BN;
  if (!(p.ablate & 32)) p.partial = nullptr;
  static bool attr_set = false;
  if (!attr_set) {
    if (lr2_allow_dynamic_lds(gemm256_nt_kernel<0, 4>, LDS_BYTES, "gemm256")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_nt_kernel<1, 4>, LDS_BYTES, "gemm256")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_nt_kernel<0, 3>, LDS_BYTES, "gemm256(192 rows)")) return LR2_ERR_LAUNCH;
    attr_set = true;
  }
  const char* se = getenv("LR2_GEMM_STRIP");          // read per call: strip width of the tile order (A/B inside one process)
  p.strip_n = se ? atoi(se) : 0;
  const char* ve = getenv("LR2_GEMM256_VARIANT");     // read per call: tools A/B the variants inside one process
  const int variant = ve ? atoi(ve) : 0;
  if (bm == 192) LR2_LAUNCH((gemm256_nt_kernel<0, 3>), dim3(p.tiles_m * p.tiles_n), dim3(512), LDS_BYTES, stream, p);
  else if (variant == 1) LR2_LAUNCH((gemm256_nt_kernel<1, 4>), dim3(p.tiles_m * p.tiles_n), dim3(512), LDS_BYTES, stream, p);
  else LR2_LAUNCH((gemm256_nt_kernel<0, 4>), dim3(p.tiles_m * p.tiles_n), dim3(512), LDS_BYTES, stream, p);
  return lr2_launch_status(__func__);
}

}  // namespace lr2gemm
