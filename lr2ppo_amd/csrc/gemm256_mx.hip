// 256 x 256 "ping-pong" product of MX-FP8 operands:  C[M,N] = A_q[M,K] . B_q[N,K]^T  (e4m3fn bytes, one E8M0 scale per 32 elements of a
// row, fp32 accumulate) on gfx950's block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 -- the "fp8 MFMA" mode of
// BASELINE.json configs[4] (ViT-L/14 swap; FeatureExtractor(precision="mxfp8")).  NOT the parity path (3 mantissa bits per element).
//
// Replaces, in that mode, the cuBLAS sgemm calls reached through nn.Linear in the encoders (tencentpretrain/layers/
// multi_headed_attn.py:55-76, position_ffn.py:12-15); same contract as lr2_gemm_mxfp8's 128 x 128 kernel (fp8.hip).
//
// Structure: the skeleton of gemm256.hip (one workgroup of 8 waves per CU, tile 256 x 256, wave (wr, wc) of a 2 x 4 grid owns
// 128 x 64 = 8 x 4 accumulator tiles, an LDS ring of 2 stages x 4 parts filled by LDS-DMA with COUNTED s_waitcnt vmcnt, bare s_barrier,
// two wave groups running one section apart) with one-byte operands:
//   * a K step is 128 elements = 128 bytes per row: a part (128 rows of A or B) is 16 KiB, laid out as TWO 8-KiB planes --
//     plane 0 = K bytes [0, 64) of every row, plane 1 = K bytes [64, 128) -- each [128 rows][4 units of 16 B] with gemm256.hip's
//     XOR swizzle on the source address.  The instruction's operand for lane (row l & 15, q = l >> 4) is bytes 16 q .. 16 q + 15 and
//     64 + 16 q .. (measured layout, fp8.hip): unit q of plane 0 and unit q of plane 1 -- the two conflict-free ds_read_b128 the
//     bf16 kernel issues for its hi / lo planes.  A K step is 32 instructions per wave (1024 matrix cycles) for 64 KiB of operands.
//   * the scales travel with the operands: per K step one 4-byte LDS-DMA per lane (waves 0-3: the tile's 256 A rows, waves 4-7: its
//     256 B rows) into a 2-stage [512 rows][4 B] region; a lane's scale (row l & 15, block q) is ONE ds_read_u8.
//   * schedule per K step t (stage t & 1), refills as early as the last reader allows:
//       phase 0 (A0 x B0): issue B0(t+1)            read A0, B0 fragments + their scales     wait vmcnt(14)
//       phase 1 (A0 x B1): issue A0(t+2)            read B1 + scales                         wait vmcnt(14)
//       phase 2 (A1 x B1): issue B1(t+2)            read A1 + scales                         wait vmcnt(13)
//       phase 3 (A1 x B0): issue A1(t+2), S(t+2)    read B0                                  wait vmcnt(7)
//     (2 DMA instructions per part, 1 per scale step; each count = the DMAs issued after the youngest part the NEXT section reads).
//   * epilogue through a wave-private LDS slab, 32 rows at a time: bias, GELU, residual; results as fp32, as bf16 hi / lo planes
//     (what the attention kernels read) and / or re-quantised to MX-FP8 (the next product's A operand).
#include <stdlib.h>

#include "fp8_common.h"
#include "gemm_common.h"

namespace lr2mx256 {

using lr2gemm::lds_addr;
using lr2gemm::tile_coords;
using lr2gemm::uniform_rsrc;

typedef int v8i_t __attribute__((ext_vector_type(8)));

constexpr int BM = 256, BN = 256, BKB = 128;   // K step in elements = bytes
constexpr int PLANE = 128 * 64;                // 128 rows x 64 bytes = 8 KiB
constexpr int PART = 2 * PLANE;
constexpr int STAGE = 4 * PART;                // parts A0, B0, B1, A1
constexpr int RING = 2 * STAGE;                // 128 KiB
constexpr int SC_STAGE = 2048;                 // A scales [256 rows][4 B], then B scales [256][4 B]
constexpr int LDS_BYTES = RING + 2 * SC_STAGE;
constexpr int SC16_STAGE = 8192;               // SC16: A scales [256 rows][16 B] (4 K steps each), then B scales [256][16 B]
constexpr int LDS_BYTES16 = RING + 2 * SC16_STAGE;
constexpr int SLOT_A0 = 0, SLOT_B0 = 1, SLOT_B1 = 2, SLOT_A1 = 3;
constexpr uint32_t OOB = 0xFFFFFF00u;          // beyond any descriptor this library builds

__device__ __forceinline__ int swz(int r) { return (4 - ((r >> 2) & 3)) & 3; }

template <int IMM>
__device__ __forceinline__ u32x4_t lds_read16(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
  return v;
}
template <int IMM>
__device__ __forceinline__ int lds_read_u8(uint32_t addr) {
  int v;
  asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(IMM));
  return v;
}

struct Ctx {
  __amdgpu_buffer_rsrc_t a, b, sc;   // sc: this wave's scale source (A scales for waves 0-3, B scales for waves 4-7)
  uint32_t voff_a[2], voff_b[2];     // per-lane source byte offsets of this wave's piece of part A(h) / B(h) at K step 0
  uint32_t voff_s;                   // per-lane source byte offset of this lane's scale row at K step 0
  uint32_t rd_a[2], rd_b[2];         // per-lane LDS read bases (fragments) for stage 0 / 1
  uint32_t rd_sa, rd_sb;             // per-lane LDS read bases (scales), stage 0 (SC16: + the step's offset, see k_step)
  int ks;                            // scale bytes per row (K / 32)
  char* smem;
  int wave, nt;
};

// Two DMA pieces (K bytes [0, 64) and [64, 128) of 16 rows) of one part for K step `tile` into stage `stage`.
template <int SLOT, bool IS_A, int HALF>
__device__ __forceinline__ void issue_part(const Ctx& c, int tile, int stage) {
  const uint32_t base = IS_A ? c.voff_a[HALF] : c.voff_b[HALF];
  const bool in = tile < c.nt && base != OOB;
  const uint32_t v0 = in ? base + (uint32_t)tile * BKB : OOB;
  const uint32_t v1 = in ? v0 + 64u : OOB;
  char* dst = c.smem + stage * STAGE + SLOT * PART + c.wave * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? c.a : c.b, LDS_PTR(dst), 16, v0, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? c.a : c.b, LDS_PTR(dst + PLANE), 16, v1, 0, 0, 0);
}
// The 4 scale bytes of K step `tile` for this wave's 64 rows (one dword per lane).
__device__ __forceinline__ void issue_scales(const Ctx& c, int tile, int stage) {
  const uint32_t v = (tile < c.nt && c.voff_s != OOB) ? c.voff_s + (uint32_t)tile * 4u : OOB;
  char* dst = c.smem + RING + stage * SC_STAGE + c.wave * 256;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.sc, LDS_PTR(dst), 4, v, 0, 0, 0);
}

template <int SLOT>
__device__ __forceinline__ void read_a_half(uint32_t base, v8i_t (&a)[4]) {
  const u32x4_t l0 = lds_read16<SLOT * PART + 0 * 1024>(base), l1 = lds_read16<SLOT * PART + 1 * 1024>(base);
  const u32x4_t l2 = lds_read16<SLOT * PART + 2 * 1024>(base), l3 = lds_read16<SLOT * PART + 3 * 1024>(base);
  const u32x4_t h0 = lds_read16<SLOT * PART + PLANE + 0 * 1024>(base), h1 = lds_read16<SLOT * PART + PLANE + 1 * 1024>(base);
  const u32x4_t h2 = lds_read16<SLOT * PART + PLANE + 2 * 1024>(base), h3 = lds_read16<SLOT * PART + PLANE + 3 * 1024>(base);
  a[0] = v8i_t{(int)l0[0], (int)l0[1], (int)l0[2], (int)l0[3], (int)h0[0], (int)h0[1], (int)h0[2], (int)h0[3]};
  a[1] = v8i_t{(int)l1[0], (int)l1[1], (int)l1[2], (int)l1[3], (int)h1[0], (int)h1[1], (int)h1[2], (int)h1[3]};
  a[2] = v8i_t{(int)l2[0], (int)l2[1], (int)l2[2], (int)l2[3], (int)h2[0], (int)h2[1], (int)h2[2], (int)h2[3]};
  a[3] = v8i_t{(int)l3[0], (int)l3[1], (int)l3[2], (int)l3[3], (int)h3[0], (int)h3[1], (int)h3[2], (int)h3[3]};
}
template <int SLOT>
__device__ __forceinline__ void read_b_half(uint32_t base, v8i_t (&b)[2]) {
  const u32x4_t l0 = lds_read16<SLOT * PART + 0 * 1024>(base), l1 = lds_read16<SLOT * PART + 1 * 1024>(base);
  const u32x4_t h0 = lds_read16<SLOT * PART + PLANE + 0 * 1024>(base), h1 = lds_read16<SLOT * PART + PLANE + 1 * 1024>(base);
  b[0] = v8i_t{(int)l0[0], (int)l0[1], (int)l0[2], (int)l0[3], (int)h0[0], (int)h0[1], (int)h0[2], (int)h0[3]};
  b[1] = v8i_t{(int)l1[0], (int)l1[1], (int)l1[2], (int)l1[3], (int)h1[0], (int)h1[1], (int)h1[2], (int)h1[3]};
}
// scales of accumulator half AH / BH for stage S: A rows wr*128 + AH*64 + 16 i + r16, B rows wc*64 + BH*32 + 16 j + r16
// SC16 form: the scale bytes of FOUR K steps per row and DMA (16 B per lane): a 4-byte gather per lane and step touched one cache
// line per row and step -- as many L2 -> L1 bytes as the operands themselves, 0.56 of 2.21 us per K step (tools/dbg/mx_ablate.py)
__device__ __forceinline__ void issue_scales16(const Ctx& c, int quad, int stage) {
  const uint32_t off = (uint32_t)quad * 16u;
  const uint32_t v = (off < (uint32_t)c.ks && c.voff_s != OOB) ? c.voff_s + off : OOB;
  char* dst = c.smem + RING + stage * SC16_STAGE + c.wave * 1024;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.sc, LDS_PTR(dst), 16, v, 0, 0, 0);
}
template <int AH>
__device__ __forceinline__ void read_sa16(uint32_t base, int (&sa)[4]) {
  sa[0] = lds_read_u8<(AH * 64 + 0) * 16>(base);
  sa[1] = lds_read_u8<(AH * 64 + 16) * 16>(base);
  sa[2] = lds_read_u8<(AH * 64 + 32) * 16>(base);
  sa[3] = lds_read_u8<(AH * 64 + 48) * 16>(base);
}
template <int BH>
__device__ __forceinline__ void read_sb16(uint32_t base, int (&sb)[2]) {
  sb[0] = lds_read_u8<(BH * 32 + 0) * 16>(base);
  sb[1] = lds_read_u8<(BH * 32 + 16) * 16>(base);
}
template <int S, int AH>
__device__ __forceinline__ void read_sa(uint32_t base, int (&sa)[4]) {
  sa[0] = lds_read_u8<S * SC_STAGE + (AH * 64 + 0) * 4>(base);
  sa[1] = lds_read_u8<S * SC_STAGE + (AH * 64 + 16) * 4>(base);
  sa[2] = lds_read_u8<S * SC_STAGE + (AH * 64 + 32) * 4>(base);
  sa[3] = lds_read_u8<S * SC_STAGE + (AH * 64 + 48) * 4>(base);
}
template <int S, int BH>
__device__ __forceinline__ void read_sb(uint32_t base, int (&sb)[2]) {
  sb[0] = lds_read_u8<S * SC_STAGE + (BH * 32 + 0) * 4>(base);
  sb[1] = lds_read_u8<S * SC_STAGE + (BH * 32 + 16) * 4>(base);
}

template <int VM>
__device__ __forceinline__ void end_load_section() {
  constexpr int imm = (VM & 15) | ((VM >> 4) << 14) | (7 << 4) | (0 << 8);     // vmcnt(VM), lgkmcnt(0)
  __builtin_amdgcn_s_waitcnt(imm);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// 8 instructions of one accumulator quadrant (K = 128 each)
template <int AH, int BH, int ABL = 0>
__device__ __forceinline__ void mfma_section(f32x4_t (&acc)[8][4], const v8i_t (&a)[4], const v8i_t (&b)[2], const int (&sa)[4],
                                             const int (&sb)[2]) {
  __builtin_amdgcn_s_setprio(1);
  if constexpr (ABL & 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[i]), "v"(sa[i]));
    asm volatile("" ::"v"(b[0]), "v"(b[1]), "v"(sb[0]), "v"(sb[1]));
  } else {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[AH * 4 + i][BH * 2 + j] =
          __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[AH * 4 + i][BH * 2 + j], 0, 0, 0, sa[i], 0, sb[j]);
  }
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// ABL (diagnostics, LR2_MX_ABLATE, never in a timed product run): 1 = no scale traffic (unit scales), 2 = no operand DMA,
// 4 = no matrix instructions, 8 = no epilogue
template <int S, int ABL, bool SC16>
__device__ __forceinline__ void k_step(const Ctx& c, int t, f32x4_t (&acc)[8][4]) {
  v8i_t a[4], b[2];
  int sa[4] = {127, 127, 127, 127}, sb0[2] = {127, 127}, sb1[2] = {127, 127};
  constexpr bool SC = !(ABL & 1);
  // SC16: the step's scales sit in stage (t >> 2) & 1 at byte 4 (t & 3) of the lane's row
  const uint32_t soff = SC16 ? (uint32_t)(((t >> 2) & 1) * SC16_STAGE + (t & 3) * 4) : 0u;
  const uint32_t rsa = c.rd_sa + soff, rsb = c.rd_sb + soff;
  // the waits: SC16 issues a scale DMA in one step of four only, so the counts are those of the operand parts alone (an extra DMA in
  // flight makes a counted wait stricter, never laxer); the 4-byte form issues one per step and counts it
  constexpr int W0 = (ABL & 2) ? 0 : (SC && !SC16) ? 14 : 12, W2 = (ABL & 2) ? 0 : (SC && !SC16) ? 13 : 12, W3 = (ABL & 2) ? 0 : (SC && !SC16) ? 7 : 6;
  // phase 0: quadrant (A0, B0)
  if (!(ABL & 2)) issue_part<SLOT_B0, false, 0>(c, t + 1, S ^ 1);
  read_a_half<SLOT_A0>(c.rd_a[S], a);
  read_b_half<SLOT_B0>(c.rd_b[S], b);
  if constexpr (SC && SC16) {
    read_sa16<0>(rsa, sa);
    read_sb16<0>(rsb, sb0);
  } else if constexpr (SC) {
    read_sa<S, 0>(c.rd_sa, sa);
    read_sb<S, 0>(c.rd_sb, sb0);
  }
  end_load_section<W0>();
  mfma_section<0, 0, ABL>(acc, a, b, sa, sb0);
  // phase 1: (A0, B1)
  if (!(ABL & 2)) issue_part<SLOT_A0, true, 0>(c, t + 2, S);
  read_b_half<SLOT_B1>(c.rd_b[S], b);
  if constexpr (SC && SC16) read_sb16<1>(rsb, sb1);
  else if constexpr (SC) read_sb<S, 1>(c.rd_sb, sb1);
  end_load_section<W0>();
  mfma_section<0, 1, ABL>(acc, a, b, sa, sb1);
  // phase 2: (A1, B1)
  if (!(ABL & 2)) issue_part<SLOT_B1, false, 1>(c, t + 2, S);
  read_a_half<SLOT_A1>(c.rd_a[S], a);
  if constexpr (SC && SC16) read_sa16<1>(rsa, sa);
  else if constexpr (SC) read_sa<S, 1>(c.rd_sa, sa);
  end_load_section<W2>();
  mfma_section<1, 1, ABL>(acc, a, b, sa, sb1);
  // phase 3: (A1, B0); every scale read of step t was retired one barrier ago by both groups.  4-byte form: the step's slot takes
  // step t + 2.  SC16: at the first step of a quad the OTHER stage (last read in step t - 1) takes the next quad.
  if (!(ABL & 2)) issue_part<SLOT_A1, true, 1>(c, t + 2, S);
  if constexpr (SC && SC16) {
    if ((t & 3) == 0) issue_scales16(c, (t >> 2) + 1, ((t >> 2) + 1) & 1);
  } else if constexpr (SC) {
    issue_scales(c, t + 2, S);
  }
  read_b_half<SLOT_B0>(c.rd_b[S], b);
  end_load_section<W3>();
  mfma_section<1, 0, ABL>(acc, a, b, sa, sb0);
}

// ---- epilogue: one 32-row slab of the wave tile (accumulator tile rows 2 HALF, 2 HALF + 1) --------------------------------------
__device__ __forceinline__ float4 mx_finish(const Mx8Params& p, float4 v, float4 b4) {
  v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
  if (p.act == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
  return v;
}
// the row's 32-column MX block = 8 consecutive lanes x 4 columns: quantise as lr2_quant_mxfp8 would the stored row
__device__ __forceinline__ void mx_quant_store(const Mx8Params& p, float4 v, int m, int n, int lane, bool ok) {
  float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  amax = group8_max(amax);
  int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 127 - 8;
  if (amax < 1.17549435e-38f) e = -127;
  if (e < -127) e = -127;
  if (e > 127) e = 127;
  const uint32_t ef = (uint32_t)(127 - e);
  const float inv = __uint_as_float(ef ? ef << 23 : 0x00400000u);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.x * inv, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.y * inv, -448.f, 448.f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.z * inv, -448.f, 448.f), __builtin_amdgcn_fmed3f(v.w * inv, -448.f, 448.f), w, true);
  if (ok) {
    *reinterpret_cast<int*>(p.out_q + (size_t)m * p.N + n) = w;
    if ((lane & 7) == 0) p.out_s[(size_t)m * (p.N / 32) + (n >> 5)] = (uint8_t)(e + 127);
  }
}

// Straight-line forms of a slab that lies inside the matrix, chosen ONCE per slab (wave-uniform) -- the encoder's four products:
//   FORM 0: (+ bias) -> ONE bf16 plane (QKV)          FORM 1: -> bf16 hi / lo planes          FORM 2: + residual -> fp32 (output
//   projection, FFN-2)   FORM 3 / 4: (GELU for 4) -> MX-FP8 (FFN-1).
// The general body below decides everything per pass: a dozen wave-uniform branches, exec-masked stores, a residual select that made
// the compiler wait (vmcnt: loads AND stores, in order) for stores two passes back, 64-bit address arithmetic per store -- 16 us per
// K = 1024 tile against 17 us of main loop.  Same operations in the same order per element as the general body: same bytes.
template <int HALF, int FORM>
__device__ __forceinline__ void epilogue_slab_fast(const Mx8Params& p, f32x4_t (&acc)[8][4], float* slab, int mw, int nw, int lane, float4 b4) {
  constexpr int NP = 8, RPP = 4, LDW = 68;
  const int row0 = lane >> 4, col = (lane & 15) * 4, n = nw + col;
  const int mbase = mw + 32 * HALF + row0;
  float4 rr[NP];
  if constexpr (FORM == 2) {
    const float* r0 = p.resid + (size_t)mbase * p.ld_resid + n;
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) rr[pass] = lr2gemm::ld4(r0 + (size_t)(pass * RPP) * p.ld_resid);
  }
  lr2gemm::epilogue_to_slab<64, 8, 4, HALF>(acc, slab, lane);
  float4 v[NP];
  lr2gemm::slab_read_all<64, NP, 0>(lds_addr(slab) + (uint32_t)((row0 * LDW + col) * 4), v);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int m = mbase + pass * RPP;
    float4 x = v[pass];
    x.x += b4.x; x.y += b4.y; x.z += b4.z; x.w += b4.w;
    if constexpr (FORM == 4) { x.x = gelu_erf(x.x); x.y = gelu_erf(x.y); x.z = gelu_erf(x.z); x.w = gelu_erf(x.w); }
    if constexpr (FORM == 2) {
      x.x += rr[pass].x; x.y += rr[pass].y; x.z += rr[pass].z; x.w += rr[pass].w;
      lr2gemm::st4(p.out + (size_t)m * p.ld_out + n, x);
    } else if constexpr (FORM == 0) {
      store_bf16x4(p.out_hi + (size_t)m * p.ld_planes + n, x);
    } else if constexpr (FORM == 1) {
      store_planes4(p.out_hi + (size_t)m * p.ld_planes + n, p.out_lo_off, x);
    } else {
      mx_quant_store(p, x, m, n, lane, true);
    }
  }
}

template <int HALF>
__device__ __forceinline__ void epilogue_slab(const Mx8Params& p, f32x4_t (&acc)[8][4], float* slab, int mw, int nw, int lane, float4 b4) {
  constexpr int NP = 8, RPP = 4, LDW = 68;
  const int row0 = lane >> 4, col = (lane & 15) * 4, n = nw + col;
  const int mbase = mw + 32 * HALF + row0;
  const bool inside = mw + 32 * HALF + 32 <= p.M;            // wave-uniform
  if (inside && p.act <= 1) {
    const bool f32 = p.out != nullptr, pl = p.out_hi != nullptr, mx = p.out_q != nullptr, rs = p.resid != nullptr;
    if (pl && !f32 && !mx && !rs && p.act == 0) {
      if (p.out_lo_off) epilogue_slab_fast<HALF, 1>(p, acc, slab, mw, nw, lane, b4);
      else epilogue_slab_fast<HALF, 0>(p, acc, slab, mw, nw, lane, b4);
      return;
    }
    if (f32 && rs && !pl && !mx && p.act == 0) { epilogue_slab_fast<HALF, 2>(p, acc, slab, mw, nw, lane, b4); return; }
    if (mx && !f32 && !pl && !rs) {
      if (p.act == 1) epilogue_slab_fast<HALF, 4>(p, acc, slab, mw, nw, lane, b4);
      else epilogue_slab_fast<HALF, 3>(p, acc, slab, mw, nw, lane, b4);
      return;
    }
  }
  float4 rr[NP];
  if (p.resid) {                                             // every residual request of the slab before its first store
    if (inside) {
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) rr[pass] = lr2gemm::ld4(p.resid + (size_t)(mbase + pass * RPP) * p.ld_resid + n);
    } else {
#pragma unroll
      for (int pass = 0; pass < NP; ++pass) {
        const int m = mbase + pass * RPP;
        rr[pass] = lr2gemm::ld4(p.resid + (size_t)(m < p.M ? m : p.M - 1) * p.ld_resid + n);
      }
    }
  }
  lr2gemm::epilogue_to_slab<64, 8, 4, HALF>(acc, slab, lane);
  float4 v[NP];
  lr2gemm::slab_read_all<64, NP, 0>(lds_addr(slab) + (uint32_t)((row0 * LDW + col) * 4), v);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int m = mbase + pass * RPP;
    const bool ok = inside || m < p.M;
    float4 x = mx_finish(p, v[pass], b4);
    if (p.resid) { x.x += rr[pass].x; x.y += rr[pass].y; x.z += rr[pass].z; x.w += rr[pass].w; }
    if (ok) {
      if (p.out) lr2gemm::st4(p.out + (size_t)m * p.ld_out + n, x);
      if (p.out_hi) { if (p.out_lo_off) store_planes4(p.out_hi + (size_t)m * p.ld_planes + n, p.out_lo_off, x); else store_bf16x4(p.out_hi + (size_t)m * p.ld_planes + n, x); }
    }
    if (p.out_q) mx_quant_store(p, x, m, n, lane, ok);
  }
}

template <int ABL, bool SC16>
__global__ __launch_bounds__(512, 2) void gemm256_mx_kernel(Mx8Params p, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  int tm, tn;
  tile_coords(tiles_m, tiles_n, blockIdx.x, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int ks = p.K / 32;                    // scale bytes per row

  Ctx c;
  c.smem = smem;
  c.wave = wave;
  c.nt = p.K / BKB;
  c.a = uniform_rsrc(p.aq, (uint32_t)((size_t)p.M * p.K));
  c.b = uniform_rsrc(p.bq, (uint32_t)((size_t)p.N * p.K));
  {
    const __amdgpu_buffer_rsrc_t sa = uniform_rsrc(p.as, (uint32_t)((size_t)p.M * ks));
    const __amdgpu_buffer_rsrc_t sb = uniform_rsrc(p.bs, (uint32_t)((size_t)p.N * ks));
    c.sc = wave < 4 ? sa : sb;
    const int srow = (wave < 4 ? m0 : n0) + (wave & 3) * 64 + lane;
    const uint64_t os = (uint64_t)srow * (uint64_t)ks;
    c.voff_s = (srow < (wave < 4 ? p.M : p.N) && os < (uint64_t)OOB) ? (uint32_t)os : OOB;
    // this wave's 1-KiB piece of a plane = local rows wave*16 .. +16; lane l fills unit slot (l & 3) of row (l >> 2), which holds the
    // row's 16-byte chunk (l & 3) ^ swz(row) of that plane's 64 bytes
    const int lr = wave * 16 + (lane >> 2);
    const uint32_t ku = (uint32_t)((lane & 3) ^ swz(lr)) * 16u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int arow = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);   // part A(h): rows wr*128 + h*64 + [0, 64) of both wr
      const int bcol = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);    // part B(h): cols wc*64 + h*32 + [0, 32) of all wc
      const uint64_t oa = (uint64_t)arow * (uint64_t)p.K + ku;
      const uint64_t ob = (uint64_t)bcol * (uint64_t)p.K + ku;
      c.voff_a[h] = (arow < p.M && oa < (uint64_t)OOB) ? (uint32_t)oa : OOB;
      c.voff_b[h] = (bcol < p.N && ob < (uint64_t)OOB) ? (uint32_t)ob : OOB;
    }
    const int r16 = lane & 15, q = lane >> 4;
    const uint32_t lane_off = (uint32_t)(r16 * 64 + ((q ^ swz(r16)) * 16));
    const uint32_t sm = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      c.rd_a[s] = sm + s * STAGE + wr * 4096 + lane_off;     // A part: local row wr*64 + i*16 + r16
      c.rd_b[s] = sm + s * STAGE + wc * 2048 + lane_off;     // B part: local row wc*32 + j*16 + r16
    }
    c.rd_sa = sm + RING + (SC16 ? (uint32_t)((wr * 128 + r16) * 16 + q) : (uint32_t)((wr * 128 + r16) * 4 + q));
    c.rd_sb = sm + RING + (SC16 ? 4096u + (uint32_t)((wc * 64 + r16) * 16 + q) : 1024u + (uint32_t)((wc * 64 + r16) * 4 + q));
    c.ks = ks;
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: K steps 0 and 1 except B0(1), in the steady-state issue order (... A1, S, B0, A0, B1, A1, S ...); SC16: the first
  // quad of scales first (the oldest request: every later counted wait covers it)
  if constexpr (SC16) issue_scales16(c, 0, 0);
  issue_part<SLOT_A0, true, 0>(c, 0, 0);
  issue_part<SLOT_B1, false, 1>(c, 0, 0);
  issue_part<SLOT_A1, true, 1>(c, 0, 0);
  if constexpr (!SC16) issue_scales(c, 0, 0);
  issue_part<SLOT_B0, false, 0>(c, 0, 0);
  issue_part<SLOT_A0, true, 0>(c, 1, 1);
  issue_part<SLOT_B1, false, 1>(c, 1, 1);
  issue_part<SLOT_A1, true, 1>(c, 1, 1);
  if constexpr (!SC16) issue_scales(c, 1, 1);
  end_load_section<SC16 ? 6 : 7>();             // everything of K step 0 has landed, everyone's
  if (wr == 1) {                                // waves 4-7 run one section behind waves 0-3
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int t = 0; t < c.nt; t += 2) {
    k_step<0, ABL, SC16>(c, t, acc);
    if (t + 1 < c.nt) k_step<1, ABL, SC16>(c, t + 1, acc);
  }
  if (wr == 0) {                                // same number of barriers for every wave
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail refills have landed: LDS is reusable
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  const int mw = m0 + wr * 128, nw = n0 + wc * 64;
  if constexpr (ABL == 8) {                     // diagnostics: no epilogue at all (the accumulators are kept alive)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if (nw + 64 > p.N || mw >= p.M) return;       // N % 128 == 0 <=> a wave's 64 columns are all inside or all outside
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * (64 + 4));
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) b4 = lr2gemm::ld4(p.bias + nw + (lane & 15) * 4);
  asm volatile("" ::"v"(b4.x), "v"(b4.y), "v"(b4.z), "v"(b4.w));      // ONE wait for the bias, here (gemm_common.h::epilogue_wave)
  epilogue_slab<0>(p, acc, slab, mw, nw, lane, b4);
  if (mw + 32 < p.M) epilogue_slab<1>(p, acc, slab, mw, nw, lane, b4);
  if (mw + 64 < p.M) epilogue_slab<2>(p, acc, slab, mw, nw, lane, b4);
  if (mw + 96 < p.M) epilogue_slab<3>(p, acc, slab, mw, nw, lane, b4);
}

}  // namespace lr2mx256

int launch_gemm256_mx(const Mx8Params& p, hipStream_t stream) {
  using namespace lr2mx256;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  static bool attr_set = false;
  static int abl = 0, sc16_env = 1;
  if (!attr_set) {
    if (lr2_allow_dynamic_lds(gemm256_mx_kernel<0, false>, LDS_BYTES, "gemm256_mx")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_mx_kernel<0, true>, LDS_BYTES16, "gemm256_mx")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_mx_kernel<1, false>, LDS_BYTES, "gemm256_mx")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_mx_kernel<2, false>, LDS_BYTES, "gemm256_mx")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_mx_kernel<4, false>, LDS_BYTES, "gemm256_mx")) return LR2_ERR_LAUNCH;
    if (lr2_allow_dynamic_lds(gemm256_mx_kernel<8, false>, LDS_BYTES, "gemm256_mx")) return LR2_ERR_LAUNCH;
    const char* e = getenv("LR2_MX_ABLATE");      // diagnostics only (wrong results): see k_step
    abl = e ? atoi(e) : 0;
    const char* e16 = getenv("LR2_MX_SC16");      // 0: one 4-byte scale gather per K step everywhere (A/B)
    sc16_env = e16 ? atoi(e16) : 1;
    attr_set = true;
  }
  const dim3 grid(tiles_m * tiles_n);
  // whole 16-byte scale chunks per row (K / 32 a multiple of 16): the SC16 form; else the 4-byte form
  // (from K = 2048: at K = 1024 -- 8 steps -- the row gather in the prologue costs more than the seven later gathers it saves)
  const bool sc16 = sc16_env && (p.K % 512) == 0 && p.K >= 2048;
  if (abl == 1) LR2_LAUNCH((gemm256_mx_kernel<1, false>), grid, dim3(512), LDS_BYTES, stream, p, tiles_m, tiles_n);
  else if (abl == 2) LR2_LAUNCH((gemm256_mx_kernel<2, false>), grid, dim3(512), LDS_BYTES, stream, p, tiles_m, tiles_n);
  else if (abl == 4) LR2_LAUNCH((gemm256_mx_kernel<4, false>), grid, dim3(512), LDS_BYTES, stream, p, tiles_m, tiles_n);
  else if (abl == 8) LR2_LAUNCH((gemm256_mx_kernel<8, false>), grid, dim3(512), LDS_BYTES, stream, p, tiles_m, tiles_n);
  else if (sc16) LR2_LAUNCH((gemm256_mx_kernel<0, true>), grid, dim3(512), LDS_BYTES16, stream, p, tiles_m, tiles_n);
  else LR2_LAUNCH((gemm256_mx_kernel<0, false>), grid, dim3(512), LDS_BYTES, stream, p, tiles_m, tiles_n);
  return lr2_launch_status(__func__);
}
