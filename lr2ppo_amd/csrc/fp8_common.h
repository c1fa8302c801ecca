// Launch parameters shared by the MX-FP8 product kernels (fp8.hip: 128 x 128 tiles; gemm256_mx.hip: the 256 x 256 LDS-DMA ring).
#pragma once
#include "common.h"

struct Mx8Params {
  const uint8_t* aq;   // A elements [M, K] (e4m3fn bytes)
  const uint8_t* as;   // A scales [M, K / 32] (E8M0 bytes)
  const uint8_t* bq;   // B elements [N, K]
  const uint8_t* bs;   // B scales [N, K / 32]
  float* out;
  const float* bias;
  const float* resid;
  int M, N, K, ld_out, ld_resid, act;
  uint8_t* out_q;      // result ALSO / INSTEAD as MX-FP8 [M, N] + scales [M, N / 32] (the next product's A operand): LDS kernels only
  uint8_t* out_s;
  bf16_t* out_hi;      // result ALSO / INSTEAD as bf16 hi / lo planes [M, ld_planes] (what the attention kernels read): LDS kernels only
  size_t out_lo_off;
  int ld_planes;
};

// gemm256_mx.hip: 256 x 256 tiles, one 8-wave workgroup per CU; requires K % 128 == 0, N % 128 == 0, operands < 4 GiB - 512 B
int launch_gemm256_mx(const Mx8Params& p, hipStream_t stream);
