// Small HBM-bound kernels of the LR2PPO hot path on gfx950: row gather / concat copies, the 768->1 head,
// position-embedding add, the fused PPO loss (forward + analytic backward), SmoothL1, multi-tensor AdamW,
// and the TencentPretrain embedding front-ends (word+pos+seg sum, patchify, ViT cls/pos assembly).
#include "common.h"
#include "lr2ppo_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ index,
                                                          float* __restrict__ dst, int t_in, int t_out, uint64_t row_elems,
                                                          uint64_t bstride, uint64_t tstride) {
  const int row = blockIdx.y;  // b * t_out + j
  const int b = row / t_out, j = row % t_out;
  int64_t src_t = index ? index[(size_t)b * t_out + j] : j;
  if (src_t < 0) src_t = 0;
  if (src_t >= t_in) src_t = t_in - 1;
  const float4* s = reinterpret_cast<const float4*>(src + (size_t)b * bstride + (size_t)src_t * tstride);
  float4* d = reinterpret_cast<float4*>(dst + (size_t)row * row_elems);
  const uint64_t n4 = row_elems / 4;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}

// dsrc[b, t, :] = sum over j with index[b, j] == t of ddst[b, j, :]   (deterministic, no atomics)
__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ ddst, const int64_t* __restrict__ index,
                                                              float* __restrict__ dsrc, int t_in, int t_out,
                                                              uint64_t row_elems) {
  const int row = blockIdx.y;  // b * t_in + t
  const int b = row / t_in, t = row % t_in;
  float4* d = reinterpret_cast<float4*>(dsrc + (size_t)row * row_elems);
  const uint64_t n4 = row_elems / 4;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < t_out; ++j) {
      const int64_t st = index ? index[(size_t)b * t_out + j] : j;
      if (st == t) {
        const float4 v = reinterpret_cast<const float4*>(ddst + ((size_t)b * t_out + j) * row_elems)[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    d[i] = a;
  }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                        size_t lo_off, int rows, int D, int group, uint64_t gstride,
                                                        uint64_t off) {
  const int d4 = D / 4;
  const size_t total = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / d4), c = (int)(i % d4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)r * D + c);
    const size_t o = (size_t)(r / group) * gstride + (size_t)(r % group) * D + off + c;
    if (planes) store_planes4(reinterpret_cast<bf16_t*>(dst) + o, lo_off, v);
    else *reinterpret_cast<float4*>(dst + o) = v;
  }
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                           size_t lo_off, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
    store_planes4(dst + i * 4, lo_off, reinterpret_cast<const float4*>(src)[i]);
}

// planes of dropout_mask(src) / (1 - p): the gradient entering a dropped branch (pre-LN encoder layers)
__global__ __launch_bounds__(256) void dropout_planes_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             size_t lo_off, size_t n4, float scale, uint32_t thr, uint64_t key) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(src)[i];
    const uint64_t e = (uint64_t)i * 4;
    v = dropout_apply4(key, e, thr, scale, v);
    store_planes4(dst + i * 4, lo_off, v);
  }
}

// dst = dropout_mask(src) / (1 - p) in fp32 (embedding dropout forward, and its backward on the incoming gradient)
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n4,
                                                            float scale, uint32_t thr, uint64_t key) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(src)[i];
    const uint64_t e = (uint64_t)i * 4;
    v = dropout_apply4(key, e, thr, scale, v);
    reinterpret_cast<float4*>(dst)[i] = v;
  }
}

// Backward of the word / segment embedding gathers, deterministic (no float atomics: run-to-run identical bits).
// Word table: the caller passes the rows sorted by token id (`order` = stable argsort of the ids, `sorted_ids` = ids in
// that order).  A run of equal ids is cut at the multiples of WORD_SEG sorted positions into PIECES; workgroup r owns the piece
// that starts at sorted position r (a run start, or a multiple of WORD_SEG inside a run; every other workgroup exits at once)
// and adds its rows in sorted order = original row order.  A run that is one piece (every ordinary token) is written straight
// to its table row.  A longer run (the padding id of a 640 x 196 batch is ~1e5 rows) leaves one partial per piece --
// slot 2c+1 for the piece holding the run's start in chunk c, slot 2c for a piece that starts chunk c -- and
// text_embed_bwd_word_finish adds the partials of a run in piece order: fixed order, chip-wide parallel.
constexpr int WORD_SEG = LR2_TEXT_EMBED_BWD_WORD_SEG;
__global__ __launch_bounds__(256) void text_embed_bwd_word_kernel(const float* __restrict__ dx, const int64_t* __restrict__ sorted_ids,
                                                                  const int64_t* __restrict__ order, float* __restrict__ dword,
                                                                  float* __restrict__ partials, int rows, int D, int64_t vocab) {
  const int r = blockIdx.x;
  const int64_t tok = sorted_ids[r];
  const bool run_start = (r == 0) || sorted_ids[r - 1] != tok;
  if (!run_start && (r % WORD_SEG) != 0) return;
  if (tok < 0 || tok >= vocab) return;          // flagged by the forward; never written
  const int cut = min(rows, (r / WORD_SEG + 1) * WORD_SEG);
  int end = r + 1;
  while (end < cut && sorted_ids[end] == tok) ++end;
  const bool run_ends = (end == rows) || sorted_ids[end] != tok;
  float* dst = (run_start && run_ends) ? dword + (size_t)tok * D
                                       : partials + ((size_t)2 * (r / WORD_SEG) + (run_start ? 1 : 0)) * D;
  for (int c = threadIdx.x * 4; c < D; c += 256 * 4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = r; j < end; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(dx + (size_t)order[j] * D + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(dst + c) = a;
  }
}
// One workgroup per chunk c of WORD_SEG sorted positions: if the run that crosses the chunk's upper boundary STARTS in this chunk,
// sum its pieces (slot 2c+1, then slots 2(c+1), 2(c+2), ... while the run goes on) into the table row.
__global__ __launch_bounds__(256) void text_embed_bwd_word_finish(const int64_t* __restrict__ sorted_ids, const float* __restrict__ partials,
                                                                  float* __restrict__ dword, int rows, int D, int64_t vocab) {
  const int c = blockIdx.x;
  const int last = (c + 1) * WORD_SEG - 1;
  if (last + 1 >= rows) return;                                  // nothing beyond this chunk
  const int64_t tok = sorted_ids[last];
  if (sorted_ids[last + 1] != tok) return;                        // no run crosses the boundary
  if (c > 0 && sorted_ids[c * WORD_SEG - 1] == tok) return;       // the run started in an earlier chunk: finished there
  if (tok < 0 || tok >= vocab) return;
  int n_cont = 0;                                                 // continuation pieces: chunks c+1 .. c+n_cont
  while ((c + 1 + n_cont) * WORD_SEG < rows && sorted_ids[(c + 1 + n_cont) * WORD_SEG] == tok) ++n_cont;
  for (int col = threadIdx.x * 4; col < D; col += 256 * 4) {
    float4 a = *reinterpret_cast<const float4*>(partials + ((size_t)2 * c + 1) * D + col);
    for (int k = 1; k <= n_cont; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(partials + (size_t)2 * (c + k) * D + col);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(dword + (size_t)tok * D + col) = a;
  }
}
// Segment table (n_seg <= 4 rows): per-workgroup partial sums over a contiguous chunk of rows, [block][n_seg][D];
// lr2_colsum_partials_finish-style tree is replaced by a fixed-order sum over blocks in text_embed_bwd_seg_finish.
__global__ __launch_bounds__(256) void text_embed_bwd_seg_kernel(const float* __restrict__ dx, const int64_t* __restrict__ seg,
                                                                 float* __restrict__ partials, int rows, int D, int n_seg,
                                                                 int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < D; c += 256) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = r0; r < r1; ++r) {
      const int sg = (int)seg[r];
      const float v = dx[(size_t)r * D + c];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] += (sg == k) ? v : 0.f;
    }
    for (int k = 0; k < n_seg; ++k) partials[((size_t)blockIdx.x * n_seg + k) * D + c] = a[k];
  }
}
__global__ __launch_bounds__(256) void text_embed_bwd_seg_finish(const float* __restrict__ partials, float* __restrict__ dseg,
                                                                 int nblocks, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float a = 0.f;
  for (int b = 0; b < nblocks; ++b) a += partials[(size_t)b * total + i];
  dseg[i] = a;
}

// dst planes [C][R] = transpose of src fp32 [R][C] (weights re-laid so that the forward GEMM can run in its NN form).
// 32 x 32 tiles through LDS; reads and writes are both row-contiguous.
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t lo_off,
                                                             int R, int C) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;             // output row c, column r
    if (c < C && r < R) {
      const float v = tile[tx][ty + 8 * i];
      const bf16_t h = f2bf(v);
      dst[(size_t)c * R + r] = h;
      dst[(size_t)c * R + r + lo_off] = f2bf(v - bf2f(h));
    }
  }
}

struct SplitChunk {
  const float* src;
  bf16_t* dst_hi;
  uint64_t lo_off;
  uint64_t count;
};
__global__ __launch_bounds__(256) void split_planes_multi_kernel(const SplitChunk* __restrict__ table) {
  const SplitChunk c = table[blockIdx.x];
  const size_t n4 = c.count / 4;
  for (size_t i = threadIdx.x; i < n4; i += 256) store_planes4(c.dst_hi + i * 4, c.lo_off, reinterpret_cast<const float4*>(c.src)[i]);
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, int rows, int D,
                                                       int row_step, int row_off) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + ((size_t)r * row_step + row_off) * D;
  float s = 0.f;
  for (int e = lane * 4; e < D; e += 256) {
    const float4 a = *reinterpret_cast<const float4*>(xr + e);
    const float4 ww = *reinterpret_cast<const float4*>(w + e);
    s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
  }
  s = wave_sum(s);
  if (lane == 0) y[r] = s + b[0];
}

__global__ __launch_bounds__(256) void head_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int D, int row_step, int row_off,
                                                          int total_rows) {
  const int d4 = D / 4;
  const size_t total = (size_t)total_rows * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / d4), c = (int)(i % d4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row % row_step == row_off) {
      const float g = dy[row / row_step];
      const float4 ww = *reinterpret_cast<const float4*>(w + c);
      v = make_float4(g * ww.x, g * ww.y, g * ww.z, g * ww.w);
    }
    *reinterpret_cast<float4*>(dx + (size_t)row * D + c) = v;
  }
}

__global__ __launch_bounds__(256) void head_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dw, float* __restrict__ db, int rows, int D,
                                                          int row_step, int row_off) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < D) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += dy[r] * x[((size_t)r * row_step + row_off) * D + c];
    dw[c] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += dy[r];
    db[0] = s;
  }
}

__global__ __launch_bounds__(256) void add_period_rows_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                              float* __restrict__ out, int rows, int D, int period) {
  const int d4 = D / 4;
  const size_t total = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / d4), c = (int)(i % d4) * 4;
    const float4 a = *reinterpret_cast<const float4*>(x + (size_t)r * D + c);
    const float4 t = *reinterpret_cast<const float4*>(table + (size_t)(r % period) * D + c);
    *reinterpret_cast<float4*>(out + (size_t)r * D + c) = make_float4(a.x + t.x, a.y + t.y, a.z + t.z, a.w + t.w);
  }
}

__global__ __launch_bounds__(256) void period_rows_grad_kernel(const float* __restrict__ dy, float* __restrict__ dtable,
                                                               int rows, int D, int period) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= period * D) return;
  const int t = idx / D, c = idx % D;
  float s = 0.f;
  for (int r = t; r < rows; r += period) s += dy[(size_t)r * D + c];
  dtable[(size_t)t * D + c] = s;
}

// ---------------------------------------------------------------------------------------------
// Fused PPO loss, one workgroup, one thread per item (finetune/ppo.py:544-584).
constexpr int PPO_MAX_T = 8;
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float clamped_log(float t) { return logf(fmaxf(t, 1e-20f)); }

__global__ __launch_bounds__(1024) void ppo_loss_kernel(const float* __restrict__ scores, const float* __restrict__ old_scores,
                                                        const float* __restrict__ rewards, const float* __restrict__ old_value,
                                                        const float* __restrict__ value, const int64_t* __restrict__ next_state,
                                                        int ns_len, int rank_len, int B, int T, float kl_w, float ent_w, float clip,
                                                        float margin, float adv_eps, float* __restrict__ scalars,
                                                        float* __restrict__ per_item, float* __restrict__ dscores,
                                                        float* __restrict__ dvalue, float* __restrict__ stats_out,
                                                        const float* __restrict__ global_stats, int world) {
  __shared__ float red[16];
  const int i = threadIdx.x;
  const bool act = i < B;
  float s[PPO_MAX_T], p[PPO_MAX_T], qo[PPO_MAX_T], dR[PPO_MAX_T];
  int order[PPO_MAX_T];
  float kl = 0.f, ent = 0.f, r = 0.f, adv = 0.f, hsum = 0.f, hcnt = 0.f, vl = 0.f, dv = 0.f;
#pragma unroll
  for (int t = 0; t < PPO_MAX_T; ++t) { s[t] = 0.f; p[t] = 0.f; qo[t] = 0.f; dR[t] = 0.f; order[t] = 0; }
  if (act) {
    float mx = -INFINITY, mo = -INFINITY;
#pragma unroll
    for (int t = 0; t < PPO_MAX_T; ++t)
      if (t < T) {
        s[t] = scores[(size_t)i * T + t];
        qo[t] = old_scores[(size_t)i * T + t];
        mx = fmaxf(mx, s[t]);
        mo = fmaxf(mo, qo[t]);
      }
    float zs = 0.f, zo = 0.f;
#pragma unroll
    for (int t = 0; t < PPO_MAX_T; ++t)
      if (t < T) {
        p[t] = expf(s[t] - mx);
        qo[t] = expf(qo[t] - mo);
        zs += p[t];
        zo += qo[t];
      }
#pragma unroll
    for (int t = 0; t < PPO_MAX_T; ++t)
      if (t < T) {
        p[t] /= zs;
        qo[t] /= zo;
        if (kl_w > 0.f) kl += qo[t] * (clamped_log(qo[t]) - clamped_log(p[t]));
        if (ent_w > 0.f) ent -= p[t] * clamped_log(p[t]);
      }
    r = rewards[i] - kl * kl_w;
    adv = r - old_value[i];
    const bool keep = adv >= adv_eps;
    // target order = the last rank_len entries of next_state (the reference hard-codes [-2:], finetune/ppo.py:565-567)
#pragma unroll
    for (int t = 0; t < PPO_MAX_T; ++t)
      if (t < rank_len) {
        const int src = keep ? t : (rank_len - 1 - t);
        order[t] = (int)next_state[(size_t)i * ns_len + (ns_len - rank_len) + src];
      }
    // pairwise hinge over positions a < b of the target order (RankLoss, finetune/ppo.py:43-55)
#pragma unroll
    for (int a = 0; a < PPO_MAX_T; ++a)
#pragma unroll
      for (int bq = 0; bq < PPO_MAX_T; ++bq)
        if (a < bq && bq < rank_len) {
          float sa = 0.f, sb = 0.f;
#pragma unroll
          for (int t = 0; t < PPO_MAX_T; ++t) {
            sa = (order[a] == t) ? s[t] : sa;
            sb = (order[bq] == t) ? s[t] : sb;
          }
          const float hgap = margin - (sa - sb);
          if (hgap > 0.f) {
            hsum += hgap;
            hcnt += 1.f;
#pragma unroll
            for (int t = 0; t < PPO_MAX_T; ++t) {
              dR[t] += (order[a] == t) ? -1.f : 0.f;
              dR[t] += (order[bq] == t) ? 1.f : 0.f;
            }
          }
        }
    // clipped value loss (finetune/ppo.py:494-498) with r' detached (:583)
    const float v = value[i], ov = old_value[i];
    const float dlt = v - ov;
    const float cl = fminf(fmaxf(dlt, -clip), clip);
    const float vc = ov + cl;
    const float l1 = (vc - r) * (vc - r), l2 = (v - r) * (v - r);
    vl = fmaxf(l1, l2);
    const float g1 = (dlt >= -clip && dlt <= clip) ? 2.f * (vc - r) : 0.f;
    const float g2 = 2.f * (v - r);
    dv = (l1 > l2) ? g1 : ((l1 == l2) ? 0.5f * (g1 + g2) : g2);
  }
  const float HS = block_sum_1024(hsum, red);
  const float CNT = block_sum_1024(hcnt, red);
  const float ABS = block_sum_1024(fabsf(adv), red);
  const float ENT = block_sum_1024(ent, red);
  const float VL = block_sum_1024(vl, red);
  const float invB = 1.0f / (float)B;
  if (stats_out) {          // pass 1 of the data-parallel form: this rank's hinge sum / positive count / sum |A|, nothing else
    if (threadIdx.x == 0) { stats_out[0] = HS; stats_out[1] = CNT; stats_out[2] = ABS; }
    return;
  }
  // Data parallel (global_stats = the three sums all-reduced over `world` ranks): RankLoss is ONE scalar over the whole
  // global batch (finetune/ppo.py:43-55), so R, its 1/count and mean |A| use the global sums; with the per-rank gradients
  // averaged afterwards, d(loss_global)/d(score) x world is what this rank must emit, and world / (world * B) = 1 / B.
  const float HSg = global_stats ? global_stats[0] : HS;
  const float CNTg = global_stats ? global_stats[1] : CNT;
  const float ABSm = global_stats ? global_stats[2] / (float)world : ABS;   // x invB below = global mean |A|
  const float R = (CNTg > 0.f) ? HSg / CNTg : 0.f;
  if (threadIdx.x == 0) {
    scalars[0] = R * ABSm * invB - ent_w * ENT * invB;
    scalars[1] = VL * invB;
    scalars[2] = R;
    scalars[3] = CNTg;
  }
  if (act) {
    per_item[0 * (size_t)B + i] = kl;
    per_item[1 * (size_t)B + i] = ent;
    per_item[2 * (size_t)B + i] = r;
    per_item[3 * (size_t)B + i] = adv;
    const float sgn = (adv > 0.f) ? 1.f : ((adv < 0.f) ? -1.f : 0.f);
    const float invC = (CNTg > 0.f) ? (global_stats ? (float)world : 1.0f) / CNTg : 0.f;
#pragma unroll
    for (int t = 0; t < PPO_MAX_T; ++t)
      if (t < T) {
        float gsc = ABSm * invB * invC * dR[t];
        if (kl_w > 0.f) gsc += R * invB * sgn * (-kl_w) * (p[t] - qo[t]);
        if (ent_w > 0.f) gsc += ent_w * invB * p[t] * (clamped_log(p[t]) + ent);
        dscores[(size_t)i * T + t] = gsc;
      }
    dvalue[i] = dv * invB;
  }
}

// ---------------------------------------------------------------------------------------------
// mode = 'cls' (finetune/ppo.py:209-210,229-230,239-242): a 768 -> C classification head, NLL of log-softmax, and the
// "expected label" score sum_k k * softmax(z)_k the PPO loop ranks by (ppo.py:532-537,859-863); C <= 8.
constexpr int CLS_MAX_C = 8;
// y[r, c] = x[r, :] . w[c, :] + b[c]; one wave per row
__global__ __launch_bounds__(256) void cls_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float* __restrict__ y, int rows, int D,
                                                           int C) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float acc[CLS_MAX_C];
#pragma unroll
  for (int c = 0; c < CLS_MAX_C; ++c) acc[c] = 0.f;
  for (int d = lane * 4; d < D; d += 256) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * D + d);
#pragma unroll
    for (int c = 0; c < CLS_MAX_C; ++c)
      if (c < C) {
        const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)c * D + d);
        acc[c] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
      }
  }
#pragma unroll
  for (int c = 0; c < CLS_MAX_C; ++c)
    if (c < C) {
      const float s = wave_sum(acc[c]);
      if (lane == 0) y[(size_t)r * C + c] = s + b[c];
    }
}
// dx[r, :] = sum_c dy[r, c] w[c, :]
__global__ __launch_bounds__(256) void cls_head_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                              float* __restrict__ dx, int rows, int D, int C) {
  const int d4 = D / 4;
  const size_t total = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / d4), d = (int)(i % d4) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float g = dy[(size_t)r * C + c];
      const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)c * D + d);
      a.x += g * wv.x; a.y += g * wv.y; a.z += g * wv.z; a.w += g * wv.w;
    }
    *reinterpret_cast<float4*>(dx + (size_t)r * D + d) = a;
  }
}
// dw[c, d] = sum_r dy[r, c] x[r, d]; db[c] = sum_r dy[r, c]   (rows is a few hundred at most: one thread per (c, d))
__global__ __launch_bounds__(256) void cls_head_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db, int rows, int D,
                                                              int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= C * D) return;
  const int c = i / D, d = i % D;
  float a = 0.f, bsum = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float g = dy[(size_t)r * C + c];
    a += g * x[(size_t)r * D + d];
    bsum += g;
  }
  dw[i] = a;
  if (d == 0) db[c] = bsum;
}
// probs = softmax(logits) (or, with use_softmax == 0, the raw logits as evaluate() uses them, ppo.py:641-643);
// scores[r] = sum_k k * probs[r, k]
__global__ __launch_bounds__(256) void cls_scores_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                         float* __restrict__ scores, int rows, int C, int use_softmax) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float z[CLS_MAX_C], mx = -INFINITY, sum = 0.f, e = 0.f;
#pragma unroll
  for (int c = 0; c < CLS_MAX_C; ++c) {
    z[c] = c < C ? logits[(size_t)r * C + c] : -INFINITY;
    mx = fmaxf(mx, z[c]);
  }
  if (use_softmax) {
#pragma unroll
    for (int c = 0; c < CLS_MAX_C; ++c) {
      z[c] = c < C ? expf(z[c] - mx) : 0.f;
      sum += z[c];
    }
  } else {
    sum = 1.f;
  }
#pragma unroll
  for (int c = 0; c < CLS_MAX_C; ++c)
    if (c < C) {
      const float p = z[c] / sum;
      if (probs) probs[(size_t)r * C + c] = p;
      e += (float)c * p;
    }
  scores[r] = e;
}
// d scores / d logits through the softmax: dlogits[r, c] = dscores[r] * p_c * (c - scores[r])
__global__ __launch_bounds__(256) void cls_scores_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ scores,
                                                             const float* __restrict__ dscores, float* __restrict__ dlogits,
                                                             int rows, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * C) return;
  const int r = i / C, c = i % C;
  dlogits[i] = dscores[r] * probs[i] * ((float)c - scores[r]);
}
// mean NLL of log-softmax (nn.NLLLoss()(nn.LogSoftmax(-1)(logits), tgts), ppo.py:240) and its gradient
__global__ __launch_bounds__(1024) void nll_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ tgts, int rows,
                                                        int C, float* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    float mx = -INFINITY, sum = 0.f;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, logits[(size_t)r * C + c]);
    for (int c = 0; c < C; ++c) sum += expf(logits[(size_t)r * C + c] - mx);
    const float lz = mx + logf(sum);
    int t = (int)tgts[r];
    t = t < 0 ? 0 : (t >= C ? C - 1 : t);
    acc += lz - logits[(size_t)r * C + t];
    if (dlogits)
      for (int c = 0; c < C; ++c)
        dlogits[(size_t)r * C + c] = (expf(logits[(size_t)r * C + c] - lz) - (c == t ? 1.f : 0.f)) / (float)rows;
  }
  const float s = block_sum_1024(acc, red);
  if (threadIdx.x == 0) loss[0] = s / (float)rows;
}

__global__ __launch_bounds__(1024) void smooth_l1_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, int n,
                                                         float beta, float* __restrict__ loss, float* __restrict__ dpred) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float d = pred[i] - tgt[i], a = fabsf(d);
    acc += (a < beta) ? 0.5f * d * d / beta : a - 0.5f * beta;
    if (dpred) dpred[i] = ((a < beta) ? d / beta : ((d > 0.f) ? 1.f : -1.f)) / (float)n;
  }
  const float s = block_sum_1024(acc, red);
  if (threadIdx.x == 0) loss[0] = s / (float)n;
}

// Pairwise hinge of the stage-2 reward training (finetune/reward_pair_dataloader.py:356-359):
// loss = mean relu(margin - (chosen - reject)), acc = mean (chosen > reject); scores = [chosen(bs) ; reject(bs)].
__global__ __launch_bounds__(1024) void pair_hinge_kernel(const float* __restrict__ scores, int bs, float margin,
                                                          float* __restrict__ out2, float* __restrict__ dscores) {
  __shared__ float red[16];
  float loss = 0.f, acc = 0.f;
  for (int i = threadIdx.x; i < bs; i += blockDim.x) {
    const float c = scores[i], r = scores[bs + i];
    const float h = margin - (c - r);
    loss += h > 0.f ? h : 0.f;
    acc += c > r ? 1.f : 0.f;
    if (dscores) {
      const float g = h > 0.f ? 1.0f / (float)bs : 0.f;
      dscores[i] = -g;
      dscores[bs + i] = g;
    }
  }
  const float ls = block_sum_1024(loss, red);
  __syncthreads();
  const float as = block_sum_1024(acc, red);
  if (threadIdx.x == 0) {
    out2[0] = ls / (float)bs;
    out2[1] = as / (float)bs;
  }
}

// ---------------------------------------------------------------------------------------------
// AdamW, correct_bias=False, decay applied after the Adam update with the same lr
// (tencentpretrain/utils/optimizers.py:381-400).  28 B of HBM traffic per parameter.
struct AdamChunk {
  float* p;
  const float* g;
  float* m;
  float* v;
  uint64_t count;
  float wd;
  float pad;
};
__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk* __restrict__ table, float lr_host, float b1, float b2,
                                                    float ob1, float ob2, float eps, const float* __restrict__ lr_dev) {
  const float lr = lr_dev ? scalar_load_f32(lr_dev) : lr_host;
  const AdamChunk c = table[blockIdx.x];
  const uint64_t n4 = c.count / 4;
  float4* p4 = reinterpret_cast<float4*>(c.p);
  const float4* g4 = reinterpret_cast<const float4*>(c.g);
  float4* m4 = reinterpret_cast<float4*>(c.m);
  float4* v4 = reinterpret_cast<float4*>(c.v);
#define ADAM1(P, G, M, V) adam_update(P, G, M, V, lr, b1, b2, ob1, ob2, eps, c.wd);
  for (uint64_t i = threadIdx.x; i < n4; i += 256) {
    float4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
    ADAM1(p.x, g.x, m.x, v.x) ADAM1(p.y, g.y, m.y, v.y) ADAM1(p.z, g.z, m.z, v.z) ADAM1(p.w, g.w, m.w, v.w)
    p4[i] = p; m4[i] = m; v4[i] = v;
  }
  for (uint64_t i = n4 * 4 + threadIdx.x; i < c.count; i += 256) {
    float p = c.p[i], g = c.g[i], m = c.m[i], v = c.v[i];
    ADAM1(p, g, m, v)
    c.p[i] = p; c.m[i] = m; c.v[i] = v;
  }
#undef ADAM1
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ seg,
                                                         const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ seg_table, float* __restrict__ out,
                                                         int rows, int L, int D, int64_t vocab, int n_seg, int* __restrict__ err) {
  const int d4 = D / 4;
  const size_t total = (size_t)rows * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / d4), c = (int)(i % d4) * 4;
    int64_t tok = src[r], sg = seg[r];
    // nn.Embedding raises on an out-of-range index; here the row is read from index 0 and the error word is set
    // (bit 0: token id, bit 1: segment id) for the host to raise -- never an out-of-bounds access.
    if (tok < 0 || tok >= vocab) { if (c == 0 && err) atomicOr(err, 1); tok = 0; }
    if (sg < 0 || sg >= n_seg) { if (c == 0 && err) atomicOr(err, 2); sg = 0; }
    const float4 a = *reinterpret_cast<const float4*>(word + (size_t)tok * D + c);
    const float4 b = *reinterpret_cast<const float4*>(pos + (size_t)(r % L) * D + c);
    const float4 s = *reinterpret_cast<const float4*>(seg_table + (size_t)sg * D + c);
    // same association as the reference: (word + pos) + seg   (embeddings/embedding.py:24-30)
    *reinterpret_cast<float4*>(out + (size_t)r * D + c) =
        make_float4((a.x + b.x) + s.x, (a.y + b.y) + s.y, (a.z + b.z) + s.z, (a.w + b.w) + s.w);
  }
}

// Image -> patch rows, written directly as the bf16 hi/lo planes the projection GEMM streams.
// out[(b*P + p), c*ps*ps + i*ps + j] = pixel(b, c, py*ps + i, px*ps + j); one thread owns 16 consecutive k (one pixel row of
// one patch and channel: 16 contiguous input pixels -> 32 contiguous bytes per plane); the 16 lanes of a group cover the 16
// pixel rows, so a group writes 512 contiguous bytes per plane, and all reads of a workgroup (one strip of patches) fall in
// 16 image rows per channel (L1/L2 hits).  U8: pixels are uint8 frames, normalised on the fly as the reference's loader
// does: x.float().div(255) then (x - mean[c]) / std[c] (tencentpretrain/utils/dataloader.py:559-561), IEEE division.
template <bool U8>
__global__ __launch_bounds__(256) void patchify_planes_kernel(const void* __restrict__ img, bf16_t* __restrict__ out, size_t lo_off,
                                                              int B, int C, int H, int W, float m0, float m1, float m2,
                                                              float s0, float s1, float s2, int normalize) {
  constexpr int ps = 16;
  const int px = W / ps, py = H / ps, P = px * py, Kd = C * ps * ps;
  const int strip = blockIdx.x;               // b * py + y
  const int b = strip / py, y = strip % py;
  const int per_strip = px * C * ps;          // (patch x, channel, pixel row) triples
  for (int t = threadIdx.x; t < per_strip; t += 256) {
    const int i = t % ps, c = (t / ps) % C, x = t / (ps * C);
    const size_t src = (((size_t)b * C + c) * H + (size_t)y * ps + i) * W + (size_t)x * ps;
    float v[16];
    if (U8) {
      const uint4 raw = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(img) + src);
      const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float f = (float)((w4[k >> 2] >> (8 * (k & 3))) & 0xffu) / 255.0f;
        if (normalize) f = (f - mean) / sd;
        v[k] = f;
      }
    } else {
      const float4* p4 = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(img) + src);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 q = p4[k];
        v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
      }
    }
    bf16_t* dst = out + ((size_t)b * P + (size_t)y * px + x) * Kd + (size_t)c * ps * ps + (size_t)i * ps;
#pragma unroll
    for (int k = 0; k < 4; ++k) store_planes4(dst + 4 * k, lo_off, make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]));
  }
}

// Any even patch size / padded row length (ViT-L/14: ps = 14, Kd = 588 padded to ld = 640 so that the projection GEMM sees
// whole K tiles): one thread per VEC consecutive k of one patch row; columns [Kd, ld) are written as zeros.
template <bool U8, int VEC>
__global__ __launch_bounds__(256) void patchify_planes_generic_kernel(const void* __restrict__ img, bf16_t* __restrict__ out,
                                                                      size_t lo_off, int B, int C, int H, int W, int ps, int ld,
                                                                      float m0, float m1, float m2, float s0, float s1, float s2,
                                                                      int normalize) {
  const int px = W / ps, py = H / ps, P = px * py, Kd = C * ps * ps;
  const int vpr = ld / VEC;                         // vectors per output row
  const size_t total = (size_t)B * P * vpr;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int k = (int)(t % vpr) * VEC;
    const size_t bp = t / vpr;
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    if (k < Kd) {
      const int p = (int)(bp % P), b = (int)(bp / P);
      const int c = k / (ps * ps), ij = k % (ps * ps), ii = ij / ps, jj = ij % ps;
      const size_t src = (((size_t)b * C + c) * H + (size_t)(p / px) * ps + ii) * W + (size_t)(p % px) * ps + jj;
      const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float f;
        if (U8) {
          f = (float)reinterpret_cast<const uint8_t*>(img)[src + e] / 255.0f;
          if (normalize) f = (f - mean) / sd;
        } else {
          f = reinterpret_cast<const float*>(img)[src + e];
        }
        v[e] = f;
      }
    }
    bf16_t* dst = out + bp * (size_t)ld + k;
    if (VEC == 4) {
      store_planes4(dst, lo_off, make_float4(v[0], v[1], v[VEC > 2 ? 2 : 0], v[VEC > 3 ? 3 : 0]));
    } else {
      const uint32_t h = cvt_pk_bf16(v[0], v[1]);
      const uint32_t l = cvt_pk_bf16(v[0] - __uint_as_float(h << 16), v[1] - __uint_as_float(h & 0xffff0000u));
      *reinterpret_cast<uint32_t*>(dst) = h;
      *reinterpret_cast<uint32_t*>(dst + lo_off) = l;
    }
  }
}

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int C,
                                                       int H, int W, int ps) {
  const int px = W / ps, py = H / ps, P = px * py, Kd = C * ps * ps;
  const size_t total = (size_t)B * P * Kd;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kd);
    const size_t bp = i / Kd;
    const int p = (int)(bp % P), b = (int)(bp / P);
    const int c = k / (ps * ps), ij = k % (ps * ps), ii = ij / ps, jj = ij % ps;
    const int y = (p / px) * ps + ii, x = (p % px) * ps + jj;
    out[i] = img[(((size_t)b * C + c) * H + y) * W + x];
  }
}

// NDCG@k per item with the reference's arithmetic (ndcg.py:28-65, finetune/ppo.py:651-659): items are ragged
// (offsets[i] .. offsets[i+1]); one thread per item: stable insertion sort by score (descending, earlier index first on
// ties), ideal order = labels sorted descending, DCG summed sequentially in fp32 with gain (2^rel - 1) and the discount
// table disc[i] = log2(i + 2) supplied by the host (so the only device arithmetic is IEEE divide / add: bit parity with
// the CPU reference); NDCG = 1 when the ideal DCG <= 1e-6.
constexpr int NDCG_MAX_T = 64;
__global__ __launch_bounds__(64) void ndcg_kernel(const float* __restrict__ scores, const int64_t* __restrict__ gold,
                                                  const int64_t* __restrict__ offsets, const float* __restrict__ disc,
                                                  const int64_t* __restrict__ ks, int n_k, float* __restrict__ out, int n_items) {
  const int item = blockIdx.x * 64 + threadIdx.x;
  if (item >= n_items) return;
  const int64_t o0 = offsets[item];
  const int64_t T64 = offsets[item + 1] - o0;
  bool bad = T64 < 0 || T64 > NDCG_MAX_T;   // an item this kernel cannot rank: its row is NaN, never a truncated NDCG
  const int T = bad ? 0 : (int)T64;
  float sc[NDCG_MAX_T];
  int64_t by_score[NDCG_MAX_T], ideal[NDCG_MAX_T];
  for (int i = 0; i < T; ++i) {           // insertion sorts (T <= 64, typically 20)
    const float s = scores[o0 + i];
    const int64_t g = gold[o0 + i];
    bad |= (g < 0 || g > 62);             // gain 2^rel - 1 in int64 (ndcg.py:28): a shift by < 0 or >= 63 is undefined
    int j = i;
    while (j > 0 && sc[j - 1] < s) { sc[j] = sc[j - 1]; by_score[j] = by_score[j - 1]; --j; }
    sc[j] = s; by_score[j] = g;
    j = i;
    while (j > 0 && ideal[j - 1] < g) { ideal[j] = ideal[j - 1]; --j; }
    ideal[j] = g;
  }
  for (int q = 0; q < n_k; ++q) {
    const int64_t k = ks[q];
    const int n = (int)(k < (int64_t)T ? k : (int64_t)T);
    float pred = 0.f, tru = 0.f;
    for (int i = 0; i < n; ++i) {
      pred += (float)((1ll << (by_score[i] & 63)) - 1) / disc[i];
      tru += (float)((1ll << (ideal[i] & 63)) - 1) / disc[i];
    }
    out[(size_t)item * n_k + q] = bad ? __builtin_nanf("") : ((tru <= 1e-6f) ? 1.0f : pred / tru);
  }
}

__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ proj, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ out, int B,
                                                           int P, int D) {
  const int d4 = D / 4;
  const size_t total = (size_t)B * (P + 1) * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % d4) * 4;
    const size_t bt = i / d4;
    const int t = (int)(bt % (P + 1)), b = (int)(bt / (P + 1));
    const float4 a = (t == 0) ? *reinterpret_cast<const float4*>(cls + c)
                              : *reinterpret_cast<const float4*>(proj + ((size_t)b * P + (t - 1)) * D + c);
    const float4 pe = *reinterpret_cast<const float4*>(pos + (size_t)t * D + c);
    *reinterpret_cast<float4*>(out + bt * D + c) = make_float4(a.x + pe.x, a.y + pe.y, a.z + pe.z, a.w + pe.w);
  }
}

inline int grid_for(size_t work_items, int cap = 4096) {
  size_t b = (work_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}
#define CHECK_LAUNCH() return lr2_launch_status(__func__)

}  // namespace

extern "C" int lr2_abi_version(void) { return LR2_ABI_VERSION; }

extern "C" int lr2_device_info(char* name, int len) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return LR2_ERR_LAUNCH;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return LR2_ERR_LAUNCH;
  if (name && len > 0) {
    int i = 0;
    for (; i < len - 1 && prop.gcnArchName[i]; ++i) name[i] = prop.gcnArchName[i];
    name[i] = 0;
  }
  return prop.multiProcessorCount;
}

extern "C" int lr2_gather_rows(const void* src, const int64_t* index, void* dst, int B, int t_in, int t_out,
                               uint64_t row_elems, uint64_t src_bstride, uint64_t src_tstride, void* stream) {
  if (!src || !dst || B <= 0 || t_in <= 0 || t_out <= 0) return LR2_ERR_ARG;
  if (row_elems % 4 || src_bstride % 4 || src_tstride % 4) return LR2_ERR_SHAPE;
  dim3 grid(grid_for(row_elems / 4, 64), B * t_out);
  LR2_LAUNCH(gather_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, index, (float*)dst,
                     t_in, t_out, row_elems, src_bstride, src_tstride);
  CHECK_LAUNCH();
}

extern "C" int lr2_gather_rows_bwd(const void* ddst, const int64_t* index, void* dsrc, int B, int t_in, int t_out,
                                   uint64_t row_elems, void* stream) {
  if (!ddst || !dsrc || B <= 0 || t_in <= 0 || t_out <= 0) return LR2_ERR_ARG;
  if (row_elems % 4) return LR2_ERR_SHAPE;
  dim3 grid(grid_for(row_elems / 4, 64), B * t_in);
  LR2_LAUNCH(gather_rows_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float*)ddst, index,
                     (float*)dsrc, t_in, t_out, row_elems);
  CHECK_LAUNCH();
}

extern "C" int lr2_copy_rows(const void* src, void* dst, int dst_planes, uint64_t dst_lo_off, int rows, int D, int group,
                             uint64_t dst_gstride, uint64_t dst_off, void* stream) {
  if (!src || !dst || rows <= 0 || D <= 0 || group <= 0) return LR2_ERR_ARG;
  if (D % 4 || dst_gstride % 4 || dst_off % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(copy_rows_kernel, dim3(grid_for((size_t)rows * D / 4)), dim3(256), 0, (hipStream_t)stream,
             (const float*)src, (float*)dst, dst_planes, (size_t)dst_lo_off, rows, D, group, dst_gstride, dst_off);
  CHECK_LAUNCH();
}

extern "C" int lr2_split_planes(const void* src, void* dst_hi, uint64_t lo_off, uint64_t n, void* stream) {
  if (!src || !dst_hi || n == 0) return LR2_ERR_ARG;
  if (n % 4 || lo_off % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(split_planes_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src,
             (bf16_t*)dst_hi, (size_t)lo_off, (size_t)(n / 4));
  CHECK_LAUNCH();
}

extern "C" int lr2_dropout_planes(const void* src, void* dst_hi, uint64_t lo_off, uint64_t n, float drop_p, uint64_t drop_seed,
                                  uint32_t drop_site, void* stream) {
  if (!src || !dst_hi || n == 0 || drop_p < 0.f || drop_p >= 1.f) return LR2_ERR_ARG;
  if (n % 4 || lo_off % 4) return LR2_ERR_SHAPE;
  if (drop_p == 0.f) return lr2_split_planes(src, dst_hi, lo_off, n, stream);
  LR2_LAUNCH(dropout_planes_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src,
             (bf16_t*)dst_hi, (size_t)lo_off, (size_t)(n / 4), 1.0f / (1.0f - drop_p), dropout_threshold(drop_p),
             (((uint64_t)drop_site) << 40) ^ (drop_seed * 0x9E3779B97F4A7C15ull));
  CHECK_LAUNCH();
}

extern "C" int lr2_dropout_apply(const void* src, void* dst, uint64_t n, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                                 void* stream) {
  if (!src || !dst || n == 0 || drop_p <= 0.f || drop_p >= 1.f) return LR2_ERR_ARG;
  if (n % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(dropout_apply_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (float*)dst,
             (size_t)(n / 4), 1.0f / (1.0f - drop_p), dropout_threshold(drop_p),
             (((uint64_t)drop_site) << 40) ^ (drop_seed * 0x9E3779B97F4A7C15ull));
  CHECK_LAUNCH();
}

extern "C" int lr2_text_embed_bwd(const void* dx, const int64_t* sorted_ids, const int64_t* order, const int64_t* seg,
                                  void* dword, void* dseg, void* seg_partials, void* word_partials, int rows, int D,
                                  int64_t vocab, int n_seg, void* stream) {
  if (!dx || !sorted_ids || !order || !seg || !dword || !dseg || !seg_partials || !word_partials || rows <= 0 || D <= 0)
    return LR2_ERR_ARG;
  if (D % 4 || n_seg < 1 || n_seg > 4) return LR2_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  LR2_LAUNCH(text_embed_bwd_word_kernel, dim3(rows), dim3(256), 0, s, (const float*)dx, sorted_ids, order, (float*)dword,
             (float*)word_partials, rows, D, vocab);
  if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  const int chunks = (rows + WORD_SEG - 1) / WORD_SEG;
  if (chunks > 1) {
    LR2_LAUNCH(text_embed_bwd_word_finish, dim3(chunks - 1), dim3(256), 0, s, sorted_ids, (const float*)word_partials,
               (float*)dword, rows, D, vocab);
    if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  }
  const int rows_per_block = LR2_TEXT_EMBED_BWD_ROWS_PER_BLOCK;
  const int nb = (rows + rows_per_block - 1) / rows_per_block;
  LR2_LAUNCH(text_embed_bwd_seg_kernel, dim3(nb), dim3(256), 0, s, (const float*)dx, seg, (float*)seg_partials, rows, D, n_seg,
             rows_per_block);
  if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  LR2_LAUNCH(text_embed_bwd_seg_finish, dim3((n_seg * D + 255) / 256), dim3(256), 0, s, (const float*)seg_partials,
             (float*)dseg, nb, n_seg * D);
  CHECK_LAUNCH();
}

extern "C" int lr2_split_planes_t(const void* src, void* dst_hi, uint64_t lo_off, int R, int C, void* stream) {
  if (!src || !dst_hi || R <= 0 || C <= 0) return LR2_ERR_ARG;
  LR2_LAUNCH(split_planes_t_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, (const float*)src,
             (bf16_t*)dst_hi, (size_t)lo_off, R, C);
  CHECK_LAUNCH();
}

extern "C" int lr2_split_planes_multi(const lr2_split_chunk* table_dev, int n_chunks, void* stream) {
  static_assert(sizeof(lr2_split_chunk) == sizeof(SplitChunk), "chunk layout");
  if (!table_dev || n_chunks <= 0) return LR2_ERR_ARG;
  LR2_LAUNCH(split_planes_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const SplitChunk*)table_dev);
  CHECK_LAUNCH();
}

extern "C" int lr2_head_fwd(const void* x, const void* w, const void* b, void* y, int rows, int D, int row_step,
                            int row_off, void* stream) {
  if (!x || !w || !b || !y || rows <= 0 || row_step <= 0) return LR2_ERR_ARG;
  if (D % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(head_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                     (const float*)w, (const float*)b, (float*)y, rows, D, row_step, row_off);
  CHECK_LAUNCH();
}

extern "C" int lr2_head_bwd(const void* x, const void* w, const void* dy, void* dx, void* dw, void* db, int rows, int D,
                            int row_step, int row_off, int total_rows, void* stream) {
  if (!x || !w || !dy || rows <= 0 || row_step <= 0) return LR2_ERR_ARG;
  if (D % 4) return LR2_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    LR2_LAUNCH(head_bwd_dx_kernel, dim3(grid_for((size_t)total_rows * D / 4)), dim3(256), 0, s, (const float*)w,
                       (const float*)dy, (float*)dx, D, row_step, row_off, total_rows);
    if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  }
  if (dw && db) {
    LR2_LAUNCH(head_bwd_dw_kernel, dim3((D + 255) / 256), dim3(256), 0, s, (const float*)x, (const float*)dy,
                       (float*)dw, (float*)db, rows, D, row_step, row_off);
  }
  CHECK_LAUNCH();
}

extern "C" int lr2_add_period_rows(const void* x, const void* table, void* out, int rows, int D, int period,
                                   void* stream) {
  if (!x || !table || !out || rows <= 0 || period <= 0) return LR2_ERR_ARG;
  if (D % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(add_period_rows_kernel, dim3(grid_for((size_t)rows * D / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (const float*)table, (float*)out, rows, D, period);
  CHECK_LAUNCH();
}

extern "C" int lr2_period_rows_grad(const void* dy, void* dtable, int rows, int D, int period, void* stream) {
  if (!dy || !dtable || rows <= 0 || period <= 0) return LR2_ERR_ARG;
  LR2_LAUNCH(period_rows_grad_kernel, dim3((period * D + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float*)dy, (float*)dtable, rows, D, period);
  CHECK_LAUNCH();
}

extern "C" int lr2_ppo_loss(const void* scores, const void* old_scores, const void* rewards, const void* old_value,
                            const void* value, const int64_t* next_state, int ns_len, int rank_len, int B, int T,
                            float kl_w, float ent_w, float value_clip, float margin, float adv_eps, void* scalars,
                            void* per_item, void* dscores, void* dvalue, void* stats_out, const void* global_stats, int world,
                            void* stream) {
  if (!scores || !old_scores || !rewards || !old_value || !value || !next_state) return LR2_ERR_ARG;
  if (!stats_out && (!scalars || !per_item || !dscores || !dvalue)) return LR2_ERR_ARG;
  if (global_stats && world < 1) return LR2_ERR_ARG;
  if (B < 1 || B > 1024 || T < 1 || T > PPO_MAX_T || rank_len < 1 || rank_len > T || ns_len < rank_len) return LR2_ERR_SHAPE;
  const int threads = ((B + 63) / 64) * 64;
  LR2_LAUNCH(ppo_loss_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, (const float*)scores,
                     (const float*)old_scores, (const float*)rewards, (const float*)old_value, (const float*)value,
                     next_state, ns_len, rank_len, B, T, kl_w, ent_w, value_clip, margin, adv_eps, (float*)scalars,
                     (float*)per_item, (float*)dscores, (float*)dvalue, (float*)stats_out, (const float*)global_stats, world);
  CHECK_LAUNCH();
}

extern "C" int lr2_cls_head_fwd(const void* x, const void* w, const void* b, void* y, int rows, int D, int C, void* stream) {
  if (!x || !w || !b || !y || rows <= 0) return LR2_ERR_ARG;
  if (D % 4 || C < 1 || C > CLS_MAX_C) return LR2_ERR_SHAPE;
  LR2_LAUNCH(cls_head_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)w,
             (const float*)b, (float*)y, rows, D, C);
  CHECK_LAUNCH();
}

extern "C" int lr2_cls_head_bwd(const void* x, const void* w, const void* dy, void* dx, void* dw, void* db, int rows, int D, int C,
                                void* stream) {
  if (!x || !w || !dy || rows <= 0) return LR2_ERR_ARG;
  if (D % 4 || C < 1 || C > CLS_MAX_C) return LR2_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    LR2_LAUNCH(cls_head_bwd_dx_kernel, dim3(grid_for((size_t)rows * D / 4)), dim3(256), 0, s, (const float*)w, (const float*)dy,
               (float*)dx, rows, D, C);
    if (lr2_launch_status(__func__)) return LR2_ERR_LAUNCH;
  }
  if (dw && db)
    LR2_LAUNCH(cls_head_bwd_dw_kernel, dim3((C * D + 255) / 256), dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dw,
               (float*)db, rows, D, C);
  CHECK_LAUNCH();
}

extern "C" int lr2_cls_scores(const void* logits, void* probs, void* scores, int rows, int C, int use_softmax, void* stream) {
  if (!logits || !scores || rows <= 0) return LR2_ERR_ARG;
  if (C < 1 || C > CLS_MAX_C) return LR2_ERR_SHAPE;
  LR2_LAUNCH(cls_scores_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)logits, (float*)probs,
             (float*)scores, rows, C, use_softmax);
  CHECK_LAUNCH();
}

extern "C" int lr2_cls_scores_bwd(const void* probs, const void* scores, const void* dscores, void* dlogits, int rows, int C,
                                  void* stream) {
  if (!probs || !scores || !dscores || !dlogits || rows <= 0) return LR2_ERR_ARG;
  if (C < 1 || C > CLS_MAX_C) return LR2_ERR_SHAPE;
  LR2_LAUNCH(cls_scores_bwd_kernel, dim3((rows * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)probs,
             (const float*)scores, (const float*)dscores, (float*)dlogits, rows, C);
  CHECK_LAUNCH();
}

extern "C" int lr2_nll_loss(const void* logits, const int64_t* tgts, int rows, int C, void* loss, void* dlogits, void* stream) {
  if (!logits || !tgts || !loss || rows <= 0) return LR2_ERR_ARG;
  if (C < 1 || C > CLS_MAX_C) return LR2_ERR_SHAPE;
  LR2_LAUNCH(nll_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)logits, tgts, rows, C, (float*)loss,
             (float*)dlogits);
  CHECK_LAUNCH();
}

extern "C" int lr2_smooth_l1(const void* pred, const void* target, int n, float beta, void* loss, void* dpred,
                             void* stream) {
  if (!pred || !target || !loss || n <= 0 || beta <= 0.f) return LR2_ERR_ARG;
  LR2_LAUNCH(smooth_l1_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)pred,
                     (const float*)target, n, beta, (float*)loss, (float*)dpred);
  CHECK_LAUNCH();
}

extern "C" int lr2_pair_hinge(const void* scores, int bs, float margin, void* loss_acc, void* dscores, void* stream) {
  if (!scores || !loss_acc || bs <= 0) return LR2_ERR_ARG;
  LR2_LAUNCH(pair_hinge_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)scores, bs, margin,
             (float*)loss_acc, (float*)dscores);
  CHECK_LAUNCH();
}

struct StepScalars {
  uint64_t seed;
  float lrs[LR2_STEP_SCALARS_MAX_LRS];
};
static_assert(sizeof(StepScalars) == LR2_STEP_SCALARS_BYTES, "step scalars layout");
__global__ void step_scalars_kernel(StepScalars* __restrict__ dst, StepScalars v) {
  if (threadIdx.x == 0) *dst = v;
}
extern "C" int lr2_step_scalars_store(void* dst_dev, uint64_t seed, const float* lrs_host, int n_lrs, void* stream) {
  if (!dst_dev || n_lrs < 0 || n_lrs > LR2_STEP_SCALARS_MAX_LRS || (n_lrs && !lrs_host)) return LR2_ERR_ARG;
  StepScalars v{};
  v.seed = seed;
  for (int i = 0; i < n_lrs; ++i) v.lrs[i] = lrs_host[i];
  LR2_LAUNCH(step_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (StepScalars*)dst_dev, v);
  CHECK_LAUNCH();
}

extern "C" int lr2_adamw_multi(const lr2_adamw_chunk* table_dev, int n_chunks, double lr, double beta1, double beta2,
                               double eps, const void* lr_dev, void* stream) {
  static_assert(sizeof(lr2_adamw_chunk) == sizeof(AdamChunk), "chunk layout");
  if (!table_dev || n_chunks <= 0) return LR2_ERR_ARG;
  // (1 - beta) is formed in double like the reference's Python scalars, then rounded once to fp32
  LR2_LAUNCH(adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, (const AdamChunk*)table_dev,
                     (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (const float*)lr_dev);
  CHECK_LAUNCH();
}

extern "C" int lr2_text_embed(const int64_t* src, const int64_t* seg, const void* word, const void* pos,
                              const void* seg_table, void* out, int rows, int L, int D, int64_t vocab, int n_seg, int* err_flag,
                              void* stream) {
  if (!src || !seg || !word || !pos || !seg_table || !out || rows <= 0 || L <= 0 || vocab <= 0 || n_seg <= 0) return LR2_ERR_ARG;
  if (D % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(text_embed_kernel, dim3(grid_for((size_t)rows * D / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     seg, (const float*)word, (const float*)pos, (const float*)seg_table, (float*)out, rows, L, D, vocab, n_seg,
                     err_flag);
  CHECK_LAUNCH();
}

extern "C" int lr2_patchify_planes(const void* img, int is_u8, void* out_hi, uint64_t lo_off, int ld, int B, int C, int H,
                                   int W, int ps, const float* mean3, const float* std3, void* stream) {
  if (!img || !out_hi || B <= 0 || C <= 0 || ps <= 0) return LR2_ERR_ARG;
  const int Kd = C * ps * ps;
  if (H % ps || W % ps || C > 3 || (ps & 1) || ld < Kd || (ld & 3) || (lo_off & 1)) return LR2_ERR_SHAPE;
  const int normalize = (is_u8 && mean3 && std3) ? 1 : 0;
  const float m0 = normalize ? mean3[0] : 0.f, m1 = normalize && C > 1 ? mean3[1] : 0.f, m2 = normalize && C > 2 ? mean3[2] : 0.f;
  const float s0 = normalize ? std3[0] : 1.f, s1 = normalize && C > 1 ? std3[1] : 1.f, s2 = normalize && C > 2 ? std3[2] : 1.f;
  hipStream_t s = (hipStream_t)stream;
  bf16_t* o = (bf16_t*)out_hi;
  if (ps == 16 && ld == Kd && (lo_off & 3) == 0) {    // ViT-B/16, ViT-L/16: the coalesced strip kernel
    const dim3 grid(B * (H / ps));
    if (is_u8) LR2_LAUNCH(patchify_planes_kernel<true>, grid, dim3(256), 0, s, img, o, (size_t)lo_off, B, C, H, W, m0, m1, m2, s0, s1, s2, normalize);
    else LR2_LAUNCH(patchify_planes_kernel<false>, grid, dim3(256), 0, s, img, o, (size_t)lo_off, B, C, H, W, m0, m1, m2, s0, s1, s2, 0);
    CHECK_LAUNCH();
  }
  const size_t P = (size_t)(H / ps) * (W / ps);
  if ((ps & 3) == 0 && (lo_off & 3) == 0) {
    const dim3 grid(grid_for((size_t)B * P * (ld / 4)));
    if (is_u8) LR2_LAUNCH((patchify_planes_generic_kernel<true, 4>), grid, dim3(256), 0, s, img, o, (size_t)lo_off, B, C, H, W, ps, ld, m0, m1, m2, s0, s1, s2, normalize);
    else LR2_LAUNCH((patchify_planes_generic_kernel<false, 4>), grid, dim3(256), 0, s, img, o, (size_t)lo_off, B, C, H, W, ps, ld, m0, m1, m2, s0, s1, s2, 0);
  } else {
    const dim3 grid(grid_for((size_t)B * P * (ld / 2)));
    if (is_u8) LR2_LAUNCH((patchify_planes_generic_kernel<true, 2>), grid, dim3(256), 0, s, img, o, (size_t)lo_off, B, C, H, W, ps, ld, m0, m1, m2, s0, s1, s2, normalize);
    else LR2_LAUNCH((patchify_planes_generic_kernel<false, 2>), grid, dim3(256), 0, s, img, o, (size_t)lo_off, B, C, H, W, ps, ld, m0, m1, m2, s0, s1, s2, 0);
  }
  CHECK_LAUNCH();
}

extern "C" int lr2_ndcg(const void* scores, const int64_t* gold, const int64_t* offsets, const void* disc, const int64_t* ks,
                        int n_k, void* out, int n_items, void* stream) {
  if (!scores || !gold || !offsets || !disc || !ks || !out || n_items <= 0 || n_k <= 0) return LR2_ERR_ARG;
  LR2_LAUNCH(ndcg_kernel, dim3((n_items + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float*)scores, gold, offsets,
             (const float*)disc, ks, n_k, (float*)out, n_items);
  CHECK_LAUNCH();
}

extern "C" int lr2_patchify(const void* img, void* out, int B, int C, int H, int W, int ps, void* stream) {
  if (!img || !out || B <= 0 || C <= 0 || ps <= 0) return LR2_ERR_ARG;
  if (H % ps || W % ps) return LR2_ERR_SHAPE;
  LR2_LAUNCH(patchify_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)img, (float*)out, B, C, H, W, ps);
  CHECK_LAUNCH();
}

extern "C" int lr2_vit_assemble(const void* patch_proj, const void* cls, const void* pos, void* out, int B, int P, int D,
                                void* stream) {
  if (!patch_proj || !cls || !pos || !out || B <= 0 || P <= 0) return LR2_ERR_ARG;
  if (D % 4) return LR2_ERR_SHAPE;
  LR2_LAUNCH(vit_assemble_kernel, dim3(grid_for((size_t)B * (P + 1) * D / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)patch_proj, (const float*)cls, (const float*)pos, (float*)out, B, P, D);
  CHECK_LAUNCH();
}
