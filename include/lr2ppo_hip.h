/* lr2ppo_hip.h -- C ABI of the MI355X (gfx950) kernels behind the LR2PPO hot path.
 *
 * The reference (ChazzyGordon/LR2PPO) is pure Python/PyTorch and has no FFI of its own:
 * every entry point below replaces a group of ATen/cuBLAS ops that the reference reaches through
 * nn.Module calls.  The "replaces" note on each function cites that reference call site
 * (paths relative to the reference repo root).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions: raw device pointers + sizes; `stream` is a hipStream_t passed as void*; every call is
 * stream-ordered and asynchronous (no internal sync, no allocation, graph-capturable); no ownership is
 * transferred; return 0 on success or a negative LR2_ERR_* code (no exceptions cross the ABI).
 * All tensors are fp32 in HBM exactly as in the reference (indices int64/int32); bf16 exists only inside
 * the GEMM kernel (LDS images of split operands).  All reductions/accumulators are fp32.
 */
#ifndef LR2PPO_HIP_H
#define LR2PPO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR2_ERR_ARG (-1)    /* null pointer / inconsistent arguments */
#define LR2_ERR_SHAPE (-2)  /* shape not supported by the kernel's tiling */
#define LR2_ERR_LAUNCH (-3) /* HIP launch failure */

#define LR2_ABI_VERSION 19
int lr2_abi_version(void);
/* Fills name[0..len) with the HIP device name and returns the CU count (or <0). */
int lr2_device_info(char* name, int len);

/* Fused GEMM epilogue, applied per element (m, n) in this order:
 *   v = acc*alpha (+bias[n]); act==1: z=v (stored to out_z if set), v=gelu_erf(v);
 *   drop_p>0: v = keep(seed,site,m*N+n) ? v/(1-p) : 0;  act==2: v *= gelu_erf'(aux_z[m,n]);
 *   resid: v += resid[m,n];  accumulate: v += out[m,n];  out[m,n] = v (fp32) and/or (out_hi, out_lo)[m,n] = split(v). */
typedef struct lr2_epilogue {
  const void* bias;   /* [N] or NULL */
  const void* resid;  /* [M, ld_resid] or NULL */
  const void* aux_z;  /* [M, ld_aux], required when act == 2 */
  void* out;          /* [M, ld_out] fp32 or NULL (then out_hi must be set) */
  void* out_z;        /* [M, ld_z] pre-activation (act == 1) or NULL */
  void* out_hi;       /* bf16 hi plane [M, ld_planes] or NULL: result ALSO/INSTEAD written as split planes */
  uint64_t out_lo_off; /* elements from the hi plane to the lo plane */
  int32_t ld_planes;
  int32_t ld_resid, ld_aux, ld_out, ld_z;
  int32_t act;        /* 0 none, 1 GELU(erf), 2 multiply by GELU'(aux_z) */
  int32_t accumulate; /* 1: out += result */
  float alpha;
  float drop_p;       /* 0 disables dropout */
  uint32_t drop_site;
  uint32_t _pad;
  uint64_t drop_seed;
  /* Fused optimizer step (weight-gradient GEMMs): when adam_p is set the result g[m,n] (after alpha) is not stored but
   * consumed by the AdamW update of lr2_adamw_multi on (adam_p, adam_m, adam_v)[m, ld_out*m + n]; `out` may be NULL.
   * The 2 GB gradient of out_layer.fc1.weight is then never written to or read back from HBM.  drop_p must be 0 with adam_p
   * (the update consumes the result: there is nothing to mask; LR2_ERR_ARG otherwise).
   * replaces: the fc1.weight slice of optimizers.py:344-402 + the .grad write of loss.backward(). */
  void* adam_p;
  void* adam_m;
  void* adam_v;
  double adam_lr, adam_beta1, adam_beta2, adam_eps, adam_weight_decay;
  /* (1,1) form only -- the weight gradient dW = dY^T X of an nn.Linear: when colsum is set, colsum[m] = sum over k of A[k, m]
   * (the gradient of the layer's BIAS, db = sum of dY rows) is produced by the same call.  On the 256 x 256 kernel (block_m =
   * 256) the sums are accumulated from the A fragments the product stages anyway -- dY is not read a second time; otherwise a
   * column-sum pass runs behind the product.  colsum_ws: fp32 scratch, max(128, splits * ceil(N / 256)) * M floats.
   * replaces: autograd of the bias of nn.Linear (tencentpretrain/layers/position_ffn.py:12-15, multi_headed_attn.py:55-76). */
  void* colsum;
  void* colsum_ws;
  /* Device-resident dropout seed (HIP-graph capture of a training step: a seed passed by value would be frozen into the graph):
   * when non-NULL, the mask stream's seed is *drop_seed_dev + drop_seed (device uint64, read by the kernel at run time);
   * drop_site must then be below 65536. */
  const void* drop_seed_dev;
  /* Device-resident learning rate of the fused optimizer step (same reason: a scheduler changes it every step): when non-NULL the
   * update uses *adam_lr_dev (device float) and adam_lr is ignored. */
  const void* adam_lr_dev;
} lr2_epilogue;

/* C[M,N] = op(A).op(B), fp32 in / fp32 out, computed on bf16 MFMA with fp32 accumulation.
 *   passes=3: split-bf16 (x = hi + lo; lo*hi + hi*lo + hi*hi): fp32-grade results (rel. err ~2^-17)
 *   passes=1: single bf16 pass (inputs rounded to bf16)
 *   trans_a=0: A is [M][K] (lda>=K);  trans_a=1: A is [K][M] (lda>=M)
 *   trans_b=0: B is [N][K] (nn.Linear weight);  trans_b=1: B is [K][N]
 * Supported forms: (0,0) forward, (0,1) input gradient, (1,1) weight gradient.
 * Constraints: K%64==0 unless both operands are strided (1,1); lda,ldb %4==0 (16-byte rows); buffers < 4 GiB.
 * M, N and (in the (1,1) form) K may be ragged.
 * Operand formats: a_planes / b_planes = 0: fp32 matrix (split into bf16 hi+lo inside the kernel);
 *   = 1: pre-split "planes" tensor: bf16 hi plane at the pointer, bf16 lo plane a_lo_off / b_lo_off BYTES later, both
 *   with the matrix's row-major shape and leading dimension (elements); streamed by LDS-DMA, no conversion work.
 *   Supported: (fp32, fp32), (planes, planes), (planes A, fp32 B).  Planes are produced by the epilogue's out_hi,
 *   by lr2_split_planes(_multi) and by the LayerNorm / attention kernels' planes outputs.
 * a_bytes/b_bytes: bytes addressable from A/B, per plane (rows past the end read as zero: ragged M, ragged K in (1,1)).
 * splits>1 uses split-K through splitk_ws (fp32 [splits][M][N]).  block_m: 128 or 64, or 256 = the 256 x 256 ping-pong kernel for
 *   planes x planes operands with passes = 3: the (0,0) form (whole 32-deep K steps, splits = 1) and the (1,1) form (any K, any
 *   splits: tiles x splits workgroups, the weight-gradient kernel of round 3); other combinations fall back to the 128-row family.
 * replaces: nn.Linear / F.linear + bias + nn.GELU + nn.Dropout + residual add in
 *   finetune/ppo.py:164-170 (Mlp), finetune/xit.py:103-110,118-122,147,
 *   tencentpretrain/layers/{multi_headed_attn.py:55-58,75, position_ffn.py:12-15} and their autograd backward. */
int lr2_gemm(const void* A, const void* B, int M, int N, int K, int lda, int ldb, int trans_a, int trans_b,
             uint64_t a_bytes, uint64_t b_bytes, int a_planes, uint64_t a_lo_off, int b_planes, uint64_t b_lo_off,
             const lr2_epilogue* epi, void* splitk_ws, int splits, int block_m, int passes, void* stream);

/* Round 4 (ABI 18).  How lr2_gemm schedules a large (0,0) product of planes asked for with block_m = 256 and no fused dropout mask:
 * *rows_256 = the leading rows that run on the 256 x 256 kernel as WHOLE rounds of one workgroup per CU (0 = one launch, no row
 * split); the remaining rows run on the general kernel with *tail_block_m-row tiles (2-3 workgroups per CU), so that a last round
 * less than half full does not hold the chip for a full tile time.  Pure host arithmetic; lr2_gemm follows this plan.
 * LR2_GEMM_ROWSPLIT=0 (read once per process) switches the split off.  Same nn.Linear call sites as lr2_gemm. */
int lr2_gemm_row_split_plan(int M, int N, int K, int* rows_256, int* tail_block_m);
/* Diagnostic: launches issued by lr2_gemm since the library was loaded -- counts[0] the 256 x 256 NT kernel, counts[1] its TN form,
 * counts[2] the general kernel family (tests assert which kernels a call reached; no device call). */
int lr2_gemm_launch_counts(uint64_t counts[3]);

/* Row gather: dst[b, j, :] = src[b, index[b, j], :]  (rows of row_elems fp32; strides in elements).
 * replaces: text_emb[batch_index, index] / img_emb[batch_index, index] (finetune/ppo.py:268-271,321-324). */
int lr2_gather_rows(const void* src, const int64_t* index, void* dst, int B, int t_in, int t_out, uint64_t row_elems,
                    uint64_t src_bstride, uint64_t src_tstride, void* stream);
/* Scatter-add of row gradients: dsrc[b, index[b,j], :] += ddst[b, j, :]  (dsrc must be zero-initialised). */
int lr2_gather_rows_bwd(const void* ddst, const int64_t* index, void* dsrc, int B, int t_in, int t_out,
                        uint64_t row_elems, void* stream);
/* Strided row copy: dst[(r / group)*dst_gstride + (r % group)*D + dst_off + c] = src[r*D + c].
 * dst_planes=1: dst is a bf16 planes tensor (same element offsets, lo plane dst_lo_off elements later).
 * replaces: torch.cat([x, img_feature], dim=1) (finetune/ppo.py:224). */
int lr2_copy_rows(const void* src, void* dst, int dst_planes, uint64_t dst_lo_off, int rows, int D, int group,
                  uint64_t dst_gstride, uint64_t dst_off, void* stream);
/* fp32 -> bf16 planes: dst_hi[i] = bf16(src[i]), dst_hi[lo_off + i] = bf16(src[i] - hi) for i < n (n % 4 == 0). */
int lr2_split_planes(const void* src, void* dst_hi, uint64_t lo_off, uint64_t n, void* stream);
/* Transposing split: src fp32 [R][C] -> planes [C][R] (hi at dst_hi, lo lo_off elements behind).  An nn.Linear weight
 * [out, in] re-laid as [in, out] lets its forward x.W^T run as the (0,1) form of lr2_gemm, the fastest of the three on wide
 * outputs; nothing in the reference corresponds to it (layout choice of this build). */
int lr2_split_planes_t(const void* src, void* dst_hi, uint64_t lo_off, int R, int C, void* stream);
/* planes of dropout_mask(src) / (1 - p), mask element index = flat element index (the gradient entering a dropped branch).
 * replaces: autograd of nn.Dropout at tencentpretrain/layers/transformer.py:55,58,65,72 on the pre-LN residual paths. */
int lr2_dropout_planes(const void* src, void* dst_hi, uint64_t lo_off, uint64_t n, float drop_p, uint64_t drop_seed,
                       uint32_t drop_site, void* stream);
/* The same for many tensors in one launch (weights after an optimizer step). table: DEVICE array of chunks. */
typedef struct lr2_split_chunk {
  const void* src;
  void* dst_hi;
  uint64_t lo_off; /* elements */
  uint64_t count;  /* elements, multiple of 4 */
} lr2_split_chunk;
int lr2_split_planes_multi(const lr2_split_chunk* table_dev, int n_chunks, void* stream);

/* LayerNorm forward over rows of length D (D%4==0, D<=1024).
 *   mode 0: nn.LayerNorm (biased variance, eps inside sqrt)      -- finetune/xit.py:37,74,93-94
 *   mode 1: TencentPretrain LayerNorm gamma*(x-mu)/(std_unbiased+eps)+beta -- tencentpretrain/layers/layer_norm.py:16-21
 * Output row r is written at out + (r / group)*group_stride + (r % group)*D (group<=0: dense), which lets the
 * final XiT LayerNorm write straight into the concat buffer of finetune/ppo.py:224.
 * out (fp32) and/or out_hi (bf16 planes, lo plane out_lo_off elements later, same row mapping) receive the result.
 * mean/rstd (fp32 [rows]) may be NULL when no backward is needed. */
int lr2_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* out, void* out_hi, uint64_t out_lo_off,
                      void* mean, void* rstd, int rows, int D, float eps, int mode, int group, uint64_t group_stride,
                      void* stream);

/* LayerNorm backward, both semantics (mode / eps as in the forward).  dy uses the same (group, stride) row mapping as the forward output.
 * dx = LN'(dy) (+ resid_grad) -> dx (fp32); optional dxm_hi = bf16 planes of dropout_mask(dx)/(1-p) (the gradient of
 * y = dropout(a) + res with respect to a, finetune/xit.py:34,40; p = 0: planes of dx), the next GEMMs' operand.  dgamma/dbeta are accumulated per block
 * into partials [nblocks][2][D]; finish with lr2_colsum_partials_finish.  drop_seed_dev (may be NULL): device uint64 added to drop_seed at run
 * time, as in lr2_epilogue.
 * replaces: autograd of nn.LayerNorm + the in-place residual adds of finetune/xit.py:45-55,77-86, and of the
 * TencentPretrain LayerNorm (tencentpretrain/layers/layer_norm.py:16-21) in the encoder layers. */
int lr2_layernorm_bwd(const void* dy, int group, uint64_t group_stride, const void* x, const void* gamma, const void* mean,
                      const void* rstd, const void* resid_grad, void* dx, void* dxm_hi, uint64_t dxm_lo_off, float drop_p,
                      uint64_t drop_seed, uint32_t drop_site, const void* drop_seed_dev, void* partials, int nblocks, int rows, int D,
                      int mode, float eps, void* stream);
/* out[c] = sum_b partials[b*ld + c] for c < cols (deterministic second stage of column reductions). */
int lr2_colsum_partials_finish(const void* partials, int nblocks, int cols, int ld, void* out, int accumulate,
                               void* stream);
/* Column sums of a [rows, cols] matrix (fp32, or bf16 planes with the lo plane lo_off elements after the hi plane)
 * -> fp32 [cols] (bias gradients); partials: workspace [nblocks][cols].
 * replaces: autograd of the nn.Linear bias add. */
int lr2_colsum(const void* x, int is_planes, uint64_t lo_off, int rows, int cols, int ld, void* partials, int nblocks,
               void* out, void* stream);

/* XiT multi-head attention core, per (sequence b, head h): S = Q K^T (no pre-scale), P = softmax(S) * post_scale,
 * O = P V.  Q/O: [batch*Lq, heads*hd]; K/V: [batch*Lk, heads*hd].  Lq<=256, Lk<=16, hd<=96, hd%4==0.
 * o_planes=1: O is written as bf16 planes (hi at O, lo o_lo_off elements later) for the projection GEMM.
 * replaces: finetune/xit.py:133-146 (einsum, softmax, "/ scaling", einsum). */
int lr2_xattn_fwd(const void* Q, const void* K, const void* V, void* O, int o_planes, uint64_t o_lo_off, int batch,
                  int heads, int Lq, int Lk, int head_dim, float post_scale, void* stream);
/* Backward of lr2_xattn_fwd: recomputes P; writes dQ [batch*Lq, E], dK, dV [batch*Lk, E] (fp32, or bf16 planes when
 * planes=1: lo planes q_lo_off / kv_lo_off elements after the hi planes). */
int lr2_xattn_bwd(const void* Q, const void* K, const void* V, const void* dO, void* dQ, void* dK, void* dV,
                  int planes, uint64_t q_lo_off, uint64_t kv_lo_off, int batch, int heads, int Lq, int Lk, int head_dim,
                  float post_scale, void* stream);

/* TencentPretrain self-attention core, per (sequence, head): S = Q K^T * scale + (seg[key]>0 ? 0 : -10000),
 * P = softmax(S), O = P V -- both products on the matrix cores (split-bf16 x3), softmax in fp32.
 * q_hi / k_hi / v_hi: bf16 hi planes of Q, K, V, element (row, h*64 + d) at ptr[row*ld + h*64 + d], the lo plane lo_off
 * ELEMENTS behind each (one fused QKV matrix [batch*L, 3*heads*64] with ld = 3*heads*64 is the intended producer);
 * seg int64 [batch*L]; o fp32 [batch*L, ld_o] and/or o_hi planes (lo plane o_lo_off elements behind); hd == 64; any L (one
 * LDS-resident key block up to 256 keys, a key-block loop with running max / sum beyond).
 * replaces: tencentpretrain/layers/multi_headed_attn.py:61-74 and the mask of encoders/transformer_encoder.py:62-68. */
int lr2_self_attn_fwd(const void* q_hi, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld, const int64_t* seg,
                      void* o, void* o_hi, uint64_t o_lo_off, int ld_o, void* lse, float drop_p, uint64_t drop_seed,
                      uint32_t drop_site, int batch, int heads, int L, int head_dim, float scale, void* stream);
/* (above) lse: optional fp32 [batch, heads, L] log-sum-exp of the masked scores; drop_p > 0 applies the counter-based
 * dropout to the probabilities (element index ((b*heads + h)*L + q)*L + key, multi_headed_attn.py:72). */

/* Self-attention for the FIRST query row of every sequence only: o[b, h*64 + d] = sum_j softmax_j(q[b] . K[b*L + j] * scale +
 * (seg>0 ? 0 : -10000)) V[b*L + j] -- what pooling 'first' keeps of the last encoder layer (the [CLS] feature of the image
 * encoder), so that layer's output projection and feed-forward run on `batch` rows instead of batch*L.
 * q fp32 [batch, ld_q] (already projected, bias included); k_hi / v_hi bf16 hi planes of K and V over ALL rows, element
 * (row, h*64 + d) at ptr[row*ld + h*64 + d], lo plane lo_off ELEMENTS behind; seg int64 [batch*L]; o fp32 [batch, ld_o];
 * L <= 4096, head_dim == 64.
 * replaces: tencentpretrain/layers/multi_headed_attn.py:61-74 restricted to query 0 (utils/misc.py:23-35 'first' pooling,
 * as consumed at finetune/ppo.py:120-127). */
int lr2_first_token_attn(const void* q, int ld_q, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld,
                         const int64_t* seg, void* o, int ld_o, int batch, int heads, int L, int head_dim, float scale,
                         void* stream);

/* Backward of lr2_self_attn_fwd: from Q, K, V planes and dO planes (same layout rules) to dQ, dK, dV planes
 * (dq_hi / dk_hi / dv_hi: hi planes, element (row, h*64 + d) at ptr[row*ld_d + h*64 + d], lo plane d_lo_off elements behind --
 * three column blocks of one dQKV matrix are the intended target).  Two kernels: dQ per 16-query sub-tile, then dK, dV per 16-key
 * sub-tile; pass the forward's drop_p / seed / site to replay its mask.
 *   o_hi == NULL: everything is recomputed from Q, K, V -- the dQ kernel writes the per-query log-sum-exp and sum_k dP*P into lse_ws /
 *     dsum_ws (fp32 [batch*heads*L] each, scratch).  L > 256: both kernels walk the other dimension in blocks of 128 rows; the dQ
 *     kernel sweeps the keys twice (running max / sum / sum of exp * dP first).
 *   o_hi != NULL (ABI 19): the forward's output planes (element (row, h*64 + d) at o_hi[row*ld_o + h*64 + d], lo plane o_lo_off elements
 *     behind) and, in lse_ws, the log-sum-exp lr2_self_attn_fwd wrote (an INPUT then): P = exp(S - lse) and D = sum_d dO O need no
 *     pass over the keys, both kernels stream over 32-row blocks and -- with at least one (sequence, head) pair per CU, L <= 224 -- run
 *     as persistent 16-wave workgroups whose K / V (Q / dO) planes are refilled by LDS-DMA under the compute (lr2_self_attn_plan
 *     says which form a shape takes).  Shapes the persistent form does not cover fall back to the recomputing kernels, which
 *     OVERWRITE lse_ws with their own (equal up to rounding) values.
 * replaces: autograd of tencentpretrain/layers/multi_headed_attn.py:61-74. */
int lr2_self_attn_bwd(const void* q_hi, const void* k_hi, const void* v_hi, uint64_t lo_off, int ld, const void* do_hi,
                      uint64_t do_lo_off, int ld_do, const int64_t* seg, void* dq_hi, void* dk_hi, void* dv_hi,
                      uint64_t d_lo_off, int ld_d, const void* o_hi, uint64_t o_lo_off, int ld_o, void* lse_ws, void* dsum_ws,
                      float drop_p, uint64_t drop_seed, uint32_t drop_site, int batch, int heads, int L, int head_dim, float scale,
                      void* stream);
/* (ABI 19) Which form of the attention kernels a call of this shape runs on this device: *fwd_persistent / *bwd_persistent = 1 when
 * lr2_self_attn_fwd / lr2_self_attn_bwd (given o_hi) take the persistent kernels (a pure function of the shape, the CU count and the
 * LR2_ATTN_PERSIST switch; either pointer may be NULL).  No reference counterpart: a test hook of this library's own scheduling. */
int lr2_self_attn_plan(int batch, int heads, int L, int ld, int ld_do, int* fwd_persistent, int* bwd_persistent);

/* y[r] = dot(x[row(r)], w) + b for r < rows, row(r) = r*row_step + row_off.
 * replaces: self.head = nn.Linear(768, 1) and the last-position select (finetune/ppo.py:228-232,293-295). */
int lr2_head_fwd(const void* x, const void* w, const void* b, void* y, int rows, int D, int row_step, int row_off,
                 void* stream);
/* dx[total_rows, D] = 0 except dx[row(r)] = dy[r]*w;  dw = sum_r dy[r]*x[row(r)];  db = sum_r dy[r]. */
int lr2_head_bwd(const void* x, const void* w, const void* dy, void* dx, void* dw, void* db, int rows, int D,
                 int row_step, int row_off, int total_rows, void* stream);

/* out[r, :] = x[r, :] + table[r % period, :].  replaces: x + pos_emb (finetune/ppo.py:286-289,339-342). */
int lr2_add_period_rows(const void* x, const void* table, void* out, int rows, int D, int period, void* stream);
/* dtable[t, :] = sum over rows r with r % period == t of dy[r, :]  (t < period; other rows of dtable untouched). */
int lr2_period_rows_grad(const void* dy, void* dtable, int rows, int D, int period, void* stream);

/* Fused PPO losses + analytic gradients for one minibatch (T = tags per item, T <= 8, B <= 1024).
 * Inputs fp32: scores[B,T] (new), old_scores[B,T], rewards[B], old_value[B], value[B]; int64 next_state[B,ns_len]
 * (its last rank_len entries give the actor's descending order; the reference hard-codes 2, ppo.py:565-567).
 * Outputs fp32:
 *   scalars[0]=policy loss, [1]=value loss, [2]=rank loss, [3]=positive-hinge count;
 *   per_item[4][B] = kl, entropy, rewards (r - w_kl*kl), advantages; dscores[B,T]; dvalue[B].
 * Data-parallel form (RankLoss is one scalar over the GLOBAL batch): call once with stats_out (fp32[3]) set -- only
 * {hinge sum, positive-hinge count, sum |A|} of this rank are written -- all-reduce(sum) those three floats over the
 * `world` ranks, then call again with global_stats = the reduced sums: R, 1/count and mean |A| are then global and
 * dscores is scaled so that the rank AVERAGE of the parameter gradients equals the gradient of the global-batch loss.
 * stats_out == NULL and global_stats == NULL: the single-rank form.
 * replaces: finetune/ppo.py:544-584 (KL, entropy, advantage, flipped order, RankLoss :43-55, clipped value
 * loss :494-498) and their autograd backward. */
int lr2_ppo_loss(const void* scores, const void* old_scores, const void* rewards, const void* old_value,
                 const void* value, const int64_t* next_state, int ns_len, int rank_len, int B, int T, float kl_w,
                 float ent_w, float value_clip, float margin, float adv_eps, void* scalars, void* per_item,
                 void* dscores, void* dvalue, void* stats_out, const void* global_stats, int world, void* stream);

/* mode = 'cls' (C classes, C <= 8): the 768 -> C head, y[r, c] = x[r, :] . w[c, :] + b[c], and its backward
 * (dx[r, :] = sum_c dy[r, c] w[c, :]; dw[c, :] = sum_r dy[r, c] x[r, :]; db[c] = sum_r dy[r, c]; dx or dw/db may be NULL).
 * replaces: nn.Linear(768, labels_num) of finetune/ppo.py:209-210,228-230 and its autograd. */
int lr2_cls_head_fwd(const void* x, const void* w, const void* b, void* y, int rows, int D, int C, void* stream);
int lr2_cls_head_bwd(const void* x, const void* w, const void* dy, void* dx, void* dw, void* db, int rows, int D, int C,
                     void* stream);
/* probs[r, :] = softmax(logits[r, :]) (use_softmax = 1) or logits[r, :] (0: evaluate(), ppo.py:641-643); scores[r] =
 * sum_k k * probs[r, k] (the expected label the PPO loop ranks by); probs may be NULL.
 * replaces: finetune/ppo.py:532-537 / :859-863 / :641-643. */
int lr2_cls_scores(const void* logits, void* probs, void* scores, int rows, int C, int use_softmax, void* stream);
/* dlogits[r, c] = dscores[r] * probs[r, c] * (c - scores[r]): autograd of the softmax form of lr2_cls_scores. */
int lr2_cls_scores_bwd(const void* probs, const void* scores, const void* dscores, void* dlogits, int rows, int C, void* stream);
/* loss[0] = mean_r (logsumexp(logits[r, :]) - logits[r, tgts[r]]); dlogits (may be NULL) = its gradient.
 * replaces: nn.NLLLoss()(nn.LogSoftmax(dim=-1)(logits), tgts) of finetune/ppo.py:239-241 and its autograd. */
int lr2_nll_loss(const void* logits, const int64_t* tgts, int rows, int C, void* loss, void* dlogits, void* stream);

/* SmoothL1(beta) mean loss + gradient (dpred may be NULL).  replaces: nn.SmoothL1Loss(beta=0.3) (finetune/ppo.py:236). */
int lr2_smooth_l1(const void* pred, const void* target, int n, float beta, void* loss, void* dpred, void* stream);

/* Pairwise hinge of the stage-2 reward training: scores = [chosen(bs) ; reject(bs)];
 * loss_acc[0] = mean relu(margin - (chosen - reject)), loss_acc[1] = mean (chosen > reject); dscores (may be NULL) = d loss / d scores.
 * replaces: finetune/reward_pair_dataloader.py:356-359 (+ the autograd of that expression). */
int lr2_pair_hinge(const void* scores, int bs, float margin, void* loss_acc, void* dscores, void* stream);

/* One chunk of the multi-tensor AdamW: `count` fp32 elements starting at p/g/m/v. */
typedef struct lr2_adamw_chunk {
  void* p;
  const void* g;
  void* m;
  void* v;
  uint64_t count;
  float weight_decay;
  float _pad;
} lr2_adamw_chunk;
/* table: DEVICE array of n_chunks lr2_adamw_chunk (one workgroup per chunk).  Update (correct_bias=False):
 *   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr * m/(sqrt(v)+eps); then p -= lr*wd*p.
 * replaces: tencentpretrain/utils/optimizers.py:344-402 (AdamW.step), including its decay-after-update order. */
int lr2_adamw_multi(const lr2_adamw_chunk* table_dev, int n_chunks, double lr, double beta1, double beta2,
                    double eps, const void* lr_dev, void* stream);
/* lr_dev (may be NULL): device float read by the kernel in place of `lr` -- a captured HIP graph of a training step then follows
 * the scheduler without re-capture.
 *
 * The per-step scalars such a graph reads -- dst_dev[0..7] = seed (uint64, the dropout seed of lr2_epilogue.drop_seed_dev /
 * lr2_layernorm_bwd), dst_dev[8 + 4 i ..] = lrs[i] (float, i < n_lrs <= 14) -- are written by ONE small kernel whose arguments
 * carry the values: no host buffer has to outlive the call, so steps can be enqueued ahead of the device. */
#define LR2_STEP_SCALARS_BYTES 64
#define LR2_STEP_SCALARS_MAX_LRS 14
int lr2_step_scalars_store(void* dst_dev, uint64_t seed, const float* lrs_host, int n_lrs, void* stream);

/* Tokens -> embeddings: out[r,:] = word[src[r]] + pos[r % L] + seg_table[seg[r]].  Ids outside [0, vocab) / [0, n_seg) never
 * index memory: the row is taken from index 0 and *err_flag (device int, may be NULL) gets bit 0 (token) / bit 1 (segment)
 * OR-ed in, for the host to raise the IndexError nn.Embedding would.
 * replaces: tencentpretrain/embeddings/{word,pos,seg}_embedding.py forward sums (embedding.py:19-30). */
int lr2_text_embed(const int64_t* src, const int64_t* seg, const void* word, const void* pos, const void* seg_table,
                   void* out, int rows, int L, int D, int64_t vocab, int n_seg, int* err_flag, void* stream);
/* Backward of lr2_text_embed's word / segment gathers (the position table's gradient is lr2_period_rows_grad), deterministic:
 * `order` = stable argsort of the token ids, `sorted_ids` = ids in that order; dword[tok, :] = sum of dx rows carrying tok (rows of
 * dword for PRESENT tokens are overwritten, rows for absent tokens are NOT written: zero the table first).  A run of equal ids is
 * summed in original row order inside pieces of LR2_TEXT_EMBED_BWD_WORD_SEG sorted positions, the pieces of a long run (the padding
 * id) in piece order through `word_partials` (fp32 scratch, 2 * ceil(rows / LR2_TEXT_EMBED_BWD_WORD_SEG) * D floats) -- a fixed
 * order, and no single workgroup walks a long run alone.  dseg[s, :] = sum of dx rows with segment s via `seg_partials` (fp32
 * scratch, ceil(rows / LR2_TEXT_EMBED_BWD_ROWS_PER_BLOCK) * n_seg * D floats).  No atomics.
 * replaces: autograd of nn.Embedding in tencentpretrain/embeddings/{word,seg}_embedding.py. */
#define LR2_TEXT_EMBED_BWD_ROWS_PER_BLOCK 64
#define LR2_TEXT_EMBED_BWD_WORD_SEG 256
int lr2_text_embed_bwd(const void* dx, const int64_t* sorted_ids, const int64_t* order, const int64_t* seg, void* dword,
                       void* dseg, void* seg_partials, void* word_partials, int rows, int D, int64_t vocab, int n_seg,
                       void* stream);
/* dst = dropout_mask(src) / (1 - p), fp32, mask element index = flat element index (src == dst allowed).
 * replaces: self.dropout of tencentpretrain/embeddings/embedding.py:33 and its autograd. */
int lr2_dropout_apply(const void* src, void* dst, uint64_t n, float drop_p, uint64_t drop_seed, uint32_t drop_site,
                      void* stream);
/* Image -> patch rows: out[(b*P + p), c*ps*ps + i*ps + j] = img[b, c, py*ps+i, px*ps+j].
 * replaces: the unfold implied by nn.Conv2d(k=s=patch) in tencentpretrain/embeddings/patch_embedding.py:18,27. */
int lr2_patchify(const void* img, void* out, int B, int C, int H, int W, int ps, void* stream);
/* Image -> patch rows as bf16 hi/lo planes (the A operand of the patch-projection GEMM); element layout of lr2_patchify with row
 * stride ld >= C*ps*ps (columns past C*ps*ps are written as zeros: ViT-L/14's 588 -> 640 so the GEMM sees whole K tiles); ps even.
 * img: fp32 [B,C,H,W] (is_u8 = 0) or uint8 frames [B,C,H,W] (is_u8 = 1); with is_u8 and mean3/std3 (HOST float[3]) given, pixels
 * are normalised on the fly: (x / 255 - mean[c]) / std[c].
 * replaces: ZeroOneNormalize + transforms.Normalize (tencentpretrain/utils/dataloader.py:559-561) + the unfold implied by
 * nn.Conv2d(k=s=patch) (embeddings/patch_embedding.py:18,27). */
int lr2_patchify_planes(const void* img, int is_u8, void* out_hi, uint64_t lo_off, int ld, int B, int C, int H, int W, int ps,
                        const float* mean3, const float* std3, void* stream);
/* NDCG@ks[q] per ragged item i (elements offsets[i] .. offsets[i+1], at most 64): sort by score descending (stable), gain
 * 2^rel - 1, discount disc[j] (= log2(j + 2), device fp32 table supplied by the caller), sequential fp32 sums, 1 when the ideal
 * DCG <= 1e-6.  out: fp32 [n_items, n_k].  An item with more than 64 elements, or with a label outside [0, 62] (2^rel - 1 in
 * int64), gets a row of NaN -- never a truncated or wrapped value (the entry point cannot see device-side sizes without a sync).
 * replaces: finetune/ppo.py:651-659 + ndcg.py:28-65 (AverageNDCGMeter.return_ndcg_at_k). */
int lr2_ndcg(const void* scores, const int64_t* gold, const int64_t* offsets, const void* disc, const int64_t* ks, int n_k,
             void* out, int n_items, void* stream);
/* ViT embedding assembly: out[b,0,:] = cls + pos[0]; out[b,1+p,:] = patch_proj[b*P+p,:] + pos[1+p].
 * replaces: patch_embedding.py:28-29 (cls concat) + pos_embedding.py:30-35 + embedding.py:27-30. */
int lr2_vit_assemble(const void* patch_proj, const void* cls, const void* pos, void* out, int B, int P, int D,
                     void* stream);

/* MX-FP8 products on gfx950's block-scaled matrix instruction (BASELINE.json configs[4]: "fp8 MFMA"; an inference-only fast mode,
 * NOT the parity path: e4m3 keeps 3 mantissa bits).  Format: OCP MX v1.0 -- e4m3fn elements, one E8M0 scale byte (2^(s - 127)) per 32
 * consecutive elements of a row, shared exponent floor(log2(amax)) - 8, round-to-nearest-even, saturating at +-448.
 *   lr2_quant_mxfp8: x fp32 [rows, K] (row stride ldx) -> q uint8 [rows, K], scales uint8 [rows, K / 32].  K % 32 == 0.
 *   lr2_gemm_mxfp8 : out[M, N] fp32 = A_q . B_q^T (+ bias[N]) (act 1: GELU) (+ resid[M, ld_resid]); A_q [M, K], B_q [N, K] and their
 *                    scales as written by lr2_quant_mxfp8.  N % 128 == 0, K % 128 == 0, any M.  out_q / out_scales (both or neither;
 *                    out may then be NULL): the result also / instead quantised to MX-FP8 [M, N] + [M, N / 32] by the same rule --
 *                    the A operand of the next product without an fp32 round trip.  out_hi / out_lo_off / ld_planes: the result also /
 *                    instead as bf16 hi / lo planes (the operand format of lr2_self_attn_fwd and of the split-bf16 products);
 *                    out_lo_off == 0 (ABI 18): ONE bf16 plane (round to nearest even), the operand format of lr2_self_attn_fwd_bf16.
 * replaces: nn.Linear forward (tencentpretrain/layers/position_ffn.py:12-15, multi_headed_attn.py:55-76) in that mode. */
int lr2_quant_mxfp8(const void* x, int ldx, void* q, void* scales, int rows, int K, void* stream);
/* LayerNorm (lr2_layernorm_fwd's semantics, modes 0 / 1) whose result leaves as MX-FP8 [rows, D] + [rows, D / 32] (and as fp32 when out is
 * given): the A operand of the projection that follows, without an fp32 round trip.  D % 32 == 0. */
int lr2_layernorm_fwd_mxfp8(const void* x, const void* gamma, const void* beta, void* out, void* out_q, void* out_scales, int rows,
                            int D, float eps, int mode, void* stream);
int lr2_gemm_mxfp8(const void* a_q, const void* a_scales, const void* b_q, const void* b_scales, void* out, int ld_out,
                   const void* bias, const void* resid, int ld_resid, int act, void* out_q, void* out_scales, void* out_hi,
                   uint64_t out_lo_off, int ld_planes, int M, int N, int K, void* stream);
/* Round 4 (ABI 18).  Encoder self-attention of the MX-FP8 mode: q / k / v are ONE bf16 plane each (row stride ld elements; what
 * lr2_gemm_mxfp8 writes with out_lo_off = 0 -- q, k, v = the three column blocks of one [rows, 3E] matrix), single-pass bf16 products,
 * fp32 softmax, key mask -10000 * (seg <= 0) after the scale; the context rows leave as fp32 (o_f32 [batch * L, ld_o]) and / or as
 * MX-FP8 (o_q [batch * L, ld_o] bytes + o_scales [batch * L, ld_o / 32]: the A operand of the output projection, no fp32 round trip, no
 * quantise pass).  head_dim 64, L <= 288, inference only (no dropout).  NOT the parity path: lr2_self_attn_fwd stays the default.
 * replaces: tencentpretrain/layers/multi_headed_attn.py:60-74 in that mode. */
int lr2_self_attn_fwd_bf16(const void* q, const void* k, const void* v, int ld, const int64_t* seg, void* o_f32, void* o_q,
                           void* o_scales, int ld_o, int batch, int heads, int L, int head_dim, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LR2PPO_HIP_H */
