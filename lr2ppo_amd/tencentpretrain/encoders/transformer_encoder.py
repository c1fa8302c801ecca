"""TransformerEncoder.forward(emb, seg) -> hidden on the gfx950 kernels (inference schedule).

Mirrors the reference's encoders/transformer_encoder.py:7-138 + layers/transformer.py:50-73 for
mask="fully_visible": additive key mask -10000 * (seg <= 0) applied AFTER the 1/sqrt(64) scale, post-LN
(RoBERTa-base) or pre-LN + final LayerNorm (ViT-B/16), exact-erf GELU, TencentPretrain LayerNorm semantics.

Per layer: ONE fused QKV GEMM (N = 3*hidden, weights concatenated once), the MFMA self-attention kernel, the output
projection (+residual), FFN1 (+GELU), FFN2 (+residual) and 2 wavefront LayerNorms.  Everything a GEMM consumes travels as
bf16 hi/lo planes written by the producing kernel (LayerNorm, attention, GEMM epilogue) and is streamed by LDS-DMA; the
residual stream stays fp32.  Weight planes are split once and re-split only when a parameter changes.
With autograd enabled (train mode or parameters that require grad) the same kernels run a saving schedule and a
hand-written backward (`_EncoderFn`): MFMA attention backward with recomputed probabilities, TencentPretrain-LayerNorm
backward, dgrad / wgrad GEMMs with fused GELU' and residual epilogues, dropout at the reference's three sites per layer
(attention probabilities, dropout_1, dropout_2) from the counter-based mask stream of lr2ppo_amd.runtime."""
import math
import os

import torch
import torch.nn as nn

from ... import engine, ops, runtime
from ..layers.layer_norm import LayerNorm
from ..layers.transformer import TransformerLayer


class _LayerPlanes(dict):
    """One layer's GEMM operands; `wqkv_f32`, `wqkv_t`, `w1_t` are produced on first access after a refresh."""

    def __missing__(self, key):
        if key == "wqkv_f32":
            v = torch.cat([lin.weight.data for lin in self._qkv], dim=0, out=self._f32)
        elif key == "wqkv_t":
            v = ops.split_planes_t(self["wqkv_f32"], self._t_qkv)
        elif key == "w1_t":
            v = ops.split_planes_t(self._w1.data.contiguous(), self._t_w1)
        else:
            raise KeyError(key)
        self[key] = v
        return v


class _StackPlanes:
    """bf16 hi / lo planes of every GEMM weight of an encoder stack in one buffer + the device table that re-splits them from the
    fp32 parameters in one launch (lr2_split_planes_multi).  Wq, Wk, Wv are split straight into the row blocks of one [3E, E] planes
    matrix (hi plane of all three, then lo plane): no fp32 concatenation on the forward path."""
    CHUNK = 1 << 16
    LAZY = ("wqkv_f32", "wqkv_t", "w1_t")

    def __init__(self, enc, dev):
        layers = list(enc.transformer)
        E = layers[0].self_attn.final_linear.out_features
        F = layers[0].feed_forward.linear_1.out_features
        per = 3 * E * E + E * E + 2 * E * F
        self.dev = dev
        self.buf = torch.empty(2 * per * len(layers), dtype=torch.int16, device=dev)
        self.ptrs = tuple(p.data_ptr() for p in enc.parameters())
        self.layers, rows, cur = [], [], 0
        base = self.buf.data_ptr()

        def add(sources, nrows, ncols):
            nonlocal cur
            n = nrows * ncols
            pl = ops.Planes(self.buf[cur:cur + 2 * n], nrows, ncols)
            r0 = 0
            for w in sources:
                if w.dim() != 2 or w.shape[1] != ncols or not w.is_contiguous() or w.numel() % 4:
                    raise ValueError("encoder GEMM weights must be contiguous 2-D fp32 tensors")
                k, o = w.numel(), 0
                while o < k:
                    c = min(self.CHUNK, k - o)
                    rows.append((w.data_ptr() + 4 * o, base + 2 * (cur + r0 * ncols + o), n, c))
                    o += c
                r0 += w.shape[0]
            cur += 2 * n
            return pl

        for layer in layers:
            att, ffn = layer.self_attn, layer.feed_forward
            ent = _LayerPlanes()
            ent["wqkv"] = add([att.linear_layers[i].weight.data for i in range(3)], 3 * E, E)
            ent["wo"] = add([att.final_linear.weight.data], E, E)
            ent["w1"] = add([ffn.linear_1.weight.data], F, E)
            ent["w2"] = add([ffn.linear_2.weight.data], E, F)
            ent["bqkv"] = torch.empty(3 * E, device=dev)
            ent._qkv, ent._w1 = [att.linear_layers[i] for i in range(3)], ffn.linear_1.weight
            ent._f32 = torch.empty(3 * E, E, device=dev)
            ent._t_qkv, ent._t_w1 = ops.Planes.empty(E, 3 * E, dev), ops.Planes.empty(E, F, dev)
            self.layers.append(ent)
        import ctypes as C  # noqa: F401
        from ... import _native
        arr = (_native.SplitChunk * len(rows))()
        for i, (src, dst, lo, cnt) in enumerate(rows):
            arr[i].src, arr[i].dst_hi, arr[i].lo_off, arr[i].count = src, dst, lo, cnt
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.n_chunks = len(rows)
        self.sig = None

    @staticmethod
    def signature(enc):
        return tuple((p._version, ops.param_write_count(p)) for p in enc.parameters())

    def matches(self, enc, dev) -> bool:
        return self.dev == dev and self.ptrs == tuple(p.data_ptr() for p in enc.parameters())

    def refresh(self, enc):
        ops.split_planes_multi(self.table, self.n_chunks)
        for ent in self.layers:
            for k in self.LAZY:
                ent.pop(k, None)
            torch.cat([lin.bias.data for lin in ent._qkv], dim=0, out=ent["bqkv"])
        self.sig = self.signature(enc)


class TransformerEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.mask = args.mask
        self.layers_num = args.layers_num
        self.layernorm_positioning = args.layernorm_positioning
        self.heads_num, self.hidden_size = args.heads_num, args.hidden_size
        unsupported = [k for k in ("parameter_sharing", "factorized_embedding_parameterization", "relative_position_embedding",
                                   "has_residual_attention") if getattr(args, k, False)]
        if unsupported or self.mask != "fully_visible":
            raise NotImplementedError(f"HIP TransformerEncoder: unsupported options {unsupported or self.mask}")
        self.transformer = nn.ModuleList([TransformerLayer(args) for _ in range(self.layers_num)])
        self.final_layernorm = self.layernorm_positioning == "pre"       # transformer_encoder.py:36-37,134-135 upstream
        if self.final_layernorm:
            self.layer_norm = LayerNorm(args.hidden_size)
        self._ws = None
        self._wplanes = None

    def _weight_planes(self, dev, cache=True):
        """Per layer a dict: Wqkv planes [3E, E] (rows Q | K | V), bqkv [3E], Wo, W1, W2 planes -- and, computed on first use after
        a refresh, the fp32 concatenation `wqkv_f32` (large-M input gradients transpose it) and the transposed planes `wqkv_t` /
        `w1_t` (the NN form of the forward on small batches).  All planes of the stack live in ONE buffer and are re-split from the
        fp32 parameters by ONE multi-tensor launch (round 3; rounds 1-2: 2 concatenations + 6 allocations + 6 split launches per layer
        and forward); re-split when any parameter was written (torch's version counters + the per-parameter write counters the HIP
        optimizer bumps: its kernels write through raw pointers).  cache=False (training): always re-split."""
        sp = self._wplanes
        if sp is None or not sp.matches(self, dev):
            sp = self._wplanes = _StackPlanes(self, dev)
            fresh = False
        else:
            fresh = cache and sp.sig == sp.signature(self)
        if not fresh:
            sp.refresh(self)
        return sp.layers

    def forward(self, emb, seg):
        if emb.dtype != torch.float32 or not emb.is_cuda:
            raise TypeError("lr2ppo_amd: emb must be a float32 tensor on the HIP device (no CPU path)")
        needs_grad = torch.is_grad_enabled() and (emb.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad:
            return _EncoderFn.apply(self, emb, seg, *list(self.parameters()))
        if self.training and any(l.dropout_1.p > 0 for l in self.transformer):
            out, _ = self._forward_train(emb, seg)       # dropout without a graph (torch.no_grad() in train mode)
            return out
        return self._forward_infer(emb, seg)

    def forward_first_token(self, emb, seg):
        """hidden[:, 0, :] of forward(emb, seg), [batch, hidden] -- for callers that pool with 'first' (utils/misc.py:23-35
        upstream; the image encoder in front of the heads, finetune/ppo.py:120-127).  In inference the LAST layer then needs
        its keys and values for every row but its query, output projection and feed-forward for row 0 only: 5/6 of that
        layer's matrix work is never consumed and is not computed.  With gradients enabled: the full forward, sliced."""
        needs_grad = torch.is_grad_enabled() and (emb.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad or (self.training and any(l.dropout_1.p > 0 for l in self.transformer)):
            return self.forward(emb, seg)[:, 0, :]
        if emb.dtype != torch.float32 or not emb.is_cuda:
            raise TypeError("lr2ppo_amd: emb must be a float32 tensor on the HIP device (no CPU path)")
        return self._forward_infer(emb, seg, first_only=True)

    @torch.no_grad()
    def _forward_infer(self, emb, seg, first_only=False):
        B, L, E = emb.shape
        H, hd = self.heads_num, E // self.heads_num
        M = B * L
        if self._ws is None or self._ws.device != emb.device:
            self._ws = engine.Workspace(emb.device)
        ws = self._ws
        seg = seg.to(device=emb.device, dtype=torch.int64).contiguous().view(-1)
        W = self._weight_planes(emb.device)
        h, h2 = ws.mat("h", M, E), ws.mat("h2", M, E)
        h.copy_(emb.contiguous().view(M, E))
        pre = self.layernorm_positioning == "pre"
        F = self.transformer[0].feed_forward.linear_1.out_features
        x_p, t_p = ws.planes("x_p", M, E), ws.planes("t_p", M, E)        # GEMM inputs: LN outputs / hidden
        qkv_p, o_p, ff_p = ws.planes("qkv_p", M, 3 * E), ws.planes("o_p", M, E), ws.planes("ff_p", M, F)
        scale = 1.0 / math.sqrt(float(hd))
        # wide outputs: NT on the weight's own layout when the 256 x 256 kernel takes the product, else NN on W^T
        big_qkv, big_ff = ops.use_gemm256(M, 3 * E, E), ops.use_gemm256(M, F, E)
        if not pre:
            ops.split_planes(h, x_p)
        for li, (layer, w) in enumerate(zip(self.transformer, W)):
            att, ffn = layer.self_attn, layer.feed_forward
            ln1, ln2 = layer.layer_norm_1, layer.layer_norm_2
            if pre:                                                   # layers/transformer.py:63-73
                ops.layernorm_fwd(h, ln1.gamma.data, ln1.beta.data, None, rows=M, D=E, eps=ln1.eps, mode=1, out_planes=x_p)
            if first_only and li == self.layers_num - 1:
                return self._last_layer_first_token(ws, layer, w, h, x_p, seg, B, L, E, F)
            engine.linear_fwd(ws, x_p, w["wqkv" if big_qkv else "wqkv_t"], w["bqkv"], None, M, 3 * E, E, out_planes=qkv_p)
            ops.self_attn_fwd(qkv_p, seg, o_p, batch=B, heads=H, L=L, head_dim=hd, scale=scale)
            engine.linear_fwd(ws, o_p, w["wo"], att.final_linear.bias.data, h2, M, E, E, resid=h)
            if pre:
                ops.layernorm_fwd(h2, ln2.gamma.data, ln2.beta.data, None, rows=M, D=E, eps=ln2.eps, mode=1, out_planes=t_p)
                engine.linear_fwd(ws, t_p, w["w1" if big_ff else "w1_t"], ffn.linear_1.bias.data, None, M, F, E, act=1, out_planes=ff_p)
                engine.linear_fwd(ws, ff_p, w["w2"], ffn.linear_2.bias.data, h, M, E, F, resid=h2)
            else:                                                     # layers/transformer.py:54-61
                ops.layernorm_fwd(h2, ln1.gamma.data, ln1.beta.data, h, rows=M, D=E, eps=ln1.eps, mode=1, out_planes=t_p)
                engine.linear_fwd(ws, t_p, w["w1" if big_ff else "w1_t"], ffn.linear_1.bias.data, None, M, F, E, act=1, out_planes=ff_p)
                engine.linear_fwd(ws, ff_p, w["w2"], ffn.linear_2.bias.data, h2, M, E, F, resid=h)
                ops.layernorm_fwd(h2, ln2.gamma.data, ln2.beta.data, h, rows=M, D=E, eps=ln2.eps, mode=1, out_planes=x_p)
        out = torch.empty(B, L, E, device=emb.device)
        if self.final_layernorm:
            ops.layernorm_fwd(h, self.layer_norm.gamma.data, self.layer_norm.beta.data, out.view(M, E), rows=M, D=E,
                              eps=self.layer_norm.eps, mode=1)
        else:
            out.view(M, E).copy_(h)
        return out

    # ---- MX-FP8 inference (BASELINE.json configs[4] "fp8 MFMA"): NOT the parity path, an explicit fast mode ------------------
    def _fp8_weights(self):
        """Per layer: the four projection weights quantised to MX-FP8 once (ops.quant_mxfp8), re-done when a parameter is rewritten."""
        sig = tuple((p.data_ptr(), p._version, ops.param_write_count(p)) for p in self.parameters())
        if getattr(self, "_fp8_sig", None) != sig:
            out = []
            for layer in self.transformer:
                att, ffn = layer.self_attn, layer.feed_forward
                wqkv = ops.quant_mxfp8(torch.cat([l.weight.data for l in att.linear_layers], 0))
                E = wqkv.cols                                          # rows [E, 3E) = [Wk; Wv]: the pruned last layer's product
                wkv = ops.Mx8(wqkv.q[E * E:], wqkv.s[E * (E // 32):], 2 * E, E)
                out.append({"wqkv": wqkv, "wkv": wkv, "bqkv": torch.cat([l.bias.data for l in att.linear_layers], 0),
                            "wo": ops.quant_mxfp8(att.final_linear.weight.data), "w1": ops.quant_mxfp8(ffn.linear_1.weight.data),
                            "w2": ops.quant_mxfp8(ffn.linear_2.weight.data)})
            self._fp8_w, self._fp8_sig = out, sig
        return self._fp8_w

    @torch.no_grad()
    def forward_fp8(self, emb, seg, first_only: bool = False):
        """forward(emb, seg) with the four projections of every layer as MX-FP8 products (csrc/fp8.hip); LayerNorm, attention and the
        residual stream stay fp32 / split-bf16.  An element keeps 3 mantissa bits: expect the output a few per cent away from
        forward()'s -- the throughput mode of BASELINE.json configs[4] (FeatureExtractor(precision="mxfp8")), never the parity path.
        first_only: -> hidden[:, 0, :] ([batch, hidden]); the last layer then computes keys / values for every row (MX-FP8) and
        everything behind the scores for row 0 only (forward_first_token's schedule, B rows: split-bf16)."""
        if emb.dtype != torch.float32 or not emb.is_cuda:
            raise TypeError("lr2ppo_amd: emb must be a float32 tensor on the HIP device (no CPU path)")
        B, L, E = emb.shape
        H, hd, M = self.heads_num, E // self.heads_num, B * L
        F = self.transformer[0].feed_forward.linear_1.out_features
        if E % 128 or F % 128:
            raise ValueError("forward_fp8: hidden and feed-forward widths must be multiples of 128")
        if self._ws is None or self._ws.device != emb.device:
            self._ws = engine.Workspace(emb.device)
        ws, dev = self._ws, emb.device
        seg = seg.to(device=dev, dtype=torch.int64).contiguous().view(-1)
        W = self._fp8_weights()
        pre = self.layernorm_positioning == "pre"
        h, h2, xn = ws.mat("h", M, E), ws.mat("h2", M, E), ws.mat("fp8_xn", M, E)
        o32 = ws.mat("fp8_o", M, E)
        qkv_p = ws.planes("qkv_p", M, 3 * E)
        fast_attn = hd == 64 and L <= 288 and os.environ.get("LR2_FP8_ATTN", "1") != "0"     # (longer sequences: the 3-pass kernels)
        qkv_b = qkv_p.buf[:M * 3 * E]
        if getattr(self, "_fp8_act", None) is None or self._fp8_act[0].rows != M:
            self._fp8_act = (ops.Mx8.empty(M, E, dev), ops.Mx8.empty(M, F, dev))
        x_q, ff_q = self._fp8_act
        h.copy_(emb.contiguous().view(M, E))
        scale = 1.0 / math.sqrt(float(hd))
        for li, (layer, w) in enumerate(zip(self.transformer, W)):
            att, ffn, ln1, ln2 = layer.self_attn, layer.feed_forward, layer.layer_norm_1, layer.layer_norm_2
            if pre:
                ops.layernorm_fwd_mxfp8(h, ln1.gamma.data, ln1.beta.data, x_q, rows=M, D=E, eps=ln1.eps, mode=1)
            else:
                ops.quant_mxfp8(h, x_q)
            if first_only and li == self.layers_num - 1:
                kv_p = ws.planes("kv_p", M, 2 * E)
                ops.gemm_mxfp8(x_q, w["wkv"], None, bias=w["bqkv"][E:], out_planes=kv_p)
                return self._last_layer_first_token(ws, layer, self._weight_planes(dev)[li], h, None, seg, B, L, E, F, kv_p=kv_p)
            if fast_attn:
                # Q | K | V as ONE bf16 plane, single-pass attention, context straight to MX-FP8 (csrc/selfattn_mx.hip)
                ops.gemm_mxfp8(x_q, w["wqkv"], None, bias=w["bqkv"], out_bf16=qkv_b)
                ops.self_attn_fwd_bf16(qkv_b, seg, batch=B, heads=H, L=L, head_dim=hd, scale=scale, out_mx=x_q)
            else:
                ops.gemm_mxfp8(x_q, w["wqkv"], None, bias=w["bqkv"], out_planes=qkv_p)     # the 3-pass attention kernels take bf16 hi / lo planes
                ops.self_attn_fwd(qkv_p, seg, o32, batch=B, heads=H, L=L, head_dim=hd, scale=scale)
                ops.quant_mxfp8(o32, x_q)
            ops.gemm_mxfp8(x_q, w["wo"], h2, bias=att.final_linear.bias.data, resid=h)
            if pre:                                                   # layers/transformer.py:63-73
                ops.layernorm_fwd_mxfp8(h2, ln2.gamma.data, ln2.beta.data, x_q, rows=M, D=E, eps=ln2.eps, mode=1)
                ops.gemm_mxfp8(x_q, w["w1"], None, bias=ffn.linear_1.bias.data, act=1, out_mx=ff_q)
                ops.gemm_mxfp8(ff_q, w["w2"], h, bias=ffn.linear_2.bias.data, resid=h2)
            else:                                                     # layers/transformer.py:54-61
                ops.layernorm_fwd_mxfp8(h2, ln1.gamma.data, ln1.beta.data, x_q, xn, rows=M, D=E, eps=ln1.eps, mode=1)
                ops.gemm_mxfp8(x_q, w["w1"], None, bias=ffn.linear_1.bias.data, act=1, out_mx=ff_q)
                ops.gemm_mxfp8(ff_q, w["w2"], h2, bias=ffn.linear_2.bias.data, resid=xn)
                ops.layernorm_fwd(h2, ln2.gamma.data, ln2.beta.data, h, rows=M, D=E, eps=ln2.eps, mode=1)
        out = torch.empty(B, L, E, device=dev)
        if self.final_layernorm:
            ops.layernorm_fwd(h, self.layer_norm.gamma.data, self.layer_norm.beta.data, out.view(M, E), rows=M, D=E,
                              eps=self.layer_norm.eps, mode=1)
        else:
            out.view(M, E).copy_(h)
        return out

    def _last_layer_first_token(self, ws, layer, w, h, x_p, seg, B, L, E, F, kv_p=None):
        """Last layer of the inference schedule for row 0 of every sequence.  h: the layer's input [B*L, E] (fp32); x_p: the
        planes its QKV projection reads (LayerNorm_1(h) for 'pre', h itself for 'post').  K, V: all rows; everything after the
        scores: B rows.  Same kernels and epilogues as the full schedule (the row-0 GEMMs run at M = B).
        kv_p: the [K | V] planes already computed by the caller (forward_fp8); x_p is then unused."""
        att, ffn, ln1, ln2 = layer.self_attn, layer.feed_forward, layer.layer_norm_1, layer.layer_norm_2
        H, hd, M = self.heads_num, E // self.heads_num, B * L
        pre = self.layernorm_positioning == "pre"
        wqkv, dev = w["wqkv"], h.device
        w_q = ops.Planes(wqkv.buf, E, E, lo_off=wqkv.lo_off)                       # rows [0, E) of [Wq; Wk; Wv]
        w_kv = ops.Planes(wqkv.buf[E * E:], 2 * E, E, lo_off=wqkv.lo_off)          # rows [E, 3E)
        if kv_p is None:
            kv_p = ws.planes("kv_p", M, 2 * E)
            engine.linear_fwd(ws, x_p, w_kv, w["bqkv"][E:], None, M, 2 * E, E, out_planes=kv_p)
        h0 = h.view(B, L, E)[:, 0, :].contiguous()                                  # the layer's input at row 0
        x0_p, q0, o0, o0_p = ws.planes("x0_p", B, E), ws.mat("q0", B, E), ws.mat("o0", B, E), ws.planes("o0_p", B, E)
        if pre:
            ops.layernorm_fwd(h0, ln1.gamma.data, ln1.beta.data, None, rows=B, D=E, eps=ln1.eps, mode=1, out_planes=x0_p)
        else:
            ops.split_planes(h0, x0_p)
        engine.linear_fwd(ws, x0_p, w_q, w["bqkv"][:E], q0, B, E, E)
        ops.first_token_attn(q0, kv_p, seg, o0, batch=B, heads=H, L=L, head_dim=hd, scale=1.0 / math.sqrt(float(hd)))
        ops.split_planes(o0, o0_p)
        a0, t0_p, ff0_p, y0 = ws.mat("a0", B, E), ws.planes("t0_p", B, E), ws.planes("ff0_p", B, F), ws.mat("y0", B, E)
        out = torch.empty(B, E, device=dev)
        engine.linear_fwd(ws, o0_p, w["wo"], att.final_linear.bias.data, a0, B, E, E, resid=h0)
        if pre:                                                       # layers/transformer.py:63-73
            ops.layernorm_fwd(a0, ln2.gamma.data, ln2.beta.data, None, rows=B, D=E, eps=ln2.eps, mode=1, out_planes=t0_p)
            engine.linear_fwd(ws, t0_p, w["w1"], ffn.linear_1.bias.data, None, B, F, E, act=1, out_planes=ff0_p)
            engine.linear_fwd(ws, ff0_p, w["w2"], ffn.linear_2.bias.data, y0, B, E, F, resid=a0)
            if self.final_layernorm:
                ops.layernorm_fwd(y0, self.layer_norm.gamma.data, self.layer_norm.beta.data, out, rows=B, D=E,
                                  eps=self.layer_norm.eps, mode=1)
            else:
                out.copy_(y0)
        else:                                                         # layers/transformer.py:54-61
            i0 = ws.mat("i0", B, E)
            ops.layernorm_fwd(a0, ln1.gamma.data, ln1.beta.data, i0, rows=B, D=E, eps=ln1.eps, mode=1, out_planes=t0_p)
            engine.linear_fwd(ws, t0_p, w["w1"], ffn.linear_1.bias.data, None, B, F, E, act=1, out_planes=ff0_p)
            engine.linear_fwd(ws, ff0_p, w["w2"], ffn.linear_2.bias.data, y0, B, E, F, resid=i0)
            ops.layernorm_fwd(y0, ln2.gamma.data, ln2.beta.data, out, rows=B, D=E, eps=ln2.eps, mode=1)
        return out

    # ---- training schedule: same kernels, activations kept for the backward -------------------------------------
    def _dims(self, emb):
        B, L, E = emb.shape
        return B, L, E, self.heads_num, E // self.heads_num, B * L, self.transformer[0].feed_forward.linear_1.out_features

    @torch.no_grad()
    def _forward_train(self, emb, seg):
        B, L, E, H, hd, M, F = self._dims(emb)
        dev = emb.device
        if self._ws is None or self._ws.device != dev:
            self._ws = engine.Workspace(dev)
        ws = self._ws                                                   # split-K / reduction scratch only
        seg = seg.to(device=dev, dtype=torch.int64).contiguous().view(-1)
        W = self._weight_planes(dev, cache=False)
        p = float(self.transformer[0].dropout_1.p) if self.training else 0.0
        seed = runtime.next_drop(p, 0).seed if p > 0 else 0
        drop = (lambda site: ops.Drop(p, seed, site)) if p > 0 else (lambda site: None)
        pre = self.layernorm_positioning == "pre"
        # everything the backward needs lives in one allocation per forward: per layer 32 (pre-LN) / 40 (post-LN) x M x E bytes
        # of hidden-width tensors, 8 x M x F of feed-forward ones, 4 row statistics; + the final LayerNorm's statistics
        per_layer = (32 if pre else 40) * M * E + 8 * M * F + 16 * M + 4 * B * H * L + 20 * 256
        arena = engine.Arena(dev, self.layers_num * per_layer + 4 * M * E + 8 * M + 8 * 256)
        mat, vec, pl = arena.mat, arena.vec, arena.planes
        scale = 1.0 / math.sqrt(float(hd))
        big_qkv, big_ff = ops.use_gemm256(M, 3 * E, E), ops.use_gemm256(M, F, E)
        h = emb.detach().contiguous().view(M, E)
        h_p = None
        if not pre:
            h_p = ops.split_planes(h, pl(M, E))
        saved = {"layers": [], "seg": seg, "dims": (B, L, E, H, hd, M, F), "drop": (p, seed), "W": W}
        for i, (layer, w) in enumerate(zip(self.transformer, W)):
            att, ffn = layer.self_attn, layer.feed_forward
            ln1, ln2 = layer.layer_norm_1, layer.layer_norm_2
            s0 = 4 * i
            S = {}
            if pre:
                x_p, S["m1"], S["r1"], S["h_in"] = pl(M, E), vec(M), vec(M), h
                ops.layernorm_fwd(h, ln1.gamma.data, ln1.beta.data, None, S["m1"], S["r1"], rows=M, D=E, eps=ln1.eps, mode=1,
                                  out_planes=x_p)
            else:
                x_p = h_p
            qkv_p, o_p, t1 = pl(M, 3 * E), pl(M, E), mat(M, E)
            engine.linear_fwd(ws, x_p, w["wqkv" if big_qkv else "wqkv_t"], w["bqkv"], None, M, 3 * E, E, out_planes=qkv_p)
            S["lse"] = vec(B * H * L)                                    # the attention's log-sum-exp: an input of its backward
            ops.self_attn_fwd(qkv_p, seg, o_p, batch=B, heads=H, L=L, head_dim=hd, scale=scale, lse=S["lse"], drop=drop(s0))
            engine.linear_fwd(ws, o_p, w["wo"], att.final_linear.bias.data, t1, M, E, E, resid=h, drop=drop(s0 + 1))
            z, ff_p = mat(M, F), pl(M, F)
            S.update(x_p=x_p, qkv_p=qkv_p, o_p=o_p, t1=t1, z=z, ff_p=ff_p)
            if pre:
                x2_p, S["m2"], S["r2"] = pl(M, E), vec(M), vec(M)
                ops.layernorm_fwd(t1, ln2.gamma.data, ln2.beta.data, None, S["m2"], S["r2"], rows=M, D=E, eps=ln2.eps, mode=1,
                                  out_planes=x2_p)
                engine.linear_fwd(ws, x2_p, w["w1" if big_ff else "w1_t"], ffn.linear_1.bias.data, None, M, F, E, act=1, out_z=z, out_planes=ff_p)
                hn = mat(M, E)
                engine.linear_fwd(ws, ff_p, w["w2"], ffn.linear_2.bias.data, hn, M, E, F, resid=t1, drop=drop(s0 + 2))
                S["x2_p"] = x2_p
                h = hn
            else:
                inter, inter_p, S["m1"], S["r1"] = mat(M, E), pl(M, E), vec(M), vec(M)
                ops.layernorm_fwd(t1, ln1.gamma.data, ln1.beta.data, inter, S["m1"], S["r1"], rows=M, D=E, eps=ln1.eps, mode=1,
                                  out_planes=inter_p)
                engine.linear_fwd(ws, inter_p, w["w1" if big_ff else "w1_t"], ffn.linear_1.bias.data, None, M, F, E, act=1, out_z=z, out_planes=ff_p)
                t2 = mat(M, E)
                engine.linear_fwd(ws, ff_p, w["w2"], ffn.linear_2.bias.data, t2, M, E, F, resid=inter, drop=drop(s0 + 2))
                hn, hn_p, S["m2"], S["r2"] = mat(M, E), pl(M, E), vec(M), vec(M)
                ops.layernorm_fwd(t2, ln2.gamma.data, ln2.beta.data, hn, S["m2"], S["r2"], rows=M, D=E, eps=ln2.eps, mode=1,
                                  out_planes=hn_p)
                S.update(inter_p=inter_p, t2=t2)
                h, h_p = hn, hn_p
            saved["layers"].append(S)
        if self.final_layernorm:
            out = torch.empty(B, L, E, device=dev)
            saved["h_final"], saved["mf"], saved["rf"] = h, vec(M), vec(M)
            ops.layernorm_fwd(h, self.layer_norm.gamma.data, self.layer_norm.beta.data, out.view(M, E), saved["mf"], saved["rf"],
                              rows=M, D=E, eps=self.layer_norm.eps, mode=1)
        else:
            out = h.view(B, L, E)
        return out, saved

    def _grad_layout(self, flat):
        """Views of `flat` (numel = all parameters) as {parameter: gradient} plus, per layer, the [3E, E] / [3E] blocks that hold the
        Q, K, V weight / bias gradients CONTIGUOUSLY: the fused QKV weight gradient (one TN GEMM + its column sums) is written
        straight into them -- no per-layer copies into three separate tensors."""
        views, qkv, off = {}, [], 0
        for layer in self.transformer:
            lin = layer.self_attn.linear_layers
            E = lin[0].weight.shape[0]
            wblk = flat[off:off + 3 * E * E].view(3 * E, E)
            off += 3 * E * E
            bblk = flat[off:off + 3 * E]
            off += 3 * E
            for j in range(3):
                views[lin[j].weight] = wblk[j * E:(j + 1) * E]
                views[lin[j].bias] = bblk[j * E:(j + 1) * E]
            qkv.append((wblk, bblk))
        for q in self.parameters():
            if q not in views:
                views[q] = flat[off:off + q.numel()].view_as(q)
                off += q.numel()
        return views, qkv

    def grad_buffers(self):
        """{parameter: gradient} views of ONE persistent flat fp32 buffer (the explicit training path of
        lr2ppo_amd.finetune.features: no per-step allocation, fixed addresses for the optimizer's chunk table and one
        contiguous all-reduce under data parallelism)."""
        params = list(self.parameters())
        dev = params[0].device
        if getattr(self, "_gflat", None) is None or self._gflat.device != dev:
            self._gflat = torch.zeros(sum(q.numel() for q in params), device=dev)
            self._gviews, self._gqkv = self._grad_layout(self._gflat)
        return self._gviews

    @torch.no_grad()
    def _backward_train(self, saved, dout, G=None):
        """-> (d emb [B, L, E], {parameter: gradient}) for the forward that produced `saved`.  G: write the parameter gradients
        into these tensors (grad_buffers()) instead of a fresh allocation."""
        B, L, E, H, hd, M, F = saved["dims"]
        dev, seg, W = dout.device, saved["seg"], saved["W"]
        ws = self._ws
        p, seed = saved["drop"]
        drop = (lambda site: ops.Drop(p, seed, site)) if p > 0 else (lambda site: None)
        # transient gradients: named workspace buffers, reused by every layer and every call (one stream, sequential); the
        # running hidden-state gradient alternates between two of them
        mat = lambda name, r, c: ws.mat("bwd:" + name, r, c)            # noqa: E731
        pl = lambda name, r, c: ws.planes("bwd:" + name, r, c)          # noqa: E731
        if G is None:
            G, qkv_blocks = self._grad_layout(torch.empty(sum(q.numel() for q in self.parameters()), device=dev))
        else:
            if G is not getattr(self, "_gviews", None):
                raise ValueError("_backward_train(G=...): pass grad_buffers()")
            qkv_blocks = self._gqkv
        partials = ws.vec("ln_partials", ops.LN_BWD_BLOCKS * 2 * E)
        dsum_ws = ws.vec("attn_dsum", B * H * L)
        pre = self.layernorm_positioning == "pre"
        scale = 1.0 / math.sqrt(float(hd))
        big_dqkv = ops.use_gemm256(M, E, 3 * E)     # the QKV input gradient goes through a transposed fp32 concatenation only then
        dh = dout.contiguous().view(M, E)
        if self.final_layernorm:
            ln = self.layer_norm
            dnew = mat("dh0", M, E)
            # pre-LN: the top layer's first consumer of this gradient is dropout_2's mask + the planes split (FFN-2's dY): the LayerNorm
            # backward writes them on the way out instead of a separate pass over [M, E]
            top = self.layers_num - 1
            ops.layernorm_bwd(dh, saved["h_final"], ln.gamma.data, saved["mf"], saved["rf"], dnew, partials, G[ln.gamma],
                              G[ln.beta], rows=M, D=E, mode=1, eps=ln.eps, dx_planes=pl("dff_p", M, E) if pre else None,
                              drop=drop(4 * top + 2) if pre else None)
            dh = dnew
        dff_ready = pre and self.final_layernorm
        flip = 1
        for i in reversed(range(self.layers_num)):
            layer, w, S = self.transformer[i], W[i], saved["layers"][i]
            att, ffn = layer.self_attn, layer.feed_forward
            ln1, ln2 = layer.layer_norm_1, layer.layer_norm_2
            s0 = 4 * i
            dff_p, dz_p = pl("dff_p", M, E), pl("dz_p", M, F)
            if pre:
                if not dff_ready:
                    ops.dropout_planes(dh, dff_p, drop(s0 + 2))
                ffn_in_p = S["x2_p"]
            else:
                d_t2 = mat("d_t2", M, E)
                ops.layernorm_bwd(dh, S["t2"], ln2.gamma.data, S["m2"], S["r2"], d_t2, partials, G[ln2.gamma], G[ln2.beta],
                                  rows=M, D=E, dx_planes=dff_p, drop=drop(s0 + 2), mode=1, eps=ln2.eps)
                ffn_in_p = S["inter_p"]
            engine.linear_wgrad(ws, dff_p, S["ff_p"], G[ffn.linear_2.weight], G[ffn.linear_2.bias], M, F, E)
            engine.linear_dgrad(ws, dff_p, w["w2"], None, M, F, E, act=2, aux_z=S["z"], out_planes=dz_p, w_f32=ffn.linear_2.weight.data)
            engine.linear_wgrad(ws, dz_p, ffn_in_p, G[ffn.linear_1.weight], G[ffn.linear_1.bias], M, E, F)
            d_t1, dao_p = mat("d_t1", M, E), pl("dao_p", M, E)
            if pre:
                d_x2 = mat("d_x2", M, E)
                engine.linear_dgrad(ws, dz_p, w["w1"], d_x2, M, E, F, w_f32=ffn.linear_1.weight.data)
                ops.layernorm_bwd(d_x2, S["t1"], ln2.gamma.data, S["m2"], S["r2"], d_t1, partials, G[ln2.gamma], G[ln2.beta],
                                  rows=M, D=E, resid_grad=dh, dx_planes=dao_p, drop=drop(s0 + 1), mode=1, eps=ln2.eps)
            else:
                d_inter = mat("d_x2", M, E)
                engine.linear_dgrad(ws, dz_p, w["w1"], d_inter, M, E, F, resid=d_t2, w_f32=ffn.linear_1.weight.data)
                ops.layernorm_bwd(d_inter, S["t1"], ln1.gamma.data, S["m1"], S["r1"], d_t1, partials, G[ln1.gamma], G[ln1.beta],
                                  rows=M, D=E, dx_planes=dao_p, drop=drop(s0 + 1), mode=1, eps=ln1.eps)
            engine.linear_wgrad(ws, dao_p, S["o_p"], G[att.final_linear.weight], G[att.final_linear.bias], M, E, E)
            do_p, dqkv_p = pl("do_p", M, E), pl("dqkv_p", M, 3 * E)
            engine.linear_dgrad(ws, dao_p, w["wo"], None, M, E, E, out_planes=do_p, w_f32=att.final_linear.weight.data)
            ops.self_attn_bwd(S["qkv_p"], do_p, seg, dqkv_p, S["lse"], dsum_ws, batch=B, heads=H, L=L, head_dim=hd, scale=scale,
                              drop=drop(s0), o=S["o_p"])
            engine.linear_wgrad(ws, dqkv_p, S["x_p"], qkv_blocks[i][0], qkv_blocks[i][1], M, E, 3 * E)     # = the three gradients
            # the gradient handed back to autograd (layer 0) gets its own storage: it outlives this call
            dprev = torch.empty(M, E, device=dev) if i == 0 else mat("dh%d" % flip, M, E)
            flip ^= 1
            if pre:
                d_x1 = mat("d_x1", M, E)
                engine.linear_dgrad(ws, dqkv_p, w["wqkv"], d_x1, M, E, 3 * E, w_f32=w["wqkv_f32"] if big_dqkv else None)
                # ... and the layer below gets its dropout_2-masked planes from this LayerNorm backward (dff_p is free again: its
                # readers of this layer ran earlier on the stream)
                nxt = i > 0
                ops.layernorm_bwd(d_x1, S["h_in"], ln1.gamma.data, S["m1"], S["r1"], dprev, partials, G[ln1.gamma], G[ln1.beta],
                                  rows=M, D=E, resid_grad=d_t1, mode=1, eps=ln1.eps, dx_planes=dff_p if nxt else None,
                                  drop=drop(s0 - 4 + 2) if nxt else None)
                dff_ready = nxt
            else:
                engine.linear_dgrad(ws, dqkv_p, w["wqkv"], dprev, M, E, 3 * E, resid=d_t1, w_f32=w["wqkv_f32"] if big_dqkv else None)
            dh = dprev
            saved["layers"][i] = None                                   # release this layer's activations
        return dh.view(B, L, E), G


class _EncoderFn(torch.autograd.Function):
    """Coarse autograd node: the whole encoder stack forward / backward on the HIP kernels."""

    @staticmethod
    def forward(ctx, enc, emb, seg, *params):
        out, saved = enc._forward_train(emb, seg)
        ctx.enc, ctx.saved = enc, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        demb, G = ctx.enc._backward_train(ctx.saved, dout.contiguous())
        ctx.saved = None
        return (None, demb, None) + tuple(G[q] if q.requires_grad else None for q in ctx.enc.parameters())


class _OneLayerStack(TransformerEncoder):
    """A TransformerLayer run on its own (`TransformerLayer.forward(hidden, mask)`, layers/transformer.py:50-73 upstream):
    the one-layer case of the encoder schedule, without the stack's final LayerNorm.  Shares the layer's parameters."""

    def __init__(self, layer):
        nn.Module.__init__(self)
        att = layer.self_attn
        self.mask, self.layers_num = "fully_visible", 1
        self.layernorm_positioning = layer.layernorm_positioning
        self.heads_num, self.hidden_size = att.heads_num, att.final_linear.out_features
        self.transformer = nn.ModuleList([layer])
        self.final_layernorm = False
        self._ws = None
        self._wplanes = None


def seg_from_additive_mask(mask):
    """[B, 1, L, L] additive mask of the 'fully_visible' form (0 where the KEY is visible, -10000 where it is padding, the
    same for every query row: encoders/transformer_encoder.py:62-68 upstream) -> seg [B, L] (1 visible / 0 padded).
    Other mask shapes (causal, per-query) are not on the HIP path."""
    if mask.dim() != 4 or mask.shape[1] != 1 or mask.shape[2] != mask.shape[3]:
        raise ValueError("mask must be [batch, 1, seq, seq]")
    row0 = mask[:, 0, 0, :]
    if not bool((mask[:, 0] == row0.unsqueeze(1)).all()):
        raise NotImplementedError("HIP attention takes key-padding masks only (mask='fully_visible'); use the reference "
                                  "for causal / per-query masks")
    if not bool(((row0 == 0) | (row0 == -10000.0)).all()):
        raise NotImplementedError("HIP attention supports additive masks with values 0 / -10000 only")
    return (row0 == 0).to(torch.int64)
