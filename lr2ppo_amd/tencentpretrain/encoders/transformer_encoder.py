"""TransformerEncoder.forward(emb, seg) -> hidden on the gfx950 kernels (inference schedule).

Mirrors the reference's encoders/transformer_encoder.py:7-138 + layers/transformer.py:50-73 for
mask="fully_visible": additive key mask -10000 * (seg <= 0) applied AFTER the 1/sqrt(64) scale, post-LN
(RoBERTa-base) or pre-LN + final LayerNorm (ViT-B/16), exact-erf GELU, TencentPretrain LayerNorm semantics.

Per layer: ONE fused QKV GEMM (N = 3*hidden, weights concatenated once), the MFMA self-attention kernel, the output
projection (+residual), FFN1 (+GELU), FFN2 (+residual) and 2 wavefront LayerNorms.  Everything a GEMM consumes travels as
bf16 hi/lo planes written by the producing kernel (LayerNorm, attention, GEMM epilogue) and is streamed by LDS-DMA; the
residual stream stays fp32.  Weight planes are split once and re-split only when a parameter changes.
Training through the encoders (dropout + backward) is not part of this round: the reference never trains them either
(features are pre-extracted)."""
import math

import torch
import torch.nn as nn

from ... import engine, ops
from ..layers.layer_norm import LayerNorm
from ..layers.transformer import TransformerLayer


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
        if self.layernorm_positioning == "pre":
            self.layer_norm = LayerNorm(args.hidden_size)
        self._ws = None
        self._wplanes = None

    def _weight_planes(self, dev):
        """Per layer: (Wqkv planes [3E, E], bqkv [3E], Wo, W1, W2 planes); rebuilt when any parameter was written."""
        sig = tuple(p._version for p in self.parameters()) + (str(dev),)
        if self._wplanes is not None and self._wplanes[0] == sig:
            return self._wplanes[1]
        out = []
        for layer in self.transformer:
            att, ffn = layer.self_attn, layer.feed_forward
            wqkv = torch.cat([att.linear_layers[i].weight.data for i in range(3)], dim=0).contiguous()
            bqkv = torch.cat([att.linear_layers[i].bias.data for i in range(3)], dim=0).contiguous()
            ent = {"bqkv": bqkv}
            for name, w in (("wqkv", wqkv), ("wo", att.final_linear.weight.data), ("w1", ffn.linear_1.weight.data),
                            ("w2", ffn.linear_2.weight.data)):
                pl = ops.Planes.empty(w.shape[0], w.shape[1], dev)
                ops.split_planes(w.contiguous(), pl)
                ent[name] = pl
            out.append(ent)
        self._wplanes = (sig, out)
        return out

    @torch.no_grad()
    def forward(self, emb, seg):
        if self.training and any(l.dropout_1.p > 0 for l in self.transformer):
            raise NotImplementedError("encoder training (dropout/backward) is outside this round's scope; call .eval()")
        if emb.dtype != torch.float32 or not emb.is_cuda:
            raise TypeError("lr2ppo_amd: emb must be a float32 tensor on the HIP device (no CPU path)")
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
        if not pre:
            ops.split_planes(h, x_p)
        for layer, w in zip(self.transformer, W):
            att, ffn = layer.self_attn, layer.feed_forward
            ln1, ln2 = layer.layer_norm_1, layer.layer_norm_2
            if pre:                                                   # layers/transformer.py:63-73
                ops.layernorm_fwd(h, ln1.gamma.data, ln1.beta.data, None, rows=M, D=E, eps=ln1.eps, mode=1, out_planes=x_p)
            engine.linear_fwd(ws, x_p, w["wqkv"], w["bqkv"], None, M, 3 * E, E, out_planes=qkv_p)
            ops.self_attn_fwd(qkv_p, seg, o_p, batch=B, heads=H, L=L, head_dim=hd, scale=scale)
            engine.linear_fwd(ws, o_p, w["wo"], att.final_linear.bias.data, h2, M, E, E, resid=h)
            if pre:
                ops.layernorm_fwd(h2, ln2.gamma.data, ln2.beta.data, None, rows=M, D=E, eps=ln2.eps, mode=1, out_planes=t_p)
                engine.linear_fwd(ws, t_p, w["w1"], ffn.linear_1.bias.data, None, M, F, E, act=1, out_planes=ff_p)
                engine.linear_fwd(ws, ff_p, w["w2"], ffn.linear_2.bias.data, h, M, E, F, resid=h2)
            else:                                                     # layers/transformer.py:54-61
                ops.layernorm_fwd(h2, ln1.gamma.data, ln1.beta.data, h, rows=M, D=E, eps=ln1.eps, mode=1, out_planes=t_p)
                engine.linear_fwd(ws, t_p, w["w1"], ffn.linear_1.bias.data, None, M, F, E, act=1, out_planes=ff_p)
                engine.linear_fwd(ws, ff_p, w["w2"], ffn.linear_2.bias.data, h2, M, E, F, resid=h)
                ops.layernorm_fwd(h2, ln2.gamma.data, ln2.beta.data, h, rows=M, D=E, eps=ln2.eps, mode=1, out_planes=x_p)
        out = torch.empty(B, L, E, device=emb.device)
        if pre:
            ops.layernorm_fwd(h, self.layer_norm.gamma.data, self.layer_norm.beta.data, out.view(M, E), rows=M, D=E,
                              eps=self.layer_norm.eps, mode=1)
        else:
            out.view(M, E).copy_(h)
        return out
