"""TransformerEncoder.forward(emb, seg) -> hidden on the gfx950 kernels (inference schedule).

Mirrors the reference's encoders/transformer_encoder.py:7-138 + layers/transformer.py:50-73 for
mask="fully_visible": additive key mask -10000 * (seg <= 0) applied AFTER the 1/sqrt(64) scale, post-LN
(RoBERTa-base) or pre-LN + final LayerNorm (ViT-B/16), exact-erf GELU, TencentPretrain LayerNorm semantics.
Per layer: 6 split-bf16 MFMA GEMMs with fused bias / GELU / residual epilogues, one LDS-resident
self-attention kernel, 2 wavefront LayerNorms.  Training through the encoders (dropout + backward) is not part of
this round: the reference never trains them either (features are pre-extracted)."""
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
        h = ws.mat("h", M, E)
        h.copy_(emb.contiguous().view(M, E))
        pre = self.layernorm_positioning == "pre"
        q, k, v, o = (ws.mat(n, M, E) for n in ("q", "k", "v", "o"))
        t1, t2 = ws.mat("t1", M, E), ws.mat("t2", M, E)
        F = self.transformer[0].feed_forward.linear_1.out_features
        ff = ws.mat("ff", M, F)
        scale = 1.0 / math.sqrt(float(hd))
        for layer in self.transformer:
            att, ffn = layer.self_attn, layer.feed_forward
            ln1, ln2 = layer.layer_norm_1, layer.layer_norm_2
            if pre:
                x_in = t1
                ops.layernorm_fwd(h, ln1.gamma.data, ln1.beta.data, x_in, rows=M, D=E, eps=ln1.eps, mode=1)
            else:
                x_in = h
            for dst, lin in ((q, att.linear_layers[0]), (k, att.linear_layers[1]), (v, att.linear_layers[2])):
                engine.linear_fwd(ws, x_in, lin.weight.data, lin.bias.data, dst, M, E, E)
            ops.self_attn_fwd(q, k, v, seg, o, batch=B, heads=H, L=L, head_dim=hd, scale=scale)
            if pre:
                engine.linear_fwd(ws, o, att.final_linear.weight.data, att.final_linear.bias.data, t2, M, E, E, resid=h)
                h, t2 = t2, h                                     # hidden = hidden + attn
                ops.layernorm_fwd(h, ln2.gamma.data, ln2.beta.data, t1, rows=M, D=E, eps=ln2.eps, mode=1)
                engine.linear_fwd(ws, t1, ffn.linear_1.weight.data, ffn.linear_1.bias.data, ff, M, F, E, act=1)
                engine.linear_fwd(ws, ff, ffn.linear_2.weight.data, ffn.linear_2.bias.data, t2, M, E, F, resid=h)
                h, t2 = t2, h
            else:
                engine.linear_fwd(ws, o, att.final_linear.weight.data, att.final_linear.bias.data, t1, M, E, E, resid=h)
                ops.layernorm_fwd(t1, ln1.gamma.data, ln1.beta.data, t2, rows=M, D=E, eps=ln1.eps, mode=1)   # inter
                engine.linear_fwd(ws, t2, ffn.linear_1.weight.data, ffn.linear_1.bias.data, ff, M, F, E, act=1)
                engine.linear_fwd(ws, ff, ffn.linear_2.weight.data, ffn.linear_2.bias.data, t1, M, E, F, resid=t2)
                ops.layernorm_fwd(t1, ln2.gamma.data, ln2.beta.data, h, rows=M, D=E, eps=ln2.eps, mode=1)
        out = torch.empty(B, L, E, device=emb.device)
        if pre:
            ops.layernorm_fwd(h, self.layer_norm.gamma.data, self.layer_norm.beta.data, out.view(M, E), rows=M, D=E,
                              eps=self.layer_norm.eps, mode=1)
        else:
            out.view(M, E).copy_(h)
        return out
