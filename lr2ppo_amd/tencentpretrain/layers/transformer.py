"""One TencentPretrain transformer layer with the reference's key layout (layers/transformer.py:8-48,
multi_headed_attn.py:6-25, position_ffn.py:4-10): self_attn.linear_layers.{0,1,2} = Q, K, V; self_attn.final_linear;
feed_forward.linear_{1,2}; layer_norm_{1,2}.{gamma,beta}.
The arithmetic is scheduled by TransformerEncoder (encoders/transformer_encoder.py of this package); the layer-level
`forward`s below run the same kernels for callers that use a layer on its own:
  TransformerLayer.forward(hidden, mask)              one-layer encoder schedule, differentiable (layers/transformer.py:50-73)
  PositionwiseFeedForward.forward(x)                  two fused GEMMs, inference only (position_ffn.py:12-15)
  MultiHeadedAttention.forward(key, value, query, mask)  self-attention (key is value is query), inference only
                                                      (multi_headed_attn.py:27-76)"""
import math

import torch
import torch.nn as nn

from ... import engine, ops
from .layer_norm import LayerNorm


def _needs_grad(mod, *xs):
    return torch.is_grad_enabled() and (any(x.requires_grad for x in xs) or any(p.requires_grad for p in mod.parameters()))


class MultiHeadedAttention(nn.Module):
    def __init__(self, hidden_size, heads_num, attention_head_size, dropout, has_bias=True, with_scale=True):
        super().__init__()
        self.heads_num, self.per_head_size, self.with_scale = heads_num, attention_head_size, with_scale
        self.inner_hidden_size = heads_num * attention_head_size
        self.linear_layers = nn.ModuleList([nn.Linear(hidden_size, self.inner_hidden_size, bias=has_bias) for _ in range(3)])
        self.dropout = nn.Dropout(dropout)
        self.final_linear = nn.Linear(self.inner_hidden_size, hidden_size, bias=has_bias)

    def forward(self, key, value, query, mask, position_bias=None, has_residual_attention=False, prev_attn=None):
        """-> (output [B, L, hidden], None).  Self-attention with a key-padding mask, evaluated without autograd; for
        gradients call the enclosing TransformerLayer / TransformerEncoder (whose backward is hand-written)."""
        from ..encoders.transformer_encoder import seg_from_additive_mask
        if not (key is value and value is query):
            raise NotImplementedError("HIP MultiHeadedAttention.forward covers self-attention (key is value is query)")
        if position_bias is not None or has_residual_attention or not self.with_scale or self.linear_layers[0].bias is None:
            raise NotImplementedError("position bias / residual attention / unscaled / bias-free attention are not on the HIP path")
        if _needs_grad(self, query):
            raise RuntimeError("MultiHeadedAttention.forward runs without autograd on the HIP path; call it under "
                               "torch.no_grad(), or use TransformerLayer / TransformerEncoder for a differentiable forward")
        B, L, E = query.shape
        H, hd, M = self.heads_num, self.per_head_size, B * L
        dev = query.device
        seg = seg_from_additive_mask(mask).to(dev).contiguous().view(-1)
        with torch.no_grad():
            ws = engine.Workspace(dev)
            x_p = ops.split_planes(query.contiguous().view(M, E), ops.Planes.empty(M, E, dev))
            wqkv = torch.cat([l.weight.data for l in self.linear_layers], dim=0).contiguous()
            bqkv = torch.cat([l.bias.data for l in self.linear_layers], dim=0).contiguous()
            w_p = ops.split_planes(wqkv, ops.Planes.empty(3 * H * hd, E, dev))
            qkv_p = ops.Planes.empty(M, 3 * H * hd, dev)
            engine.linear_fwd(ws, x_p, w_p, bqkv, None, M, 3 * H * hd, E, out_planes=qkv_p)
            o_p = ops.Planes.empty(M, H * hd, dev)
            ops.self_attn_fwd(qkv_p, seg, o_p, batch=B, heads=H, L=L, head_dim=hd, scale=1.0 / math.sqrt(float(hd)))
            wo_p = ops.split_planes(self.final_linear.weight.data.contiguous(), ops.Planes.empty(E, H * hd, dev))
            out = torch.empty(M, E, device=dev)
            engine.linear_fwd(ws, o_p, wo_p, self.final_linear.bias.data, out, M, E, H * hd)
        return out.view(B, L, E), None


class PositionwiseFeedForward(nn.Module):
    def __init__(self, hidden_size, feedforward_size, hidden_act, has_bias=True):
        super().__init__()
        if hidden_act != "gelu":
            raise NotImplementedError("the HIP FFN epilogue implements exact-erf GELU (both reference encoder configs)")
        self.linear_1 = nn.Linear(hidden_size, feedforward_size, bias=has_bias)
        self.linear_2 = nn.Linear(feedforward_size, hidden_size, bias=has_bias)

    def forward(self, x):
        """linear_2(gelu(linear_1(x))) (position_ffn.py:12-15), evaluated without autograd."""
        if _needs_grad(self, x):
            raise RuntimeError("PositionwiseFeedForward.forward runs without autograd on the HIP path; use TransformerLayer / "
                               "TransformerEncoder for a differentiable forward")
        E, F = self.linear_1.in_features, self.linear_1.out_features
        x2 = x.contiguous().view(-1, E)
        M, dev = x2.shape[0], x.device
        with torch.no_grad():
            ws = engine.Workspace(dev)
            h_p = ops.Planes.empty(M, F, dev)
            engine.linear_fwd(ws, x2, self.linear_1.weight.data, self.linear_1.bias.data, None, M, F, E, act=1, out_planes=h_p)
            w2_p = ops.split_planes(self.linear_2.weight.data.contiguous(), ops.Planes.empty(E, F, dev))
            out = torch.empty(M, E, device=dev)
            engine.linear_fwd(ws, h_p, w2_p, self.linear_2.bias.data, out, M, E, F)
        return out.view(*x.shape[:-1], E)


class TransformerLayer(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.layernorm_positioning = args.layernorm_positioning
        head = getattr(args, "attention_head_size", args.hidden_size // args.heads_num)
        has_bias = not bool(args.remove_transformer_bias)
        if args.feed_forward != "dense" or args.layernorm != "normal" or not has_bias or bool(args.remove_attention_scale):
            raise NotImplementedError("HIP path covers dense FFN + TencentPretrain LayerNorm + biases + scaled attention "
                                      "(ViT-B/16 and RoBERTa-base configs)")
        self.self_attn = MultiHeadedAttention(args.hidden_size, args.heads_num, head, args.dropout, has_bias=has_bias)
        self.dropout_1 = nn.Dropout(args.dropout)
        self.feed_forward = PositionwiseFeedForward(args.hidden_size, args.feedforward_size, args.hidden_act, has_bias)
        self.dropout_2 = nn.Dropout(args.dropout)
        self.layer_norm_1 = LayerNorm(args.hidden_size)
        self.layer_norm_2 = LayerNorm(args.hidden_size)
        self._stack = None

    def forward(self, hidden, mask, position_bias=None, has_residual_attention=False, prev_attn=None):
        """-> (output [B, L, hidden], None): layers/transformer.py:50-73 for a key-padding `mask` [B, 1, L, L]; differentiable
        (the one-layer case of TransformerEncoder's saving forward + hand-written backward)."""
        from ..encoders.transformer_encoder import _OneLayerStack, seg_from_additive_mask
        if position_bias is not None or has_residual_attention:
            raise NotImplementedError("position bias / residual attention are not on the HIP path")
        if self._stack is None:
            object.__setattr__(self, "_stack", _OneLayerStack(self))     # not a registered child: no parameter cycle
        self._stack.training = self.training
        return self._stack(hidden, seg_from_additive_mask(mask)), None
