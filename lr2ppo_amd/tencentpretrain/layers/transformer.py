"""Parameter containers of one TencentPretrain transformer layer with the reference's key layout
(layers/transformer.py:8-48, multi_headed_attn.py:6-25, position_ffn.py:4-10): self_attn.linear_layers.{0,1,2}
= Q, K, V; self_attn.final_linear; feed_forward.linear_{1,2}; layer_norm_{1,2}.{gamma,beta}.
The arithmetic is scheduled by TransformerEncoder (encoders/transformer_encoder.py of this package)."""
import torch.nn as nn

from .layer_norm import LayerNorm


class MultiHeadedAttention(nn.Module):
    def __init__(self, hidden_size, heads_num, attention_head_size, dropout, has_bias=True, with_scale=True):
        super().__init__()
        self.heads_num, self.per_head_size, self.with_scale = heads_num, attention_head_size, with_scale
        self.inner_hidden_size = heads_num * attention_head_size
        self.linear_layers = nn.ModuleList([nn.Linear(hidden_size, self.inner_hidden_size, bias=has_bias) for _ in range(3)])
        self.dropout = nn.Dropout(dropout)
        self.final_linear = nn.Linear(self.inner_hidden_size, hidden_size, bias=has_bias)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, hidden_size, feedforward_size, hidden_act, has_bias=True):
        super().__init__()
        if hidden_act != "gelu":
            raise NotImplementedError("the HIP FFN epilogue implements exact-erf GELU (both reference encoder configs)")
        self.linear_1 = nn.Linear(hidden_size, feedforward_size, bias=has_bias)
        self.linear_2 = nn.Linear(feedforward_size, hidden_size, bias=has_bias)


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
