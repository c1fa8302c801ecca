"""TencentPretrain LayerNorm: gamma * (x - mean) / (std_unbiased + eps) + beta  (layers/layer_norm.py:5-21 of
the reference; NOT nn.LayerNorm -- the two differ by up to 2.5e-3).  Parameters are named
gamma / beta so checkpoints and the decay exemption by name (finetune/ppo.py:381) carry over."""
import torch
import torch.nn as nn

from ... import ops


class LayerNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(hidden_size))
        self.beta = nn.Parameter(torch.zeros(hidden_size))

    @torch.no_grad()
    def forward(self, x):
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        out = torch.empty_like(x2)
        ops.layernorm_fwd(x2, self.gamma.data, self.beta.data, out, rows=x2.shape[0], D=D, eps=self.eps, mode=1)
        return out.view_as(x)
