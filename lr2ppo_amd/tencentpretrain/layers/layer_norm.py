"""TencentPretrain LayerNorm: gamma * (x - mean) / (std_unbiased + eps) + beta  (layers/layer_norm.py:5-21 of
the reference; NOT nn.LayerNorm -- the two differ by up to 2.5e-3).  Parameters are named
gamma / beta so checkpoints and the decay exemption by name (finetune/ppo.py:381) carry over.

Differentiable: with autograd enabled the forward keeps (x, mean, 1/(std+eps)) and the backward runs
lr2_layernorm_bwd (mode 1), so a stand-alone LayerNorm -- e.g. the stream LayerNorms of DualEmbedding
(embeddings/dual_embedding.py:21-33) -- passes gradients to whatever produced x and to gamma / beta."""
import torch
import torch.nn as nn

from ... import ops


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        D = x.shape[-1]
        x2 = x.detach().contiguous().view(-1, D)
        M = x2.shape[0]
        out = torch.empty_like(x2)
        mean, rstd = torch.empty(M, device=x.device), torch.empty(M, device=x.device)
        ops.layernorm_fwd(x2, gamma.detach(), beta.detach(), out, mean, rstd, rows=M, D=D, eps=eps, mode=1)
        ctx.save_for_backward(x2, gamma.detach(), mean, rstd)
        ctx.eps, ctx.shape = eps, x.shape
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, gamma, mean, rstd = ctx.saved_tensors
        M, D = x2.shape
        dy = dout.contiguous().view(M, D)
        dx = torch.empty_like(x2)
        dgamma, dbeta = torch.empty(D, device=dy.device), torch.empty(D, device=dy.device)
        partials = torch.empty(ops.LN_BWD_BLOCKS * 2 * D, device=dy.device)
        ops.layernorm_bwd(dy, x2, gamma, mean, rstd, dx, partials, dgamma, dbeta, rows=M, D=D, mode=1, eps=ctx.eps)
        return dx.view(ctx.shape), dgamma, dbeta, None


class LayerNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.gamma = nn.Parameter(torch.ones(hidden_size))
        self.beta = nn.Parameter(torch.zeros(hidden_size))

    def forward(self, x):
        if x.dtype != torch.float32 or not x.is_cuda:
            raise TypeError("lr2ppo_amd: LayerNorm input must be a float32 tensor on the HIP device (no CPU path)")
        if torch.is_grad_enabled() and (x.requires_grad or self.gamma.requires_grad or self.beta.requires_grad):
            return _LayerNormFn.apply(x, self.gamma, self.beta, self.eps)
        D = x.shape[-1]
        x2 = x.contiguous().view(-1, D)
        out = torch.empty_like(x2)
        with torch.no_grad():
            ops.layernorm_fwd(x2, self.gamma.data, self.beta.data, out, rows=x2.shape[0], D=D, eps=self.eps, mode=1)
        return out.view_as(x)


class _DropoutFn(torch.autograd.Function):
    """nn.Dropout on the device with the package's counter-based masks (csrc/common.h::dropout_keep): the same mask is
    applied to the gradient.  `drop` is an ops.Drop (p, seed, site)."""

    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return ops.dropout_apply(x.detach().contiguous(), torch.empty_like(x, memory_format=torch.contiguous_format), drop)

    @staticmethod
    def backward(ctx, dout):
        return ops.dropout_apply(dout.contiguous(), torch.empty_like(dout, memory_format=torch.contiguous_format), ctx.drop), None


def device_dropout(x, p: float, training: bool, site: int = 0):
    """x -> dropout(x) in train mode (one fresh seed from the runtime's mask stream per call), identity otherwise."""
    if not training or p <= 0.0:
        return x
    from ... import runtime
    drop = ops.Drop(p, runtime.next_drop(p, 0).seed, site)
    if torch.is_grad_enabled() and x.requires_grad:
        return _DropoutFn.apply(x, drop)
    with torch.no_grad():
        return ops.dropout_apply(x.contiguous(), torch.empty_like(x, memory_format=torch.contiguous_format), drop)
