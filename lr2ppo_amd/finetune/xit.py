"""XiT cross-attention block -- parameter containers with the reference's state_dict layout.

Mirrors the module tree of the reference's finetune/xit.py:9-148 so checkpoints interchange
(`xit.0.0.0.fn.1.queries.weight`, ...), but the sub-modules only hold parameters: the whole block
(3 LayerNorms, Q/K/V/out projections, softmax(QK^T)/sqrt(E) attention, GELU FFN, 3 dropouts, 2 residuals,
final LayerNorm) is executed by lr2ppo_amd.engine.xit_forward / xit_backward on the gfx950 kernels.

Reference quirks kept: 8 heads; no 1/sqrt(d) on the energies; probabilities divided by
sqrt(feat_size) after the softmax; attention_mask='causal' accepted and ignored (it is a no-op upstream).
"""
import torch
import torch.nn as nn

from .. import engine
from ..ops import Drop  # noqa: F401  (re-export for tests)


class _Fn(nn.Module):
    """`.fn` holder used by the two residual wrappers."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class ResidualAddFusion(_Fn):   # xit.py:45-55
    pass


class ResidualAdd(_Fn):         # xit.py:77-86
    pass


class LayerNormBlock(nn.Module):   # xit.py:89-100
    def __init__(self, emb_size):
        super().__init__()
        self.emb_size = emb_size
        self.ln_x = nn.LayerNorm(emb_size)
        self.ln_y = nn.LayerNorm(emb_size)


class MultiHeadAttention(nn.Module):   # xit.py:113-123 (declaration order fixes the checkpoint key order)
    def __init__(self, feat_size=768, num_heads=8, dropout=0, attention_mask="fully_visiable"):
        super().__init__()
        if dropout:
            raise NotImplementedError("attention-probability dropout is 0 in every reference configuration")
        self.emb_size, self.num_heads, self.attention_mask = feat_size, num_heads, attention_mask
        self.keys = nn.Linear(feat_size, feat_size)
        self.queries = nn.Linear(feat_size, feat_size)
        self.values = nn.Linear(feat_size, feat_size)
        self.att_drop = nn.Dropout(dropout)
        self.projection = nn.Linear(feat_size, feat_size)


class FeedForwardBlock(nn.Sequential):   # xit.py:103-110 -> keys "0" and "3"
    def __init__(self, emb_size, expansion=4, drop_p=0.0):
        super().__init__(nn.Linear(emb_size, expansion * emb_size), nn.GELU(), nn.Dropout(drop_p),
                         nn.Linear(expansion * emb_size, emb_size))


class XEncoderBlock(nn.Sequential):   # xit.py:23-42
    def __init__(self, feat_size=768, drop_p=0.1, forward_expansion=4, forward_drop_p=0.1, **kwargs):
        if forward_expansion != 4 or drop_p != forward_drop_p:
            raise NotImplementedError("the fused XiT kernel schedule assumes expansion 4 and one dropout rate")
        super().__init__(
            ResidualAddFusion(nn.Sequential(LayerNormBlock(feat_size), MultiHeadAttention(feat_size, **kwargs),
                                            nn.Dropout(drop_p))),
            ResidualAdd(nn.Sequential(nn.LayerNorm(feat_size),
                                      FeedForwardBlock(feat_size, expansion=forward_expansion, drop_p=forward_drop_p),
                                      nn.Dropout(drop_p))))
        self.drop_p = drop_p


class XEncoder(nn.Sequential):   # xit.py:18-20
    def __init__(self, **kwargs):
        super().__init__(XEncoderBlock(**kwargs))


class XFeatureLayer(nn.Sequential):   # xit.py:71-74
    def __init__(self, feat_size=768):
        super().__init__(nn.LayerNorm(feat_size))


class XiT(nn.Sequential):
    """XiT(feat_size)((x, y)) -> LN(block(x, y));  x: [b, Lq, E], y: [b, Lk, E] fp32 on a HIP device."""

    def __init__(self, feat_size: int = 768, **kwargs):
        super().__init__(XEncoder(feat_size=feat_size, **kwargs), XFeatureLayer(feat_size=feat_size))
        self.feat_size = feat_size
        self._keys = engine.XitKeys("xit")
        self._ws = None

    @property
    def drop_p(self):
        return self[0][0].drop_p

    def forward(self, x_y):
        x, y = x_y
        params = [p for _, p in self.named_parameters()]
        return _XitFn.apply(self, x, y, *params)


class _XitFn(torch.autograd.Function):
    """Stand-alone autograd entry for one XiT block (the heads call the engine directly)."""

    @staticmethod
    def forward(ctx, mod, x, y, *params):
        from .. import runtime
        b, Lq, E = x.shape
        Lk = y.shape[1]
        if mod._ws is None:
            mod._ws = engine.Workspace(x.device)
        ws = mod._ws
        P = {"xit." + n: p.data for n, p in mod.named_parameters()}
        wp = engine.WeightPlanes(P, mod._keys.gemm_weights())
        wp.refresh()
        x2 = x.detach().contiguous().view(b * Lq, E).clone()
        same = y is x
        y2 = x2 if same else y.detach().contiguous().view(b * Lk, E).clone()
        drop = runtime.next_drop(mod.drop_p, 0) if mod.training else None
        out = torch.empty(b * Lq, E, device=x.device)
        engine.xit_forward(ws, "s.", P, wp.planes, mod._keys, x2, y2, b, Lq, Lk, E, out, save=True, drop=drop, heads=8)
        ctx.mod, ctx.drop, ctx.dims, ctx.same, ctx.wp = mod, drop, (b, Lq, Lk, E), same, wp
        ctx.save_for_backward(x2, y2)
        return out.view(b, Lq, E)

    @staticmethod
    def backward(ctx, d_out):
        mod = ctx.mod
        b, Lq, Lk, E = ctx.dims
        x2, y2 = ctx.saved_tensors
        ws = mod._ws
        P = {"xit." + n: p.data for n, p in mod.named_parameters()}
        G = {"xit." + n: torch.empty_like(p.data) for n, p in mod.named_parameters()}
        dx = torch.empty(b * Lq, E, device=x2.device)
        dy = None if ctx.same else torch.empty(b * Lk, E, device=x2.device)
        engine.xit_backward(ws, "s.", P, ctx.wp.planes, G, mod._keys, x2, y2, d_out.contiguous().view(b * Lq, E), b, Lq, Lk,
                            E, dx, dy, drop=ctx.drop, same_xy=ctx.same, heads=8)
        grads = [G["xit." + n] for n, _ in mod.named_parameters()]
        return (None, dx.view(b, Lq, E), None if ctx.same else dy.view(b, Lk, E), *grads)
