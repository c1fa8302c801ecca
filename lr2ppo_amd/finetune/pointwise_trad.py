"""The `_trad` pointwise ranker on MI355X -- drop-in for the model / step of the reference's finetune/pointwise_trad.py
(BASELINE.json configs[0], the reference's own small "plumbing" case).

`Classifier` (pointwise_trad.py:132-177) is the LR2PPO head at sequence length 1: one pre-projected 768-d feature per
document serves as both streams of the XiT block, is concatenated behind the block's output, and goes through
out_layer = Mlp(1536, 3072, 768) and a Linear(768, 1) head; SmoothL1(beta=0.3) against the relevance label, AdamW,
per-batch scheduler.  Same kernels and engine schedule as stage 3 (`engine.xit_forward / xit_backward`, fused GEMM epilogues,
`lr2_smooth_l1`, `lr2_adamw_multi`); mode 'reg' only.  `LTRDataset` (pointwise_trad.py:88-109) reads the LETOR `train.h5` /
`test.h5` files (`finetune/letor.py`); `SyntheticLTR` provides data of its shapes.  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn
from torch.utils.data import Dataset

from .. import engine, ops, runtime
from ..tencentpretrain.utils.optimizers import str2optimizer, str2scheduler
from .letor import QueryRows
from .ppo import FEAT, Mlp, _grouped, _init_normal
from .xit import XiT

OUT_FC1, OUT_FC2 = "out_layer.fc1.weight", "out_layer.fc2.weight"


class Classifier(nn.Module):
    """pointwise_trad.py:132-177.  forward(text_emb [bs, docs, 768], img_emb (ignored, as upstream), tgts) ->
    (loss, logits [bs*docs, 1]) or logits."""

    def __init__(self, args, vit_args=None):
        super().__init__()
        self.mode, self.labels_num = args.mode, args.labels_num
        if self.mode != "reg":
            raise NotImplementedError("the HIP path implements mode='reg'")
        self.xit = XiT(feat_size=FEAT)
        self.out_layer = Mlp(2 * FEAT, 4 * FEAT, FEAT, nn.GELU, 0)
        self.head = nn.Linear(FEAT, 1)
        self._ws: Optional[engine.Workspace] = None
        self._wp: Optional[engine.WeightPlanes] = None
        self._G: Optional[Dict[str, torch.Tensor]] = None
        self._saved = None

    # ---- plumbing ----
    def _P(self):
        return {n: p.data for n, p in self.named_parameters()}

    def _weights(self, P):
        if self._wp is None or not self._wp.matches(P):
            self._wp = engine.WeightPlanes(P, engine.XIT.gemm_weights() + [OUT_FC1, OUT_FC2],
                                           transposed=[engine.XIT.f1_w, OUT_FC1])
        self._wp.refresh()
        return self._wp.planes

    def grad_buffers(self):
        dev = next(self.parameters()).device
        if self._G is None or next(iter(self._G.values())).device != dev:
            self._G = {n: torch.zeros_like(p) for n, p in self.named_parameters()}
        return self._G

    def bind_grads(self):
        for n, p in self.named_parameters():
            p.grad = self.grad_buffers()[n]

    # ---- engine schedule ----
    @torch.no_grad()
    def engine_forward(self, text_emb, *, save: bool):
        if text_emb.dtype != torch.float32 or not text_emb.is_cuda:
            raise TypeError("lr2ppo_amd: text_emb must be a float32 tensor on the HIP device (no CPU path)")
        if text_emb.dim() != 3 or text_emb.shape[-1] != FEAT:
            raise ValueError(f"text_emb must be [bs, docs, {FEAT}] (pointwise_trad.py:146-152)")
        dev = text_emb.device
        if self._ws is None or self._ws.device != dev:
            self._ws = engine.Workspace(dev)
        ws, P = self._ws, self._P()
        W = self._weights(P)
        N, E = text_emb.shape[0] * text_emb.shape[1], FEAT
        x0 = text_emb.contiguous().view(N, E)
        drop = runtime.next_drop(engine.DROP_P, 0) if self.training else None
        g2 = engine.trad_trunk_forward(ws, P, W, x0, N, E, save=save, drop=drop)
        logits = torch.empty(N, device=dev)
        ops.head_fwd(g2, P["head.weight"], P["head.bias"], logits, rows=N, D=E)
        if save:
            self._saved = (x0, N, drop)
            self._in_shape = tuple(text_emb.shape)
        return logits.view(-1, 1)

    @torch.no_grad()
    def engine_backward(self, dlogits, input_grads: bool = False):
        """input_grads: -> d text_emb (a fresh tensor in the forward's shape), for a caller that trains what produced the
        features; else None."""
        x0, N, drop = self._saved
        ws, P, G = self._ws, self._P(), self.grad_buffers()
        W = self._wp.planes
        E = FEAT
        g2 = ws.mat("g2", N, E)
        dg2 = ws.mat("dg2", N, E)
        ops.head_bwd(g2, P["head.weight"], dlogits.contiguous().view(-1), dg2, G["head.weight"], G["head.bias"], rows=N, D=E)
        dx0 = engine.trad_trunk_backward(ws, P, W, G, x0, dg2, N, E, drop=drop, want_dx=input_grads)
        self._saved = None
        return dx0.clone().view(self._in_shape) if input_grads else None

    def forward(self, text_emb, img_emb=None, tgts=None):
        if torch.is_grad_enabled() and (text_emb.requires_grad or any(p.requires_grad for p in self.parameters())):
            logits = _TradFn.apply(self, text_emb, *list(self.parameters()))
        else:
            logits = self.engine_forward(text_emb, save=False)
        if tgts is None:
            return logits
        return nn.SmoothL1Loss(beta=0.3)(logits.view(-1), tgts.view(-1).to(torch.float32)), logits


class _TradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, text_emb, *params):
        ctx.model = model
        return model.engine_forward(text_emb, save=True)

    @staticmethod
    def backward(ctx, dlogits):
        m = ctx.model
        dx = m.engine_backward(dlogits.contiguous(), input_grads=ctx.needs_input_grad[1])    # the gradient, or a raise: never a silent None
        G = m.grad_buffers()
        unused = tuple(getattr(m, "_unused_prefixes", ()))      # pointwise_2data_trad: the projection this batch did not go through
        return (None, dx) + tuple(G[n].clone() if (p.requires_grad and not n.startswith(unused)) else None
                                    for n, p in m.named_parameters())


def load_or_initialize_parameters(args, model):
    """pointwise_trad.py:179-211 (see lr2ppo_amd/finetune/pointwise.py for the strict=False quirk)."""
    if getattr(args, "pretrained_model_path", None) is not None:
        model.load_state_dict(torch.load(args.pretrained_model_path, map_location="cpu"), strict=False)
    else:
        _init_normal(model)


def build_optimizer(args, model):
    """pointwise_trad.py:214-237."""
    if args.optimizer not in str2optimizer:
        raise NotImplementedError(f"optimizer {args.optimizer!r}: only adamw is on the HIP path")
    optimizer = str2optimizer[args.optimizer](_grouped(list(model.named_parameters())), lr=args.learning_rate,
                                              correct_bias=False)
    if args.scheduler in ["constant"]:
        scheduler = str2scheduler[args.scheduler](optimizer)
    elif args.scheduler in ["constant_with_warmup"]:
        scheduler = str2scheduler[args.scheduler](optimizer, args.train_steps * args.warmup)
    else:
        scheduler = str2scheduler[args.scheduler](optimizer, args.train_steps * args.warmup, args.train_steps)
    return optimizer, scheduler


def train_model(args, model, optimizer, scheduler, text_emb_batch, img_emb_batch, tgts_batch):
    """One batch (pointwise_trad.py:240-255) on the engine path -> loss (0-dim device tensor)."""
    dev = text_emb_batch.device
    model.bind_grads()
    logits = model.engine_forward(text_emb_batch, save=True).view(-1)
    target = tgts_batch.to(device=dev, dtype=torch.float32).contiguous().view(-1)
    loss, dlogits = torch.empty(1, device=dev), torch.empty_like(logits)
    ops.smooth_l1(logits, target, loss, dlogits, n=logits.numel(), beta=0.3)
    model.engine_backward(dlogits)
    optimizer.step()
    scheduler.step()
    return loss[0]


class LTRDataset(QueryRows):
    """pointwise_trad.py:88-109: LTRDataset(args, path, is_train) over `path`/train.h5 | test.h5, one item per query."""


class SyntheticLTR(Dataset):
    """Seeded stand-in with LTRDataset's item shape (pointwise_trad.py:88-109): 20 documents per query, 768-d features,
    labels in {0, 1, 2}."""

    def __init__(self, n_queries, docs=20, seed=7):
        self.n, self.docs, self.seed = n_queries, docs, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        return torch.randn(self.docs, FEAT, generator=g), torch.zeros(1), torch.randint(0, 3, (self.docs,), generator=g)
