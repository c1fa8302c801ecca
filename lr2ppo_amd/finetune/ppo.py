"""LR2PPO stage 3 on MI355X -- drop-in for the reference's finetune/ppo.py.

Same public surface (names, argument meaning, state_dict keys, CLI flags, returned metric order):
`RankLoss, Mlp, ActorCritic, Actor, Critic, Reward, build_optimizer, clipped_value_loss, train_model,
evaluate, get_dataloader, main`.  What differs is underneath: the model forwards, their backward, the PPO
loss and the AdamW step run on the hand-written gfx950 kernels of lr2ppo_amd/csrc through
lr2ppo_amd.engine (explicit schedule, persistent workspace, no autograd graph on the training path).
There is no CPU fallback: without the native library every forward raises.

Reference behaviours reproduced on purpose (SURVEY.md 8a "quirks"): one-step advantage A = (r - w_kl KL) - V_old
with r' NOT detached in the policy loss; target order flipped when A < -0.1; batch-level RankLoss scalar;
`--eps_clip` parsed and unused; schedulers stepped once per train_model call (first cycle at lr 0);
Reward hard-codes 4 positions; Critic/Reward read the LAST position; decay exemption by substring
bias|gamma|beta (so LayerNorm weights and pos_emb are decayed).
Deliberate deviations: (1) gradients are averaged across ranks before each optimizer step (the reference
trains independent replicas); (2) NaN loss raises FloatingPointError on every rank instead
of dropping rank 0 into pdb; (3) trunk work that is provably identical is not repeated (shared image tokens,
duplicate tags in the reward model's index) -- results are bit-identical.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import random
import sys
import time as _time
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from .. import engine, h5lite, ops, runtime
from ..ndcg import AverageNDCGMeter, ndcg_rows
from ..tencentpretrain.model_saver import save_model
from ..tencentpretrain.opts import adv_opts, finetune_opts, tokenizer_opts
from ..tencentpretrain.utils.config import load_hyperparam
from ..tencentpretrain.utils.logging import init_logger
from ..tencentpretrain.utils.optimizers import str2optimizer, str2scheduler
from . import misc
from .xit import XiT

FEAT = 768        # reference hard-codes 768 / 196 (finetune/ppo.py:202-208,219-220)
SEQ_LEN = engine.SEQ_LEN


# ---------------------------------------------------------------------------------------------
# small host-side helpers of the reference's API (finetune/ppo.py:38-55, 422-498)
# ---------------------------------------------------------------------------------------------
class RankLoss(nn.Module):
    """Pairwise hinge over a target order; mean over the POSITIVE entries of the whole batch.
    Stand-alone module for API parity; train_model uses the fused lr2_ppo_loss kernel instead."""

    def __init__(self, margin=1):
        super().__init__()
        self.margin = margin

    def forward(self, scores, indices):
        s = torch.gather(scores, 1, indices)
        hinge = torch.relu(torch.triu(self.margin - (s.unsqueeze(2) - s.unsqueeze(1)), diagonal=1))
        total, cnt = hinge.sum(), torch.sign(hinge).sum()
        return total if cnt == 0 else total / cnt


def log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))


def log_prob(prob):
    return log(prob.max(dim=-1).values)


def masked_entropy(prob, dim=-1, mask=None):
    return (prob * log(prob)).sum(dim=-1)


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else (d() if callable(d) else d)


def masked_mean(seq, mask=None, dim=1, keepdim=False):
    if mask is None:
        return seq.mean(dim=dim)
    if seq.ndim == 3:
        mask = mask.unsqueeze(-1)
    numer = seq.masked_fill(~mask, 0.0).sum(dim=dim, keepdim=keepdim)
    denom = mask.sum(dim=dim, keepdim=keepdim)
    return (numer / denom.clamp(min=1e-3)).masked_fill(denom == 0, 0.0)


def masked_kl_div(prob1, prob2, mask=None, reduce_batch=False):
    kl = (prob1 * (log(prob1) - log(prob2))).sum(dim=-1)
    return kl.mean() if reduce_batch else kl


def masked_normalize(t, eps=1e-5, mask=None, dim=None):
    centred = t - t.mean()
    return centred * (centred ** 2).mean().clamp(min=eps).rsqrt()


def clipped_value_loss(values, rewards, old_values, clip):
    clipped = old_values + (values - old_values).clamp(-clip, clip)
    return torch.mean(torch.max((clipped.flatten() - rewards) ** 2, (values.flatten() - rewards) ** 2))


def get_inds(indices, tgts, cls):
    return [i for i, t in zip(indices, tgts) if t == cls]


def freeze_layer(layer):
    for p in layer.parameters():
        p.requires_grad = False


# ---------------------------------------------------------------------------------------------
# models
# ---------------------------------------------------------------------------------------------
class Mlp(nn.Module):
    """fc1 -> GELU(erf) -> fc2 parameter holder (finetune/ppo.py:154-170; drop is 0 everywhere upstream).
    Stand-alone calls run two fused GEMMs; inside the heads the engine schedules them."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        if drop:
            raise NotImplementedError("Mlp dropout is 0 in every reference configuration")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    @torch.no_grad()
    def forward(self, x):
        K, F, Nout = self.fc1.in_features, self.fc1.out_features, self.fc2.out_features
        x2 = x.contiguous().view(-1, K)
        M = x2.shape[0]
        ws = engine.Workspace(x.device)
        h = torch.empty(M, F, device=x.device)
        out = torch.empty(M, Nout, device=x.device)
        engine.linear_fwd(ws, x2, self.fc1.weight.data, self.fc1.bias.data, h, M, F, K, act=1)
        engine.linear_fwd(ws, h, self.fc2.weight.data, self.fc2.bias.data, out, M, Nout, F)
        return out.view(*x.shape[:-1], Nout)


class _Head(nn.Module):
    """Shared machinery of Actor / Critic / Reward: parameters, flat gradient buffer, workspace, engine calls.
    TRAD = True (finetune/ppo_trad.py here and upstream, :142-281): the same heads at sequence length 1 -- no text_proj /
    img_proj, the [bs, tags, 768] document feature is both streams of the XiT block and is concatenated behind its output,
    out_layer = Mlp(2 * 768, 3072, 768); img_emb is ignored."""
    has_tail = False
    fixed_positions: Optional[int] = None
    TRAD = False

    def __init__(self, args, vit_args=None):
        super().__init__()
        self.mode = args.mode
        self.labels_num = args.labels_num
        if self.mode not in ("reg", "cls"):
            raise ValueError(f"mode must be 'reg' or 'cls' (finetune/ppo.py:209-212), got {self.mode!r}")
        if not self.TRAD:
            if args.visual_feat_dim != FEAT:
                raise ValueError("visual_feat_dim must be 768 (hard-coded in the reference, finetune/ppo.py:202-208)")
            self.seq_length, self.max_imgs = args.seq_length, args.max_imgs
            if self.seq_length != SEQ_LEN:
                raise ValueError("seq_length must be 196 (hard-coded in the reference, finetune/ppo.py:219-220)")
            self.text_proj = Mlp(FEAT, FEAT * 4, FEAT, nn.GELU, 0)
            self.img_proj = Mlp(FEAT, FEAT * 4, FEAT, nn.GELU, 0)
        if self.has_tail:
            self.pos_emb = nn.Embedding(4, FEAT)
        self.xit = XiT(feat_size=FEAT)
        if self.has_tail:
            self.xitt = XiT(feat_size=FEAT, attention_mask="causal")
        if self.TRAD:
            self.out_layer = Mlp(2 * FEAT, FEAT * 4, FEAT, nn.GELU, 0)               # ppo_trad.py:150
        else:
            self.out_layer = Mlp((args.seq_length + args.max_imgs) * args.visual_feat_dim, FEAT * 4, FEAT, nn.GELU, 0)
        # 'cls': the ACTOR scores through a labels_num-way classifier (finetune/ppo.py:209-210); Critic / Reward keep 768 -> 1
        self.n_out = self.labels_num if (self.mode == "cls" and not self.has_tail) else 1
        if self.n_out > 8:
            raise ValueError("mode='cls' supports at most 8 labels (lr2_cls_head_fwd)")
        self.head = nn.Linear(FEAT, self.n_out)
        self._P_cache: Optional[Dict[str, torch.Tensor]] = None
        self._P_sig = []
        self._ws: Optional[engine.Workspace] = None
        self._wp: Optional[engine.WeightPlanes] = None
        self._G: Optional[Dict[str, torch.Tensor]] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._saved = None

    # ---- plumbing ----
    def _workspace(self, device) -> engine.Workspace:
        if self._ws is None or self._ws.device != device:
            self._ws = engine.Workspace(device)
        return self._ws

    def _P(self) -> Dict[str, torch.Tensor]:
        """name -> parameter storage.  Cached: walking named_parameters() costs ~1 ms per call and a PPO step makes seven.  The
        cache is checked against EVERY parameter's current storage address (one data_ptr() per parameter, ~5 us), so anything
        that re-seats a parameter -- .to() / .cuda(), load_state_dict(assign=True), `p.data = ...` -- rebuilds it."""
        c = self._P_cache
        if c is not None:
            for p, ptr in self._P_sig:
                if p.data_ptr() != ptr:
                    c = None
                    break
        if c is None:
            named = list(self.named_parameters())
            c = self._P_cache = {n: p.data for n, p in named}
            self._P_sig = [(p, p.data_ptr()) for _, p in named]
        return c

    def _apply(self, fn, *a, **kw):
        self._P_cache = None
        out = super()._apply(fn, *a, **kw)
        self._P_cache = None
        return out

    def _weights(self, P, refresh: bool = True) -> Dict[str, ops.Planes]:
        """bf16 hi/lo planes of the token-GEMM weights (everything but the 2 GB out_layer.fc1), re-split from the
        fp32 parameters at the start of every forward."""
        if self._wp is None or not self._wp.matches(P):
            if self.TRAD:
                names = engine.XIT.gemm_weights() + [engine.TRAD_FC1, engine.TRAD_FC2]
                tnames = [engine.XIT.f1_w, engine.TRAD_FC1]
            else:
                names, tnames = list(engine.TRUNK_GEMM_WEIGHTS), list(engine.TRUNK_T_WEIGHTS)
            if self.has_tail:
                names, tnames = names + engine.XITT.gemm_weights(), tnames + [engine.XITT.f1_w]
            self._wp = engine.WeightPlanes(P, names, transposed=tnames)
            refresh = True
        if refresh:
            self._wp.refresh()
        return self._wp.planes

    GRAD_ORDER_FIRST = ("out_layer.fc1.weight",)

    def grad_buffers(self) -> Dict[str, torch.Tensor]:
        """Persistent gradient storage: one flat fp32 buffer with out_layer.fc1.weight (96 % of the bytes) in front.
        In data-parallel runs that block is never communicated (its rank-64 factors are, engine.trunk_backward);
        everything behind `_bucket_split` is averaged with ONE all-reduce."""
        dev = next(self.parameters()).device
        if self._G is not None and self._flat_grad.device == dev:
            return self._G
        named = dict(self.named_parameters())
        first = () if self.TRAD else self.GRAD_ORDER_FIRST     # `_trad`: no 2-GB matrix, every gradient goes through the all-reduce
        order = [n for n in first if n in named] + [n for n in named if n not in first]
        total = sum((named[n].numel() + 3) // 4 * 4 for n in order)
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self._G, off = {}, 0
        self._bucket_split = 0
        for n in order:
            k = named[n].numel()
            self._G[n] = self._flat_grad[off:off + k].view_as(named[n])
            off += (k + 3) // 4 * 4
            if first and n == first[-1]:
                self._bucket_split = off
        return self._G

    def bind_grads(self):
        """Point every parameter's .grad at its slice of the flat buffer (no copies, no per-step allocation)."""
        G = self.grad_buffers()
        for n, p in self.named_parameters():
            if p.grad is None or p.grad.data_ptr() != G[n].data_ptr():
                p.grad = G[n]

    def _prep_inputs(self, text_emb, img_emb):
        if text_emb.dtype != torch.float32 or not text_emb.is_cuda:
            raise TypeError("lr2ppo_amd: text_emb must be a float32 tensor on the HIP device (no CPU path)")
        bs, tags = text_emb.shape[:2]
        if text_emb.shape[2] != SEQ_LEN or text_emb.shape[3] != FEAT:
            raise ValueError(f"text_emb must be [bs, tags, {SEQ_LEN}, {FEAT}]")
        shared = engine._img_shared(img_emb)
        n_img = img_emb.shape[-2]
        if n_img != self.max_imgs:
            raise ValueError(f"img_emb carries {n_img} image tokens, model was built for max_imgs={self.max_imgs}")
        if shared:
            img2 = (img_emb if img_emb.dim() == 3 else img_emb[:, 0]).contiguous().view(bs * n_img, FEAT)
        else:
            img2 = img_emb.contiguous().view(bs * tags * n_img, FEAT)
        text2 = text_emb.contiguous().view(bs * tags * SEQ_LEN, FEAT)
        return engine.input_planes(text_emb, text2), engine.input_planes(img_emb, img2), bs, tags, n_img, shared

    def _prep_trad(self, text_emb):
        """[bs, tags, 768] document features -> ([bs*tags, 768] fp32 view, bs, tags) (ppo_trad.py:160-166)."""
        if text_emb.dtype != torch.float32 or not text_emb.is_cuda:
            raise TypeError("lr2ppo_amd: text_emb must be a float32 tensor on the HIP device (no CPU path)")
        if text_emb.dim() != 3 or text_emb.shape[-1] != FEAT:
            raise ValueError(f"text_emb must be [bs, tags, {FEAT}] (ppo_trad.py:160-166)")
        bs, tags = text_emb.shape[:2]
        return text_emb.contiguous().view(bs * tags, FEAT), bs, tags

    def _drop_cfg(self, site_base=0):
        return runtime.next_drop(engine.DROP_P, site_base) if self.training else None


class Actor(_Head):
    """finetune/ppo.py:196-244: forward(text_emb, img_emb, tgts) -> (loss, logits) or logits when tgts is None.
    mode 'reg': logits [bs*tags], SmoothL1(beta = 0.3); mode 'cls': logits [bs*tags, labels_num], NLL of log-softmax."""

    def forward(self, text_emb, img_emb, tgts=None):
        if torch.is_grad_enabled() and (_wants_grad(text_emb, img_emb) or any(p.requires_grad for p in self.parameters())):
            logits = _ActorFn.apply(self, text_emb, img_emb, *list(self.parameters()))
        else:
            logits = self.engine_forward(text_emb, img_emb, save=False)
        if tgts is None:
            return logits
        if self.mode == "cls":
            return _NllFn.apply(logits, tgts.reshape(-1).to(torch.int64).contiguous()), logits
        return _SmoothL1Fn.apply(logits, tgts.reshape(-1).to(torch.float32).contiguous()), logits

    def engine_forward(self, text_emb, img_emb, *, save: bool) -> torch.Tensor:
        ws, P = self._workspace(text_emb.device), self._P()
        if self.TRAD:
            x0, bs, tags = self._prep_trad(text_emb)
            W = self._weights(P)
            drop = self._drop_cfg(0)
            g2 = engine.trad_trunk_forward(ws, P, W, x0, bs * tags, FEAT, save=save, drop=drop)
            text2, img2, n_img, shared = x0, None, 0, False
        else:
            text2, img2, bs, tags, n_img, shared = self._prep_inputs(text_emb, img_emb)
            W = self._weights(P)
            drop = self._drop_cfg(0)
            g2 = engine.trunk_forward(ws, P, W, text2, img2, bs, tags, n_img, FEAT, save=save, drop=drop, img_shared=shared)
        if self.n_out == 1:
            logits = torch.empty(bs * tags, device=text_emb.device)
            ops.head_fwd(g2, P["head.weight"], P["head.bias"], logits, rows=bs * tags, D=FEAT)
        else:
            logits = torch.empty(bs * tags, self.n_out, device=text_emb.device)
            ops.cls_head_fwd(g2, P["head.weight"], P["head.bias"], logits, rows=bs * tags, D=FEAT, C=self.n_out)
        if save:
            self._saved = (text2, img2, bs, tags, n_img, shared, drop)
            self._in_shapes = (tuple(text_emb.shape), None if img_emb is None else tuple(img_emb.shape))
        return logits

    def engine_backward(self, dlogits: torch.Tensor, dp=None, fc1_update=None, input_grads: bool = False):
        """Gradients of sum(dlogits * logits) into the flat gradient buffer (call after engine_forward(save=True)).
        input_grads: -> (d text_emb, d img_emb) in the shapes the forward was given (what the reference's autograd passes on
        to a producer of the features, finetune/ppo.py:214-232); otherwise (None, None)."""
        text2, img2, bs, tags, n_img, shared, drop = self._saved
        P, G = self._P(), self.grad_buffers()
        ws, W = self._workspace(dlogits.device), self._weights(P, refresh=False)
        N = bs * tags
        g2 = ws.mat("g2", N, FEAT)
        dg2 = ws.mat("dg2", N, FEAT)
        if self.n_out == 1:
            ops.head_bwd(g2, P["head.weight"], dlogits.contiguous().view(-1), dg2, G["head.weight"], G["head.bias"], rows=N, D=FEAT)
        else:
            ops.cls_head_bwd(g2, P["head.weight"], dlogits.contiguous().view(N, self.n_out), dg2, G["head.weight"], G["head.bias"],
                             rows=N, D=FEAT, C=self.n_out)
        if self.TRAD:
            if fc1_update is not None:
                raise ValueError("the fused out_layer.fc1 update belongs to the 2-GB matrix of the full heads; pass fuse_fc1_update=False")
            dx0 = engine.trad_trunk_backward(ws, P, W, G, text2, dg2, N, FEAT, drop=drop, want_dx=input_grads)
            d_text, d_img = (dx0.clone().view(self._in_shapes[0]) if input_grads else None), None
        else:
            d_text, d_img = engine.trunk_backward(ws, P, W, G, text2, img2, dg2, bs, tags, n_img, FEAT, drop=drop,
                                                  img_shared=shared, dp=dp, fc1_update=fc1_update, input_grads=input_grads)
            if input_grads:
                d_text = d_text.view(self._in_shapes[0])
                d_img = _shape_img_grad(d_img, self._in_shapes[1], bs, n_img)
        self._saved = None
        return d_text, d_img

    def action_scores(self, logits: torch.Tensor, bs: int, tags: int, want_probs: bool = False):
        """The per-tag score the PPO loop ranks by: the logit itself ('reg'), or the expected label under softmax(logits)
        ('cls', finetune/ppo.py:532-537,859-863) -> (scores [bs, tags], probs [bs*tags, C] or None)."""
        if self.n_out == 1:
            return logits.view(bs, tags), None
        scores = torch.empty(bs * tags, device=logits.device)
        probs = torch.empty(bs * tags, self.n_out, device=logits.device) if want_probs else None
        ops.cls_scores(logits, probs, scores, rows=bs * tags, C=self.n_out, softmax=True)
        return scores.view(bs, tags), probs


class _TailHead(_Head):
    has_tail = True

    def forward(self, text_emb, img_emb, tgts, index):
        if torch.is_grad_enabled() and (_wants_grad(text_emb, img_emb) or any(p.requires_grad for p in self.parameters())):
            return _CriticFn.apply(self, text_emb, img_emb, index, *list(self.parameters()))
        return self.engine_forward(text_emb, img_emb, index, save=False)

    def _n_pos(self, t_out: int) -> int:
        if self.fixed_positions is not None:
            if t_out != self.fixed_positions:
                raise ValueError(f"Reward adds pos_emb(arange(4)) (finetune/ppo.py:339): index must have 4 columns, got {t_out}")
            return self.fixed_positions
        if t_out > 4:
            raise IndexError("pos_emb has 4 rows (finetune/ppo.py:256): at most 4 positions")
        return t_out

    @torch.no_grad()
    def trunk_no_grad(self, text_emb, img_emb):
        """The index-free part of the no-grad forward: every (item, tag) pair through the trunk once -> ([bs*tags, 768] in a
        workspace buffer, the dropout configuration drawn for this forward).  The trunk is per pair, so gathering ITS OUTPUT
        by `index` is bit-identical to gathering the inputs (finetune/ppo.py:267-271) -- and the rollout can run it before
        the actor has produced the ordering."""
        dev = text_emb.device
        bs, tags_in = text_emb.shape[:2]
        ws, P = self._workspace(dev), self._P()
        W = self._weights(P)
        n_img = img_emb.shape[-2]
        text2, img2, _, _, _, shared = self._prep_inputs(text_emb, img_emb)
        drop = self._drop_cfg(0)
        return engine.trunk_forward(ws, P, W, text2, img2, bs, tags_in, n_img, FEAT, save=False, drop=drop, img_shared=shared), drop

    def _gather_trunk(self, g2_all, index, bs, tags_in, t_out):
        g2 = self._workspace(g2_all.device).mat("g2_g", bs * t_out, FEAT)
        ops.gather_rows(g2_all, index, g2, B=bs, t_in=tags_in, t_out=t_out, row_elems=FEAT)
        return g2

    def engine_forward(self, text_emb, img_emb, index, *, save: bool, trunk_out=None) -> torch.Tensor:
        """trunk_out: the result of trunk_no_grad(text_emb, img_emb) when the caller already ran it (no-grad only)."""
        dev = text_emb.device
        bs, tags_in = text_emb.shape[:2]
        index = index.to(device=dev, dtype=torch.int64).contiguous()
        t_out = index.shape[1]
        self._n_pos(t_out)
        ws, P = self._workspace(dev), self._P()
        W = self._weights(P)
        n_img = 0 if self.TRAD else img_emb.shape[-2]
        if self.TRAD:
            # sequence length 1 (ppo_trad.py:214-232): gather the document features by index, then the trunk
            x_all, _, _ = self._prep_trad(text_emb)
            text_p = ws.mat("x_g", bs * t_out, FEAT)
            ops.gather_rows(x_all.view(bs, tags_in, FEAT), index, text_p.view(bs, t_out, FEAT), B=bs, t_in=tags_in, t_out=t_out,
                            row_elems=FEAT)
            img_p = None
            drop = self._drop_cfg(0)
            g2 = engine.trad_trunk_forward(ws, P, W, text_p, bs * t_out, FEAT, save=save, drop=drop)
            drop_t = drop.at(3) if drop else None
        elif save:
            # train mode: gather the inputs by index exactly like the reference (ppo.py:267-271), then run the trunk
            shared_in = engine._img_shared(img_emb)
            text_g = ws.mat("text_g", bs * t_out * SEQ_LEN, FEAT)
            ops.gather_rows(text_emb.contiguous(), index, text_g, B=bs, t_in=tags_in, t_out=t_out, row_elems=SEQ_LEN * FEAT)
            img_src = (img_emb if img_emb.dim() == 3 else img_emb[:, 0]).contiguous() if shared_in else img_emb.contiguous()
            img_g = ws.mat("img_g", bs * t_out * n_img, FEAT)
            ops.gather_rows(img_src, index, img_g, B=bs, t_in=tags_in, t_out=t_out, row_elems=n_img * FEAT,
                            src_bstride=(n_img * FEAT) if shared_in else tags_in * n_img * FEAT,
                            src_tstride=0 if shared_in else n_img * FEAT)
            text_p, img_p = ws.planes("text_gp", bs * t_out * SEQ_LEN, FEAT), ws.planes("img_gp", bs * t_out * n_img, FEAT)
            ops.split_planes(text_g, text_p)
            ops.split_planes(img_g, img_p)
            drop = self._drop_cfg(0)
            g2 = engine.trunk_forward(ws, P, W, text_p, img_p, bs, t_out, n_img, FEAT, save=True, drop=drop, img_shared=False)
            drop_t = drop.at(3) if drop else None
        else:
            g2_all, drop = trunk_out if trunk_out is not None else self.trunk_no_grad(text_emb, img_emb)
            g2 = self._gather_trunk(g2_all, index, bs, tags_in, t_out)
            drop_t = drop.at(3) if drop else None
        M = bs * t_out
        xin = ws.mat("xin", M, FEAT)
        ops.add_period_rows(g2, P["pos_emb.weight"], xin, rows=M, D=FEAT, period=t_out)
        xo = ws.mat("xo", M, FEAT)
        engine.xit_forward(ws, "xitt.", P, W, engine.XITT, xin, xin, bs, t_out, t_out, FEAT, xo, save=save, drop=drop_t)
        value = torch.empty(bs, device=dev)
        ops.head_fwd(xo, P["head.weight"], P["head.bias"], value, rows=bs, D=FEAT, row_step=t_out, row_off=t_out - 1)
        if save:
            self._saved = (text_p, img_p, bs, t_out, n_img, drop, drop_t)
            self._in_shapes = (tuple(text_emb.shape), None if img_emb is None else tuple(img_emb.shape), index, tags_in,
                               None if self.TRAD else engine._img_shared(img_emb))
        return value

    def engine_backward(self, dvalue: torch.Tensor, dp=None, fc1_update=None, fc1_early: bool = False,
                        input_grads: bool = False):
        """input_grads: -> (d text_emb, d img_emb) in the forward's input shapes: the trunk's input gradients scattered back
        through the `index` gather of finetune/ppo.py:267-271 (a tag picked twice receives the sum); else (None, None)."""
        text_g, img_g, bs, t_out, n_img, drop, drop_t = self._saved
        P, G = self._P(), self.grad_buffers()
        ws, W = self._workspace(dvalue.device), self._weights(P, refresh=False)
        M = bs * t_out
        xo, xin = ws.mat("xo", M, FEAT), ws.mat("xin", M, FEAT)
        dxo = ws.mat("dxo", M, FEAT)
        ops.head_bwd(xo, P["head.weight"], dvalue.contiguous().view(-1), dxo, G["head.weight"], G["head.bias"], rows=bs,
                     D=FEAT, row_step=t_out, row_off=t_out - 1, total_rows=M)
        dxin = ws.mat("dxin", M, FEAT)
        engine.xit_backward(ws, "xitt.", P, W, G, engine.XITT, xin, xin, dxo, bs, t_out, t_out, FEAT, dxin, None,
                            drop=drop_t, same_xy=True)
        G["pos_emb.weight"].zero_()
        ops.period_rows_grad(dxin, G["pos_emb.weight"], rows=M, D=FEAT, period=t_out)
        if self.TRAD:
            if fc1_update is not None:
                raise ValueError("the fused out_layer.fc1 update belongs to the 2-GB matrix of the full heads; pass fuse_fc1_update=False")
            dx0 = engine.trad_trunk_backward(ws, P, W, G, text_g, dxin, M, FEAT, drop=drop, want_dx=input_grads)
            d_text = d_img = None
            if input_grads:
                t_shape, _, index, tags_in, _ = self._in_shapes
                d_text = torch.empty(t_shape, device=dx0.device)
                ops.gather_rows_bwd(dx0.view(bs, t_out, FEAT), index, d_text, B=bs, t_in=tags_in, t_out=t_out, row_elems=FEAT)
        else:
            d_text, d_img = engine.trunk_backward(ws, P, W, G, text_g, img_g, dxin, bs, t_out, n_img, FEAT, drop=drop,
                                                  img_shared=False, dp=dp, fc1_update=fc1_update, fc1_early=fc1_early,
                                                  input_grads=input_grads)
            if input_grads:
                t_shape, i_shape, index, tags_in, shared_in = self._in_shapes
                dev = d_text.device
                dt = torch.empty(t_shape, device=dev)
                ops.gather_rows_bwd(d_text.view(bs, t_out, SEQ_LEN * FEAT), index, dt, B=bs, t_in=tags_in, t_out=t_out,
                                    row_elems=SEQ_LEN * FEAT)
                if shared_in:         # every position reads the item's one set of image tokens: sum over positions
                    di = torch.empty(bs, n_img, FEAT, device=dev)
                    ops.gather_rows_bwd(d_img.view(bs, t_out, n_img * FEAT), torch.zeros(bs, t_out, dtype=torch.int64, device=dev),
                                        di, B=bs, t_in=1, t_out=t_out, row_elems=n_img * FEAT)
                    di = _shape_img_grad(di.view(bs * n_img, FEAT), i_shape, bs, n_img)
                else:
                    di = torch.empty(i_shape, device=dev)
                    ops.gather_rows_bwd(d_img.view(bs, t_out, n_img * FEAT), index, di, B=bs, t_in=tags_in, t_out=t_out,
                                        row_elems=n_img * FEAT)
                d_text, d_img = dt, di
        self._saved = None
        return d_text, d_img


class Critic(_TailHead):
    """finetune/ppo.py:247-297: value of the ordering `index` -> [bs] (head output at the last position)."""


class Reward(_TailHead):
    """finetune/ppo.py:300-350: as Critic, with pos_emb(arange(4)) hard-coded (index must be [bs, 4])."""
    fixed_positions = 4


class ActorCritic(nn.Module):
    """finetune/ppo.py:173-193."""

    def __init__(self, args, vit_args=None):
        super().__init__()
        self.actor = Actor(args, vit_args)
        self.critic = Critic(args, vit_args)

    def enable_actor(self):
        for p in self.actor.parameters():
            p.requires_grad = True

    def disable_actor(self):
        for p in self.actor.parameters():
            p.requires_grad = False

    def enable_critic(self):
        for p in self.critic.parameters():
            p.requires_grad = True

    def disable_critic(self):
        for p in self.critic.parameters():
            p.requires_grad = False


class _SmoothL1Fn(torch.autograd.Function):
    """nn.SmoothL1Loss(beta=0.3) of finetune/ppo.py:236 (mean reduction), differentiable w.r.t. the logits."""

    @staticmethod
    def forward(ctx, logits, targets):
        loss = torch.empty(1, device=logits.device)
        dpred = torch.empty_like(logits) if logits.requires_grad else None
        ops.smooth_l1(logits.detach().contiguous(), targets, loss, dpred, n=logits.numel(), beta=0.3)
        ctx.dpred = dpred
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        return (ctx.dpred * dloss if ctx.dpred is not None else None), None


class _NllFn(torch.autograd.Function):
    """nn.NLLLoss()(nn.LogSoftmax(dim=-1)(logits), tgts) of finetune/ppo.py:239-241 (mean), differentiable w.r.t. the logits."""

    @staticmethod
    def forward(ctx, logits, targets):
        loss = torch.empty(1, device=logits.device)
        dl = torch.empty_like(logits) if logits.requires_grad else None
        ops.nll_loss(logits.detach().contiguous(), targets, loss, dl, rows=logits.shape[0], C=logits.shape[1])
        ctx.dl = dl
        return loss[0]

    @staticmethod
    def backward(ctx, dloss):
        return (ctx.dl * dloss if ctx.dl is not None else None), None


def _wants_grad(*tensors) -> bool:
    return any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def _shape_img_grad(d_img: torch.Tensor, shape, bs: int, n_img: int) -> torch.Tensor:
    """d_img [bs*n_img or bs*tags*n_img, E] -> the shape of the img_emb the forward was given.  A stride-0 expand over tags
    ([bs, tags, n_img, E] sharing one set of rows) gets the whole gradient in its tag-0 slice: autograd sums the slices."""
    rows = 1
    for d in shape[:-1]:
        rows *= d
    if rows == d_img.shape[0]:
        return d_img.view(shape)
    full = torch.zeros(shape, dtype=d_img.dtype, device=d_img.device)
    full[:, 0] = d_img.view(bs, n_img, shape[-1])
    return full


def _param_grads(ctx, mod, first_param_arg: int):
    """Gradients for the parameter arguments of the coarse autograd nodes: a copy of each flat-buffer slice for parameters that
    require grad (the buffer is reused by the next backward), None for frozen ones."""
    G = mod.grad_buffers()
    return [G[n].clone() if ctx.needs_input_grad[first_param_arg + i] else None for i, (n, _) in enumerate(mod.named_parameters())]


class _ActorFn(torch.autograd.Function):
    """Autograd entry of the drop-in nn.Module path: `loss.backward()` works as with the reference -- parameters AND the two
    feature inputs receive gradients (the reference's heads are plain autograd, finetune/ppo.py:214-232)."""

    @staticmethod
    def forward(ctx, mod, text_emb, img_emb, *params):
        ctx.mod = mod
        with torch.no_grad():
            return mod.engine_forward(text_emb, img_emb, save=True)

    @staticmethod
    def backward(ctx, dlogits):
        mod = ctx.mod
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        d_text, d_img = mod.engine_backward(dlogits, input_grads=want)
        return (None, d_text if ctx.needs_input_grad[1] else None, d_img if ctx.needs_input_grad[2] else None,
                *_param_grads(ctx, mod, 3))


class _CriticFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, text_emb, img_emb, index, *params):
        ctx.mod = mod
        with torch.no_grad():
            return mod.engine_forward(text_emb, img_emb, index, save=True)

    @staticmethod
    def backward(ctx, dvalue):
        mod = ctx.mod
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        d_text, d_img = mod.engine_backward(dvalue, input_grads=want)
        return (None, d_text if ctx.needs_input_grad[1] else None, d_img if ctx.needs_input_grad[2] else None, None,
                *_param_grads(ctx, mod, 4))


# ---------------------------------------------------------------------------------------------
# parameter init / optimizer (finetune/ppo.py:358-419)
# ---------------------------------------------------------------------------------------------
def _init_normal(model, generator=None):
    for n, p in list(model.named_parameters()):
        if "gamma" not in n and "beta" not in n:
            p.data.normal_(0, 0.02, generator=generator)


def load_or_initialize_parameters(args, model):
    if getattr(args, "pretrained_model_path", None) is not None:
        model.load_state_dict(torch.load(args.pretrained_model_path, map_location="cpu"), strict=True)
    else:
        _init_normal(model)


def load_or_initialize_parameters_reward(args, model):
    if getattr(args, "reward_model_path", None) is not None:
        model.load_state_dict(torch.load(args.reward_model_path, map_location="cpu"), strict=True)
    else:
        _init_normal(model)


def _grouped(named):
    no_decay = ["bias", "gamma", "beta"]
    return [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
            {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]


def build_optimizer(args, model):
    if args.optimizer not in str2optimizer:
        raise NotImplementedError(f"optimizer {args.optimizer!r}: only adamw is on the HIP path (every LR2PPO launcher uses it)")
    optimizer = str2optimizer[args.optimizer](_grouped(list(model.actor.named_parameters())), lr=args.learning_rate,
                                              correct_bias=False)
    critic_optimizer = str2optimizer[args.optimizer](_grouped(list(model.critic.named_parameters())),
                                                     lr=args.critic_learning_rate, correct_bias=False)
    if args.scheduler in ["constant"]:
        scheduler = str2scheduler[args.scheduler](optimizer)
        critic_scheduler = str2scheduler[args.scheduler](critic_optimizer)
    elif args.scheduler in ["constant_with_warmup"]:
        scheduler = str2scheduler[args.scheduler](optimizer, args.train_steps * args.warmup)
        critic_scheduler = str2scheduler[args.scheduler](critic_optimizer, args.train_steps * args.warmup)
    else:
        scheduler = str2scheduler[args.scheduler](optimizer, args.train_steps * args.warmup, args.train_steps)
        critic_scheduler = str2scheduler[args.scheduler](critic_optimizer, args.train_steps * args.warmup, args.train_steps)
    return optimizer, critic_optimizer, scheduler, critic_scheduler


# ---------------------------------------------------------------------------------------------
# rollout + update  (finetune/ppo.py:501-617, 844-883)
# ---------------------------------------------------------------------------------------------
class _Side:
    """The critic's HIP stream beside the actor's.  Actor and critic share nothing between the inputs and the PPO loss, and
    nothing again between the loss and the end of the step, so their launches may interleave: the critic's HBM-bound
    out_layer.fc1 passes (forward 2 GB, fused gradient + AdamW 12 GB) run beside the actor's MFMA-bound token GEMMs and the
    other way round, and partial last rounds of one model's launches are filled by the other's.  Same kernels, same order
    per model, same bits.  LR2_PPO_STREAMS=0 puts everything back on one stream."""
    _streams: Dict[int, "torch.cuda.Stream"] = {}

    def __init__(self, device):
        env = os.environ.get("LR2_PPO_STREAMS")
        # data parallel: the critic's collectives would be issued from the side stream.  Their order on RCCL's stream is the host
        # issue order on every rank (no cycle), but that schedule has run on no multi-GPU hardware yet: the default
        # with more than one rank is ONE stream, LR2_PPO_STREAMS=1 opts in.
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.on = (env == "1" if multi else env != "0") and device.type == "cuda"
        if self.on:
            key = device.index if device.index is not None else torch.cuda.current_device()
            if key not in _Side._streams:
                _Side._streams[key] = torch.cuda.Stream(device=device)
            self.side = _Side._streams[key]
            self.main = torch.cuda.current_stream(device)

    def fork_point(self):
        """Work issued on the side stream from now on starts after everything issued on the main stream SO FAR (and not after
        what the main stream is given later)."""
        if self.on:
            self.side.wait_stream(self.main)

    def run(self):
        """-> context manager: work issued inside goes to the side stream."""
        return torch.cuda.stream(self.side) if self.on else contextlib.nullcontext()

    def join(self):
        """What follows on the main stream sees everything issued on the side stream so far."""
        if self.on:
            self.main.wait_stream(self.side)


@torch.no_grad()
def rollout_step(model, reward_model, text_emb, img_emb, tgts, state=None):
    """One timestep of the rollout loop (finetune/ppo.py:844-883) -> the 8-entry memory record."""
    bs, tags = text_emb.shape[:2]
    dev = text_emb.device
    if state is None:
        state = torch.arange(tags, device=dev).unsqueeze(0).repeat(bs, 1)
    # the critic beside the actor and the reward model (host order stays actor, critic, reward: a train-mode caller draws its
    # dropout seeds in call order).  A third stream for the reward model's trunk, which needs the actor's ordering only at its
    # end, was measured: no further gain (18.35 vs 18.20 ms per step).
    side = _Side(dev)
    if not model.actor.TRAD:
        # the input planes all three models share are produced HERE, on the main stream and before the fork: the critic's
        # first GEMMs on the side stream read them through the cache, and a split launched by the actor after the fork would
        # be invisible to that stream (a NEW batch tensor every step -- the training loop -- is exactly that case)
        model.actor._prep_inputs(text_emb, img_emb)
    side.fork_point()
    logits = model.actor.engine_forward(text_emb, img_emb, save=False)
    with side.run():
        value = model.critic.engine_forward(text_emb, img_emb, state, save=False)
    scores, _ = model.actor.action_scores(logits, bs, tags)
    _, order = torch.sort(scores, dim=-1, descending=True)
    next_state = torch.cat([torch.arange(2, device=dev).unsqueeze(0).repeat(bs, 1), torch.gather(state, 1, order)], dim=1)
    rewards = reward_model.engine_forward(text_emb, img_emb, next_state, save=False)
    side.join()
    return [state, next_state, scores.clone(), rewards, value, text_emb, img_emb, tgts]


class _DoneWork:
    def wait(self):
        return True


class _DataParallel:
    """Gradient exchange of the north-star data-parallel mode (the reference trains independent replicas,
    finetune/ppo.py has no DDP).  Per model and minibatch:
      * out_layer.fc1.weight (2 GB of the 2.08 GB of gradients) is NOT all-reduced: its gradient is the product of two
        thin factors, dW = dZ^T X with dZ [N,3072] and X [N,162816] per rank, so the ranks all-gather the factors
        (42 MB each, as bf16 hi/lo planes) and every rank multiplies the concatenation locally (engine.trunk_backward);
      * everything else (19-26 M parameters) is averaged with ONE all-reduce over the tail of the flat gradient buffer.
    Collectives are issued with async_op on RCCL's stream and waited for right before their consumer, so they overlap
    the rest of backward."""

    def __init__(self):
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if inited else 1
        self.backend = dist.get_backend() if inited else None
        # LR2_DP_FORCE=1: take the exchange path even with ONE rank (factor all-gathers, tail all-reduce, the global RankLoss
        # statistics, the K = 64 * world fused update with alpha = 1 / world): every collective is the identity there, so the
        # step must reproduce the plain one bit for bit -- the way RCCL and this path are exercised on a one-GPU box
        # (tests/test_rccl_world1_gpu.py).  Never set in production.
        self._forced = inited and os.environ.get("LR2_DP_FORCE", "0") == "1"

    @property
    def active(self) -> bool:
        """True when the gradient-exchange path runs: more than one rank (or forced, see __init__)."""
        return self.world > 1 or self._forced

    def _all_gather(self, out: torch.Tensor, inp: torch.Tensor):
        out, inp = out.view(torch.uint8), inp.view(torch.uint8)   # raw bytes: int16 is not a NCCL/gloo element type
        if self.backend == "nccl":
            return dist.all_gather_into_tensor(out, inp, async_op=True)
        if inp.is_cuda:      # gloo has no device all_gather: stage through the host (rehearsals of the N > 1 path only)
            host = [torch.empty(inp.numel(), dtype=torch.uint8) for _ in range(self.world)]
            dist.all_gather(host, inp.cpu())
            out.copy_(torch.cat(host).to(out.device))
            return _DoneWork()
        return dist.all_gather(list(out.view(self.world, -1).unbind(0)), inp, async_op=True)

    def gather_planes_start(self, pl: ops.Planes, ws, name: str):
        """Start all-gathering a [rows, cols] planes tensor along rows -> ([world*rows, cols] planes, handles)."""
        n = pl.rows * pl.cols
        out = ws.planes(name, pl.rows * self.world, pl.cols)
        works = [self._all_gather(out.buf[:n * self.world], pl.buf[:n]),
                 self._all_gather(out.buf[out.lo_off:out.lo_off + n * self.world], pl.buf[pl.lo_off:pl.lo_off + n])]
        return out, works

    def gather_planes_finish(self, pending):
        out, works = pending
        for w in works:
            w.wait()
        return out

    def reduce_start(self, head: "_Head"):
        """Average every gradient except out_layer.fc1.weight (handled through its factors)."""
        if not self.active:
            return None
        tail = head._flat_grad[head._bucket_split:]
        tail.div_(self.world)
        return dist.all_reduce(tail, async_op=True)

    @staticmethod
    def finish(work):
        if work is not None:
            work.wait()

    def reduce(self, head: "_Head"):
        self.finish(self.reduce_start(head))


def update_minibatch(args, model, optimizer, critic_optim, record, dp=None, input_grads: bool = False):
    """The body of one train_model iteration (finetune/ppo.py:518-598) on one stored rollout record:
    actor + critic train forwards, fused PPO loss, actor backward + AdamW, critic backward + AdamW.
    Returns the 10 logged metrics of this minibatch as a device tensor (no host sync).
    input_grads: -> (metrics, d text_emb, d img_emb): the gradient of (policy loss + value loss) with respect to the record's
    features, actor's and critic's contributions summed, in the shapes of record[5] / record[6] -- what the reference's autograd
    would hand to an encoder in front of the heads (features.finetune_ppo_step)."""
    state, next_state, old_scores, rewards, old_value, text, img, tgts = record
    dev = text.device
    dp = dp or _DataParallel()
    actor, critic = model.actor, model.critic
    bs, tags = old_scores.shape[:2]
    scal, per = torch.empty(4, device=dev), torch.empty(4, bs, device=dev)
    dscores, dvalue = torch.empty(bs, tags, device=dev), torch.empty(bs, device=dev)
    side = _Side(dev)
    side.fork_point()
    logits = actor.engine_forward(text, img, save=True)         # host order actor, critic: the order the dropout seeds are drawn in
    with side.run():
        value = critic.engine_forward(text, img, state, save=True)
    side.join()
    loss_kw = dict(B=bs, T=tags, kl_w=args.kl_div_loss_weight, ent_w=args.entropy_weight, value_clip=args.value_clip,
                   margin=0.01, adv_eps=-0.1)
    scores, probs = actor.action_scores(logits, bs, tags, want_probs=True)
    loss_in = (scores, old_scores.contiguous(), rewards.contiguous(), old_value.contiguous(), value, next_state.contiguous())
    if dp.active and getattr(args, "global_rank_loss", True):
        # RankLoss is one scalar over the whole batch (finetune/ppo.py:43-55): with the batch sharded over ranks its hinge
        # sum / positive count (and mean |A|, which multiplies it) must be global before R is formed, or the rank-averaged
        # gradient is not the gradient of the global-batch loss (SURVEY.md 8e caveat).  Three floats, one all-reduce.
        stats = torch.empty(3, device=dev)
        ops.ppo_loss(*loss_in, None, None, None, None, stats_out=stats, **loss_kw)
        dist.all_reduce(stats)
        ops.ppo_loss(*loss_in, scal, per, dscores, dvalue, global_stats=stats, world=dp.world, **loss_kw)
    else:
        ops.ppo_loss(*loss_in, scal, per, dscores, dvalue, **loss_kw)
    # out_layer.fc1.weight (96 % of each model): gradient GEMM and AdamW step in one kernel, the 2 GB gradient is never
    # materialised (args.fuse_fc1_update=False restores the separate wgrad + optimizer passes; same bits either way)
    fuse = getattr(args, "fuse_fc1_update", True) and hasattr(optimizer, "external_update") \
        and hasattr(critic_optim, "external_update") and not actor.TRAD          # (`_trad` heads have no 2-GB matrix to fuse)
    fa = optimizer.external_update(actor.out_layer.fc1.weight) if fuse else None
    fc = critic_optim.external_update(critic.out_layer.fc1.weight) if fuse else None
    if probs is not None:        # 'cls': chain d loss / d scores through the expected-label softmax to the class logits
        dscores = ops.cls_scores_bwd(probs, scores.view(-1), dscores.view(-1), torch.empty_like(probs), rows=bs * tags,
                                     C=actor.n_out)
    side.fork_point()
    ga = actor.engine_backward(dscores, dp, fc1_update=fa, input_grads=input_grads)
    wa = dp.reduce_start(actor)            # overlaps the critic's backward
    with side.run():                       # the critic's backward, gradient exchange and optimizer step beside the actor's
        # (its out_layer.fc1 update first, the actor's last: the two HBM-bound passes fall beside the other model's GEMMs)
        gc = critic.engine_backward(dvalue, dp, fc1_update=fc, fc1_early=side.on and os.environ.get("LR2_FC1_EARLY", "1") != "0",
                                    input_grads=input_grads)
        wc = dp.reduce_start(critic)
        if side.on:
            dp.finish(wc)
            critic_optim.step()
    dp.finish(wa)
    optimizer.step()
    if not side.on:
        dp.finish(wc)
        critic_optim.step()
    side.join()
    pm = per.mean(dim=1)
    metrics = torch.stack([scal[0], scal[1], pm[0], old_value.mean(), value.mean(), rewards.mean(), pm[2], pm[3], scal[2],
                           pm[1]])
    if dp.active:              # the reference's 10 logging all-reduces (ppo.py:589-598), packed into one
        metrics.div_(dp.world)
        dist.all_reduce(metrics)
    if input_grads:            # after side.join(): both models' input gradients are complete on the main stream
        d_text = ga[0].add_(gc[0])
        d_img = None if ga[1] is None else ga[1].add_(gc[1])
        return metrics, d_text, d_img
    return metrics


class GraphedPPOStep:
    """rollout_step + update_minibatch on one batch (the per-batch body of the reference's loop, finetune/ppo.py:844-883 then
    :518-598) captured ONCE in a HIP graph and replayed: ~300 kernel launches, two streams and their joins become one
    hipGraphLaunch, the host cost per step drops from ~4 ms to the cost of one small launch plus the replay.

    What a graph would freeze is moved to device memory first: the dropout seeds (runtime.device_seed: every train-mode forward
    inside draws *seed + i) and the learning rates (AdamW.use_device_lr) live in one 64-byte ops.StepScalars block that a single
    small launch rewrites before each replay from the host's dropout counter and the schedulers' current rates.  Same kernels,
    same order, same arguments otherwise: the step gives the bits of the eager step (tests/test_graph_gpu.py).

    Call 1 runs eagerly (it sizes the workspaces), call 2 captures and replays, later calls replay.  Inputs are copied into
    static buffers (pass the buffers themselves -- .text / .img / .tgts -- to skip the copies).  The returned metrics tensor is
    static too: read it before the next call.  Data parallel (round 4): with the RCCL backend the step's collectives -- the factor
    all-gathers, the tail all-reduces, the 3-float RankLoss statistics, the packed metric all-reduce -- are captured with it (RCCL
    enqueues device kernels on its own stream, forked from and joined to the capture by events; checked with one rank and the
    exchange forced: tests/workers/rccl_graph_worker.py); every rank must capture and replay in lock step.  Other backends stage
    through the host and cannot be captured."""

    def __init__(self, args, model, reward_model, optimizer, critic_optim):
        dp = _DataParallel()
        if dp.active and dp.backend != "nccl":
            raise NotImplementedError("GraphedPPOStep: the data-parallel step can be captured with the RCCL ('nccl') backend only "
                                      f"(backend {dp.backend!r} stages its collectives through the host)")
        for o in (optimizer, critic_optim):
            if not hasattr(o, "use_device_lr"):
                raise TypeError("GraphedPPOStep needs lr2ppo_amd's AdamW (device-resident learning rates)")
        self.args, self.model, self.reward_model, self.opt, self.copt = args, model, reward_model, optimizer, critic_optim
        self.graph = None
        self.calls = 0
        self.draws = 0
        self.text = self.img = self.tgts = self.metrics = self.scalars = None

    def _setup(self, text, img, tgts):
        dev = text.device
        # (`_trad` heads take no image features: img_emb is None there, ppo_trad.py:431-433)
        self.text, self.img, self.tgts = torch.empty_like(text), None if img is None else torch.empty_like(img), torch.empty_like(tgts)
        self.scalars = ops.StepScalars(dev)
        for o in (self.opt, self.copt):
            o.use_device_lr([self.scalars.lr_tensor(self.scalars.new_lr()) for _ in o.param_groups])

    def _store_scalars(self):
        self.scalars.store(runtime.peek_drop_seed(), [g["lr"] for o in (self.opt, self.copt) for g in o.param_groups])

    def _body(self):
        self.model.eval()
        rec = rollout_step(self.model, self.reward_model, self.text, self.img, self.tgts)
        self.model.train()
        return update_minibatch(self.args, self.model, self.opt, self.copt, rec)

    def _traced_body(self):
        with runtime.device_seed(self.scalars.seed) as ds:
            out = self._body()
        return out, ds.draws

    def __call__(self, text, img, tgts):
        if self.text is None:
            self._setup(text, img, tgts)
        for dst, src in ((self.text, text), (self.img, img), (self.tgts, tgts)):
            if src is not dst:
                if (src is None) != (dst is None):
                    raise ValueError("GraphedPPOStep: img_emb given for one call and None for another")
                if src.shape != dst.shape or src.dtype != dst.dtype:
                    raise ValueError("GraphedPPOStep: batch shape changed (one graph per shape: build another GraphedPPOStep)")
                dst.copy_(src)
        self._store_scalars()
        self.calls += 1
        if self.graph is None and self.calls == 1:
            self.metrics, self.draws = self._traced_body()              # eager, through the same device-resident scalars
        elif self.graph is None:
            engine._INPUT_PLANES.clear()       # the splits of the static inputs must be IN the graph, whatever the cache holds
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.metrics, draws = self._traced_body()
            if draws != self.draws:
                raise RuntimeError("GraphedPPOStep: the captured step drew a different number of dropout seeds than the eager one")
            self.graph = graph
            graph.replay()
        else:
            self.graph.replay()
            self.opt.count_replayed_step()
            self.copt.count_replayed_step()
        runtime.advance(self.draws)
        return self.metrics

    def release(self):
        """Back to by-value learning rates; drops the graph and its static buffers."""
        self.opt.use_device_lr(None)
        self.copt.use_device_lr(None)
        self.graph = None
        self.text = self.img = self.tgts = self.metrics = self.scalars = None
        self.calls = 0


def train_model(args, model, optimizer, critic_optim, scheduler, critic_scheduler, memories, epoch):
    """One PPO update cycle over the stored rollouts; returns the reference's 10 averaged metrics
    [policy, value, kl, old_value, value, rewards_ori, rewards, advantages, rank_loss, entropy] (ppo.py:615-617)."""
    dev = next(model.parameters()).device
    dp = _DataParallel()
    model.actor.bind_grads()
    model.critic.bind_grads()
    total = torch.zeros(10, device=dev)
    for record in memories:
        total += update_minibatch(args, model, optimizer, critic_optim, record, dp)
    scheduler.step()
    critic_scheduler.step()
    out = (total / max(len(memories), 1)).tolist()
    if any(v != v for v in out):
        raise FloatingPointError("NaN in PPO metrics (the reference drops into pdb here, finetune/ppo.py:576-578)")
    return out


# ---------------------------------------------------------------------------------------------
# evaluation (finetune/ppo.py:620-681)
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def evaluate(args, val_loader, step, split="test", num_tasks=None):
    ndcg_obj = AverageNDCGMeter()
    args.model.eval()
    scores, golds = [], []
    for text_emb, img_emb, tgts in val_loader:
        text_emb = text_emb.to(args.device)
        img_emb = img_emb.to(args.device) if img_emb is not None else None   # [1, n_img, 768]: shared by all tags (None: `_trad`)
        logits = args.model.actor.engine_forward(text_emb, img_emb, save=False)
        if args.model.actor.n_out > 1:               # 'cls': 0 * z0 + 1 * z1 + 2 * z2 on the RAW logits, as upstream (ppo.py:641-643)
            logits = ops.cls_scores(logits, None, torch.empty(logits.shape[0], device=logits.device), rows=logits.shape[0],
                                    C=logits.shape[1], softmax=False)
        scores.append(logits.view(-1))
        golds.append(tgts.view(-1))
    # scores never leave the device: one batched NDCG kernel over the whole split (lr2_ndcg: sort by score, gain 2^rel - 1,
    # discount log2(i + 2), ideal DCG <= 1e-6 -> 1) and ONE device-to-host copy of the [items, 6] result -- the reference
    # synchronises and loops in Python once per item (finetune/ppo.py:640-659, ndcg.py:28-65)
    mine = ndcg_rows(scores, golds, args.device, tuple(ndcg_obj.ndcg_at_k))
    world = num_tasks or 1
    if world > 1 and dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        # interleave like the reference's per-item all_gather (rank-major within each step)
        rows = [g[i] for i in range(max(len(g) for g in gathered)) for g in gathered if i < len(g)]
        mine = torch.stack(rows) if rows else mine
    if getattr(args, "is_master", True):
        for row in mine:
            for i, k in enumerate(ndcg_obj.ndcg_at_k):
                ndcg_obj.ndcg[k].append(row[i])
        vals = ndcg_obj.value()
        if hasattr(args, "logger"):
            args.logger.info("NDCG:")
            args.logger.info("".join("\nNDCG@{}={:.4f}".format(k, vals[k]) for k in sorted(vals.keys())))
        args.last_ndcg = {int(k): float(v) for k, v in vals.items()}
        return vals[100000000]
    return None


# ---------------------------------------------------------------------------------------------
# data (finetune/ppo.py:58-151, 684-699)
# ---------------------------------------------------------------------------------------------
class MovieNet(Dataset):
    """LRMovieNet reader with the reference's sampling (80 random ordered tag pairs per item in training,
    shuffled image features cyclically padded to max_imgs).  Reads LRMovieNet/clean_feat.h5 (h5py, or `lr2ppo_amd.h5lite` on libhdf5)."""

    def __init__(self, args, path, is_train=False):
        with open(path) as f:
            self.data = json.load(f)
        self.embed_data = h5lite.open_file(os.path.join("LRMovieNet", "clean_feat.h5"), "r")    # h5py, or libhdf5 via ctypes
        self.max_imgs, self.is_train, self.max_tags = args.max_imgs, is_train, args.max_tags
        self.items = []
        for item in self.data:
            tags = item["tags"]
            if is_train:
                for _ in range(self.max_tags):
                    idx = list(range(len(tags)))
                    random.shuffle(idx)
                    self.items.append((item["id"], idx[:2], [tags[i] for i in idx[:2]]))
            else:
                self.items.append((item["id"], list(range(len(tags))), tags))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        item_id, tag_index, tag_list = self.items[i]
        grp = self.embed_data[f"{item_id}"]
        text = torch.tensor(grp["text_emb"][:])[torch.tensor(tag_index)]
        loaded = torch.tensor(grp["img_emb"][:][0])
        loaded = loaded[torch.randperm(loaded.shape[0])]
        n = loaded.shape[0]
        img = loaded[: self.max_imgs] if n > self.max_imgs else loaded[torch.arange(self.max_imgs) % n]
        return text, img, torch.tensor([int(t["target"]) for t in tag_list])


class SyntheticMovieNet(Dataset):
    """Seeded stand-in with LRMovieNet's shapes: text_emb ~ N(0,1) [tags,196,768], img_emb ~ N(0,1)
    [16,768], targets in {0,1,2}.  Train items are ordered tag pairs, val items carry all `val_tags` tags."""

    def __init__(self, n_items, tags, max_imgs=16, seed=7):
        self.n, self.tags, self.max_imgs, self.seed = n_items, tags, max_imgs, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        text = torch.randn(self.tags, SEQ_LEN, FEAT, generator=g)
        img = torch.randn(self.max_imgs, FEAT, generator=g)
        return text, img, torch.randint(0, 3, (self.tags,), generator=g)


def get_dataloader(args, dataset, num_tasks, global_rank, is_train=False):
    sampler = DistributedSampler(dataset, num_replicas=num_tasks, rank=global_rank, shuffle=is_train)
    workers = getattr(args, "num_workers", 32 if isinstance(dataset, MovieNet) else 2)        # synthetic sets: 2 workers
    return DataLoader(dataset=dataset, batch_size=args.batch_size if is_train else 1, sampler=sampler,
                      num_workers=workers, drop_last=False)


# ---------------------------------------------------------------------------------------------
# entry point (finetune/ppo.py:702-915)
# ---------------------------------------------------------------------------------------------
def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    finetune_opts(parser)
    tokenizer_opts(parser)
    parser.add_argument("--mode", type=str, default="reg")
    adv_opts(parser)
    parser.add_argument("--vit_pretrained_model_path", default=None, type=str)
    parser.add_argument("--vit_config_path", default="models/bert/base_config.json", type=str)
    parser.add_argument("--vit_tokenizer", choices=["bert", "bpe", "char", "space", "xlmroberta", "image", "text_image", "virtual"])
    parser.add_argument("--vit_encoder", choices=["transformer", "rnn", "lstm", "gru", "birnn", "bilstm", "bigru", "gatedcnn", "dual"])
    parser.add_argument("--dist_url", type=str, default="env://")
    parser.add_argument("--max_tags", type=int, default=32)
    parser.add_argument("--exp_name", type=str)
    parser.add_argument("--use_pairwise", action="store_true")
    parser.add_argument("--reward_model_path", type=str)
    parser.add_argument("--max_timesteps", type=int, default=5)
    parser.add_argument("--update_timesteps", type=int, default=300)
    parser.add_argument("--eps_clip", type=float, default=0.2)        # parsed, never read -- as upstream
    parser.add_argument("--kl_div_loss_weight", type=float, default=0.1)
    parser.add_argument("--entropy_weight", type=float, default=0.1)
    parser.add_argument("--value_clip", type=float, default=0.4)
    parser.add_argument("--critic_learning_rate", type=float, default=2e-6, help="Learning rate.")
    # additions of this build (not in the reference): synthetic data + bounded runs for boxes without LRMovieNet
    parser.add_argument("--synthetic_items", type=int, default=0, help="use SyntheticMovieNet with this many train items")
    parser.add_argument("--synthetic_val_items", type=int, default=16)
    parser.add_argument("--max_cycles", type=int, default=0, help="stop after this many update cycles (0 = run all epochs)")
    from .features import raw_input_opts
    raw_input_opts(parser)      # --raw_inputs [--image_tower vit_large_14_224] [--fp8_features]: frozen encoder stacks in line
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    vit_args_dict = dict(vars(args))
    for k, v in vars(args).items():
        if "vit_" in k:
            vit_args_dict[k[4:]] = v
    vit_args = argparse.Namespace(**vit_args_dict)
    args = load_hyperparam(args)
    if os.path.exists(vit_args.config_path):
        vit_args = load_hyperparam(vit_args)
    args.labels_num = 3

    misc.init_distributed_mode(args)
    misc.setup_seed(args.seed + misc.get_rank())
    args.is_master = misc.is_main_process()
    if not args.raw_inputs:
        return run_training(args, vit_args, ActorCritic, Reward, None, None)
    # the metric's own composition (BASELINE.json: "PPO steps/sec (ViT-B+RoBERTa-base, batch 32)"; configs[4] with
    # --image_tower vit_large_14_224 --fp8_features): raw items, frozen encoder stacks in line, then the unchanged stage-3 loop
    from .features import SyntheticRawItems, build_extractor
    if args.synthetic_items <= 0:
        raise RuntimeError("--raw_inputs: the repository holds no raw LRMovieNet reader (the reference reads pre-extracted features, "
                           "finetune/ppo.py:58-151); use --synthetic_items N")
    args.device = torch.device("cuda", torch.cuda.current_device())
    fx = build_extractor(args, misc.get_world_size())

    def make_sets():
        return (SyntheticRawItems(args.synthetic_items, 2, args.max_imgs, args.seed),
                SyntheticRawItems(args.synthetic_val_items, 20, args.max_imgs, args.seed + 1))

    def batch_map(batch):
        frames, ids, seg, tgts = batch
        text_emb, img_emb = fx.extract(frames.to(args.device), ids.to(args.device), seg.to(args.device))
        return text_emb, img_emb, tgts
    return run_training(args, vit_args, ActorCritic, Reward, make_sets, batch_map)


class _MappedLoader:
    """A DataLoader whose batches go through `fn` (the `_trad` readers yield (ground_truths, query_id, features))."""

    def __init__(self, loader, fn):
        self.loader, self.fn, self.sampler = loader, fn, loader.sampler

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        return (self.fn(b) for b in self.loader)


def run_training(args, vit_args, actor_critic_cls, reward_cls, make_sets, batch_map):
    """The stage-3 loop of finetune/ppo.py:790-915 (and of ppo_trad.py:700-849, which repeats it): models, optimizers, rollouts
    into `memories`, an update cycle every `update_timesteps` rollouts, validation and checkpoint after each cycle.
    make_sets() -> (trainset, valset) (None: LRMovieNet / SyntheticMovieNet); batch_map: loader batch -> (text_emb, img_emb,
    tgts) (None: identity)."""
    num_tasks, global_rank = misc.get_world_size(), misc.get_rank()
    model = actor_critic_cls(args, vit_args)
    reward_model = reward_cls(args, vit_args)
    load_or_initialize_parameters(args, model.actor)
    load_or_initialize_parameters_reward(args, model.critic)
    load_or_initialize_parameters_reward(args, reward_model)
    if args.is_master:
        args.logger = init_logger(args)
    args.device = torch.device("cuda", torch.cuda.current_device())
    model = model.to(args.device)
    reward_model = reward_model.to(args.device).eval()
    if num_tasks > 1:   # the reference initialises each rank from its own seed (quirk 17); replicas must start equal
        for p in list(model.parameters()) + list(reward_model.parameters()):
            dist.broadcast(p.data, src=0)

    if make_sets is None:
        def make_sets():
            if args.synthetic_items > 0:
                return (SyntheticMovieNet(args.synthetic_items, 2, args.max_imgs, args.seed),
                        SyntheticMovieNet(args.synthetic_val_items, 20, args.max_imgs, args.seed + 1))
            return MovieNet(args, args.train_path, is_train=True), MovieNet(args, args.dev_path, is_train=False)

    def loader(dataset, is_train):
        dl = get_dataloader(args, dataset, num_tasks, global_rank, is_train=is_train)
        return dl if batch_map is None else _MappedLoader(dl, batch_map)

    trainset, valset = make_sets()
    val_loader = loader(valset, False)
    args.train_steps = int(len(trainset) * args.epochs_num / args.batch_size) + 1
    if args.is_master:
        args.logger.info("Batch size: {}".format(args.batch_size))
        args.logger.info("The number of training instances: {}".format(len(trainset)))
    optimizer, critic_optimizer, scheduler, critic_scheduler = build_optimizer(args, model)
    args.model = model
    best_result, step, time, cycles = 0.0, 0, 0, 0
    if args.is_master:
        args.logger.info("Start training.")
    for epoch in range(1, args.epochs_num):            # range(1, N): as upstream (quirk 18)
        trainset, _ = make_sets()
        train_loader = loader(trainset, True)
        train_loader.sampler.set_epoch(epoch)
        memories = []
        for text_emb, img_emb, tgts in train_loader:
            text_emb, tgts = text_emb.to(args.device), tgts.to(args.device)
            img_emb = img_emb.to(args.device) if img_emb is not None else None
            model.eval()
            state = None
            for timestep in range(args.max_timesteps):
                time += 1
                rec = rollout_step(model, reward_model, text_emb, img_emb, tgts, state)
                state = rec[1]
                memories.append(rec)
                if time % args.update_timesteps == 0:
                    model.train()
                    t0 = _time.time()
                    vals = train_model(args, model, optimizer, critic_optimizer, scheduler, critic_scheduler, memories, epoch)
                    memories = []
                    model.eval()
                    cycles += 1
                    names = ["Policy loss", "Critic Loss", "KL Penalty", "Old Values", "Values", "Rewards Ori", "Reward",
                             "Rank Loss", "Advantages", "Entropy"]
                    order = [0, 1, 2, 3, 4, 5, 6, 8, 7, 9]
                    if args.is_master:
                        args.logger.info(f"Training step: {step}")
                        for n, i in zip(names, order):
                            args.logger.info(f"{n}: {vals[i]}")
                        args.logger.info(f"update cycle wall time {_time.time() - t0:.2f}s")
                        args.logger.info("\nVal set evaluation.")
                    result = evaluate(args, val_loader, step, split="val", num_tasks=num_tasks)
                    if args.is_master and result > best_result:
                        best_result = result
                        save_model(model, args.output_model_path)
                        args.logger.info("Best val indicator until now!")
                    if args.max_cycles and cycles >= args.max_cycles:
                        return best_result
    return best_result


if __name__ == "__main__":
    main()
