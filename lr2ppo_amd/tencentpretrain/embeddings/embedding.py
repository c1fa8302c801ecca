"""Composable embeddings with the reference's module/key layout, fused on the HIP kernels.

Reference: tencentpretrain/embeddings/{embedding,word_embedding,pos_embedding,seg_embedding,patch_embedding,
dual_embedding}.py.  `Embedding.update(sub, name)` registers sub-embeddings; `forward(src, seg)` sums them,
applies the TencentPretrain LayerNorm unless `remove_embedding_layernorm`, then dropout (inference: identity).
Two compositions are recognised and run as fused kernels:
  ["patch", "pos"]        image -> patch rows (lr2_patchify) -> split-bf16 GEMM with the conv weight viewed as
                          [E, C*p*p] (Conv2d k=s=p is exactly that GEMM) -> cls + pos assembly (lr2_vit_assemble)
  ["word", "pos", "seg"]  one gather-sum kernel (lr2_text_embed) -> LayerNorm
"""
import copy
from argparse import Namespace

import torch
import torch.nn as nn

from ... import engine, ops
from ..layers.layer_norm import LayerNorm


class WordEmbedding(nn.Module):
    def __init__(self, args, vocab_size):
        super().__init__()
        self.embedding = nn.Embedding(vocab_size, args.emb_size)
        self.emb_size = args.emb_size
        if "sinusoidalpos" in args.embedding:
            raise NotImplementedError("sinusoidal positions are outside the LR2PPO configs")


class PosEmbedding(nn.Module):
    def __init__(self, args, _):
        super().__init__()
        self.max_seq_length = args.max_seq_length
        self.embedding = nn.Embedding(self.max_seq_length, args.emb_size)


class SegEmbedding(nn.Module):
    def __init__(self, args, _):
        super().__init__()
        self.embedding = nn.Embedding(3, args.emb_size)


class PatchEmbedding(nn.Module):
    def __init__(self, args, _):
        super().__init__()
        self.cls_emb = nn.Parameter(torch.zeros(1, 1, args.emb_size))
        self.image_height, self.image_width = args.image_height, args.image_width
        self.patch_size, self.channels_num = args.patch_size, args.channels_num
        self.projection = nn.Conv2d(args.channels_num, args.emb_size, kernel_size=(args.patch_size, args.patch_size),
                                    stride=(args.patch_size, args.patch_size), bias=False)

    def check(self, src):
        _, _, height, width = src.shape
        if height != self.image_height or width != self.image_width:
            raise ValueError(f"Input image size ({height}*{width}) doesn't match model ({self.image_height}*{self.image_width}).")


class Embedding(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.embedding_name_list = []
        self.dropout = nn.Dropout(args.dropout)
        self.remove_embedding_layernorm = args.remove_embedding_layernorm
        if not self.remove_embedding_layernorm and "dual" not in args.embedding:
            self.layer_norm = LayerNorm(args.emb_size)
        self._ws = None

    def update(self, embedding, embedding_name):
        setattr(self, embedding_name, embedding)
        self.embedding_name_list.append(embedding_name)

    @torch.no_grad()
    def forward(self, src, seg):
        names = self.embedding_name_list
        if names and names[0] == "dual":
            return self.dual(src, seg)
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("embedding dropout (training) is outside this round's scope; call .eval()")
        dev = src.device
        if not src.is_cuda:
            raise TypeError("lr2ppo_amd: inputs must live on the HIP device (no CPU path)")
        if self._ws is None or self._ws.device != dev:
            self._ws = engine.Workspace(dev)
        ws = self._ws
        if names == ["patch", "pos"]:
            pe = self.patch
            pe.check(src)
            B, C, H, W = src.shape
            ps = pe.patch_size
            P, E, Kd = (H // ps) * (W // ps), pe.cls_emb.shape[-1], C * ps * ps
            patches = ws.mat("patches", B * P, Kd)
            ops.patchify(src.contiguous().float(), patches, B=B, Cc=C, H=H, W=W, ps=ps)
            proj = ws.mat("proj", B * P, E)
            engine.linear_fwd(ws, patches, pe.projection.weight.data.view(E, Kd), None, proj, B * P, E, Kd)
            out = torch.empty(B, P + 1, E, device=dev)
            ops.vit_assemble(proj, pe.cls_emb.data.view(-1), self.pos.embedding.weight.data, out, B=B, P=P, D=E)
        elif names == ["word", "pos", "seg"]:
            B, L = src.shape
            E = self.word.emb_size
            out = torch.empty(B, L, E, device=dev)
            ops.text_embed(src.contiguous().view(-1).long(), seg.to(dev).contiguous().view(-1).long(),
                           self.word.embedding.weight.data, self.pos.embedding.weight.data, self.seg.embedding.weight.data,
                           out.view(B * L, E), rows=B * L, L=L, D=E)
        else:
            raise NotImplementedError(f"embedding composition {names}: only ['patch','pos'] and ['word','pos','seg'] "
                                      "(ViT-B/16, RoBERTa-base) are on the HIP path")
        if not self.remove_embedding_layernorm:
            out = self.layer_norm(out)
        return out


class DualEmbedding(nn.Module):
    """Two-stream embedding (embeddings/dual_embedding.py:7-66 of the reference)."""

    def __init__(self, args, vocab_size):
        super().__init__()
        from . import str2embedding
        built = []
        for over in (args.stream_0, args.stream_1):
            d = copy.deepcopy(vars(args))
            d.update(over)
            ns = Namespace(**d)
            emb = Embedding(ns)
            for name in ns.embedding:
                emb.update(str2embedding[name](ns, vocab_size), name)
            built.append((emb, ns))
        (self.embedding_0, a0), (self.embedding_1, a1) = built
        self.stream_0_remove_embedding_layernorm = a0.remove_embedding_layernorm
        if not self.stream_0_remove_embedding_layernorm:
            self.stream_0_layer_norm = LayerNorm(a0.emb_size)
        self.stream_1_remove_embedding_layernorm = a1.remove_embedding_layernorm
        if not self.stream_1_remove_embedding_layernorm:
            self.stream_1_layer_norm = LayerNorm(a1.emb_size)
        self.dropout = nn.Dropout(args.dropout)
        if args.tie_weights:
            self.embedding_0 = self.embedding_1

    def forward(self, src, seg):
        return self.get_embedding_0(src[0], seg[0]), self.get_embedding_1(src[1], seg[1])

    def get_embedding_0(self, src, seg):
        emb = self.embedding_0(src, seg)
        return emb if self.stream_0_remove_embedding_layernorm else self.stream_0_layer_norm(emb)

    def get_embedding_1(self, src, seg):
        emb = self.embedding_1(src, seg)
        return emb if self.stream_1_remove_embedding_layernorm else self.stream_1_layer_norm(emb)
