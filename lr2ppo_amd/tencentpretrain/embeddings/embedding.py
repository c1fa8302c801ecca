"""Composable embeddings with the reference's module/key layout, fused on the HIP kernels.

Reference: tencentpretrain/embeddings/{embedding,word_embedding,pos_embedding,seg_embedding,patch_embedding,
dual_embedding}.py.  `Embedding.update(sub, name)` registers sub-embeddings; `forward(src, seg)` sums them,
applies the TencentPretrain LayerNorm unless `remove_embedding_layernorm`, then dropout (inference: identity; train
mode: counter-based masks of lr2ppo_amd.runtime).  With autograd enabled the forward keeps what the hand-written backward
needs (`_EmbeddingFn`): LayerNorm backward, position-table column sums, word / segment scatter-adds, and for the patch
embedding the weight gradient of the Conv2d(k = s = patch) as one TN GEMM over the patch rows.
Two compositions are recognised and run as fused kernels:
  ["patch", "pos"]        image -> patch rows (lr2_patchify) -> split-bf16 GEMM with the conv weight viewed as
                          [E, C*p*p] (Conv2d k=s=p is exactly that GEMM) -> cls + pos assembly (lr2_vit_assemble)
  ["word", "pos", "seg"]  one gather-sum kernel (lr2_text_embed) -> LayerNorm
"""
import copy
from argparse import Namespace

import torch
import torch.nn as nn

from ... import engine, ops, runtime
from ..layers.layer_norm import LayerNorm, device_dropout


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
        # uint8 frames are converted as the reference's image loader does (tencentpretrain/utils/dataloader.py:559-561):
        # x / 255 then Normalize(mean, std) with the CLIP statistics; set both to None for ZeroOneNormalize only
        self.u8_mean, self.u8_std = ops.CLIP_MEAN, ops.CLIP_STD

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
        self._err = None
        # True: the out-of-range-id check (one device -> host read) is left to an explicit check_ids() call, so that a
        # caller with its own synchronisation point (lr2ppo_amd.finetune.features) keeps the stream asynchronous
        self.defer_id_check = False

    def check_ids(self):
        """Raise the IndexError nn.Embedding raises for an out-of-range token / segment id (the kernel itself never
        indexes out of bounds: it reads row 0 and sets a device error word)."""
        if self._err is None:
            return
        code = int(self._err.item())
        if code:
            self._err.zero_()
            what = [w for bit, w in ((1, "token id outside [0, vocab_size)"), (2, "segment id outside [0, 3)")) if code & bit]
            raise IndexError("lr2ppo_amd Embedding: " + " and ".join(what))

    def update(self, embedding, embedding_name):
        setattr(self, embedding_name, embedding)
        self.embedding_name_list.append(embedding_name)

    def forward(self, src, seg):
        names = self.embedding_name_list
        if names and names[0] == "dual":
            return self.dual(src, seg)
        if not src.is_cuda:
            raise TypeError("lr2ppo_amd: inputs must live on the HIP device (no CPU path)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _EmbeddingFn.apply(self, src, seg, *list(self.parameters()))
        out, _ = self._run(src, seg, save=False)
        return out

    @torch.no_grad()
    def _run(self, src, seg, save):
        """-> (out, saved).  Train mode applies dropout (site 0 of a fresh seed from the runtime's mask stream)."""
        names = self.embedding_name_list
        dev = src.device
        if self._ws is None or self._ws.device != dev:
            self._ws = engine.Workspace(dev)
        ws = self._ws
        saved = {"kind": tuple(names)}
        if names == ["patch", "pos"]:
            pe = self.patch
            pe.check(src)
            B, C, H, W = src.shape
            ps = pe.patch_size
            P, E, Kd = (H // ps) * (W // ps), pe.cls_emb.shape[-1], C * ps * ps
            Kp = (Kd + 63) // 64 * 64        # whole K tiles for the projection GEMM (ViT-L/14: 588 -> 640, zero padded)
            # patch rows straight to bf16 hi/lo planes (uint8 frames are normalised on the fly); kept for the backward
            patches = ops.Planes.empty(B * P, Kp, dev) if save else ws.planes("patches", B * P, Kp)
            img = src.contiguous() if src.dtype == torch.uint8 else src.contiguous().float()
            ops.patchify_planes(img, patches, B=B, Cc=C, H=H, W=W, ps=ps, mean=pe.u8_mean, std=pe.u8_std)
            w2 = pe.projection.weight.data.view(E, Kd)
            if Kp != Kd:
                wpad = ws.mat("patch_w_pad", E, Kp)
                wpad.zero_()
                wpad[:, :Kd].copy_(w2)
                w2 = wpad
            w_p = ops.split_planes(w2, ws.planes("patch_w", E, Kp))
            proj = ws.mat("proj", B * P, E)
            engine.linear_fwd(ws, patches, w_p, None, proj, B * P, E, Kp)
            out = torch.empty(B, P + 1, E, device=dev)
            ops.vit_assemble(proj, pe.cls_emb.data.view(-1), self.pos.embedding.weight.data, out, B=B, P=P, D=E)
            saved.update(patches=patches, dims=(B, P, E, Kd))
        elif names == ["word", "pos", "seg"]:
            B, L = src.shape
            E = self.word.emb_size
            ids = src.contiguous().view(-1).long()
            sg = seg.to(dev).contiguous().view(-1).long()
            out = torch.empty(B, L, E, device=dev)
            if self._err is None or self._err.device != dev:
                self._err = torch.zeros(1, dtype=torch.int32, device=dev)
            ops.text_embed(ids, sg, self.word.embedding.weight.data, self.pos.embedding.weight.data,
                           self.seg.embedding.weight.data, out.view(B * L, E), rows=B * L, L=L, D=E, err=self._err)
            if not self.defer_id_check:
                self.check_ids()
            saved.update(ids=ids, seg=sg, dims=(B, L, E))
        else:
            raise NotImplementedError(f"embedding composition {names}: only ['patch','pos'] and ['word','pos','seg'] "
                                      "(ViT-B/16, RoBERTa-base) are on the HIP path")
        if not self.remove_embedding_layernorm:
            ln = self.layer_norm
            M, E = out.shape[0] * out.shape[1], out.shape[2]
            y = torch.empty_like(out)
            mean, rstd = (torch.empty(M, device=dev), torch.empty(M, device=dev)) if save else (None, None)
            ops.layernorm_fwd(out.view(M, E), ln.gamma.data, ln.beta.data, y.view(M, E), mean, rstd, rows=M, D=E, eps=ln.eps,
                              mode=1)
            saved.update(x=out, mean=mean, rstd=rstd)
            out = y
        p = float(self.dropout.p) if self.training else 0.0
        saved["drop"] = None
        if p > 0:
            saved["drop"] = ops.Drop(p, runtime.next_drop(p, 0).seed, 0)
            ops.dropout_apply(out, out, saved["drop"])
        return out, saved

    def grad_buffers(self):
        """{parameter: gradient}: persistent tensors for the explicit training path (see TransformerEncoder.grad_buffers)."""
        params = list(self.parameters())
        dev = params[0].device
        if getattr(self, "_gflat", None) is None or self._gflat.device != dev:
            self._gflat = torch.zeros(sum(q.numel() for q in params), device=dev)
            self._gviews, off = {}, 0
            for q in params:
                self._gviews[q] = self._gflat[off:off + q.numel()].view_as(q)
                off += q.numel()
        return self._gviews

    @torch.no_grad()
    def _backward(self, saved, dout, G_out=None):
        """-> {parameter: gradient} (inputs are data: token ids / pixels get no gradient).  G_out (grad_buffers()): the
        gradients are written there instead of into fresh tensors."""
        dev = dout.device
        ws = self._ws
        G = {}

        def buf(q, zero=False):
            if G_out is None:
                return torch.zeros_like(q) if zero else torch.empty_like(q)
            return G_out[q].zero_() if zero else G_out[q]
        dy = dout.contiguous()
        if saved["drop"] is not None:
            dy = ops.dropout_apply(dy, torch.empty_like(dy), saved["drop"])
        if not self.remove_embedding_layernorm:
            ln = self.layer_norm
            M, E = dy.shape[0] * dy.shape[1], dy.shape[2]
            dx = torch.empty(M, E, device=dev)
            G[ln.gamma], G[ln.beta] = buf(ln.gamma), buf(ln.beta)
            ops.layernorm_bwd(dy.view(M, E), saved["x"].view(M, E), ln.gamma.data, saved["mean"], saved["rstd"], dx,
                              ws.vec("ln_partials", ops.LN_BWD_BLOCKS * 2 * E), G[ln.gamma], G[ln.beta], rows=M, D=E, mode=1, eps=ln.eps)
            dy = dx.view_as(dy)
        if saved["kind"] == ("patch", "pos"):
            pe = self.patch
            B, P, E, Kd = saved["dims"]
            dpos = buf(self.pos.embedding.weight, zero=True)
            ops.period_rows_grad(dy.view(B * (P + 1), E), dpos, rows=B * (P + 1), D=E, period=P + 1)
            G[self.pos.embedding.weight] = dpos
            G[pe.cls_emb] = buf(pe.cls_emb).copy_(dpos[0].view_as(pe.cls_emb))    # out[b, 0] = cls + pos[0]: same gradient rows
            dproj = torch.empty(B, 1, P * E, device=dev)
            ops.gather_rows(dy.view(B, (P + 1) * E)[:, E:], None, dproj, B=B, t_in=1, t_out=1, row_elems=P * E,
                            src_bstride=(P + 1) * E, src_tstride=0)
            dproj_p = ops.split_planes(dproj.view(B * P, E), ops.Planes.empty(B * P, E, dev))
            patches_p = saved["patches"]
            Kp = patches_p.cols
            dw = torch.empty(E, Kp, device=dev)
            engine.linear_wgrad(ws, dproj_p, patches_p, dw, None, B * P, Kp, E)
            G[pe.projection.weight] = buf(pe.projection.weight).copy_((dw if Kp == Kd else dw[:, :Kd]).reshape(pe.projection.weight.shape))
        else:
            B, L, E = saved["dims"]
            dword, dseg = buf(self.word.embedding.weight, zero=True), buf(self.seg.embedding.weight, zero=True)
            dpos = buf(self.pos.embedding.weight, zero=True)
            ops.text_embed_bwd(dy.view(B * L, E), saved["ids"], saved["seg"], dword, dseg, rows=B * L, D=E)
            ops.period_rows_grad(dy.view(B * L, E), dpos, rows=B * L, D=E, period=L)
            G[self.word.embedding.weight], G[self.seg.embedding.weight], G[self.pos.embedding.weight] = dword, dseg, dpos
        return G


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, src, seg, *params):
        out, saved = emb._run(src, seg, save=True)
        ctx.emb, ctx.saved = emb, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        G = ctx.emb._backward(ctx.saved, dout)
        ctx.saved = None
        return (None, None, None) + tuple(G.get(q) if q.requires_grad else None for q in ctx.emb.parameters())


class DualEmbedding(nn.Module):
    """Two-stream embedding (embeddings/dual_embedding.py:7-66 of the reference).  As upstream, each stream is a full
    `Embedding` (with its own LayerNorm unless removed, and its own dropout) followed by the stream LayerNorm (unless
    `remove_embedding_layernorm` in the stream's options) and the shared dropout (:51-52): in train mode every stream is
    dropped twice.  Both LayerNorms and both dropouts are differentiable (autograd nodes over the HIP kernels)."""

    def __init__(self, args, vocab_size):
        super().__init__()
        from . import str2embedding
        # registration order = the reference's (dual_embedding.py:15-33): embedding_0, stream_0_layer_norm, embedding_1,
        # stream_1_layer_norm -- named_parameters() / optimizer parameter order follow it
        for i, over in enumerate((args.stream_0, args.stream_1)):
            d = copy.deepcopy(vars(args))
            d.update(over)
            ns = Namespace(**d)
            emb = Embedding(ns)
            for name in ns.embedding:
                emb.update(str2embedding[name](ns, vocab_size), name)
            setattr(self, f"embedding_{i}", emb)
            setattr(self, f"stream_{i}_remove_embedding_layernorm", ns.remove_embedding_layernorm)
            if not ns.remove_embedding_layernorm:
                setattr(self, f"stream_{i}_layer_norm", LayerNorm(ns.emb_size))
        self.dropout = nn.Dropout(args.dropout)
        if args.tie_weights:
            self.embedding_0 = self.embedding_1

    def forward(self, src, seg):
        emb_0 = self.get_embedding_0(src[0], seg[0])
        emb_1 = self.get_embedding_1(src[1], seg[1])
        p = float(self.dropout.p)                        # dual_embedding.py:51-52: dropout on both streams
        return device_dropout(emb_0, p, self.training), device_dropout(emb_1, p, self.training)

    def get_embedding_0(self, src, seg):
        emb = self.embedding_0(src, seg)
        return emb if self.stream_0_remove_embedding_layernorm else self.stream_0_layer_norm(emb)

    def get_embedding_1(self, src, seg):
        emb = self.embedding_1(src, seg)
        return emb if self.stream_1_remove_embedding_layernorm else self.stream_1_layer_norm(emb)
