"""Online feature extraction in front of the LR2PPO heads: frames + tag token ids -> ViT-B/16 + RoBERTa-base ->
(text_emb, img_emb) in the layout the heads consume.

The reference's finetune scripts read these tensors pre-extracted from LRMovieNet/clean_feat.h5 (finetune/ppo.py:115-148:
'{id}'/text_emb [tags, 196, 768], '{id}'/img_emb [1, n_img, 768]); the extraction script is not in the repository.  This
module is that missing stage, built from the reference's own TencentPretrain composition
`memory_bank = encoder(embedding(src, seg), seg)` (tencentpretrain/models/model.py:32-41) with the shipped configs
(models/vit/base-16-224_config.json, models/xlm-roberta/base_config.json) and the mapping of SURVEY.md 8d:

    frames  uint8 [B, n_img, 3, 224, 224] -> /255 -> CLIP mean/std (utils/dataloader.py:559-561, fused into the patchify
            kernel) -> Embedding(patch, pos) -> TransformerEncoder (pre-LN, 12 x 768) -> pooling(first)   (utils/misc.py:23-35)
            -> img_emb [B, n_img, 768]
    ids     int64 [B, T, 196] (+ seg [B, T, 196], 1 on real tokens / 0 on padding) -> Embedding(word, pos, seg)
            -> TransformerEncoder (post-LN, 12 x 768) -> text_emb [B, T, 196, 768]

`img_emb` is shared by the T tags of an item (the reference materialises the repeat at finetune/ppo.py:831; the heads
accept the shared form).  Everything runs on the HIP kernels of this package; there is no CPU path.
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..tencentpretrain.embeddings import Embedding, str2embedding
from ..tencentpretrain.encoders import str2encoder
from ..tencentpretrain.opts import finetune_opts, tokenizer_opts

_CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
VIT_CONFIG = os.path.join(_CFG_DIR, "vit_base_16_224.json")
TEXT_CONFIG = os.path.join(_CFG_DIR, "roberta_base.json")
ROBERTA_VOCAB = 50265      # models/huggingface_gpt2_vocab.txt + the xlm-roberta specials used by roberta_base_en_model


def encoder_args(config_path: str, **over) -> argparse.Namespace:
    """argparse defaults of finetune_opts + tokenizer_opts overlaid with a model JSON, as load_hyperparam composes them
    (tencentpretrain/utils/config.py:6-23)."""
    p = argparse.ArgumentParser()
    finetune_opts(p)
    tokenizer_opts(p)
    d = vars(p.parse_args([]))
    with open(config_path) as f:
        d.update(json.load(f))
    d.update(over)
    return argparse.Namespace(**d)


class EncoderStack(nn.Module):
    """embedding -> encoder of tencentpretrain/models/model.py:32-41 (the part in front of the target)."""

    def __init__(self, args, vocab_size: int):
        super().__init__()
        self.embedding = Embedding(args)
        for name in args.embedding:
            self.embedding.update(str2embedding[name](args, vocab_size), name)
        self.encoder = str2encoder[args.encoder](args)

    def forward(self, src, seg):
        return self.encoder(self.embedding(src, seg), seg)

    def forward_first_token(self, src, seg):
        """encoder output at token 0 only, [batch, hidden] (TransformerEncoder.forward_first_token)."""
        return self.encoder.forward_first_token(self.embedding(src, seg), seg)


class FeatureExtractor(nn.Module):
    """ViT-B/16 over the frames of an item and RoBERTa-base over its tags' token sequences (module docstring).
    `image` / `text` are EncoderStacks whose state_dict keys are the reference's (`embedding.patch.*`,
    `encoder.transformer.{i}.*`, ...), so `vit_base_patch16_224_model.bin` / `roberta_base_en_model.bin` load with
    `load_pretrained`."""

    def __init__(self, vit_args: Optional[argparse.Namespace] = None, text_args: Optional[argparse.Namespace] = None,
                 vocab_size: int = ROBERTA_VOCAB, seq_length: int = 196):
        super().__init__()
        self.vit_args = vit_args or encoder_args(VIT_CONFIG)
        self.text_args = text_args or encoder_args(TEXT_CONFIG)
        self.seq_length = seq_length
        if self.vit_args.hidden_size != self.text_args.hidden_size:
            raise ValueError("image and text encoders must share the feature width (the heads take one visual_feat_dim)")
        self.image = EncoderStack(self.vit_args, vocab_size)
        self.text = EncoderStack(self.text_args, vocab_size)
        self.text.embedding.defer_id_check = True          # one check per extract() at its end, not one sync per call

    def load_pretrained(self, vit_path: Optional[str] = None, text_path: Optional[str] = None):
        """Released TencentPretrain checkpoints carry a `target.*` head next to embedding.* / encoder.*: dropped here."""
        for stack, path in ((self.image, vit_path), (self.text, text_path)):
            if path:
                sd = torch.load(path, map_location="cpu")
                stack.load_state_dict({k: v for k, v in sd.items() if k.startswith(("embedding.", "encoder."))}, strict=True)

    def init_normal(self, std: float = 0.02, generator=None):
        """The reference's initialiser for models without a checkpoint (finetune/ppo.py:362-365): N(0, 0.02) on everything
        but gamma / beta."""
        for n, p in self.named_parameters():
            if "gamma" not in n and "beta" not in n:
                p.data.normal_(0, std, generator=generator)

    def image_features(self, frames: torch.Tensor) -> torch.Tensor:
        """frames [B, n_img, 3, H, W] (uint8, or fp32 already normalised) -> img_emb [B, n_img, hidden]."""
        B, n_img = frames.shape[:2]
        flat = frames.reshape(B * n_img, *frames.shape[2:])
        L = self.vit_args.max_seq_length
        seg = torch.ones(B * n_img, L, dtype=torch.int64, device=frames.device)
        # pooling(h, seg, "first") of utils/misc.py:23-35 = (h * seg)[:, 0, :]: only token 0 of the encoder output is consumed,
        # so the last layer is evaluated for that token only (inference; bit-for-bit the full schedule's kernels on B rows)
        h0 = self.image.forward_first_token(flat, seg)
        return (h0 * seg[:, :1].type_as(h0)).reshape(B, n_img, -1)

    def text_features(self, ids: torch.Tensor, seg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids [B, T, L] int64 (+ seg [B, T, L]; default all ones) -> text_emb [B, T, L, hidden]."""
        B, T, L = ids.shape
        if L != self.seq_length:
            raise ValueError(f"token sequences must have length {self.seq_length} (finetune/ppo.py:219-220), got {L}")
        if seg is None:
            seg = torch.ones_like(ids)
        h = self.text(ids.reshape(B * T, L), seg.reshape(B * T, L))
        return h.reshape(B, T, L, -1)

    def forward(self, frames, ids, seg=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (text_emb [B, T, 196, 768], img_emb [B, n_img, 768]) -- the arguments of Actor / Critic / Reward.forward,
        rollout_step and pointwise.train_model."""
        if not frames.is_cuda or not ids.is_cuda:
            raise TypeError("lr2ppo_amd: frames / ids must live on the HIP device (no CPU path)")
        img_emb = self.image_features(frames)
        text_emb = self.text_features(ids, seg)
        return text_emb, img_emb

    # ---- explicit training schedule: the encoders fine-tuned behind a head, no autograd graph ------------------------------
    # forward_train keeps what the hand-written backward needs (one arena per stack), backward_train takes the head's input
    # gradients (engine_backward(input_grads=True)) and writes every encoder / embedding gradient into persistent flat buffers
    # that bind_grads() exposes as p.grad.  Same kernels and numerics as the autograd route (Actor(*fx(frames, ids, seg)) with
    # loss.backward()), which stays available; this one allocates nothing per step but the activation arenas.
    def _stacks(self):
        return ((self.image.embedding, self.image.encoder), (self.text.embedding, self.text.encoder))

    def bind_grads(self):
        """p.grad of every encoder / embedding parameter -> its slice of the module's persistent gradient buffer."""
        for emb, enc in self._stacks():
            for mod in (emb, enc):
                for q, g in mod.grad_buffers().items():
                    if q.grad is None or q.grad.data_ptr() != g.data_ptr():
                        q.grad = g

    def grad_flats(self):
        """The four flat gradient buffers (image embedding / encoder, text embedding / encoder): what a data-parallel run
        all-reduces, one collective each."""
        out = []
        for emb, enc in self._stacks():
            for mod in (emb, enc):
                mod.grad_buffers()
                out.append(mod._gflat)
        return out

    @torch.no_grad()
    def forward_train(self, frames, ids, seg=None):
        """-> (text_emb [B, T, L, E], img_emb [B, n_img, E], ctx) with dropout at the reference's sites when self.training
        (embedding dropout, attention probabilities, dropout_1, dropout_2: embeddings/embedding.py:33, layers/transformer.py:50-73,
        multi_headed_attn.py:68); ctx goes to backward_train."""
        if not frames.is_cuda or not ids.is_cuda:
            raise TypeError("lr2ppo_amd: frames / ids must live on the HIP device (no CPU path)")
        B, n_img = frames.shape[:2]
        T, L = ids.shape[1:]
        if L != self.seq_length:
            raise ValueError(f"token sequences must have length {self.seq_length} (finetune/ppo.py:219-220), got {L}")
        if seg is None:
            seg = torch.ones_like(ids)
        vseg = torch.ones(B * n_img, self.vit_args.max_seq_length, dtype=torch.int64, device=frames.device)
        e_img, s_img_emb = self.image.embedding._run(frames.reshape(B * n_img, *frames.shape[2:]), vseg, save=True)
        h_img, s_img_enc = self.image.encoder._forward_train(e_img, vseg)
        img_emb = h_img[:, 0, :].reshape(B, n_img, -1).contiguous()      # pooling 'first' with seg == 1 (utils/misc.py:23-35)
        tseg = seg.reshape(B * T, L)
        e_txt, s_txt_emb = self.text.embedding._run(ids.reshape(B * T, L), tseg, save=True)
        h_txt, s_txt_enc = self.text.encoder._forward_train(e_txt, tseg)
        text_emb = h_txt.reshape(B, T, L, -1).clone()                    # not a view of the activation arena
        ctx = {"img": (s_img_emb, s_img_enc, tuple(h_img.shape)), "txt": (s_txt_emb, s_txt_enc)}
        return text_emb, img_emb, ctx

    @torch.no_grad()
    def backward_train(self, ctx, d_text, d_img):
        """Gradients of every encoder / embedding parameter for the forward that produced ctx, given d loss / d text_emb and
        d loss / d img_emb; written into the persistent buffers (bind_grads())."""
        s_txt_emb, s_txt_enc = ctx.pop("txt")
        B_, L, E = s_txt_enc["dims"][:3]
        d_emb, _ = self.text.encoder._backward_train(s_txt_enc, d_text.reshape(B_, L, E).contiguous(),
                                                     G=self.text.encoder.grad_buffers())
        self.text.embedding._backward(s_txt_emb, d_emb, G_out=self.text.embedding.grad_buffers())
        del s_txt_enc, s_txt_emb, d_emb
        s_img_emb, s_img_enc, hshape = ctx.pop("img")
        dout = torch.zeros(hshape, device=d_img.device)                  # only the pooled [CLS] row carries a gradient
        dout[:, 0, :] = d_img.reshape(-1, hshape[-1])
        d_emb, _ = self.image.encoder._backward_train(s_img_enc, dout, G=self.image.encoder.grad_buffers())
        self.image.embedding._backward(s_img_emb, d_emb, G_out=self.image.embedding.grad_buffers())

    @torch.no_grad()
    def extract(self, frames, ids, seg=None, check_ids: bool = True):
        """Inference-mode forward (the encoders are frozen feature extractors in front of the PPO loop)."""
        was = self.training
        self.eval()
        try:
            out = self.forward(frames, ids, seg)
        finally:
            self.train(was)
        if check_ids:
            self.text.embedding.check_ids()
        return out


def synthetic_raw_batch(batch: int, tags: int, n_img: int = 16, seq_length: int = 196, vocab: int = ROBERTA_VOCAB,
                        device=None, generator=None):
    """Synthetic raw inputs of SURVEY.md 8d: frames uint8 [B, n_img, 3, 224, 224] uniform; token ids uniform in [5, vocab);
    seg = 1 on the first len ~ U{4..seq_length} tokens, 0 after; targets uniform in {0, 1, 2}."""
    frames = torch.randint(0, 256, (batch, n_img, 3, 224, 224), dtype=torch.uint8, device=device, generator=generator)
    ids = torch.randint(5, vocab, (batch, tags, seq_length), device=device, generator=generator)
    lens = torch.randint(4, seq_length + 1, (batch, tags, 1), device=device, generator=generator)
    seg = (torch.arange(seq_length, device=device).view(1, 1, -1) < lens).to(torch.int64)
    tgts = torch.randint(0, 3, (batch, tags), device=device, generator=generator)
    return frames, ids, seg, tgts


def build_encoder_optimizer(args, fx: FeatureExtractor):
    """AdamW + schedule over the two encoder stacks with the reference's grouping (weight decay 0.01 except names containing
    bias / gamma / beta: finetune/ppo.py:381-393, the rule of every TencentPretrain fine-tuning script) -> (optimizer, scheduler)."""
    from ..tencentpretrain.utils.optimizers import str2optimizer, str2scheduler
    from .ppo import _grouped
    opt = str2optimizer[args.optimizer](_grouped(list(fx.named_parameters())), lr=args.learning_rate, correct_bias=False)
    if args.scheduler == "constant":
        sch = str2scheduler[args.scheduler](opt)
    elif args.scheduler == "constant_with_warmup":
        sch = str2scheduler[args.scheduler](opt, args.train_steps * args.warmup)
    else:
        sch = str2scheduler[args.scheduler](opt, args.train_steps * args.warmup, args.train_steps)
    return opt, sch


def finetune_pointwise_step(args, fx: FeatureExtractor, model, optimizer, scheduler, enc_optimizer, enc_scheduler, frames, ids, seg,
                            tgts):
    """BASELINE configs[1] with the encoders trained end to end: frames + token ids -> ViT-B/16 + RoBERTa-base (train mode) ->
    finetune/pointwise.py's Classifier -> SmoothL1 (NLL in 'cls') -> head backward WITH input gradients -> encoder + embedding
    backward -> AdamW over the head and over both stacks, one scheduler step each.  The composition is the reference's own
    `encoder(embedding(src, seg), seg)` feeding a target (tencentpretrain/models/model.py:32-41) with the stage-1 head as the
    target (finetune/pointwise.py:300-313); gradients are averaged over ranks before the steps.  -> loss (0-dim device tensor)."""
    import torch.distributed as dist
    from .ppo import _DataParallel
    dev = frames.device
    model.bind_grads()
    fx.bind_grads()
    dp = _DataParallel()
    text_emb, img_emb, ctx = fx.forward_train(frames, ids, seg)
    logits = model.engine_forward(text_emb, img_emb, save=True)
    loss, dlogits = torch.empty(1, device=dev), torch.empty_like(logits)
    if model.n_out > 1:
        ops.nll_loss(logits, tgts.to(device=dev, dtype=torch.int64).contiguous().view(-1), loss, dlogits, rows=logits.shape[0],
                     C=model.n_out)
    else:
        ops.smooth_l1(logits.view(-1), tgts.to(device=dev, dtype=torch.float32).contiguous().view(-1), loss, dlogits.view(-1),
                      n=logits.numel(), beta=0.3)
    fuse = getattr(args, "fuse_fc1_update", True) and hasattr(optimizer, "external_update")
    fa = optimizer.external_update(model.out_layer.fc1.weight) if fuse else None
    d_text, d_img = model.engine_backward(dlogits, dp, fc1_update=fa, input_grads=True)
    wh = dp.reduce_start(model)                      # the head's tail all-reduce overlaps the encoder backward
    del text_emb, logits
    fx.backward_train(ctx, d_text, d_img)
    works = []
    if dp.active:
        for g in fx.grad_flats():
            g.div_(dp.world)
            works.append(dist.all_reduce(g, async_op=True))
    dp.finish(wh)
    optimizer.step()
    for w in works:
        w.wait()
    enc_optimizer.step()
    scheduler.step()
    enc_scheduler.step()
    return loss[0]
