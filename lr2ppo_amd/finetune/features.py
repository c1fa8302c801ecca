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

An image tower wider than the heads' feature width (BASELINE.json configs[4]: ViT-L/14, hidden 1024, in front of heads that
hard-code 768: finetune/ppo.py:202-208,219-220) ends in a `VisualProjection` -- the bias-free `[hidden_img -> visual_feat_dim]`
map CLIP applies to its pooled [CLS] row (`x = ln_post(x[:, 0, :]) @ proj`, the `encode_image` preprocess.py:59-61,83 calls to make
the reference's img_emb); absent when the widths agree.  `precision="mxfp8"` routes both stacks through the MX-FP8 products of
csrc/fp8.hip (inference only; the parity path stays split-bf16).
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import engine, ops
from ..tencentpretrain.embeddings import Embedding, str2embedding
from ..tencentpretrain.encoders import str2encoder
from ..tencentpretrain.opts import finetune_opts, tokenizer_opts

_CFG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
VIT_CONFIG = os.path.join(_CFG_DIR, "vit_base_16_224.json")
VIT_L14_CONFIG = os.path.join(_CFG_DIR, "vit_large_14_224.json")       # BASELINE.json configs[4]: the image-tower swap
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


class VisualProjection(nn.Module):
    """The map from the image tower's width to the heads' feature width: `y = x @ weight.T`, bias-free, on the pooled [CLS]
    rows -- CLIP's `proj` behind `ln_post` (the encoder stack's final LayerNorm plays ln_post here).  `weight` keeps nn.Linear's
    `[out, in]` orientation; state_dict key `visual_projection.weight`.  Forward, input gradient and weight gradient are the
    package's GEMM kernel (fp32 operands, split-bf16 products: engine.linear_fwd / linear_dgrad / linear_wgrad)."""

    def __init__(self, hidden: int, feat: int):
        super().__init__()
        self.in_features, self.out_features = hidden, feat
        self.weight = nn.Parameter(torch.empty(feat, hidden))
        nn.init.normal_(self.weight, 0.0, hidden ** -0.5)       # CLIP's own initialiser for `proj` (scale = width ** -0.5)
        self._ws = None
        self._gflat = None

    def _workspace(self, dev):
        if self._ws is None or self._ws.device != dev:
            self._ws = engine.Workspace(dev)
        return self._ws

    @torch.no_grad()
    def _fwd(self, x):
        if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 2 or x.shape[1] != self.in_features:
            raise TypeError(f"VisualProjection: a float32 [rows, {self.in_features}] tensor on the HIP device")
        x = x.contiguous()
        n = x.shape[0]
        out = torch.empty(n, self.out_features, device=x.device)
        engine.linear_fwd(self._workspace(x.device), x, self.weight.data, None, out, n, self.out_features, self.in_features)
        return out

    @torch.no_grad()
    def _bwd(self, x, dy, dw, need_dx=True):
        """dw [out, in] <- dy^T x (overwritten); -> dx [rows, in] = dy W, or None."""
        ws, n = self._workspace(x.device), x.shape[0]
        x, dy = x.contiguous(), dy.contiguous()
        engine.linear_wgrad(ws, dy, x, dw, None, n, self.in_features, self.out_features)
        if not need_dx:
            return None
        dx = torch.empty(n, self.in_features, device=x.device)
        engine.linear_dgrad(ws, dy, self.weight.data, dx, n, self.in_features, self.out_features)
        return dx

    def grad_buffers(self):
        """{parameter: gradient}: one persistent tensor (see TransformerEncoder.grad_buffers)."""
        dev = self.weight.device
        if self._gflat is None or self._gflat.device != dev:
            self._gflat = torch.zeros(self.weight.numel(), device=dev)
            self._gviews = {self.weight: self._gflat.view_as(self.weight)}
        return self._gviews

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return _ProjectionFn.apply(self, x, self.weight)
        return self._fwd(x)


class _ProjectionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, weight):
        ctx.mod, ctx.need_dx = mod, x.requires_grad
        ctx.save_for_backward(x)
        return mod._fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dw = torch.empty_like(ctx.mod.weight)
        dx = ctx.mod._bwd(x, dy, dw, need_dx=ctx.need_dx)
        return None, dx, (dw if ctx.mod.weight.requires_grad else None)


class FeatureExtractor(nn.Module):
    """An image tower (ViT-B/16; ViT-L/14 with `vit_args=encoder_args(VIT_L14_CONFIG)`) over the frames of an item and RoBERTa-base
    over its tags' token sequences (module docstring).  `image` / `text` are EncoderStacks whose state_dict keys are the reference's
    (`embedding.patch.*`, `encoder.transformer.{i}.*`, ...), so `vit_base_patch16_224_model.bin` / `roberta_base_en_model.bin` load
    with `load_pretrained`.  feat_dim: the width of both outputs = the heads' `visual_feat_dim` (default: the text stack's hidden
    size); an image tower of another width gets `visual_projection` (VisualProjection) behind its pooled row.
    precision: "split_bf16" (default: the fp32-grade parity path) or "mxfp8" -- extract() / no-grad forward() run every
    projection of both stacks as an MX-FP8 product (TransformerEncoder.forward_fp8); training paths refuse that mode."""

    PRECISIONS = ("split_bf16", "mxfp8")

    def __init__(self, vit_args: Optional[argparse.Namespace] = None, text_args: Optional[argparse.Namespace] = None,
                 vocab_size: int = ROBERTA_VOCAB, seq_length: int = 196, feat_dim: Optional[int] = None,
                 precision: str = "split_bf16"):
        super().__init__()
        self.vit_args = vit_args or encoder_args(VIT_CONFIG)
        self.text_args = text_args or encoder_args(TEXT_CONFIG)
        self.seq_length = seq_length
        self.feat_dim = int(feat_dim or self.text_args.hidden_size)
        if self.text_args.hidden_size != self.feat_dim:
            raise ValueError("the text stack feeds text_proj directly (finetune/ppo.py:202,215): its hidden size must equal the "
                             f"heads' feature width {self.feat_dim}, got {self.text_args.hidden_size}")
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {self.PRECISIONS}")
        self.precision = precision
        self.image = EncoderStack(self.vit_args, vocab_size)
        self.text = EncoderStack(self.text_args, vocab_size)
        self.visual_projection = (VisualProjection(self.vit_args.hidden_size, self.feat_dim)
                                  if self.vit_args.hidden_size != self.feat_dim else None)
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
        if self._fp8_now():
            h0 = self.image.encoder.forward_fp8(self.image.embedding(flat, seg), seg, first_only=True)
        else:
            h0 = self.image.forward_first_token(flat, seg)
        h0 = h0 * seg[:, :1].type_as(h0)
        if self.visual_projection is not None:
            h0 = self.visual_projection(h0)
        return h0.reshape(B, n_img, -1)

    def _fp8_now(self) -> bool:
        """True when this call takes the MX-FP8 route: precision "mxfp8" and nothing asks for gradients or dropout."""
        if self.precision != "mxfp8":
            return False
        if self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("FeatureExtractor(precision='mxfp8') is an inference mode (frozen feature extraction: "
                                      "extract(), or eval() under torch.no_grad()); train the encoders in 'split_bf16'")
        return True

    def text_features(self, ids: torch.Tensor, seg: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ids [B, T, L] int64 (+ seg [B, T, L]; default all ones) -> text_emb [B, T, L, hidden]."""
        B, T, L = ids.shape
        if L != self.seq_length:
            raise ValueError(f"token sequences must have length {self.seq_length} (finetune/ppo.py:219-220), got {L}")
        if seg is None:
            seg = torch.ones_like(ids)
        ids2, seg2 = ids.reshape(B * T, L), seg.reshape(B * T, L)
        if self._fp8_now():
            h = self.text.encoder.forward_fp8(self.text.embedding(ids2, seg2), seg2)
        else:
            h = self.text(ids2, seg2)
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
    def _grad_modules(self):
        """Modules with a persistent flat gradient buffer, in backward order of completion (text stack first)."""
        mods = [self.text.encoder, self.text.embedding]
        if self.visual_projection is not None:
            mods.append(self.visual_projection)
        return mods + [self.image.encoder, self.image.embedding]

    def bind_grads(self):
        """p.grad of every encoder / embedding / projection parameter -> its slice of the module's persistent gradient buffer."""
        for mod in self._grad_modules():
            for q, g in mod.grad_buffers().items():
                if q.grad is None or q.grad.data_ptr() != g.data_ptr():
                    q.grad = g

    def grad_flats(self, part: Optional[str] = None):
        """The flat gradient buffers (text encoder / embedding, [visual projection,] image encoder / embedding): what a
        data-parallel run all-reduces, one collective each.  part: "text" / "image" = that tower's buffers only."""
        out = []
        for mod in self._grad_modules():
            is_text = mod is self.text.encoder or mod is self.text.embedding
            if part is None or (part == "text") == is_text:
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
        if self.precision != "split_bf16":
            raise NotImplementedError("forward_train: the encoders train in 'split_bf16' (precision='mxfp8' is inference only)")
        B, n_img = frames.shape[:2]
        T, L = ids.shape[1:]
        if L != self.seq_length:
            raise ValueError(f"token sequences must have length {self.seq_length} (finetune/ppo.py:219-220), got {L}")
        if seg is None:
            seg = torch.ones_like(ids)
        vseg = torch.ones(B * n_img, self.vit_args.max_seq_length, dtype=torch.int64, device=frames.device)
        e_img, s_img_emb = self.image.embedding._run(frames.reshape(B * n_img, *frames.shape[2:]), vseg, save=True)
        h_img, s_img_enc = self.image.encoder._forward_train(e_img, vseg)
        cls = h_img[:, 0, :].contiguous()                                # pooling 'first' with seg == 1 (utils/misc.py:23-35)
        img_emb = (cls if self.visual_projection is None else self.visual_projection._fwd(cls)).reshape(B, n_img, -1)
        tseg = seg.reshape(B * T, L)
        e_txt, s_txt_emb = self.text.embedding._run(ids.reshape(B * T, L), tseg, save=True)
        h_txt, s_txt_enc = self.text.encoder._forward_train(e_txt, tseg)
        text_emb = h_txt.reshape(B, T, L, -1).clone()                    # not a view of the activation arena
        ctx = {"img": (s_img_emb, s_img_enc, tuple(h_img.shape), cls if self.visual_projection is not None else None),
               "txt": (s_txt_emb, s_txt_enc)}
        return text_emb, img_emb, ctx

    @torch.no_grad()
    def backward_train(self, ctx, d_text, d_img, after_text=None):
        """Gradients of every encoder / embedding / projection parameter for the forward that produced ctx, given
        d loss / d text_emb and d loss / d img_emb; written into the persistent buffers (bind_grads()).
        after_text: called once the text tower's gradients are complete (its buffers can start travelling -- the data-parallel
        all-reduce of ~500 MB -- while the image tower's backward still runs)."""
        s_txt_emb, s_txt_enc = ctx.pop("txt")
        B_, L, E = s_txt_enc["dims"][:3]
        d_emb, _ = self.text.encoder._backward_train(s_txt_enc, d_text.reshape(B_, L, E).contiguous(),
                                                     G=self.text.encoder.grad_buffers())
        self.text.embedding._backward(s_txt_emb, d_emb, G_out=self.text.embedding.grad_buffers())
        del s_txt_enc, s_txt_emb, d_emb
        if after_text is not None:
            after_text()
        s_img_emb, s_img_enc, hshape, cls = ctx.pop("img")
        d_cls = d_img.reshape(hshape[0], -1)
        if self.visual_projection is not None:
            vp = self.visual_projection
            d_cls = vp._bwd(cls, d_cls, vp.grad_buffers()[vp.weight])
        dout = torch.zeros(hshape, device=d_img.device)                  # only the pooled [CLS] row carries a gradient
        dout[:, 0, :] = d_cls
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


# ---- launcher plumbing shared by the three stages' main(): raw items in, features extracted in line ---------------------------
IMAGE_TOWERS = {"vit_base_16_224": VIT_CONFIG, "vit_large_14_224": VIT_L14_CONFIG}


def finetune_ppo_step(args, fx: FeatureExtractor, model, reward_model, optimizer, critic_optim, enc_optimizer, frames, ids, seg, tgts):
    """One PPO step (a rollout timestep + an update minibatch) with the encoders TRAINED through it -- the stage-3 twin of
    finetune_pointwise_step, an explicit schedule (no autograd graph):
        rollout    frames + ids -> both stacks in EVAL mode, no gradient -> ppo.rollout_step (finetune/ppo.py:844-883 on its features)
        update     the same inputs -> both stacks in TRAIN mode, activations kept -> ppo.update_minibatch on those features
                   (finetune/ppo.py:518-598; actor / critic AdamW inside) with the heads handing back d loss / d features
                   (policy loss through the actor, value loss through the critic, summed) -> encoder + embedding backward ->
                   AdamW over both stacks.
    The reference trains stage 3 on pre-extracted features only (finetune/ppo.py:827-835); the composition is its own
    `encoder(embedding(src, seg), seg)` (tencentpretrain/models/model.py:32-41) in front of its train_model body.  Gradients are
    averaged over ranks before the steps (the text stack's all-reduce travels under the image stack's backward).
    -> the update's 10 metrics (device tensor)."""
    import torch.distributed as dist
    from . import ppo
    dp = ppo._DataParallel()
    dev = frames.device
    was_training = (model.training, fx.training)
    model.eval(), fx.eval()
    with torch.no_grad():
        text0, img0 = fx.extract(frames, ids, seg)
        record = ppo.rollout_step(model, reward_model, text0, img0, tgts.to(dev))
    del text0, img0
    model.train(), fx.train()
    model.actor.bind_grads(), model.critic.bind_grads()
    fx.bind_grads()
    text_emb, img_emb, ctx = fx.forward_train(frames, ids, seg)
    record = tuple(record[:5]) + (text_emb, img_emb, record[7])
    metrics, d_text, d_img = ppo.update_minibatch(args, model, optimizer, critic_optim, record, dp, input_grads=True)
    del text_emb, record
    works = []

    def exchange(part):
        if dp.active:
            for g in fx.grad_flats(part):
                g.div_(dp.world)
                works.append(dist.all_reduce(g, async_op=True))
    fx.backward_train(ctx, d_text, d_img, after_text=lambda: exchange("text"))
    exchange("image")
    for w in works:
        w.wait()
    enc_optimizer.step()
    model.train(was_training[0]), fx.train(was_training[1])
    return metrics


def raw_input_opts(parser):
    """Flags of this build (not in the reference, whose loaders read pre-extracted features: finetune/ppo.py:115-148): run the
    encoder stacks in front of a stage's head.  The reference's launchers already carry --pretrained_model_path /
    --vit_pretrained_model_path for the RoBERTa and ViT checkpoints; with --raw_inputs they load into the two stacks."""
    parser.add_argument("--raw_inputs", action="store_true",
                        help="items are raw (uint8 frames, tag token ids, seg): features come from the image tower + RoBERTa-base in line")
    parser.add_argument("--encoder_layers", type=int, default=0, help="override layers_num of both encoder configs (0 = as shipped)")
    parser.add_argument("--image_tower", type=str, default="vit_base_16_224",
                        help="with --raw_inputs: vit_base_16_224 | vit_large_14_224 (BASELINE configs[4]; ends in the 1024 -> "
                             "visual_feat_dim projection) | path of a TencentPretrain model JSON")
    parser.add_argument("--fp8_features", action="store_true",
                        help="with --raw_inputs and frozen encoders: every projection of both stacks as an MX-FP8 product on the "
                             "block-scaled MFMA (FeatureExtractor(precision='mxfp8'); a few per cent from the default features)")
    return parser


def build_extractor(args, num_tasks: int = 1, trainable: bool = False) -> FeatureExtractor:
    """The FeatureExtractor a launcher's flags describe, on args.device, identical on every rank."""
    import torch.distributed as dist
    over = {"layers_num": args.encoder_layers} if getattr(args, "encoder_layers", 0) else {}
    tower = getattr(args, "image_tower", "vit_base_16_224")
    fp8 = bool(getattr(args, "fp8_features", False))
    if fp8 and trainable:
        raise ValueError("--fp8_features is for frozen feature extraction; drop it or --finetune_encoders")
    fx = FeatureExtractor(encoder_args(IMAGE_TOWERS.get(tower, tower), **over), encoder_args(TEXT_CONFIG, **over),
                          seq_length=args.seq_length, feat_dim=args.visual_feat_dim, precision="mxfp8" if fp8 else "split_bf16")
    text_path, vit_path = getattr(args, "pretrained_model_path", None), getattr(args, "vit_pretrained_model_path", None)
    if text_path or vit_path:
        if not (text_path and vit_path):
            raise ValueError("--raw_inputs: give both --pretrained_model_path (RoBERTa) and --vit_pretrained_model_path, or neither")
        fx.load_pretrained(vit_path, text_path)
    else:
        fx.init_normal()
    fx = fx.to(args.device)
    if num_tasks > 1:
        for p in fx.parameters():
            dist.broadcast(p.data, src=0)
    return fx


class SyntheticRawItems(torch.utils.data.Dataset):
    """Seeded raw stand-in for LRMovieNet items: uint8 frames [max_imgs, 3, H, W], tag token ids [tags, 196] uniform
    in [5, vocab), seg = 1 on the first len ~ U{4..196} tokens, targets in {0, 1, 2} -> (frames, ids, seg, tgts, *extra(i, tgts, g)).
    extra: a stage's additional per-item fields (stage 2: chosen / reject index)."""

    def __init__(self, n_items, tags, max_imgs=16, seed=7, extra=None):
        self.n, self.tags, self.max_imgs, self.seed, self.extra = n_items, tags, max_imgs, seed, extra

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        frames, ids, seg, tgts = synthetic_raw_batch(1, self.tags, n_img=self.max_imgs, generator=g)
        item = (frames[0], ids[0], seg[0], tgts[0])
        return item + tuple(self.extra(i, tgts[0], g)) if self.extra is not None else item


class ExtractingLoader:
    """A loader of raw batches (frames, ids, seg, *rest) seen as the reference's loader of (text_emb, img_emb, *rest): the frozen
    extractor runs in line, on the device, as each batch is drawn."""

    def __init__(self, loader, fx: FeatureExtractor, device):
        self.loader, self.fx, self.device, self.sampler = loader, fx, device, getattr(loader, "sampler", None)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for frames, ids, seg, *rest in self.loader:
            text_emb, img_emb = self.fx.extract(frames.to(self.device), ids.to(self.device), seg.to(self.device))
            yield (text_emb, img_emb, *rest)


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
    works = []

    def exchange(part):
        # one all-reduce per flat buffer, issued as soon as that tower's backward has been enqueued: RCCL's stream waits for the
        # compute stream at that point, so the text tower's ~500 MB travel while the image tower's backward runs
        if dp.active:
            for g in fx.grad_flats(part):
                g.div_(dp.world)
                works.append(dist.all_reduce(g, async_op=True))
    fx.backward_train(ctx, d_text, d_img, after_text=lambda: exchange("text"))
    exchange("image")
    dp.finish(wh)
    optimizer.step()
    for w in works:
        w.wait()
    enc_optimizer.step()
    scheduler.step()
    enc_scheduler.step()
    return loss[0]
