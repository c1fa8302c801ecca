"""The two-dataset `_trad` pointwise ranker -- drop-in for the model / step of the reference's
finetune/pointwise_2data_trad.py (BASELINE.json configs[0]'s "136-dim MLP ranker": MSLR-WEB10K rows carry 136 raw LETOR
features, MQ2008 rows 46).

`Classifier` (pointwise_2data_trad.py:130-177): text_proj = Mlp(46, 3072, 768) or text_proj3 = Mlp(136, 3072, 768), chosen by
the width of the batch, in front of pointwise_trad's sequence-length-1 head (XiT over the feature with itself, concat,
out_layer = Mlp(1536, 3072, 768), Linear(768, 1)); SmoothL1(beta = 0.3), AdamW, per-batch scheduler.  The projection that a
batch does not use gets NO gradient upstream (`.grad is None`: AdamW skips it, no weight decay either): reproduced by
unbinding its gradients for that step.  Same kernels as every other head (`engine.feature_proj_forward / backward`,
`engine.trad_trunk_forward / backward`); mode 'reg'.  `LTRDataset` (pointwise_2data_trad.py:87-108) = pointwise_trad's reader, one
instance per data set (`--train_path`, `--train_path2`).  No CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import engine, ops, runtime
from . import pointwise_trad as pt
from .pointwise_trad import OUT_FC1, OUT_FC2, LTRDataset, build_optimizer, load_or_initialize_parameters  # noqa: F401
from .ppo import FEAT, Mlp
from .xit import XiT

PROJ = {46: "text_proj", 136: "text_proj3"}          # pointwise_2data_trad.py:146-150


class Classifier(pt.Classifier):
    """forward(text_emb [bs, docs, 46 | 136], img_emb (ignored), tgts) -> (loss, logits [bs*docs, 1]) or logits."""

    def __init__(self, args, vit_args=None):
        nn.Module.__init__(self)
        self.mode, self.labels_num = args.mode, args.labels_num
        if self.mode != "reg":
            raise NotImplementedError("the HIP path implements mode='reg'")
        self.text_proj = Mlp(46, 4 * FEAT, FEAT, nn.GELU, 0)
        self.text_proj3 = Mlp(136, 4 * FEAT, FEAT, nn.GELU, 0)
        self.xit = XiT(feat_size=FEAT)
        self.out_layer = Mlp(2 * FEAT, 4 * FEAT, FEAT, nn.GELU, 0)
        self.head = nn.Linear(FEAT, 1)
        self._ws: Optional[engine.Workspace] = None
        self._wp: Optional[engine.WeightPlanes] = None
        self._G: Optional[Dict[str, torch.Tensor]] = None
        self._saved = None

    def bind_grads(self, width: Optional[int] = None):
        """.grad of every parameter -> its gradient buffer; the projection a `width`-wide batch does not use: None."""
        unused = [v for k, v in PROJ.items() if width is not None and k != width]
        for n, p in self.named_parameters():
            p.grad = None if any(n.startswith(u + ".") for u in unused) else self.grad_buffers()[n]

    @torch.no_grad()
    def engine_forward(self, text_emb, *, save: bool):
        if text_emb.dtype != torch.float32 or not text_emb.is_cuda:
            raise TypeError("lr2ppo_amd: text_emb must be a float32 tensor on the HIP device (no CPU path)")
        if text_emb.dim() != 3 or text_emb.shape[-1] not in PROJ:
            raise ValueError("text_emb must be [bs, docs, 46] (MQ2008) or [bs, docs, 136] (MSLR-WEB10K) (pointwise_2data_trad.py:146-150)")
        dev, Kin = text_emb.device, text_emb.shape[-1]
        if self._ws is None or self._ws.device != dev:
            self._ws = engine.Workspace(dev)
        ws, P = self._ws, self._P()
        W = self._weights(P)
        N = text_emb.shape[0] * text_emb.shape[1]
        x0 = engine.feature_proj_forward(ws, P, PROJ[Kin], text_emb.contiguous(), N, Kin, FEAT, save=save)
        drop = runtime.next_drop(engine.DROP_P, 0) if self.training else None
        g2 = engine.trad_trunk_forward(ws, P, W, x0, N, FEAT, save=save, drop=drop)
        logits = torch.empty(N, device=dev)
        ops.head_fwd(g2, P["head.weight"], P["head.bias"], logits, rows=N, D=FEAT)
        if save:
            self._saved = (x0, N, drop, Kin)
            self._unused_prefixes = tuple(v + "." for k, v in PROJ.items() if k != Kin)     # no gradient upstream: .grad stays None
        return logits.view(-1, 1)

    @torch.no_grad()
    def engine_backward(self, dlogits, input_grads: bool = False):
        if input_grads:
            raise NotImplementedError("pointwise_2data_trad.Classifier: the raw LETOR feature rows (46 / 136 columns) are data; "
                                      "their gradient is not built -- detach text_emb")
        x0, N, drop, Kin = self._saved
        ws, P, G = self._ws, self._P(), self.grad_buffers()
        W = self._wp.planes
        g2, dg2 = ws.mat("g2", N, FEAT), ws.mat("dg2", N, FEAT)
        ops.head_bwd(g2, P["head.weight"], dlogits.contiguous().view(-1), dg2, G["head.weight"], G["head.bias"], rows=N, D=FEAT)
        dx0 = engine.trad_trunk_backward(ws, P, W, G, x0, dg2, N, FEAT, drop=drop, want_dx=True)
        engine.feature_proj_backward(ws, P, G, PROJ[Kin], dx0, N, Kin, FEAT)
        self._saved = None


def train_model(args, model, optimizer, scheduler, text_emb_batch, img_emb_batch, tgts_batch):
    """One batch (pointwise_2data_trad.py:240-253): forward, SmoothL1, backward, AdamW step (the unused projection is
    skipped, as upstream where its .grad is None), scheduler step -> the loss as a 0-dim device tensor."""
    dev = text_emb_batch.device
    model.bind_grads(text_emb_batch.shape[-1])
    logits = model.engine_forward(text_emb_batch, save=True)
    loss, dlogits = torch.empty(1, device=dev), torch.empty_like(logits)
    target = tgts_batch.to(device=dev, dtype=torch.float32).contiguous().view(-1)
    ops.smooth_l1(logits.view(-1), target, loss, dlogits.view(-1), n=logits.numel(), beta=0.3)
    model.engine_backward(dlogits)
    pt.average_grads(model)
    optimizer.step()
    scheduler.step()
    return loss[0]


evaluate, get_dataloader = pt.evaluate, pt.get_dataloader        # pointwise_2data_trad.py:255-372 == pointwise_trad's


def main(argv=None):
    """Entry point: finetune/pointwise_2data_trad.py:374-535 -- two LETOR training sets (--train_path: MQ2008, 46 features;
    --train_path2: MSLR-WEB10K, 136 features), one step from each per iteration through the projection of its width."""
    return pt.main(argv, classifier=Classifier, step_fn=train_model, two_sets=True)


if __name__ == "__main__":
    main()
