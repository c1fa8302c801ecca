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
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

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
    average_grads(model)
    optimizer.step()
    scheduler.step()
    return loss[0]


def average_grads(model):
    """Mean of the bound gradients over the ranks before the step -- what DistributedDataParallel does for the reference
    (pointwise_trad.py:446-448, the one DDP use under finetune/); a no-op on one rank.  The twins' models are small (no 2-GB
    matrix): one all-reduce per gradient buffer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    works = [dist.all_reduce(p.grad.div_(world), async_op=True) for p in model.parameters() if p.grad is not None]
    for w in works:
        w.wait()


@torch.no_grad()
def evaluate(args, model, dataloader, step=0, split="test", num_tasks=None):
    """pointwise_trad.py:257-340, mode 'reg': `dataloader` yields (ground_truths [1, docs], query_id, features [1, docs, F]) like
    LTRDataset; per query the NDCG@k of the labels re-ordered by predicted score -> (NDCG@all, 0) on the master."""
    from .pointwise import report_ndcg
    from ..ndcg import AverageNDCGMeter
    model.eval()
    scores, golds = [], []
    for ground_truths, _, features in dataloader:
        logits = model.engine_forward(features.to(device=args.device, dtype=torch.float32), save=False)
        scores.append(logits.view(-1))
        golds.append(ground_truths.view(-1).to(torch.int64))
    return report_ndcg(args, AverageNDCGMeter(), scores, golds, num_tasks)


def get_dataloader(args, dataset, num_tasks, global_rank, is_train=False):
    """pointwise_trad.py:342-357: shuffled batches of args.batch_size without the ragged last one, validation one query at a time."""
    sampler = DistributedSampler(dataset, num_replicas=num_tasks, rank=global_rank, shuffle=is_train)
    return DataLoader(dataset=dataset, batch_size=args.batch_size if is_train else 1, sampler=sampler,
                      num_workers=getattr(args, "num_workers", 2), drop_last=is_train)


def letor_batch(batch):
    """LTRDataset batch (ground_truths, query ids, features f64) -> (text_emb f32, None, tgts) (pointwise_trad.py:489-491)."""
    ground_truths, _, features = batch
    return features.to(torch.float32), None, ground_truths


def build_parser():
    """finetune/pointwise_trad.py:376-404 = finetune/pointwise.py's flags (+ this build's --synthetic_items / --max_steps)."""
    from .pointwise import build_parser as stage1_parser
    parser = stage1_parser()
    parser.add_argument("--train_path2", type=str, required=False, help="second training set (pointwise_2data_trad.py:401)")
    return parser


def main(argv=None, classifier=None, step_fn=None, two_sets=False):
    """Entry point: finetune/pointwise_trad.py:376-538 -- BASELINE configs[0].  --train_path / --dev_path name directories that hold
    train.h5 / test.h5 (LTRDataset); --synthetic_items N runs on seeded queries instead.
        python -m lr2ppo_amd.finetune.pointwise_trad --train_path DATA --dev_path DATA --batch_size 8 --epochs_num 1 --report_steps 10 ..."""
    import argparse
    from copy import copy
    from . import misc, pointwise as pw
    from ..tencentpretrain.utils.config import load_hyperparam
    from ..tencentpretrain.utils.logging import init_logger
    args = build_parser().parse_args(argv)
    vit_args_dict = copy(vars(args))
    for k, v in vars(args).items():
        if "vit_" in k:
            vit_args_dict[k[4:]] = v
    args = load_hyperparam(args)
    args.labels_num = 3
    misc.init_distributed_mode(args)
    misc.setup_seed(args.seed + misc.get_rank())
    args.is_master = misc.is_main_process()
    num_tasks, global_rank = misc.get_world_size(), misc.get_rank()
    model = (classifier or Classifier)(args, argparse.Namespace(**vit_args_dict))
    load_or_initialize_parameters(args, model)
    if args.is_master:
        args.logger = init_logger(args)
    args.device = torch.device("cuda", torch.cuda.current_device())
    model = model.to(args.device)
    if num_tasks > 1:          # DDP broadcasts rank 0's parameters when it wraps the model (pointwise_trad.py:448)
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    if args.synthetic_items > 0:
        widths = (46, 136) if two_sets else (FEAT,)          # the two-data-set twin projects raw LETOR rows (46 | 136 features)
        trainsets = [_SyntheticRows(args.synthetic_items, 20, args.seed + 7 * k, w) for k, w in enumerate(widths)]
        valset = _SyntheticRows(args.synthetic_val_items, 20, args.seed + 1, widths[0])
    else:
        paths = [args.train_path, args.train_path2] if two_sets else [args.train_path]
        trainsets = [LTRDataset(args, path, is_train=True) for path in paths]
        valset = LTRDataset(args, args.dev_path, is_train=False)
    loaders = [get_dataloader(args, ts, num_tasks, global_rank, is_train=True) for ts in trainsets]
    val_loader = get_dataloader(args, valset, num_tasks, global_rank, is_train=False)
    return pw.run_training(args, model, loaders, val_loader, len(trainsets[0]), num_tasks, build_optimizer=build_optimizer,
                           train_model=step_fn or train_model, evaluate=evaluate, batch_map=letor_batch)


class _SyntheticRows(Dataset):
    """SyntheticLTR in LTRDataset's item order: (ground_truths [docs], query id, features [docs, 768])."""

    def __init__(self, n_queries, docs=20, seed=7, width=FEAT):
        self.inner, self.width = SyntheticLTR(n_queries, docs, seed), width

    def __len__(self):
        return len(self.inner)

    def __getitem__(self, i):
        feats, _, gt = self.inner[i]
        return gt, str(i), feats[:, :self.width].contiguous()


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


if __name__ == "__main__":
    main()
