"""LR2PPO stage 1 (pointwise relevance regression) on MI355X -- drop-in for the reference's finetune/pointwise.py.

Same public surface: `get_scores, log_sig, get_def_cls, MovieNet, Mlp, Classifier, load_or_initialize_parameters,
build_optimizer, train_model, evaluate, get_dataloader, main`.  `Classifier` is the Actor architecture (identical
state_dict keys, finetune/pointwise.py:189-236 == finetune/ppo.py:196-244), trained with SmoothL1(beta=0.3) on the tag
relevance labels, one optimizer + scheduler step per batch (pointwise.py:300-313).  The model forward / backward, the
loss and AdamW run on the gfx950 kernels through lr2ppo_amd.engine exactly as in stage 3; only this host loop is new.
mode 'reg' (every LR2PPO launcher) and 'cls'; there is no CPU fallback.
"""
from __future__ import annotations

import argparse
import json
import os
from copy import copy

import torch
import torch.distributed as dist
import torch.nn as nn
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from .. import h5lite, ops
from ..ndcg import AverageNDCGMeter, ndcg_rows
from ..tencentpretrain.model_saver import save_model
from ..tencentpretrain.opts import adv_opts, finetune_opts, tokenizer_opts
from ..tencentpretrain.utils.config import load_hyperparam
from ..tencentpretrain.utils.logging import init_logger
from ..tencentpretrain.utils.optimizers import str2optimizer, str2scheduler
from . import misc
from .ppo import FEAT, SEQ_LEN, Actor, Mlp, SyntheticMovieNet, _DataParallel, _grouped, _init_normal  # noqa: F401


def get_scores(score, mode, use_pair_wise=False):
    """finetune/pointwise.py:44-59 (helper of the upstream pairwise experiments; unused by train_model)."""
    if mode == "cls":
        score = nn.Softmax(dim=-1)(score)
        scores = score if use_pair_wise else nn.Softmax(dim=-1)(score)
    elif mode == "reg":
        scores = score if use_pair_wise else torch.log(nn.Softmax(dim=-1)(score) + 1e-10)
    else:
        raise ValueError(mode)
    return scores


def log_sig(chosen_score, reject_score):
    """finetune/pointwise.py:62-66."""
    return -torch.log(torch.sigmoid(chosen_score - reject_score) + 1e-10).mean()


def get_def_cls(tgts_lst):
    """finetune/pointwise.py:69-74."""
    rand_indices = torch.randperm(3)[:2]
    if tgts_lst[rand_indices[0]] < tgts_lst[rand_indices[1]]:
        return rand_indices.flip(dims=[-1])
    return rand_indices


def train_tag_index(targets, max_tags: int):
    """Tag selection of the training reader (finetune/pointwise.py:96-119): items are cut / padded to exactly
    `max_tags` tags; padding cycles through the tags with a non-zero label when there is one (label-aware
    augmentation), through all tags otherwise.  targets: the item's integer labels in file order."""
    n = len(targets)
    if n > max_tags:
        return [i % n for i in range(max_tags)]
    index = list(range(n))
    add = [i for i in range(n) if int(targets[i]) != 0]
    for i in range(n, max_tags):
        index.append(add[i % len(add)] if add else i % n)
    return index


class MovieNet(Dataset):
    """LRMovieNet reader of stage 1 (finetune/pointwise.py:77-167).  Reads LRMovieNet/clean_feat.h5 (h5py, or `lr2ppo_amd.h5lite` on libhdf5)."""

    def __init__(self, args, path, is_train=False):
        with open(path) as f:
            self.data = json.load(f)
        self.embed_data = h5lite.open_file(os.path.join("LRMovieNet", "clean_feat.h5"), "r")    # h5py, or libhdf5 via ctypes
        self.max_imgs, self.is_train, self.max_tags = args.max_imgs, is_train, args.max_tags
        self.items = []
        for item in self.data:
            tags = item["tags"]
            labels = [int(t["target"]) for t in tags]
            index = train_tag_index(labels, self.max_tags) if is_train else list(range(len(tags)))
            self.items.append((item["id"], index, [labels[i] for i in index]))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        item_id, tag_index, labels = self.items[i]
        grp = self.embed_data[f"{item_id}"]
        text = torch.tensor(grp["text_emb"][:])[torch.tensor(tag_index)]
        loaded = torch.tensor(grp["img_emb"][:][0])
        loaded = loaded[torch.randperm(loaded.shape[0])]
        n = loaded.shape[0]
        img = loaded[: self.max_imgs] if n > self.max_imgs else loaded[torch.arange(self.max_imgs) % n]
        return text, img, torch.tensor(labels)


class Classifier(Actor):
    """finetune/pointwise.py:189-236, mode 'reg': forward(text_emb, img_emb, tgts) -> (SmoothL1 loss, logits) or logits."""


def load_or_initialize_parameters(args, model):
    """finetune/pointwise.py:239-271.  Upstream loads the RoBERTa and (key-prefixed) ViT checkpoints with strict=False
    into a module that holds only the head, so no key matches and the head keeps torch's default nn.Linear / LayerNorm /
    Embedding initialisation; normal(0, 0.02) is used only when no --pretrained_model_path is given.  Reproduced as is.
    --head_model_path (addition of this build) loads a checkpoint of the head itself, strictly."""
    head = getattr(args, "head_model_path", None)
    if head is not None:
        model.load_state_dict(torch.load(head, map_location="cpu"), strict=True)
    elif getattr(args, "pretrained_model_path", None) is not None:
        model.load_state_dict(torch.load(args.pretrained_model_path, map_location="cpu"), strict=False)
        vit_path = getattr(args, "vit_pretrained_model_path", None)
        if vit_path is not None:
            vit = torch.load(vit_path, map_location="cpu")
            model.load_state_dict({f"vit_{k}": v for k, v in vit.items()}, strict=False)
    else:
        _init_normal(model)


def build_optimizer(args, model):
    """finetune/pointwise.py:274-297."""
    if args.optimizer not in str2optimizer:
        raise NotImplementedError(f"optimizer {args.optimizer!r}: only adamw is on the HIP path (every LR2PPO launcher uses it)")
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
    """One batch (finetune/pointwise.py:300-313): forward, SmoothL1, backward, AdamW step, scheduler step.
    Returns the loss as a 0-dim device tensor.  Gradients are averaged over ranks before the step (pointwise.py wraps
    nothing in DDP and trains independent replicas; the deviation is the same as in stage 3, see finetune/ppo.py)."""
    dev = text_emb_batch.device
    model.bind_grads()
    dp = _DataParallel()
    logits = model.engine_forward(text_emb_batch, img_emb_batch, save=True)
    loss, dlogits = torch.empty(1, device=dev), torch.empty_like(logits)
    if model.n_out > 1:      # mode 'cls' (pointwise.py:228-232): NLL of log-softmax over labels_num classes
        target = tgts_batch.to(device=dev, dtype=torch.int64).contiguous().view(-1)
        ops.nll_loss(logits, target, loss, dlogits, rows=logits.shape[0], C=model.n_out)
    else:
        target = tgts_batch.to(device=dev, dtype=torch.float32).contiguous().view(-1)
        ops.smooth_l1(logits.view(-1), target, loss, dlogits.view(-1), n=logits.numel(), beta=0.3)
    fuse = getattr(args, "fuse_fc1_update", True) and hasattr(optimizer, "external_update")
    fa = optimizer.external_update(model.out_layer.fc1.weight) if fuse else None
    model.engine_backward(dlogits, dp, fc1_update=fa)
    dp.finish(dp.reduce_start(model))
    optimizer.step()
    scheduler.step()
    return loss[0]


@torch.no_grad()
def evaluate(args, model, dataloader, step, split="test", num_tasks=None):
    """finetune/pointwise.py:316-412, mode 'reg': per validation item, NDCG@k of the gold labels re-ordered by predicted
    score; master returns (NDCG@all, 0)."""
    ndcg_obj = AverageNDCGMeter()
    model.eval()
    scores, golds = [], []
    for text_emb, img_emb, tgts in dataloader:
        logits = model.engine_forward(text_emb.to(args.device), img_emb.to(args.device), save=False)
        if model.n_out > 1:     # 'cls': 0 * z0 + 1 * z1 + 2 * z2 on the raw logits (pointwise.py:342-345)
            logits = ops.cls_scores(logits, None, torch.empty(logits.shape[0], device=logits.device), rows=logits.shape[0],
                                    C=logits.shape[1], softmax=False)
        scores.append(logits.view(-1))
        golds.append(tgts.view(-1))
    return report_ndcg(args, ndcg_obj, scores, golds, num_tasks)


def report_ndcg(args, ndcg_obj, scores, golds, num_tasks=None):
    """Tail of every pointwise `evaluate` (finetune/pointwise.py:352-412, pointwise_trad.py:296-340): NDCG@k per item -- one
    kernel + one copy for the whole split --, gathered over ranks in the reference's item order, averaged and logged by the
    master -> (NDCG@all, 0), (None, None) on the other ranks."""
    mine = ndcg_rows(scores, golds, args.device, tuple(ndcg_obj.ndcg_at_k))
    world = num_tasks or 1
    if world > 1 and dist.is_initialized():
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
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
        return vals[100000000], 0
    return None, None


def get_dataloader(args, dataset, num_tasks, global_rank, is_train=False):
    """finetune/pointwise.py:415-430: train batches of args.batch_size, validation one item at a time."""
    sampler = DistributedSampler(dataset, num_replicas=num_tasks, rank=global_rank, shuffle=is_train)
    workers = getattr(args, "num_workers", 32 if isinstance(dataset, MovieNet) else 2)        # synthetic sets: 2 workers
    return DataLoader(dataset=dataset, batch_size=args.batch_size if is_train else 1, sampler=sampler,
                      num_workers=workers, drop_last=False)


def build_parser():
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    finetune_opts(parser)
    tokenizer_opts(parser)
    parser.add_argument("--soft_targets", action="store_true", help="Train model with logits.")
    parser.add_argument("--soft_alpha", type=float, default=0.5, help="Weight of the soft targets loss.")
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
    # additions of this build (not in the reference)
    parser.add_argument("--head_model_path", type=str, default=None, help="optional checkpoint of the stage-1 head itself")
    parser.add_argument("--synthetic_items", type=int, default=0, help="use SyntheticMovieNet with this many train items")
    parser.add_argument("--synthetic_val_items", type=int, default=16)
    parser.add_argument("--max_steps", type=int, default=0, help="stop after this many training steps (0 = run all epochs)")
    # the composed model (encoder(embedding(src, seg), seg) -> head: tencentpretrain/models/model.py:32-41 upstream).  The reference's
    # launcher already carries --pretrained_model_path / --vit_pretrained_model_path for the RoBERTa and ViT checkpoints but loads them
    # into a module that holds only the head (pointwise.py:239-271); with --raw_inputs they load into the two encoder stacks that run
    # in front of it, and --finetune_encoders trains those stacks from the head's loss.
    from .features import raw_input_opts
    raw_input_opts(parser)
    parser.add_argument("--finetune_encoders", action="store_true",
                        help="with --raw_inputs: train both encoder stacks end to end (AdamW, the head's schedule) instead of freezing them")
    return parser


def SyntheticRawMovieNet(n_items, tags, max_imgs=16, seed=7):
    """Seeded raw stand-in for an LRMovieNet item (features.SyntheticRawItems): (frames, ids, seg, tgts)."""
    from .features import SyntheticRawItems
    return SyntheticRawItems(n_items, tags, max_imgs, seed)


def _run_raw(args, model, num_tasks, global_rank):
    """The training loop of main() on raw items: FeatureExtractor in front of the head, frozen (features extracted in line, then
    train_model) or fine-tuned (features.finetune_pointwise_step); validation extracts in line and calls evaluate()."""
    from .features import build_encoder_optimizer, build_extractor, finetune_pointwise_step
    fx = build_extractor(args, num_tasks, trainable=args.finetune_encoders)
    if args.synthetic_items <= 0:
        raise RuntimeError("--raw_inputs: the repository holds no raw LRMovieNet reader (the reference reads pre-extracted features, "
                           "finetune/pointwise.py:77-167); use --synthetic_items N")
    trainset = SyntheticRawMovieNet(args.synthetic_items, args.max_tags, args.max_imgs, args.seed)
    valset = SyntheticRawMovieNet(args.synthetic_val_items, 20, args.max_imgs, args.seed + 1)
    train_loader = get_dataloader(args, trainset, num_tasks, global_rank, is_train=True)
    val_loader = get_dataloader(args, valset, num_tasks, global_rank, is_train=False)
    args.train_steps = int(len(trainset) * args.epochs_num / args.batch_size) + 1
    optimizer, scheduler = build_optimizer(args, model)
    enc_opt, enc_sch = build_encoder_optimizer(args, fx) if args.finetune_encoders else (None, None)
    args.model = model

    class _Features:                       # validation loader of (text_emb, img_emb, tgts) from the raw one
        def __iter__(self_):
            for frames, ids, seg, tgts in val_loader:
                t, i = fx.extract(frames.to(args.device), ids.to(args.device), seg.to(args.device))
                yield t, i, tgts

    best, step, total_loss = 0.0, 0, 0.0
    for epoch in range(1, args.epochs_num + 1):
        train_loader.sampler.set_epoch(epoch)
        for i, (frames, ids, seg, tgts) in enumerate(train_loader):
            frames, ids, seg, tgts = (t.to(args.device) for t in (frames, ids, seg, tgts))
            model.train()
            if args.finetune_encoders:
                fx.train()
                loss = finetune_pointwise_step(args, fx, model, optimizer, scheduler, enc_opt, enc_sch, frames, ids, seg, tgts)
            else:
                text_emb, img_emb = fx.extract(frames, ids, seg)
                loss = train_model(args, model, optimizer, scheduler, text_emb, img_emb, tgts)
            if num_tasks > 1:
                dist.all_reduce(loss.div_(num_tasks))
            total_loss += loss.item()
            step += 1
            if (i + 1) % args.report_steps == 0 or (args.max_steps and step >= args.max_steps):
                if args.is_master:
                    args.logger.info("Epoch id: {}, Training steps: {}, Avg loss: {:.3f}".format(epoch, i + 1, total_loss / args.report_steps))
                total_loss = 0.0
                result, _ = evaluate(args, model, _Features(), step, split="val", num_tasks=num_tasks)
                if args.is_master and result.item() > best:
                    best = result.item()
                    save_model(model, args.output_model_path)
            if args.max_steps and step >= args.max_steps:
                return best
    return best


def main(argv=None):
    """finetune/pointwise.py:433-584."""
    parser = build_parser()
    args = parser.parse_args(argv)
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
    model = Classifier(args, argparse.Namespace(**vit_args_dict))
    load_or_initialize_parameters(args, model)
    if args.is_master:
        args.logger = init_logger(args)
    args.device = torch.device("cuda", torch.cuda.current_device())
    model = model.to(args.device)
    if num_tasks > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    if args.raw_inputs:
        return _run_raw(args, model, num_tasks, global_rank)
    if args.synthetic_items > 0:
        trainset = SyntheticMovieNet(args.synthetic_items, args.max_tags, args.max_imgs, args.seed)
        valset = SyntheticMovieNet(args.synthetic_val_items, 20, args.max_imgs, args.seed + 1)
    else:
        trainset, valset = MovieNet(args, args.train_path, is_train=True), MovieNet(args, args.dev_path, is_train=False)
    train_loader = get_dataloader(args, trainset, num_tasks, global_rank, is_train=True)
    val_loader = get_dataloader(args, valset, num_tasks, global_rank, is_train=False)
    return run_training(args, model, [train_loader], val_loader, len(trainset), num_tasks)


def run_training(args, model, train_loaders, val_loader, instances_num, num_tasks, *, build_optimizer=build_optimizer,
                 train_model=train_model, evaluate=evaluate, batch_map=None):
    """The epoch loop of finetune/pointwise.py:515-584 -- and of pointwise_trad.py:479-538 / pointwise_2data_trad.py:470-535, which
    repeat it over LETOR queries: optimizer + schedule from `instances_num`, one `train_model` step per batch (per batch of EACH of
    the zipped loaders, in turn, for the two-data-set twin), loss averaged over ranks for the log, validation + best-NDCG checkpoint
    every `report_steps` iterations.  batch_map: loader batch -> (text_emb, img_emb | None, tgts)."""
    args.train_steps = int(instances_num * args.epochs_num / args.batch_size) + 1
    if args.is_master:
        args.logger.info("Batch size: {}".format(args.batch_size))
        args.logger.info("The number of training instances: {}".format(instances_num))
    optimizer, scheduler = build_optimizer(args, model)
    args.model = model
    total_loss, best_result, step = 0.0, 0.0, 0
    if args.is_master:
        args.logger.info("Start training.")
    for epoch in range(1, args.epochs_num + 1):
        for loader in train_loaders:
            loader.sampler.set_epoch(epoch)
        model.train()
        for i, batches in enumerate(zip(*train_loaders)):
            for j, batch in enumerate(batches):
                text_emb, img_emb, tgts = batch if batch_map is None else batch_map(batch)
                loss = train_model(args, model, optimizer, scheduler, text_emb.to(args.device),
                                   img_emb.to(args.device) if img_emb is not None else None, tgts.to(args.device))
                if num_tasks > 1:
                    dist.all_reduce(loss.div_(num_tasks))
                total_loss += loss.item()
                step += 1
                last = j == len(batches) - 1
                stop = bool(getattr(args, "max_steps", 0)) and step >= args.max_steps
                if ((i + 1) % args.report_steps == 0 and last) or stop:
                    if args.is_master:
                        args.logger.info("Epoch id: {}, Training steps: {}, Avg loss: {:.3f}".format(
                            epoch, i + 1, total_loss / args.report_steps))
                        args.logger.info("Val set evaluation.")
                    total_loss = 0.0
                    result, _ = evaluate(args, model, val_loader, step, split="val", num_tasks=num_tasks)
                    if args.is_master:
                        if result.item() > best_result:
                            best_result = result.item()
                            save_model(model, args.output_model_path)
                            args.logger.info("Best NDCG until now!\n")
                        args.logger.info("Best NDCG: {}".format(best_result))
                    model.train()
                if stop:
                    return best_result
    return best_result


if __name__ == "__main__":
    main()
