"""LR2PPO stage 2 (pairwise reward model) on MI355X -- drop-in for the reference's finetune/reward_pair_dataloader.py.

Same public surface: `log_sig, get_def_cls, get_index, MovieNet, Mlp, Classifier, load_or_initialize_parameters,
build_optimizer, train_model, evaluate, get_dataloader, main`.  `Classifier` is the Critic/Reward architecture with
pos_emb(arange(4)) hard-coded (identical state_dict keys, reward_pair_dataloader.py:233-283 == finetune/ppo.py:300-350);
its checkpoint is what stage 3 loads as `--reward_model_path`.  One training step (:347-365) scores every item under two
4-long tag orderings -- `chosen_index` and `reject_index`, equal in their first two entries (the shown order) and
swapped in the last two (the candidate next order) -- and minimises relu(1 - (chosen - reject)).

The two forwards are issued as ONE forward over the batch [chosen ; reject] (items are independent of each other), so the
trunk GEMMs run at M = 2*bs*4*196 token rows; model forward / backward, the hinge and AdamW run on the gfx950 kernels.
There is no CPU fallback.
"""
from __future__ import annotations

import argparse
import json
import os
import random
from copy import copy

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

from .. import h5lite, ops
from ..tencentpretrain.model_saver import save_model
from ..tencentpretrain.opts import adv_opts, finetune_opts, tokenizer_opts
from ..tencentpretrain.utils.config import load_hyperparam
from ..tencentpretrain.utils.logging import init_logger
from ..tencentpretrain.utils.optimizers import str2optimizer, str2scheduler
from . import misc
from .pointwise import get_def_cls, log_sig  # noqa: F401  (identical helpers upstream, :60-74)
from .ppo import FEAT, SEQ_LEN, Mlp, Reward, _DataParallel, _grouped, _init_normal  # noqa: F401

TRAIN_LAYOUTS = (([0, 1, 0, 1], [0, 1, 1, 0]), ([1, 0, 0, 1], [1, 0, 1, 0]))   # (chosen, reject), :126-139


def get_index(tag_list):
    """finetune/reward_pair_dataloader.py:77-84: shuffle the item's tags, keep two; chosen repeats the shown order when the
    first is at least as relevant, reject swaps it (and vice versa)."""
    index = list(range(len(tag_list)))
    random.shuffle(index)
    index = index[:2]
    if tag_list[index[0]]["target"] >= tag_list[index[1]]["target"]:
        return index + index, index + [index[1], index[0]]
    return index + [index[1], index[0]], index + index


class MovieNet(Dataset):
    """LRMovieNet reader of stage 2 (reward_pair_dataloader.py:87-211).  Training items are the ranked tag pairs stored
    under item["index"], shown in either order with probability 1/2; validation draws max_tags=100 label-stratified
    triples per item and orders two of them with get_index.  Reads LRMovieNet/clean_feat.h5 (h5py, or `lr2ppo_amd.h5lite` on libhdf5)."""

    def __init__(self, args, path, is_train=False):
        with open(path) as f:
            self.data = json.load(f)
        self.embed_data = h5lite.open_file(os.path.join("LRMovieNet", "clean_feat.h5"), "r")    # h5py, or libhdf5 via ctypes
        self.max_imgs, self.is_train = args.max_imgs, is_train
        self.max_tags = args.max_tags if is_train else 100
        self.items = []       # (item id, tag index, labels, chosen, reject)
        for item in self.data:
            tags = item["tags"]
            if is_train:
                for pair in item["index"]:
                    chosen, reject = TRAIN_LAYOUTS[0] if np.random.random() < 0.5 else TRAIN_LAYOUTS[1]
                    self.items.append((item["id"], list(pair), [int(tags[i]["target"]) for i in pair], chosen, reject))
            else:
                by_label = {c: [i for i, t in enumerate(tags) if int(t["target"]) == c] for c in range(3)}
                if min(len(v) for v in by_label.values()) == 0:
                    continue
                triples = [[by_label[c][random.randint(0, len(by_label[c]) - 1)] for c in range(3)]
                           for _ in range(self.max_tags)]        # upstream draws all triples first, then orders each
                for triple in triples:
                    chosen, reject = get_index([tags[i] for i in triple])
                    self.items.append((item["id"], triple, [int(tags[i]["target"]) for i in triple], chosen, reject))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        item_id, tag_index, labels, chosen, reject = self.items[i]
        grp = self.embed_data[f"{item_id}"]
        text = torch.tensor(grp["text_emb"][:])[torch.tensor(tag_index)]
        loaded = torch.tensor(grp["img_emb"][:][0])
        loaded = loaded[torch.randperm(loaded.shape[0])]
        n = loaded.shape[0]
        img = loaded[: self.max_imgs] if n > self.max_imgs else loaded[torch.arange(self.max_imgs) % n]
        return text, img, torch.tensor(labels), torch.tensor(chosen), torch.tensor(reject)


class SyntheticPairs(Dataset):
    """Seeded stand-in with the reader's shapes: train items carry 2 tags and one of the two training layouts, validation
    items 3 tags ordered by get_index's rule."""

    def __init__(self, n_items, is_train, max_imgs=16, seed=7):
        self.n, self.is_train, self.max_imgs, self.seed = n_items, is_train, max_imgs, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        tags = 2 if self.is_train else 3
        text = torch.randn(tags, SEQ_LEN, FEAT, generator=g)
        img = torch.randn(self.max_imgs, FEAT, generator=g)
        labels = torch.randint(0, 3, (tags,), generator=g)
        if self.is_train:
            chosen, reject = TRAIN_LAYOUTS[int(torch.randint(0, 2, (1,), generator=g))]
        else:
            order = torch.randperm(3, generator=g)[:2].tolist()
            keep, swap = order + order, order + order[::-1]
            chosen, reject = (keep, swap) if labels[order[0]] >= labels[order[1]] else (swap, keep)
        return text, img, labels, torch.tensor(chosen), torch.tensor(reject)


def _synthetic_layout(is_train):
    """chosen / reject index of a synthetic item with the given labels (SyntheticPairs' rule), for features.SyntheticRawItems."""
    def extra(i, labels, g):
        if is_train:
            chosen, reject = TRAIN_LAYOUTS[int(torch.randint(0, 2, (1,), generator=g))]
        else:
            order = torch.randperm(3, generator=g)[:2].tolist()
            keep, swap = order + order, order + order[::-1]
            chosen, reject = (keep, swap) if labels[order[0]] >= labels[order[1]] else (swap, keep)
        return torch.tensor(chosen), torch.tensor(reject)
    return extra


class Classifier(Reward):
    """reward_pair_dataloader.py:233-283: forward(text_emb, img_emb, tgts, index[bs,4]) -> score[bs] (last position).
    `args.mode` is stored and never read by the upstream forward (its launcher passes --mode cls): any value is accepted."""

    def __init__(self, args, vit_args=None):
        a = copy(args)
        a.mode = "reg"
        super().__init__(a, vit_args)
        self.mode = args.mode


def load_or_initialize_parameters(args, model):
    """finetune/reward_pair_dataloader.py:286-318.  Upstream loads the RoBERTa and (key-prefixed) ViT checkpoints with strict=False
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
    """reward_pair_dataloader.py:321-344."""
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


def _pair_batch(text_emb, img_emb, chosen_index, reject_index):
    if chosen_index.shape != reject_index.shape or chosen_index.shape[1] != 4:
        raise ValueError("chosen_index / reject_index must both be [bs, 4] (pos_emb has 4 rows, reward_pair_dataloader.py:269)")
    return (torch.cat([text_emb, text_emb]), torch.cat([img_emb, img_emb]) if img_emb is not None else None,
            torch.cat([chosen_index, reject_index]).to(torch.int64))


def train_model(args, model, optimizer, scheduler, text_emb_batch, img_emb_batch, tgts_batch, chosen_index_batch,
                reject_index_batch, margin: float = 1.0):
    """One batch (reward_pair_dataloader.py:347-365) -> (loss, acc) as 0-dim device tensors.  Gradients are averaged
    over ranks before the step (upstream trains independent replicas; same deviation as stage 3).
    margin: 1 upstream here; finetune/reward_trad.py:270 uses 0.01 (lr2ppo_amd.finetune.reward_trad passes it)."""
    dev = text_emb_batch.device
    bs = text_emb_batch.shape[0]
    model.bind_grads()
    dp = _DataParallel()
    text2, img2, index2 = _pair_batch(text_emb_batch, img_emb_batch, chosen_index_batch, reject_index_batch)
    scores = model.engine_forward(text2, img2, index2, save=True).view(-1)
    loss_acc, dscores = torch.empty(2, device=dev), torch.empty_like(scores)
    ops.pair_hinge(scores, loss_acc, dscores, bs=bs, margin=margin)
    fuse = getattr(args, "fuse_fc1_update", True) and hasattr(optimizer, "external_update") and not model.TRAD
    fa = optimizer.external_update(model.out_layer.fc1.weight) if fuse else None
    model.engine_backward(dscores, dp, fc1_update=fa)
    dp.finish(dp.reduce_start(model))
    optimizer.step()
    scheduler.step()
    return loss_acc[0], loss_acc[1]


@torch.no_grad()
def evaluate(args, model, dataloader, step, split="test", num_tasks=None):
    """reward_pair_dataloader.py:367-415: fraction of validation pairs with chosen > reject, summed over ranks."""
    model.eval()
    dev = args.device
    counts = torch.zeros(2, device=dev, dtype=torch.float64)       # correct, samples
    for text_emb, img_emb, tgts, chosen_index, reject_index in dataloader:
        bs = text_emb.shape[0]
        text2, img2, index2 = _pair_batch(text_emb.to(dev), img_emb.to(dev) if img_emb is not None else None, chosen_index.to(dev),
                                          reject_index.to(dev))
        scores = model.engine_forward(text2, img2, index2, save=False).view(-1)
        counts[0] += (scores[:bs] > scores[bs:]).sum()
        counts[1] += bs
    if (num_tasks or 1) > 1 and dist.is_initialized():
        dist.all_reduce(counts)
    if getattr(args, "is_master", True):
        correct, total = counts.tolist()
        accuracy = correct / total if total > 0 else 0
        if hasattr(args, "logger"):
            args.logger.info(f"{split} accuracy: {accuracy:.4f}")
        return accuracy
    return None


def get_dataloader(args, dataset, num_tasks, global_rank, is_train=False):
    """reward_pair_dataloader.py:418-434: args.batch_size for both splits."""
    sampler = DistributedSampler(dataset, num_replicas=num_tasks, rank=global_rank, shuffle=is_train)
    workers = getattr(args, "num_workers", 32 if isinstance(dataset, MovieNet) else 2)
    return DataLoader(dataset=dataset, batch_size=args.batch_size, sampler=sampler, num_workers=workers, drop_last=False)


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
    parser.add_argument("--head_model_path", type=str, default=None, help="optional checkpoint of the stage-2 head itself")
    parser.add_argument("--synthetic_items", type=int, default=0, help="use SyntheticPairs with this many train items")
    parser.add_argument("--synthetic_val_items", type=int, default=16)
    parser.add_argument("--max_steps", type=int, default=0, help="stop after this many training steps (0 = run all epochs)")
    from .features import raw_input_opts
    raw_input_opts(parser)      # --raw_inputs [--image_tower vit_large_14_224] [--fp8_features]: frozen encoder stacks in line
    return parser


def main(argv=None):
    """reward_pair_dataloader.py:437-592."""
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
    fx = None
    if args.raw_inputs:
        # BASELINE.json configs[4]'s first half: reward-pair training behind frozen encoder stacks (ViT-L/14 swap, MX-FP8 products)
        from .features import ExtractingLoader, SyntheticRawItems, build_extractor
        if args.synthetic_items <= 0:
            raise RuntimeError("--raw_inputs: the repository holds no raw LRMovieNet reader (the reference reads pre-extracted "
                               "features, finetune/reward_pair_dataloader.py:87-211); use --synthetic_items N")
        fx = build_extractor(args, num_tasks)
        trainset = SyntheticRawItems(args.synthetic_items, 2, args.max_imgs, args.seed, extra=_synthetic_layout(True))
        valset = SyntheticRawItems(args.synthetic_val_items, 3, args.max_imgs, args.seed + 1, extra=_synthetic_layout(False))
    elif args.synthetic_items > 0:
        trainset = SyntheticPairs(args.synthetic_items, True, args.max_imgs, args.seed)
        valset = SyntheticPairs(args.synthetic_val_items, False, args.max_imgs, args.seed + 1)
    else:
        trainset, valset = MovieNet(args, args.train_path, is_train=True), MovieNet(args, args.dev_path, is_train=False)
    train_loader = get_dataloader(args, trainset, num_tasks, global_rank, is_train=True)
    val_loader = get_dataloader(args, valset, num_tasks, global_rank, is_train=False)
    if fx is not None:
        train_loader, val_loader = ExtractingLoader(train_loader, fx, args.device), ExtractingLoader(val_loader, fx, args.device)
    return run_training(args, model, train_loader, val_loader, len(trainset), num_tasks)


def run_training(args, model, train_loader, val_loader, instances_num, num_tasks, *, train_model=train_model, evaluate=evaluate,
                 batch_map=None):
    """The epoch loop of reward_pair_dataloader.py:531-592 -- and of reward_trad.py:440-503, which repeats it over LETOR pairs:
    one `train_model` step per batch, loss / accuracy averaged over ranks for the log, validation + best-accuracy checkpoint every
    `report_steps` batches.  batch_map: loader batch -> (text_emb, img_emb | None, tgts, chosen_index, reject_index)."""
    args.train_steps = int(instances_num * args.epochs_num / args.batch_size) + 1
    if args.is_master:
        args.logger.info("Batch size: {}".format(args.batch_size))
        args.logger.info("The number of training instances: {}".format(instances_num))
    optimizer, scheduler = build_optimizer(args, model)
    args.model = model
    total_loss, total_acc, total_cnt, best_acc, step = 0.0, 0.0, 0, 0.0, 0
    if args.is_master:
        args.logger.info("Start training.")
    for epoch in range(1, args.epochs_num + 1):
        train_loader.sampler.set_epoch(epoch)
        model.train()
        for i, batch in enumerate(train_loader):
            text_emb, img_emb, tgts, chosen_index, reject_index = batch if batch_map is None else batch_map(batch)
            loss, acc = train_model(args, model, optimizer, scheduler, text_emb.to(args.device),
                                    img_emb.to(args.device) if img_emb is not None else None,
                                    tgts.to(args.device), chosen_index.to(args.device), reject_index.to(args.device))
            if num_tasks > 1:
                dist.all_reduce(loss.div_(num_tasks))
                dist.all_reduce(acc.div_(num_tasks))
            total_loss += loss.item()
            total_acc += acc.item()
            total_cnt += 1
            step += 1
            stop = bool(getattr(args, "max_steps", 0)) and step >= args.max_steps
            if (i + 1) % args.report_steps == 0 or stop:
                if args.is_master:
                    args.logger.info("Epoch id: {}, Training steps: {}, Avg loss: {:.3f}, Acc: {:.3f}".format(
                        epoch, i + 1, total_loss / total_cnt, total_acc / total_cnt))
                    args.logger.info("Val set evaluation.")
                total_loss, total_acc, total_cnt = 0.0, 0.0, 0
                val_acc = evaluate(args, model, val_loader, step, split="val", num_tasks=num_tasks)
                if args.is_master:
                    if val_acc > best_acc:
                        best_acc = val_acc
                        save_model(model, args.output_model_path)
                        args.logger.info("Best Acc until now!\n")
                    args.logger.info("Best Acc: {}".format(best_acc))
                model.train()
            if stop:
                return best_acc
    return best_acc


if __name__ == "__main__":
    main()
