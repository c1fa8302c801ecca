"""Stage 2 at sequence length 1 -- drop-in for the model and the training step of the reference's finetune/reward_trad.py
(the pairwise reward model of the MSLR-WEB10K -> MQ2008 transfer; twin of finetune/reward_pair_dataloader.py).

`Classifier` (reward_trad.py:156-201) is ppo_trad's Reward architecture: pos_emb(arange(4)), XiT over the document feature
with itself, out_layer = Mlp(1536, 3072, 768), the second XiT over the 4 indexed documents, head at the last position.
`train_model` (:263-281): chosen / reject scores, hinge relu(0.01 - (s+ - s-)) (margin 0.01, not stage 2's 1), AdamW step,
scheduler step -- `reward_pair_dataloader.train_model` with that margin and `img_emb = None` ([chosen ; reject] as one batch,
`lr2_pair_hinge`).  `evaluate` (:283-322): fraction of validation pairs with chosen > reject.  `LTRDataset` (:87-134) draws the
label-stratified pairs from the LETOR h5 files (`finetune/letor.py`); `SyntheticTradPairs` provides its item layout.  No CPU fallback.
"""
from __future__ import annotations

import torch
from torch.utils.data import Dataset

from . import ppo_trad
from .letor import RewardPairs
from . import reward_pair_dataloader as rp
from .ppo import FEAT
from .reward_pair_dataloader import build_optimizer  # noqa: F401  (reward_trad.py:238-260 == reward_pair_dataloader.py:321-344)


class Classifier(ppo_trad.Reward):
    """reward_trad.py:156-201: forward(text_emb [bs, docs, 768], img_emb (ignored), tgts, index [bs, 4]) -> score [bs]."""


load_or_initialize_parameters = ppo_trad.load_or_initialize_parameters


def train_model(args, model, optimizer, scheduler, text_emb_batch, img_emb_batch, tgts_batch, chosen_index_batch,
                reject_index_batch):
    """reward_trad.py:263-281 -> (loss, acc) as 0-dim device tensors."""
    return rp.train_model(args, model, optimizer, scheduler, text_emb_batch, None, tgts_batch, chosen_index_batch,
                          reject_index_batch, margin=0.01)


@torch.no_grad()
def evaluate(args, model, dataloader, step=0, split="test", num_tasks=None):
    """reward_trad.py:283-322; `dataloader` yields (ground_truths, query_id, features, chosen_index, reject_index)."""
    def as_stage2():
        for ground_truths, _, features, chosen_index, reject_index in dataloader:
            yield features.to(torch.float32), None, ground_truths, chosen_index, reject_index
    return rp.evaluate(args, model, ppo_trad._Loader(as_stage2), step, split=split, num_tasks=num_tasks)


class LTRDataset(RewardPairs):
    """reward_trad.py:87-134: LTRDataset(args, path, is_train, max_tags=20)."""


class SyntheticTradPairs(Dataset):
    """Items of reward_trad.py's LTRDataset layout: (ground_truths [docs], query_id, features [docs, 768], chosen [4], reject [4])."""

    def __init__(self, n: int, docs: int = 20, seed: int = 11):
        self.n, self.docs, self.seed = n, docs, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        gt = torch.randint(0, 3, (self.docs,), generator=g)
        feats = torch.randn(self.docs, FEAT, generator=g)
        chosen = torch.randint(0, self.docs, (4,), generator=g)
        reject = torch.randint(0, self.docs, (4,), generator=g)
        return gt, i, feats, chosen, reject


def letor_batch(batch):
    """LTRDataset batch -> (text_emb f32, None, tgts, chosen_index, reject_index) (reward_trad.py:458-463)."""
    ground_truths, _, features, chosen_index, reject_index = batch
    return features.to(torch.float32), None, ground_truths, chosen_index, reject_index


def main(argv=None):
    """Entry point: finetune/reward_trad.py:352-503 -- the pairwise reward model on LETOR queries.  --train_path / --dev_path
    name directories holding train.h5 / test.h5 (LTRDataset draws max_tags = 20 label-stratified pairs per query for either
    split); --synthetic_items N runs on seeded pairs instead.
        python -m lr2ppo_amd.finetune.reward_trad --train_path DATA --dev_path DATA --batch_size 8 --epochs_num 1 --report_steps 10 ..."""
    import argparse
    from copy import copy
    from . import misc
    from ..tencentpretrain.utils.config import load_hyperparam
    from ..tencentpretrain.utils.logging import init_logger
    args = rp.build_parser().parse_args(argv)
    vit_args_dict = copy(vars(args))
    for k, v in vars(args).items():
        if "vit_" in k:
            vit_args_dict[k[4:]] = v
    args = load_hyperparam(args)
    args.labels_num = 3
    args.fuse_fc1_update = False                 # no 2-GB out_layer.fc1 at sequence length 1: every gradient through lr2_adamw_multi
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
        import torch.distributed as dist
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    if args.synthetic_items > 0:
        trainset, valset = SyntheticTradPairs(args.synthetic_items, 20, args.seed), SyntheticTradPairs(args.synthetic_val_items, 20, args.seed + 1)
    else:
        trainset, valset = LTRDataset(args, args.train_path, is_train=True), LTRDataset(args, args.dev_path, is_train=False)
    train_loader = rp.get_dataloader(args, trainset, num_tasks, global_rank, is_train=True)
    val_loader = rp.get_dataloader(args, valset, num_tasks, global_rank, is_train=False)
    return rp.run_training(args, model, train_loader, val_loader, len(trainset), num_tasks, train_model=train_model,
                           evaluate=evaluate, batch_map=letor_batch)


if __name__ == "__main__":
    main()
