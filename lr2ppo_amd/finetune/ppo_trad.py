"""Stage 3 at sequence length 1 -- drop-in for the models and the PPO step of the reference's finetune/ppo_trad.py (the
MSLR-WEB10K -> MQ2008 transfer twin of finetune/ppo.py; SURVEY.md 8(f)-4's smoke configuration).

`Actor`, `Critic`, `Reward` (ppo_trad.py:142-281) are the LR2PPO heads without text_proj / img_proj: a pre-projected 768-d
feature per document serves as both streams of the XiT block and is concatenated behind its output, out_layer =
Mlp(1536, 3072, 768); Critic / Reward add pos_emb, the second XiT over the documents of a query ('causal' = the reference's
no-op mask) and read the head at the last position.  The rollout / update / loss / optimizer code is finetune/ppo.py's own
(`rollout_step`, `update_minibatch`, `train_model`, `build_optimizer`: ppo_trad.py:309-345,431-560,760-829 repeat ppo.py line
for line with `img_emb = None`), on the same HIP kernels: `engine.trad_trunk_forward / backward`, `engine.xit_forward /
backward`, `lr2_ppo_loss`, `lr2_adamw_multi`.  `LTRDataset` (ppo_trad.py:63-98) reads the LETOR `train.h5` / `test.h5`
files (`finetune/letor.py`: h5py, or libhdf5 through `lr2ppo_amd.h5lite`); `SyntheticLTR` provides queries of its shape.  No CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ppo
from torch.utils.data import Dataset

from .letor import QueryPairs
from .ppo import RankLoss, clipped_value_loss, build_optimizer, rollout_step, update_minibatch   # noqa: F401


class Actor(ppo.Actor):
    """ppo_trad.py:142-189.  forward(text_emb [bs, tags, 768], img_emb (ignored), tgts) -> (loss, logits) or logits."""
    TRAD = True


class Critic(ppo.Critic):
    """ppo_trad.py:192-236.  forward(text_emb, img_emb (ignored), tgts, index [bs, t <= 4]) -> value [bs]."""
    TRAD = True


class Reward(ppo.Reward):
    """ppo_trad.py:239-281: as Critic with pos_emb(arange(4)) hard-coded (index must have 4 columns)."""
    TRAD = True


class ActorCritic(nn.Module):
    """ppo_trad.py:113-139."""

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


load_or_initialize_parameters = ppo.load_or_initialize_parameters
load_or_initialize_parameters_reward = ppo.load_or_initialize_parameters_reward


def train_model(args, model, optimizer, critic_optim, scheduler, critic_scheduler, memories, epoch):
    """ppo_trad.py:431-560: one PPO update cycle over the stored rollouts (records carry img_emb = None)."""
    return ppo.train_model(args, model, optimizer, critic_optim, scheduler, critic_scheduler, memories, epoch)


@torch.no_grad()
def evaluate(args, val_loader, step=0, split="test", num_tasks=None):
    """ppo_trad.py:563-640: NDCG of the gold labels re-ordered by the actor's scores, one query per item.  `val_loader`
    yields (ground_truths [1, docs], query_id, features [1, docs, 768]) like LTRDataset / SyntheticLTR."""
    def as_ppo_batches():
        for ground_truths, _, features in val_loader:
            yield features.to(torch.float32), None, ground_truths
    return ppo.evaluate(args, _Loader(as_ppo_batches), step, split=split, num_tasks=num_tasks)


class LTRDataset(QueryPairs):
    """ppo_trad.py:63-98: LTRDataset(args, path, is_train, max_tags=20) -- training items are random ordered document pairs
    (max_tags per query), validation items whole queries."""


class SyntheticLTR(Dataset):
    """Queries of LTRDataset's item layout (ppo_trad.py:89-94): (ground_truths [docs], query_id, features [docs, 768]); every
    query resampled to exactly `docs` documents like datasets_trad/convert_to_h5py.py:17-23.  pairs=True: training items, a
    random ordered pair of the query's documents (ppo_trad.py:77-82)."""

    def __init__(self, n_queries: int, docs: int = 20, seed: int = 7, pairs: bool = False):
        self.n, self.docs, self.seed, self.pairs = n_queries, docs, seed, pairs

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        gt, feats = torch.randint(0, 3, (self.docs,), generator=g), torch.randn(self.docs, ppo.FEAT, generator=g)
        if self.pairs:
            pair = torch.randperm(self.docs, generator=g)[:2]
            gt, feats = gt[pair], feats[pair]
        return gt, i, feats


def main(argv=None):
    """Entry point: finetune/ppo_trad.py's loop (:700-849 = finetune/ppo.py's) over the LETOR files under --train_path /
    --dev_path (each a directory holding train.h5 / test.h5, ppo_trad.py:712-713,746) or over SyntheticLTR queries:
        python -m lr2ppo_amd.finetune.ppo_trad --synthetic_items 64 --batch_size 8 --update_timesteps 4 --max_cycles 2 ..."""
    import argparse
    from . import misc
    parser = ppo.build_parser()
    args = parser.parse_args(argv)
    args.labels_num = 3
    args.fuse_fc1_update = False
    misc.init_distributed_mode(args)
    misc.setup_seed(args.seed + misc.get_rank())
    args.is_master = misc.is_main_process()

    def make_sets():
        if args.synthetic_items > 0:
            return (SyntheticLTR(args.synthetic_items, 20, args.seed, pairs=True), SyntheticLTR(args.synthetic_val_items, 20, args.seed + 1))
        return LTRDataset(args, args.train_path, is_train=True), LTRDataset(args, args.dev_path, is_train=False)

    def batch_map(b):
        ground_truths, _, features = b
        return features.to(torch.float32), None, ground_truths

    return ppo.run_training(args, argparse.Namespace(**vars(args)), ActorCritic, Reward, make_sets, batch_map)


class _Loader:
    def __init__(self, make):
        self.make = make

    def __iter__(self):
        return self.make()


if __name__ == "__main__":
    main()
