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
