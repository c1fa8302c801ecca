"""Evaluation-only entry point of the sequence-length-1 stage 3 -- drop-in for the reference's finetune/ppo_eval_trad.py: load the
ActorCritic checkpoint finetune/ppo_trad.py wrote (`--pretrained_model_path`, strict: ppo_eval_trad.py:283-286,497), score every query
of `<--dev_path>/test.h5` with the actor and log NDCG@k (ppo_eval_trad.py:347-409,506-520).  Models, reader and `evaluate` are
finetune/ppo_trad.py's (the reference repeats them in this file); NDCG for the whole split comes from one `lr2_ndcg` launch.
No CPU fallback.
"""
from __future__ import annotations

import argparse

import torch

from . import misc, ppo, ppo_trad
from .ppo_trad import ActorCritic, LTRDataset, Reward, evaluate  # noqa: F401


def load_or_initialize_parameters(args, model):
    """ppo_eval_trad.py:283-290: the whole ActorCritic, strict."""
    if getattr(args, "pretrained_model_path", None) is not None:
        model.load_state_dict(torch.load(args.pretrained_model_path, map_location="cpu"), strict=True)
    else:
        ppo._init_normal(model)


def main(argv=None):
    """python -m lr2ppo_amd.finetune.ppo_eval_trad <flags of ppo_trad.sh> --pretrained_model_path stage3_trad.bin --dev_path DIR"""
    from ..tencentpretrain.utils.logging import init_logger
    args = ppo.build_parser().parse_args(argv)
    args.labels_num = 3
    misc.init_distributed_mode(args)
    misc.setup_seed(args.seed + misc.get_rank())
    args.is_master = misc.is_main_process()
    model = ActorCritic(args, argparse.Namespace(**vars(args)))
    load_or_initialize_parameters(args, model)
    if args.is_master:
        args.logger = init_logger(args)
    args.device = torch.device("cuda", torch.cuda.current_device())
    args.model = model.to(args.device)
    valset = LTRDataset(args, args.dev_path, is_train=False)
    val_loader = ppo.get_dataloader(args, valset, misc.get_world_size(), misc.get_rank(), is_train=False)
    return evaluate(args, val_loader, 0, split="val", num_tasks=misc.get_world_size())


if __name__ == "__main__":
    main()
