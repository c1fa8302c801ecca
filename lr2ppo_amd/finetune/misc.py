"""Process-group bootstrap and rank helpers (finetune/misc.py:22-107 of the reference), on RCCL.

One process per GPU under torchrun (RANK / WORLD_SIZE / LOCAL_RANK from the environment); backend "nccl" is RCCL
on ROCm and runs over xGMI inside a node.  On a machine without a GPU (CPU tests) the backend falls back to gloo
for the *collectives only* -- model kernels still require the HIP library.
"""
import builtins
import datetime
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def mkdir(path):
    os.makedirs(path, exist_ok=True)


def setup_seed(seed):
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def init_distributed_mode(args):
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        args.rank, args.world_size, args.gpu, args.distributed = 0, 1, 0, False
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return
    args.rank = int(os.environ["RANK"])
    args.world_size = int(os.environ["WORLD_SIZE"])
    args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    args.distributed = True
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required for RCCL on this driver
    if torch.cuda.is_available():
        torch.cuda.set_device(args.gpu)
        args.dist_backend = "nccl"
    else:
        args.dist_backend = "gloo"
    print("| distributed init (rank {}): {}, gpu {}".format(args.rank, getattr(args, "dist_url", "env://"), args.gpu), flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank)
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def setup_for_distributed(is_master):
    """Silence print() on non-master ranks (pass force=True to override), with a timestamp prefix."""
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False) or get_world_size() > 8
        if is_master or force:
            builtin_print("[{}] ".format(datetime.datetime.now().time()), end="")
            builtin_print(*args, **kwargs)

    builtins.print = print
