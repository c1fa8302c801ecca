"""pooling() of tencentpretrain/utils/misc.py:23-35 (mean / max / last / first over the sequence axis)."""
import sys

import torch


def pooling(memory_bank, seg, pooling_type):
    seg = torch.unsqueeze(seg, dim=-1).type_as(memory_bank)
    memory_bank = memory_bank * seg
    if pooling_type == "mean":
        return torch.sum(memory_bank, dim=1) / torch.sum(seg, dim=1)
    if pooling_type == "last":
        last = torch.squeeze(torch.sum(seg, dim=1).type(torch.int64) - 1)
        return memory_bank[torch.arange(memory_bank.shape[0]), last, :]
    if pooling_type == "max":
        return torch.max(memory_bank + (seg - 1) * sys.maxsize, dim=1)[0]
    return memory_bank[:, 0, :]
