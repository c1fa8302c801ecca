"""AdamW + LR schedules with the TencentPretrain call signatures, stepping on one fused HIP kernel.

Drop-in for the symbols finetune/ppo.py uses from tencentpretrain/utils/optimizers.py of the reference:
`AdamW(params, lr, betas, eps, weight_decay, correct_bias)` (:305-402), `get_linear_schedule_with_warmup`
(:62-86), `get_constant_schedule(_with_warmup)` and the `str2optimizer` / `str2scheduler` registries.
The update is the reference's exactly (eps 1e-6 outside the sqrt, optional bias correction folded into the
step size, decoupled decay applied AFTER the Adam update with the same lr) but runs as ONE launch per
parameter group over a device-resident chunk table instead of ~5 small kernels per tensor.
Adafactor and the other schedules of the reference are out of scope (no LR2PPO launcher uses them).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Iterable, Tuple

import torch
from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR

from ... import _native, ops

CHUNK_ELEMS = 1 << 18   # 1 MiB of fp32 per workgroup: 4000 workgroups for the 1.05 B parameters of actor+critic


class AdamW(Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-6,
                 weight_decay: float = 0.0, correct_bias: bool = True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._tables = {}
        self._external = set()
        self._external_seen = set()     # ids of parameters ever updated through external_update
        self._lr_dev = None      # per group: float32[1] device tensor the kernels read the learning rate from (use_device_lr)

    def _ensure_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p.data)
            st["exp_avg_sq"] = torch.zeros_like(p.data)
        return st

    def use_device_lr(self, tensors):
        """tensors: one float32[1] device tensor per param group (ops.StepScalars.lr_tensor), or None to go back to by-value
        learning rates.  While set, step() / external_update() launch kernels that read the rate from there -- a HIP graph
        captured around them follows the scheduler; the CALLER stores group["lr"] there before every step."""
        if tensors is not None and len(tensors) != len(self.param_groups):
            raise ValueError("use_device_lr: one tensor per param group")
        self._lr_dev = list(tensors) if tensors is not None else None

    def count_replayed_step(self):
        """Book-keeping of one step() that ran inside a replayed graph: per-parameter step counters and write marks."""
        ps = [p for g in self.param_groups for p in g["params"] if p.grad is not None or id(p) in self._external_seen]
        for p in ps:
            self._ensure_state(p)["step"] += 1
        ops.mark_params_written(ps)

    def external_update(self, p) -> "ops.AdamArgs":
        """Take parameter `p` out of the next step(): its update is done by a kernel that produces the gradient and
        applies this optimizer's rule in one pass (ops.gemm(adam=...), used for the 2 GB out_layer.fc1.weight).
        Returns the state tensors and the hyper-parameters the next step() would have used; p.grad is not read."""
        for gi, group in enumerate(self.param_groups):
            if any(q is p for q in group["params"]):
                break
        else:
            raise ValueError("external_update: parameter is not managed by this optimizer")
        if group["correct_bias"]:
            raise NotImplementedError("external_update needs correct_bias=False (what LR2PPO uses)")
        if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
            raise RuntimeError("lr2ppo_amd AdamW needs contiguous float32 HIP parameters")
        st = self._ensure_state(p)
        st["step"] += 1
        self._external.add(id(p))
        self._external_seen.add(id(p))
        return ops.AdamArgs(p.data, st["exp_avg"], st["exp_avg_sq"], group["lr"], group["betas"][0], group["betas"][1],
                            group["eps"], group["weight_decay"], lr_dev=self._lr_dev[gi] if self._lr_dev else None)

    def _table(self, gi: int, group):
        ps = [p for p in group["params"] if p.grad is not None and id(p) not in self._external]
        if not ps:
            return None, 0, ps, 0
        # the table holds raw pointers: parameters, gradients AND both moment tensors (load_state_dict replaces the latter)
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), p.numel(), self._ensure_state(p)["exp_avg"].data_ptr(),
                     self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        gi = (gi, len(ps))        # separate cached tables with / without externally updated parameters
        cached = self._tables.get(gi)
        if cached is not None and cached[0] == sig:
            return cached[1], cached[2], ps, cached[3]
        rows = []
        # one workgroup per chunk: keep >= ~2000 workgroups in flight even when the group is small (the 20 M parameters
        # left after out_layer.fc1.weight is updated inside its GEMM would otherwise run on 90 workgroups)
        total = sum(p.numel() for p in ps)
        chunk_elems = min(CHUNK_ELEMS, max(1 << 13, (-(-total // 2048) + 1023) // 1024 * 1024))
        for p in ps:
            if p.grad.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError("lr2ppo_amd AdamW needs contiguous float32 HIP parameters and gradients")
            st = self._ensure_state(p)
            n, off = p.numel(), 0
            while off < n:
                c = min(chunk_elems, n - off)
                rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr() + 4 * off,
                             st["exp_avg_sq"].data_ptr() + 4 * off, c, group["weight_decay"]))
                off += c
        arr = (_native.AdamChunk * len(rows))()
        for i, (a, b, c_, d, cnt, wd) in enumerate(rows):
            arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].count, arr[i].weight_decay = a, b, c_, d, cnt, wd
        dev = ps[0].device
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._tables[gi] = (sig, table, len(rows), sum(p.numel() for p in ps))
        return table, len(rows), ps, self._tables[gi][3]

    @torch.no_grad()
    def step(self, closure: Callable = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            table, n, ps, n_params = self._table(gi, group)
            if table is None:
                continue
            beta1, beta2 = group["betas"]
            step_size = group["lr"]
            for p in ps:
                self.state[p]["step"] += 1
            if group["correct_bias"]:
                t = self.state[ps[0]]["step"]
                step_size = step_size * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
                if group["weight_decay"] > 0.0 and step_size != group["lr"]:
                    # decay must use the raw lr (optimizers.py:399-400): run the decay-free update, then decay
                    raise NotImplementedError("correct_bias=True with weight decay is not used by LR2PPO")
            lr_dev = self._lr_dev[gi] if self._lr_dev else None
            if lr_dev is not None and group["correct_bias"]:
                raise NotImplementedError("device learning rates need correct_bias=False (what LR2PPO uses)")
            ops.adamw_multi(table, n, step_size, beta1, beta2, group["eps"], n_params=n_params, lr_dev=lr_dev)
            ops.mark_params_written(ps)
        if self._external:          # parameters a fused kernel updated since the last step (external_update)
            ops.mark_params_written([p for g_ in self.param_groups for p in g_["params"] if id(p) in self._external])
        self._external.clear()
        return loss


def get_constant_schedule(optimizer, last_epoch=-1):
    return LambdaLR(optimizer, lambda _: 1, last_epoch=last_epoch)


def get_constant_schedule_with_warmup(optimizer, num_warmup_steps, last_epoch=-1):
    def lr_lambda(current_step):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1.0, num_warmup_steps))
        return 1.0
    return LambdaLR(optimizer, lr_lambda, last_epoch=last_epoch)


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    """lambda(s) = s/warm for s < warm, then linear decay to 0 at num_training_steps.  lambda(0) = 0, so an
    optimizer used before the first scheduler.step() runs at lr 0 -- finetune/ppo.py steps its schedulers once
    per train_model call, hence the whole first PPO cycle trains at lr 0."""
    def lr_lambda(current_step: int):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - current_step) / float(max(1, num_training_steps - num_warmup_steps)))
    return LambdaLR(optimizer, lr_lambda, last_epoch)


str2optimizer = {"adamw": AdamW}
str2scheduler = {"linear": get_linear_schedule_with_warmup, "constant": get_constant_schedule,
                 "constant_with_warmup": get_constant_schedule_with_warmup}
