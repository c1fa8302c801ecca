"""CPU baseline leg of bench.py: the oracle's restatement of one PPO step, timed on the host cores.

TEST / MEASUREMENT INFRASTRUCTURE (kind = "port"): plain torch-CPU fp32 running oracle/lr2ppo_oracle.py --
the same math as the reference's rollout (finetune/ppo.py:844-883, eval mode) and update (:518-587, train mode: dropout
0.1 at the three XiT sites) with autograd for the backward and the reference's in-place AdamW op sequence
(optimizers.py:381-400).  Procedure of BASELINE.md section 3: one untimed warm-up step (allocates the Adam state, warms the
thread pool), then the measured step(s) at the benchmark batch.
"""
import time

import torch

from . import lr2ppo_oracle as O


def _adamw_inplace(params, grads, state, lr, names, beta1=0.9, beta2=0.999, eps=1e-6):
    for n in names:
        p, g = params[n], grads[n]
        m, v = state[n]
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        p.addcdiv_(m, v.sqrt().add_(eps), value=-lr)
        if not O.no_decay(n):
            p.add_(p, alpha=-lr * 0.01)


class PpoCpu:
    """Actor + critic + reward parameters and persistent AdamW state; step(bs) runs one PPO step and returns its phases."""

    def __init__(self, tags: int = 2, seed: int = 7, lr: float = 1e-4):
        self.tags, self.seed, self.lr, self.calls = tags, seed, lr, 0
        self.pa = O.seeded_params(O.head_param_spec("actor"), seed=seed)
        self.pc = O.seeded_params(O.head_param_spec("critic"), seed=seed + 1)
        self.pr = O.seeded_params(O.head_param_spec("reward"), seed=seed + 2)
        self.state = None

    def step(self, bs: int):
        tags = self.tags
        text, img, tgts = O.seeded_head_inputs(self.seed + 3 + self.calls, bs, tags)
        self.calls += 1
        st = torch.arange(tags).unsqueeze(0).repeat(bs, 1)
        t0 = time.time()
        with torch.no_grad():
            logits = O.actor_forward(self.pa, text, img, None)
            value = O.critic_forward(self.pc, text, img, st)
            scores = logits.view(bs, tags)
            nxt = O.rollout_next_state(scores, st)
            rewards = O.reward_forward(self.pr, text, img, nxt)
        t_roll = time.time() - t0
        pa_g = {k: v.detach().requires_grad_(True) for k, v in self.pa.items()}
        pc_g = {k: v.detach().requires_grad_(True) for k, v in self.pc.items()}
        drop = {"p": 0.1, "seed": 1000 + self.calls, "site_base": 0}
        t0 = time.time()
        new_scores = O.actor_forward(pa_g, text, img, None, drop=drop).view(bs, tags)
        new_value = O.critic_forward(pc_g, text, img, st, drop=drop)
        loss, vloss, _ = O.ppo_update_math(new_scores, new_value, scores, rewards, value, nxt, 0.001, 0.001, 0.5)
        loss.backward()
        vloss.backward()
        t_fb = time.time() - t0
        if self.state is None:          # first call: Adam state allocation (untimed warm-up step absorbs it)
            self.state = [{k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in p.items()} for p in (self.pa, self.pc)]
        t0 = time.time()
        with torch.no_grad():
            for params, pg, stt in ((self.pa, pa_g, self.state[0]), (self.pc, pc_g, self.state[1])):
                _adamw_inplace(params, {k: v.grad for k, v in pg.items()}, stt, self.lr, list(params))
        t_opt = time.time() - t0
        return dict(rollout_s=t_roll, fwd_bwd_s=t_fb, adamw_s=t_opt, total_s=t_roll + t_fb + t_opt, bs=bs)


def time_ppo_steps(bs: int = 32, tags: int = 2, steps: int = 1, warmup_bs: int = 2, seed: int = 7):
    """-> dict: median seconds per PPO step at batch `bs` over `steps` measured steps, after one untimed warm-up step at
    batch `warmup_bs` (same code path, allocates the 2 x 1.045 B-element Adam state)."""
    m = PpoCpu(tags, seed)
    m.step(warmup_bs)
    runs = [m.step(bs) for _ in range(max(1, steps))]
    runs.sort(key=lambda r: r["total_s"])
    med = runs[len(runs) // 2]
    return dict(med, steps=len(runs), warmup_bs=warmup_bs, threads=torch.get_num_threads())


def time_ppo_step(bs: int, tags: int = 2, seed: int = 7, lr: float = 1e-4):
    """One cold step (kept for callers of the round-1 interface)."""
    r = PpoCpu(tags, seed, lr).step(bs)
    return dict(r, threads=torch.get_num_threads())
