"""CPU baseline leg of bench.py: the oracle's restatement of one PPO step, timed on the host cores.

TEST / MEASUREMENT INFRASTRUCTURE (kind = "port"): plain torch-CPU fp32 running oracle/lr2ppo_oracle.py --
the same math as the reference's rollout (finetune/ppo.py:844-883) and update (:518-587) with dropout off,
autograd for the backward and the reference's in-place AdamW op sequence (optimizers.py:381-400).
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


def time_ppo_step(bs: int, tags: int = 2, seed: int = 7, lr: float = 1e-4):
    """-> dict(seconds per phase, threads).  One rollout batch + one update minibatch at batch `bs`."""
    torch.manual_seed(seed)
    pa = O.seeded_params(O.head_param_spec("actor"), seed=seed)
    pc = O.seeded_params(O.head_param_spec("critic"), seed=seed + 1)
    pr = O.seeded_params(O.head_param_spec("reward"), seed=seed + 2)
    text, img, tgts = O.seeded_head_inputs(seed + 3, bs, tags)
    state = torch.arange(tags).unsqueeze(0).repeat(bs, 1)
    t0 = time.time()
    with torch.no_grad():
        logits = O.actor_forward(pa, text, img, None)
        value = O.critic_forward(pc, text, img, state)
        scores = logits.view(bs, tags)
        nxt = O.rollout_next_state(scores, state)
        rewards = O.reward_forward(pr, text, img, nxt)
    t_roll = time.time() - t0
    del pr
    pa_g = {k: v.requires_grad_(True) for k, v in pa.items()}
    pc_g = {k: v.requires_grad_(True) for k, v in pc.items()}
    t0 = time.time()
    new_scores = O.actor_forward(pa_g, text, img, None).view(bs, tags)
    new_value = O.critic_forward(pc_g, text, img, state)
    loss, vloss, _ = O.ppo_update_math(new_scores, new_value, scores, rewards, value, nxt, 0.001, 0.001, 0.5)
    loss.backward()
    vloss.backward()
    t_fb = time.time() - t0
    t0 = time.time()
    with torch.no_grad():
        for params in (pa_g, pc_g):
            grads = {k: v.grad for k, v in params.items()}
            st = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in params.items()}
            _adamw_inplace(params, grads, st, lr, list(params))
    t_opt = time.time() - t0
    return dict(rollout_s=t_roll, fwd_bwd_s=t_fb, adamw_s=t_opt, threads=torch.get_num_threads(), bs=bs)
