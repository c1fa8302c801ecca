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
from .cpu_threads import fit_torch_threads


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
    fit_torch_threads()                 # one thread per usable core (cgroup quota), reported as `threads`
    m = PpoCpu(tags, seed)
    m.step(warmup_bs)
    runs = [m.step(bs) for _ in range(max(1, steps))]
    runs.sort(key=lambda r: r["total_s"])
    med = runs[len(runs) // 2]
    return dict(med, steps=len(runs), warmup_bs=warmup_bs, threads=torch.get_num_threads())


class FeaturesCpu:
    """The oracle's dual-encoder forward (ViT-B/16 + RoBERTa-base, 12 layers each, random N(0, 0.02) weights, inference) on the
    host cores: uint8 frames -> /255 -> CLIP mean / std -> vit_embedding -> transformer_encoder -> pooling_first, token ids ->
    text_embedding -> transformer_encoder (tencentpretrain/models/model.py:32-41 with the shipped configs)."""

    MEAN = (0.48145466, 0.4578275, 0.40821073)       # tencentpretrain/utils/dataloader.py:561
    STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, seed: int = 60):
        self.pve = O.seeded_params(O.vit_embedding_spec(768, 3, 16, 197), seed=seed + 1)
        self.pvn = O.seeded_params(O.encoder_param_spec(12, 768, 3072, True), seed=seed + 2)
        self.pte = O.seeded_params(O.text_embedding_spec(768, 50265, 514), seed=seed + 4)
        self.ptn = O.seeded_params(O.encoder_param_spec(12, 768, 3072, False), seed=seed + 5)

    @torch.no_grad()
    def extract(self, frames, ids, seg, chunk: int = 64):
        B, n_img = frames.shape[:2]
        T, L = ids.shape[1:]
        mean, std = torch.tensor(self.MEAN).view(1, 3, 1, 1), torch.tensor(self.STD).view(1, 3, 1, 1)
        flat = frames.reshape(B * n_img, 3, 224, 224)
        out = []
        for i in range(0, flat.shape[0], chunk):              # frames in chunks: the [b, 12, 197, 197] score tensors stay small
            x = (flat[i:i + chunk].float().div(255) - mean) / std
            vseg = torch.ones(x.shape[0], 197, dtype=torch.long)
            h = O.transformer_encoder(self.pvn, O.vit_embedding(self.pve, x, 16), vseg, 12, 12, True)
            out.append(O.pooling_first(h, vseg))
        img_emb = torch.cat(out).reshape(B, n_img, 768)
        s2 = seg.reshape(B * T, L)
        text_emb = O.transformer_encoder(self.ptn, O.text_embedding(self.pte, ids.reshape(B * T, L), s2), s2, 12, 12, False)
        return text_emb.reshape(B, T, L, 768), img_emb


def time_feature_extraction(bs: int = 32, tags: int = 2, n_img: int = 16, seed: int = 7):
    """-> seconds of ONE dual-encoder forward at the PPO step's shapes (bs x n_img frames, bs x tags sequences) after a small
    untimed warm-up call."""
    fit_torch_threads()
    fx = FeaturesCpu()
    g = torch.Generator().manual_seed(seed)

    def raw(b):
        frames = torch.randint(0, 256, (b, n_img, 3, 224, 224), dtype=torch.uint8, generator=g)
        ids = torch.randint(5, 50265, (b, tags, 196), generator=g)
        lens = torch.randint(4, 197, (b, tags, 1), generator=g)
        return frames, ids, (torch.arange(196).view(1, 1, -1) < lens).to(torch.int64)
    fx.extract(*raw(1))
    batch = raw(bs)
    t0 = time.time()
    fx.extract(*batch)
    return time.time() - t0


def time_ppo_step(bs: int, tags: int = 2, seed: int = 7, lr: float = 1e-4):
    """One cold step (kept for callers of the round-1 interface)."""
    r = PpoCpu(tags, seed, lr).step(bs)
    return dict(r, threads=torch.get_num_threads())
