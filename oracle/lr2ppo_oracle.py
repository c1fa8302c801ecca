"""CPU oracle for the LR2PPO hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, single-file CPU (torch fp32) restatement of the reference algorithm for the path
named in BASELINE.json `north_star`.  It exists only to check
the HIP path: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it.  Nothing under `lr2ppo_amd/` imports it, and the product fails loudly when the HIP
library is missing instead of falling back to this file.

Parity pin: every function here is checked against fixtures under `tests/golden/` that were
produced by *importing the reference itself* in the build container (`oracle/gen_golden.py`);
see `tests/test_oracle_golden.py`.  The reference ships no tests of its own,
so those fixtures are the pin.

All parameters are passed as a flat ``dict[str, Tensor]`` using the reference's state_dict key
names, so the same dict can be loaded into the reference, this oracle and the product modules.
File:line citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Params = Dict[str, torch.Tensor]

SEQ_LEN = 196      # hard-coded in the reference: finetune/ppo.py:219-220
FEAT = 768         # finetune/ppo.py:202-208
XIT_HEADS = 8      # finetune/xit.py:114


# --------------------------------------------------------------------------------------------
# parameter specs (names, shapes, order == reference named_parameters() order)
# --------------------------------------------------------------------------------------------
def _mlp_spec(prefix: str, din: int, dh: int, dout: int):
    return [(f"{prefix}.fc1.weight", (dh, din)), (f"{prefix}.fc1.bias", (dh,)),
            (f"{prefix}.fc2.weight", (dout, dh)), (f"{prefix}.fc2.bias", (dout,))]


def _xit_spec(prefix: str, d: int):
    """Key layout of finetune/xit.py:9-42 (nn.Sequential nesting gives the numeric path)."""
    a = f"{prefix}.0.0.0.fn"
    f = f"{prefix}.0.0.1.fn"
    out = []
    for ln in ("ln_x", "ln_y"):
        out += [(f"{a}.0.{ln}.weight", (d,)), (f"{a}.0.{ln}.bias", (d,))]
    for lin in ("keys", "queries", "values", "projection"):     # declaration order xit.py:118-122
        out += [(f"{a}.1.{lin}.weight", (d, d)), (f"{a}.1.{lin}.bias", (d,))]
    out += [(f"{f}.0.weight", (d,)), (f"{f}.0.bias", (d,))]
    out += [(f"{f}.1.0.weight", (4 * d, d)), (f"{f}.1.0.bias", (4 * d,))]
    out += [(f"{f}.1.3.weight", (d, 4 * d)), (f"{f}.1.3.bias", (d,))]
    out += [(f"{prefix}.1.0.weight", (d,)), (f"{prefix}.1.0.bias", (d,))]
    return out


def head_param_spec(kind: str, seq_length: int = SEQ_LEN, max_imgs: int = 16, feat: int = FEAT, n_out: int = 1):
    """(name, shape) list for ``kind`` in {"actor", "critic", "reward"}.

    Order follows module declaration order in finetune/ppo.py:196-212 (Actor) and
    :247-263 / :300-316 (Critic / Reward: pos_emb is declared between img_proj and xit).
    """
    d = feat
    spec = _mlp_spec("text_proj", d, 4 * d, d) + _mlp_spec("img_proj", d, 4 * d, d)
    if kind in ("critic", "reward"):
        spec += [("pos_emb.weight", (4, d))]
    spec += _xit_spec("xit", d)
    if kind in ("critic", "reward"):
        spec += _xit_spec("xitt", d)
    spec += _mlp_spec("out_layer", (seq_length + max_imgs) * d, 4 * d, d)
    spec += [("head.weight", (n_out, d)), ("head.bias", (n_out,))]     # n_out = labels_num for the 'cls' actor (ppo.py:209-210)
    return spec


def seeded_params(spec, seed: int, std: float = 0.02, skip_gamma_beta: bool = True) -> Params:
    """normal_(0, std) for every parameter in spec order -- the reference's own initialiser
    (finetune/ppo.py:362-365), drawn from one torch CPU generator so both sides can rebuild it."""
    g = torch.Generator().manual_seed(seed)
    out: Params = {}
    for name, shape in spec:
        if skip_gamma_beta and name.endswith("gamma"):
            out[name] = torch.ones(shape)
        elif skip_gamma_beta and name.endswith("beta"):
            out[name] = torch.zeros(shape)
        else:
            out[name] = torch.empty(shape).normal_(0, std, generator=g)
    return out


# --------------------------------------------------------------------------------------------
# A1  Mlp, building blocks
# --------------------------------------------------------------------------------------------
def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """Exact-erf GELU (nn.GELU() default, finetune/ppo.py:155; tencentpretrain/utils/act_fun.py:7-8)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    y = x @ P[prefix + ".weight"].t()
    b = P.get(prefix + ".bias")
    return y if b is None else y + b


def mlp(P: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """finetune/ppo.py:164-170 with drop=0."""
    return linear(P, prefix + ".fc2", gelu_erf(linear(P, prefix + ".fc1", x)))


def layernorm_torch(x, w, b, eps: float = 1e-5):
    """nn.LayerNorm semantics used by XiT (finetune/xit.py:37,74,93-94): biased variance, eps in sqrt."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def layernorm_tp(x, gamma, beta, eps: float = 1e-6):
    """TencentPretrain LayerNorm (layers/layer_norm.py:16-21): unbiased std, eps added to std."""
    mu = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)          # unbiased (N-1)
    return gamma * (x - mu) / (std + eps) + beta


# --------------------------------------------------------------------------------------------
# dropout mask shared with the HIP kernels (counter-based hash; not a reference algorithm --
# the reference uses torch's Philox stream which no other backend can reproduce)
# --------------------------------------------------------------------------------------------
def dropout_keep_mask(seed: int, site: int, numel: int, p: float) -> np.ndarray:
    """keep[i] for flat element index i; mirrors lr2ppo_amd/csrc/common.h::dropout_keep (round 3: one 32-bit hash -- the
    'lowbias32' finaliser on (i >> 1) ^ k32 -- serves two consecutive elements, 16 bits each, against floor(p * 65536))."""
    M32 = np.uint64(0xFFFFFFFF)
    key = ((int(site) << 40) ^ ((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    k32 = np.uint64((key ^ (key >> 32)) & 0xFFFFFFFF)
    idx = np.arange(numel, dtype=np.uint64)
    x = ((idx >> np.uint64(1)) & M32) ^ k32                     # 32-bit arithmetic carried in uint64 lanes
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    field = np.where((idx & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
    # the C ABI carries p as a float: the threshold is formed from float32(p)
    thr = np.uint64(min(int(float(np.float32(p)) * 65536.0), 65535))
    return field >= thr


def attention_keep_mask(seed: int, site: int, batch: int, heads: int, L: int, p: float) -> np.ndarray:
    """keep[b, h, q, key] of the encoders' probability dropout (lr2_self_attn_fwd / bwd): the mask stream is laid out with the key
    dimension pitched to a multiple of 4 (csrc/selfattn.hip::mask_pitch)."""
    pitch = (L + 3) // 4 * 4
    return np.ascontiguousarray(dropout_keep_mask(seed, site, batch * heads * L * pitch, p).reshape(batch, heads, L, pitch)[..., :L])


def _apply_dropout(x, drop: Optional[dict], site: int, pitch4: bool = False):
    """pitch4: the mask stream pitches the LAST dimension to a multiple of 4 (the encoders' attention probabilities [b, heads, L, L]:
    element (b, h, q, key) is index ((b * heads + h) * L + q) * pitch + key, csrc/selfattn.hip::mask_pitch)."""
    if drop is None or drop.get("p", 0.0) <= 0.0:
        return x
    p = float(drop["p"])
    site_id = int(drop.get("site_base", 0)) + site
    if pitch4:
        L = x.shape[-1]
        pitch = (L + 3) // 4 * 4
        rows = x.numel() // L
        keep = dropout_keep_mask(int(drop["seed"]), site_id, rows * pitch, p).reshape(rows, pitch)[:, :L]
    else:
        keep = dropout_keep_mask(int(drop["seed"]), site_id, x.numel(), p)
    m = torch.from_numpy(np.ascontiguousarray(keep).astype(np.float32)).view_as(x) / (1.0 - p)
    return x * m


# --------------------------------------------------------------------------------------------
# A2  XiT  (finetune/xit.py)
# --------------------------------------------------------------------------------------------
def xit_attention(P: Params, prefix: str, x: torch.Tensor, y: torch.Tensor, heads: int = XIT_HEADS):
    """MultiHeadAttention.forward (finetune/xit.py:125-148).

    Quirks kept: no 1/sqrt(d) on the energies; softmax first, THEN divide by sqrt(emb_size)
    (:142-143); the 'causal' mask is computed but masked_fill is not in-place so it is a no-op (:140).
    """
    b, n, e = x.shape
    m = y.shape[1]
    d = e // heads
    q = linear(P, prefix + ".queries", x).view(b, n, heads, d).permute(0, 2, 1, 3)
    k = linear(P, prefix + ".keys", y).view(b, m, heads, d).permute(0, 2, 1, 3)
    v = linear(P, prefix + ".values", y).view(b, m, heads, d).permute(0, 2, 1, 3)
    energy = q @ k.transpose(-1, -2)
    att = torch.softmax(energy, dim=-1) / (e ** 0.5)
    out = (att @ v).permute(0, 2, 1, 3).reshape(b, n, e)
    return linear(P, prefix + ".projection", out)


def xit(P: Params, prefix: str, x: torch.Tensor, y: torch.Tensor, drop: Optional[dict] = None,
        heads: int = XIT_HEADS) -> torch.Tensor:
    """XiT = XEncoderBlock + final LayerNorm (finetune/xit.py:9-42,71-74).

    drop = {"p": 0.1, "seed": s, "site_base": k} enables the three train-time dropouts
    (xit.py:34,108 [inside FFN, after GELU],40) with the build's counter-based mask.
    """
    a = f"{prefix}.0.0.0.fn"
    f = f"{prefix}.0.0.1.fn"
    xn = layernorm_torch(x, P[f"{a}.0.ln_x.weight"], P[f"{a}.0.ln_x.bias"])
    yn = layernorm_torch(y, P[f"{a}.0.ln_y.weight"], P[f"{a}.0.ln_y.bias"])
    att = xit_attention(P, f"{a}.1", xn, yn, heads)
    x1 = _apply_dropout(att, drop, 0) + x                                  # xit.py:45-55
    h = layernorm_torch(x1, P[f"{f}.0.weight"], P[f"{f}.0.bias"])
    h = gelu_erf(linear(P, f"{f}.1.0", h))
    h = _apply_dropout(h, drop, 1)
    h = linear(P, f"{f}.1.3", h)
    x2 = _apply_dropout(h, drop, 2) + x1                                   # xit.py:77-86
    return layernorm_torch(x2, P[f"{prefix}.1.0.weight"], P[f"{prefix}.1.0.bias"])


# --------------------------------------------------------------------------------------------
# A3-A5  Actor / Critic / Reward  (finetune/ppo.py:196-350)
# --------------------------------------------------------------------------------------------
def trunk(P: Params, text_emb: torch.Tensor, img_emb: torch.Tensor, drop: Optional[dict] = None):
    """Shared trunk: text_proj, img_proj, xit, concat, out_layer -> [bs, tags, 768]
    (finetune/ppo.py:215-227 == :273-285 == :326-338)."""
    bs, tags = text_emb.shape[:2]
    tf = mlp(P, "text_proj", text_emb).reshape(bs * tags, SEQ_LEN, FEAT)
    imf = mlp(P, "img_proj", img_emb).reshape(bs * tags, -1, FEAT)
    x = xit(P, "xit", tf, imf, drop)
    x = torch.cat([x, imf], dim=1)
    x = mlp(P, "out_layer", x.reshape(x.shape[0], -1))
    return x.view(bs, tags, FEAT)


def smooth_l1(pred, tgt, beta: float = 0.3):
    d = (pred - tgt).abs()
    return torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta).mean()


def actor_forward(P: Params, text_emb, img_emb, tgts=None, drop=None):
    """Actor.forward, mode 'reg' (finetune/ppo.py:214-238) -> (loss, logits[bs*tags]) or logits."""
    x = trunk(P, text_emb, img_emb, drop)
    logits = linear(P, "head", x).view(-1)
    if tgts is None:
        return logits
    loss = smooth_l1(logits, tgts.reshape(-1).to(logits.dtype))
    return loss, logits


def actor_forward_cls(P: Params, text_emb, img_emb, tgts=None, drop=None):
    """Actor.forward, mode 'cls' (finetune/ppo.py:214-244): logits [bs*tags, labels_num]; with targets the mean NLL of
    log-softmax (:239-241)."""
    x = trunk(P, text_emb, img_emb, drop)
    logits = linear(P, "head", x).reshape(-1, P["head.weight"].shape[0])
    if tgts is None:
        return logits
    lz = torch.log_softmax(logits, dim=-1)
    loss = -lz[torch.arange(logits.shape[0]), tgts.reshape(-1)].mean()
    return loss, logits


def cls_action_scores(logits: torch.Tensor, bs: int, tags: int, softmax: bool = True) -> torch.Tensor:
    """The score the PPO loop ranks by in mode 'cls': sum_k k * softmax(logits)_k (finetune/ppo.py:532-537,859-863);
    evaluate() applies the same weights to the RAW logits (:641-643) -> softmax=False."""
    z = logits.view(bs, tags, -1)
    p = z.softmax(dim=-1) if softmax else z
    k = torch.arange(p.shape[-1], dtype=p.dtype)
    return (p * k).sum(-1)


def critic_forward(P: Params, text_emb, img_emb, index, n_pos: Optional[int] = None, drop=None):
    """Critic.forward (finetune/ppo.py:265-297); Reward.forward (:318-350) is the same with
    pos_emb(arange(4)) hard-coded (pass n_pos=4)."""
    bs = text_emb.shape[0]
    bi = torch.arange(bs).view(bs, 1)
    text_emb = text_emb[bi, index]
    img_emb = img_emb[bi, index]
    x = trunk(P, text_emb, img_emb, drop)
    tags = x.shape[1]
    n_pos = tags if n_pos is None else n_pos
    x = x + P["pos_emb.weight"][:n_pos].unsqueeze(0)
    drop2 = None if drop is None else dict(drop, site_base=int(drop.get("site_base", 0)) + 3)
    x = xit(P, "xitt", x, x, drop2)
    logits = linear(P, "head", x)          # [bs, tags, 1]
    return logits[:, -1].reshape(bs)


def reward_forward(P: Params, text_emb, img_emb, index):
    return critic_forward(P, text_emb, img_emb, index, n_pos=4)


# --------------------------------------------------------------------------------------------
# A6  rollout glue (finetune/ppo.py:844-874)
# --------------------------------------------------------------------------------------------
def rollout_next_state(scores: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
    """sort desc, permute state, prepend [0,1] (finetune/ppo.py:865-874)."""
    _, idx = torch.sort(scores, dim=-1, descending=True)
    nxt = torch.gather(state, 1, idx)
    bs = scores.shape[0]
    return torch.cat([torch.arange(2).unsqueeze(0).repeat(bs, 1), nxt], dim=1)


# --------------------------------------------------------------------------------------------
# A7-A9  losses and the PPO update math
# --------------------------------------------------------------------------------------------
def rank_loss(scores: torch.Tensor, indices: torch.Tensor, margin: float = 0.01) -> torch.Tensor:
    """RankLoss.forward (finetune/ppo.py:43-55): mean over *positive* hinge entries of the batch."""
    s = torch.gather(scores, 1, indices)
    diff = margin - (s.unsqueeze(2) - s.unsqueeze(1))
    hinge = torch.relu(torch.triu(diff, diagonal=1))
    cnt = torch.sign(hinge).sum()
    if cnt == 0:
        return hinge.sum()
    return hinge.sum() / cnt


def clipped_value_loss(values, rewards, old_values, clip):
    """finetune/ppo.py:494-498."""
    vc = old_values + (values - old_values).clamp(-clip, clip)
    l1 = (vc.flatten() - rewards) ** 2
    l2 = (values.flatten() - rewards) ** 2
    return torch.mean(torch.max(l1, l2))


def _log(t, eps=1e-20):
    return torch.log(t.clamp(min=eps))      # finetune/ppo.py:431-432


def ppo_update_math(scores, value, old_scores, rewards, old_value, next_state,
                    kl_w: float, ent_w: float, value_clip: float):
    """Everything between the two model forwards and the two backward() calls of one
    train_model minibatch (finetune/ppo.py:539-584).  Returns (loss, value_loss, dict of the
    per-sample tensors the reference logs)."""
    old_p = old_scores.softmax(dim=-1)
    new_p = scores.softmax(dim=-1)
    kl = (old_p * (_log(old_p) - _log(new_p))).sum(dim=-1) if kl_w > 0 else torch.zeros(scores.shape[0])
    ent = -(new_p * _log(new_p)).sum(dim=-1) if ent_w > 0 else torch.zeros(scores.shape[0])
    rewards_ori = rewards.clone()
    r = rewards - kl * kl_w                                   # not detached (:556)
    adv = r - old_value
    eps = -0.1
    tail = next_state[:, -2:]
    order = torch.where((adv >= eps).unsqueeze(1), tail, tail.flip(dims=[-1]))
    abs_adv = adv.abs()                                       # "< eps -> 0" is a no-op (:570)
    rl = rank_loss(scores, order, 0.01)
    loss = (rl * abs_adv - ent_w * ent).mean()
    vloss = clipped_value_loss(value, r.detach(), old_value, value_clip)
    extras = dict(kl=kl, entropy=ent, rewards_ori=rewards_ori, rewards=r, advantages=adv,
                  rank_loss=rl, order=order)
    return loss, vloss, extras


def masked_normalize(t, eps=1e-5):
    """finetune/ppo.py:485-491 (defined, never called by the reference; kept for completeness)."""
    mc = t - t.mean()
    var = (mc ** 2).mean()
    return mc * var.clamp(min=eps).rsqrt()


# --------------------------------------------------------------------------------------------
# A10-A11  AdamW and the linear schedule  (tencentpretrain/utils/optimizers.py)
# --------------------------------------------------------------------------------------------
def adamw_step(p, g, m, v, lr, wd, beta1=0.9, beta2=0.999, eps=1e-6):
    """One AdamW.step for one tensor, correct_bias=False (optimizers.py:381-400).
    Returns new (p, m, v).  Decay uses the already-updated p and the same lr."""
    m = m * beta1 + g * (1.0 - beta1)
    v = v * beta2 + g * g * (1.0 - beta2)
    p = p - lr * (m / (v.sqrt() + eps))
    if wd > 0.0:
        p = p + p * (-lr * wd)
    return p, m, v


def no_decay(name: str) -> bool:
    """finetune/ppo.py:381-393: substring match on bias|gamma|beta."""
    return any(nd in name for nd in ("bias", "gamma", "beta"))


def linear_schedule_lambda(step: int, warmup_steps: float, train_steps: float) -> float:
    """optimizers.py:77-84."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(train_steps - step) / float(max(1, train_steps - warmup_steps)))


# --------------------------------------------------------------------------------------------
# A12  NDCG  (ndcg.py:28-65, finetune/ppo.py:645-659)
# --------------------------------------------------------------------------------------------
NDCG_KS = (1, 3, 5, 10, 20, 100000000)


def dcg_at_k(rel: torch.Tensor, k: int) -> torch.Tensor:
    n = min(len(rel), k)
    dcg = torch.zeros((), dtype=torch.float32)
    for i in range(n):
        dcg = dcg + (2 ** rel[i] - 1) / torch.log2(torch.tensor(i + 2))
    return dcg


def ndcg_vector(scores: torch.Tensor, gold: torch.Tensor) -> torch.Tensor:
    """[6] NDCG@{1,3,5,10,20,1e8} for one item; 1 when the ideal DCG <= 1e-6 (ndcg.py:60-63)."""
    _, idx = torch.sort(scores, dim=-1, descending=True)
    pred_rel = gold[idx]
    true_rel, _ = torch.sort(gold, dim=-1, descending=True)
    out = []
    for k in NDCG_KS:
        p = dcg_at_k(pred_rel, k)
        t = dcg_at_k(true_rel, k)
        out.append(torch.ones(()) if t <= 1e-6 else (p / t).to(torch.float32))
    return torch.stack(out).to(torch.float32)


# --------------------------------------------------------------------------------------------
# A14-A15  TencentPretrain transformer encoder + embeddings
# --------------------------------------------------------------------------------------------
def encoder_param_spec(layers: int, hidden: int, ff: int, pre_ln: bool, prefix: str = ""):
    """transformer.{i}.* keys in module order (layers/transformer.py:27-48, multi_headed_attn.py:18-23)."""
    spec = []
    for i in range(layers):
        t = f"{prefix}transformer.{i}"
        for j in range(3):
            spec += [(f"{t}.self_attn.linear_layers.{j}.weight", (hidden, hidden)),
                     (f"{t}.self_attn.linear_layers.{j}.bias", (hidden,))]
        spec += [(f"{t}.self_attn.final_linear.weight", (hidden, hidden)),
                 (f"{t}.self_attn.final_linear.bias", (hidden,))]
        spec += [(f"{t}.feed_forward.linear_1.weight", (ff, hidden)), (f"{t}.feed_forward.linear_1.bias", (ff,)),
                 (f"{t}.feed_forward.linear_2.weight", (hidden, ff)), (f"{t}.feed_forward.linear_2.bias", (hidden,))]
        spec += [(f"{t}.layer_norm_1.gamma", (hidden,)), (f"{t}.layer_norm_1.beta", (hidden,)),
                 (f"{t}.layer_norm_2.gamma", (hidden,)), (f"{t}.layer_norm_2.beta", (hidden,))]
    if pre_ln:
        spec += [(f"{prefix}layer_norm.gamma", (hidden,)), (f"{prefix}layer_norm.beta", (hidden,))]
    return spec


def tp_attention(P: Params, prefix: str, h: torch.Tensor, mask: torch.Tensor, heads: int, drop=None, site: int = 0):
    """MultiHeadedAttention.forward (layers/multi_headed_attn.py:27-76): linear_layers[0,1,2] =
    Q,K,V; scores / sqrt(d) THEN + mask(-10000); dropout on the probabilities (:72), mask indexed over [b, heads, L, L]."""
    b, L, e = h.shape
    d = e // heads
    q = linear(P, prefix + ".linear_layers.0", h).view(b, L, heads, d).transpose(1, 2)
    k = linear(P, prefix + ".linear_layers.1", h).view(b, L, heads, d).transpose(1, 2)
    v = linear(P, prefix + ".linear_layers.2", h).view(b, L, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-2, -1)) / math.sqrt(float(d)) + mask
    p = _apply_dropout(torch.softmax(s, dim=-1), drop, site, pitch4=True)
    o = (p @ v).transpose(1, 2).contiguous().view(b, L, e)
    return linear(P, prefix + ".final_linear", o)


ENC_SITES_PER_LAYER = 4    # dropout sites of layer i: 4i (attention probabilities), 4i+1 (dropout_1), 4i+2 (dropout_2)


def transformer_encoder(P: Params, emb: torch.Tensor, seg: torch.Tensor, layers: int, heads: int,
                        pre_ln: bool, prefix: str = "", drop=None) -> torch.Tensor:
    """TransformerEncoder.forward with mask='fully_visible' (encoders/transformer_encoder.py:48-138,
    layers/transformer.py:50-73).  drop = {"p", "seed", "site_base"}: train-mode dropout with the counter-based masks of
    the HIP path (three sites per layer, see ENC_SITES_PER_LAYER); None = dropout off."""
    b, L, _ = emb.shape
    mask = (seg > 0).unsqueeze(1).repeat(1, L, 1).unsqueeze(1).float()
    mask = (1.0 - mask) * -10000.0
    h = emb
    for i in range(layers):
        t = f"{prefix}transformer.{i}"
        g1, b1 = P[f"{t}.layer_norm_1.gamma"], P[f"{t}.layer_norm_1.beta"]
        g2, b2 = P[f"{t}.layer_norm_2.gamma"], P[f"{t}.layer_norm_2.beta"]
        s0 = ENC_SITES_PER_LAYER * i
        if not pre_ln:
            att = _apply_dropout(tp_attention(P, f"{t}.self_attn", h, mask, heads, drop, s0), drop, s0 + 1)
            inter = layernorm_tp(att + h, g1, b1)
            ffn = linear(P, f"{t}.feed_forward.linear_2", gelu_erf(linear(P, f"{t}.feed_forward.linear_1", inter)))
            h = layernorm_tp(_apply_dropout(ffn, drop, s0 + 2) + inter, g2, b2)
        else:
            inter = layernorm_tp(h, g1, b1)
            h = h + _apply_dropout(tp_attention(P, f"{t}.self_attn", inter, mask, heads, drop, s0), drop, s0 + 1)
            o = layernorm_tp(h, g2, b2)
            ffn = linear(P, f"{t}.feed_forward.linear_2", gelu_erf(linear(P, f"{t}.feed_forward.linear_1", o)))
            h = _apply_dropout(ffn, drop, s0 + 2) + h
    if pre_ln:
        h = layernorm_tp(h, P[f"{prefix}layer_norm.gamma"], P[f"{prefix}layer_norm.beta"])
    return h


def patch_embedding(P: Params, prefix: str, img: torch.Tensor, patch: int) -> torch.Tensor:
    """PatchEmbedding.forward (embeddings/patch_embedding.py:20-31): conv(k=s=patch, no bias) as a
    GEMM over unfolded patches, then prepend cls_emb."""
    b, c, H, W = img.shape
    w = P[prefix + ".projection.weight"]                         # [E, C, p, p]
    x = img.view(b, c, H // patch, patch, W // patch, patch).permute(0, 2, 4, 1, 3, 5)
    x = x.reshape(b, (H // patch) * (W // patch), c * patch * patch)
    pe = x @ w.view(w.shape[0], -1).t()
    cls = P[prefix + ".cls_emb"].expand(b, -1, -1)
    return torch.cat([cls, pe], dim=1)


def vit_embedding(P: Params, img: torch.Tensor, patch: int, prefix: str = "", drop=None) -> torch.Tensor:
    """Embedding(['patch','pos'], remove_embedding_layernorm) (embeddings/embedding.py:19-34); drop: train-mode dropout
    of embedding.py:33 with the HIP path's counter-based mask (site 0 of the embedding's own seed)."""
    e = patch_embedding(P, prefix + "patch", img, patch)
    L = e.shape[1]
    return _apply_dropout(e + P[prefix + "pos.embedding.weight"][:L].unsqueeze(0), drop, 0)


def text_embedding(P: Params, src: torch.Tensor, seg: torch.Tensor, prefix: str = "", drop=None) -> torch.Tensor:
    """Embedding(['word','pos','seg']) + TP LayerNorm (embeddings/embedding.py:19-34) + dropout (:33)."""
    L = src.shape[1]
    e = P[prefix + "word.embedding.weight"][src] + P[prefix + "pos.embedding.weight"][:L].unsqueeze(0) \
        + P[prefix + "seg.embedding.weight"][seg]
    return _apply_dropout(layernorm_tp(e, P[prefix + "layer_norm.gamma"], P[prefix + "layer_norm.beta"]), drop, 0)


def vit_embedding_spec(emb: int, channels: int, patch: int, max_seq: int, prefix: str = ""):
    return [(f"{prefix}patch.cls_emb", (1, 1, emb)), (f"{prefix}patch.projection.weight", (emb, channels, patch, patch)),
            (f"{prefix}pos.embedding.weight", (max_seq, emb))]


def text_embedding_spec(emb: int, vocab: int, max_seq: int, prefix: str = ""):
    # Embedding.__init__ registers layer_norm first, then update() appends word/pos/seg (embedding.py:6-17)
    return [(f"{prefix}layer_norm.gamma", (emb,)), (f"{prefix}layer_norm.beta", (emb,)),
            (f"{prefix}word.embedding.weight", (vocab, emb)), (f"{prefix}pos.embedding.weight", (max_seq, emb)),
            (f"{prefix}seg.embedding.weight", (3, emb))]


def dual_embedding(P: Params, src, seg, kinds, patch: int = 16, tied: bool = False):
    """DualEmbedding.forward in eval mode (embeddings/dual_embedding.py:39-66): stream i = Embedding(kinds[i]) -- with its own
    LayerNorm for the text composition (embedding.py:31-32) -- then `stream_i_layer_norm` when present (:56-58, :63-65).
    kinds[i] in {"text", "vit"}; with tie_weights both streams use embedding_1's parameters, which the reference's
    state_dict lists first under embedding_0 (:36-37)."""
    outs = []
    for i, kind in enumerate(kinds):
        pre = "embedding_0." if tied else f"embedding_{i}."
        e = text_embedding(P, src[i], seg[i], prefix=pre) if kind == "text" else vit_embedding(P, src[i], patch, prefix=pre)
        if f"stream_{i}_layer_norm.gamma" in P:
            e = layernorm_tp(e, P[f"stream_{i}_layer_norm.gamma"], P[f"stream_{i}_layer_norm.beta"])
        outs.append(e)
    return tuple(outs)


def dual_encoder(P: Params, emb, seg, layers: int, heads: int, pre_ln, tied: bool = False):
    """DualEncoder.forward (encoders/dual_encoder.py:27-47); pre_ln: one flag per stream."""
    return tuple(transformer_encoder(P, emb[i], seg[i], layers, heads, pre_ln[i], prefix="encoder_0." if tied else f"encoder_{i}.")
                 for i in range(2))


def seeded_head_inputs(seed: int, bs: int, tags: int, n_img: int = 16, n_cls: int = 3):
    """Synthetic head inputs of the reference's shapes: text_emb ~ N(0,1) [bs,tags,196,768],
    img_emb ~ N(0,1) [bs,n_img,768] repeated over tags (finetune/ppo.py:831), tgts in {0..n_cls-1}."""
    g = torch.Generator().manual_seed(seed)
    text = torch.randn(bs, tags, SEQ_LEN, FEAT, generator=g)
    img = torch.randn(bs, n_img, FEAT, generator=g).unsqueeze(1).repeat(1, tags, 1, 1)
    tgts = torch.randint(0, n_cls, (bs, tags), generator=g)
    return text, img, tgts


def pooling_first(hidden: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
    """utils/misc.py:23-35 default branch: multiply by seg then take token 0."""
    return (hidden * seg.unsqueeze(-1).type_as(hidden))[:, 0, :]


# ---------------------------------------------------------------------------------------------
# the stage-1 (pointwise) and stage-2 (pairwise reward) training steps
# ---------------------------------------------------------------------------------------------
def visual_projection(weight: torch.Tensor, cls_rows: torch.Tensor) -> torch.Tensor:
    """The bias-free map from an image tower's width to the feature width the heads hard-code (finetune/ppo.py:202-208): CLIP's
    `x = ln_post(x[:, 0, :]); x = x @ proj` -- the `model.encode_image` the reference's preprocess.py:59-61,83 calls to produce
    img_emb.  `clip` is a third-party dependency absent from /root/reference (pip3_list.txt pins clip 1.0, openai/CLIP
    clip/model.py VisionTransformer.forward); restated from its published form.  weight: [feat, hidden] (nn.Linear orientation =
    proj^T); cls_rows: [N, hidden], the pooled row behind the stack's final LayerNorm (= ln_post)."""
    return cls_rows @ weight.t()


def feature_chain(pv: Params, pt: Params, frames_u8: torch.Tensor, ids: torch.Tensor, seg: torch.Tensor, *, patch: int, vit_layers: int,
                  vit_heads: int, text_layers: int, text_heads: int = 12, proj: Optional[torch.Tensor] = None,
                  mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711), drop=None):
    """uint8 frames [B, n_img, 3, H, W] + token ids / seg [B, T, L] -> (text_emb [B, T, L, E], img_emb [B, n_img, feat]): the
    composition tencentpretrain/models/model.py:32-41 (`encoder(embedding(src, seg), seg)`) of both stacks, `/255` + CLIP mean / std
    (tencentpretrain/utils/dataloader.py:559-561), pooling 'first' (utils/misc.py:23-35) and, for a tower wider than the heads,
    visual_projection.  pv / pt: {"embedding.*", "encoder.*"} parameters of the image / text stack.  drop: None, or a function
    k -> dropout spec for module call k (0 image embedding, 1 image encoder, 2 text embedding, 3 text encoder)."""
    B, n_img, _, H, W = frames_u8.shape
    T, L = ids.shape[1:]
    d = drop or (lambda k: None)
    sub = lambda P, pre: {k[len(pre):]: v for k, v in P.items() if k.startswith(pre)}      # noqa: E731
    x = frames_u8.float().div(255)
    x = ((x - torch.tensor(mean).view(1, 1, 3, 1, 1)) / torch.tensor(std).view(1, 1, 3, 1, 1)).reshape(B * n_img, 3, H, W)
    n_tok = (H // patch) * (W // patch) + 1
    vseg = torch.ones(B * n_img, n_tok, dtype=torch.long)
    h = transformer_encoder(sub(pv, "encoder."), vit_embedding(sub(pv, "embedding."), x, patch, drop=d(0)), vseg, vit_layers, vit_heads,
                            True, drop=d(1))
    cls = pooling_first(h, vseg)
    if proj is not None:
        cls = visual_projection(proj, cls)
    img_emb = cls.reshape(B, n_img, -1)
    s2 = seg.reshape(B * T, L)
    e = text_embedding(sub(pt, "embedding."), ids.reshape(B * T, L), s2, drop=d(2))
    text_emb = transformer_encoder(sub(pt, "encoder."), e, s2, text_layers, text_heads, False, drop=d(3))
    return text_emb.reshape(B, T, L, -1), img_emb


def pair_hinge(chosen: torch.Tensor, reject: torch.Tensor, margin: float = 1.0):
    """finetune/reward_pair_dataloader.py:356-359: loss = relu(m_R - (chosen - reject)).mean(), acc = (chosen > reject).mean()."""
    return torch.relu(margin - (chosen - reject)).mean(), (chosen > reject).float().mean()


def pair_index(tag_targets, order):
    """get_index (reward_pair_dataloader.py:77-84) for an already shuffled two-element `order`:
    -> (chosen_index, reject_index), each 4 long; the first two entries are the shown order, the last two the
    candidate next order (kept when the first tag's target >= the second's, swapped otherwise)."""
    a, b = order
    keep, swap = [a, b, a, b], [a, b, b, a]
    return (keep, swap) if tag_targets[a] >= tag_targets[b] else (swap, keep)


def sgd_free_train_steps(P: Params, loss_fn, batches, base_lr: float, warmup_steps: float, train_steps: float):
    """The reference's per-batch recipe shared by pointwise.py:300-313 and reward_pair_dataloader.py:347-365:
    zero_grad -> loss -> backward -> AdamW.step (lr of the CURRENT schedule position, decay groups by the
    bias|gamma|beta substring rule) -> scheduler.step.  P is updated in place; returns the per-step outputs of loss_fn.
    (LambdaLR applies lambda(0) at construction, so the first step runs at lr = base_lr * lambda(0) = 0.)"""
    state = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in P.items()}
    outs = []
    for step, batch in enumerate(batches):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        out = loss_fn(Pg, batch)
        loss = out[0] if isinstance(out, tuple) else out
        loss.backward()
        lr = base_lr * linear_schedule_lambda(step, warmup_steps, train_steps)
        for k in P:
            m, v = state[k]
            p_new, m_new, v_new = adamw_step(P[k], Pg[k].grad, m, v, lr, 0.0 if no_decay(k) else 0.01)
            P[k], state[k] = p_new, (m_new, v_new)
        outs.append(tuple(o.detach() for o in out) if isinstance(out, tuple) else out.detach())
    return outs


def stage1_loss(P: Params, batch, drop=None):
    """pointwise.py:300-303 + Classifier.forward mode 'reg' (:203-232) == Actor.forward with targets."""
    text, img, tgts = batch
    loss, logits = actor_forward(P, text, img, tgts, drop=drop)
    return loss, logits


def stage2_loss(P: Params, batch, drop=None):
    """reward_pair_dataloader.py:347-359: two Classifier forwards (chosen / reject orderings) + hinge.
    Evaluated as ONE forward over the batch [chosen ; reject] (items are independent), which is also how the HIP path
    runs it; `drop` therefore indexes its masks over the concatenated batch."""
    text, img, chosen, reject = batch
    bs = text.shape[0]
    scores = critic_forward(P, torch.cat([text, text]), torch.cat([img, img]), torch.cat([chosen, reject]), n_pos=4, drop=drop)
    loss, acc = pair_hinge(scores[:bs], scores[bs:])
    return loss, acc, scores


# ---------------------------------------------------------------------------------------------
# A13 / 8f-3: a small in-memory stand-in for LRMovieNet (clean_feat.h5 + the split json) so that the three readers can be
# exercised without h5py or data.  Feature VALUES encode their origin: text_emb[t] == 100 * item + t everywhere and image
# row j == 1000 * item + j, so a reader's tag selection, image shuffle and cyclic padding can be read back from its output.
# ---------------------------------------------------------------------------------------------
def fake_movienet(seed: int = 3, n_items: int = 5):
    """-> (items, h5) where items is the split json (id, tags[{target}], index[[a, b], ...]) and h5 mimics
    h5py.File: h5[str(id)]["text_emb"][:] -> [n_tags, 2, 4], h5[str(id)]["img_emb"][:] -> [1, n_img, 768]."""
    rng = np.random.RandomState(seed)
    items, h5 = [], {}
    for i in range(n_items):
        n_tags = int(rng.randint(3, 9))
        n_img = int(rng.randint(3, 24))
        targets = [int(t) for t in rng.randint(0, 3, size=n_tags)]
        targets[:3] = [0, 1, 2]                                   # every label present: the stage-2 val reader keeps the item
        pairs = [[int(a), int(b)] for a, b in (rng.choice(n_tags, 2, replace=False) for _ in range(3))]
        items.append({"id": i, "tags": [{"target": t} for t in targets], "index": pairs})
        text = np.zeros((n_tags, 2, 4), dtype=np.float32) + (100 * i + np.arange(n_tags, dtype=np.float32))[:, None, None]
        img = np.zeros((1, n_img, 768), dtype=np.float32) + (1000 * i + np.arange(n_img, dtype=np.float32))[None, :, None]
        h5[str(i)] = {"text_emb": text, "img_emb": img}
    return items, h5


def describe_reader_item(sample):
    """Reader output -> plain ints: which tags / image rows it holds (see fake_movienet's value encoding)."""
    text, img, tgt = sample[0], sample[1], sample[2]
    out = {"tags": [int(v) for v in text[:, 0, 0].tolist()], "img_rows": [int(v) for v in img[:, 0].tolist()],
           "tgts": [int(v) for v in tgt.tolist()]}
    if len(sample) > 3:
        out["chosen"], out["reject"] = [int(v) for v in sample[3].tolist()], [int(v) for v in sample[4].tolist()]
    return out


def fake_letor(seed: int = 5, n_queries: int = 6, docs: int = 20, feats: int = 46):
    """A small LETOR split in the layout datasets_trad/convert_to_h5py.py:17-43 writes: {query id: float64 [docs, 2 + feats]}
    with column 0 = label (0..4), column 1 = query id, column 2 = 1000 * query number + row (so a reader's row selection can be
    read back from its output), the rest seeded noise.  Ids are chosen so that name order differs from numeric order."""
    rng = np.random.RandomState(seed)
    ids = [10002, 7, 345, 18, 9001, 23, 4, 77, 1200, 31][:n_queries]
    tables = {}
    for qn, qid in enumerate(ids):
        t = rng.standard_normal((docs, 2 + feats))
        t[:, 0] = rng.randint(0, 5, size=docs)
        if qn == 1:
            t[:, 0] = 2.0                                          # one query with a single relevance class: reward_trad keeps no pair
        t[:, 1] = qid
        t[:, 2] = 1000 * qn + np.arange(docs)
        tables[qid] = t
    return tables


def describe_letor_item(sample):
    """LTRDataset item -> plain numbers: labels, query id, the row-identifying feature column, shape / dtype, index layouts."""
    gt, qid, feats = (np.asarray(sample[0]), sample[1], np.asarray(sample[2]))
    out = {"gt": [int(v) for v in gt.tolist()], "qid": str(qid), "rows": [int(round(v)) for v in feats[:, 0].tolist()],
           "shape": list(feats.shape), "dtype": str(feats.dtype), "checksum": float(np.round(feats.sum(), 6))}
    if len(sample) > 3:
        out["chosen"], out["reject"] = [int(v) for v in np.asarray(sample[3]).tolist()], [int(v) for v in np.asarray(sample[4]).tolist()]
    return out


# ---------------------------------------------------------------------------------------------
# BASELINE.json configs[0] ("plumbing" case): the *_trad Classifier -- the same XiT / Mlp head on one pre-projected
# 768-d feature per document, sequence length 1 (finetune/pointwise_trad.py:132-177).
# ---------------------------------------------------------------------------------------------
def trad_param_spec(feat: int = FEAT):
    return _xit_spec("xit", feat) + _mlp_spec("out_layer", 2 * feat, 4 * feat, feat) + [("head.weight", (1, feat)), ("head.bias", (1,))]


def trad_forward(P: Params, text_emb: torch.Tensor, tgts=None, drop=None):
    """pointwise_trad.Classifier.forward, mode 'reg': text_emb [bs, docs, 768] -> (SmoothL1 loss, logits[bs*docs, 1]) or
    logits.  The document feature is both streams of the XiT block (:154) and is concatenated behind its output (:155)."""
    bs, docs = text_emb.shape[:2]
    f = text_emb.to(torch.float32).reshape(bs * docs, 1, FEAT)
    x = xit(P, "xit", f, f, drop=drop)
    x = mlp(P, "out_layer", torch.cat([x, f], dim=1).reshape(bs * docs, -1))
    logits = linear(P, "head", x).view(-1, 1)
    if tgts is None:
        return logits
    return smooth_l1(logits.view(-1), tgts.view(-1).to(torch.float32)), logits


def trad_head_param_spec(kind: str, feat: int = FEAT, n_out: int = 1):
    """(name, shape) list of finetune/ppo_trad.py's Actor (:143-156) / Critic (:193-205) / Reward (:240-252), in module
    declaration order: [pos_emb,] xit, [xitt,] out_layer = Mlp(2 * 768, 3072, 768), head."""
    spec = []
    if kind in ("critic", "reward"):
        spec += [("pos_emb.weight", (4, feat))]
    spec += _xit_spec("xit", feat)
    if kind in ("critic", "reward"):
        spec += _xit_spec("xitt", feat)
    spec += _mlp_spec("out_layer", 2 * feat, 4 * feat, feat)
    spec += [("head.weight", (n_out, feat)), ("head.bias", (n_out,))]
    return spec


def trad_trunk(P: Params, text_emb: torch.Tensor, drop=None) -> torch.Tensor:
    """ppo_trad.py:160-171: [bs, tags, 768] -> [bs, tags, 768]; the feature is both streams of the XiT block and is
    concatenated behind its output."""
    bs, tags = text_emb.shape[:2]
    f = text_emb.to(torch.float32).reshape(bs * tags, 1, FEAT)
    x = xit(P, "xit", f, f, drop=drop)
    x = mlp(P, "out_layer", torch.cat([x, f], dim=1).reshape(bs * tags, -1))
    return x.view(bs, tags, FEAT)


def trad_actor_forward(P: Params, text_emb, tgts=None, drop=None):
    """ppo_trad.Actor.forward, mode 'reg' (:158-183): logits [bs*tags] (+ SmoothL1 loss with targets)."""
    logits = linear(P, "head", trad_trunk(P, text_emb, drop)).view(-1)
    if tgts is None:
        return logits
    return smooth_l1(logits, tgts.view(-1).to(torch.float32)), logits


def trad_critic_forward(P: Params, text_emb, index, n_pos: Optional[int] = None, drop=None):
    """ppo_trad.Critic.forward (:207-236); Reward.forward (:254-281) is the same with pos_emb(arange(4)) (n_pos=4)."""
    bs = text_emb.shape[0]
    bi = torch.arange(bs).view(bs, 1)
    x = trad_trunk(P, text_emb[bi, index], drop)
    tags = x.shape[1]
    n_pos = tags if n_pos is None else n_pos
    x = x + P["pos_emb.weight"][:n_pos].unsqueeze(0)
    drop2 = None if drop is None else dict(drop, site_base=int(drop.get("site_base", 0)) + 3)
    x = xit(P, "xitt", x, x, drop2)
    return linear(P, "head", x)[:, -1].reshape(bs)


def trad2_param_spec(feat: int = FEAT):
    """finetune/pointwise_2data_trad.py:130-145: text_proj = Mlp(46, 3072, 768), text_proj3 = Mlp(136, 3072, 768), then
    pointwise_trad's head."""
    return _mlp_spec("text_proj", 46, 4 * feat, feat) + _mlp_spec("text_proj3", 136, 4 * feat, feat) + trad_param_spec(feat)


def trad2_forward(P: Params, text_emb: torch.Tensor, tgts=None, drop=None):
    """pointwise_2data_trad.Classifier.forward, mode 'reg' (:146-170): raw LETOR rows [bs, docs, 46 | 136] through the
    projection of their width, then trad_forward's body."""
    proj = {46: "text_proj", 136: "text_proj3"}[text_emb.shape[-1]]
    feat = mlp(P, proj, text_emb.to(torch.float32))
    return trad_forward(P, feat, tgts, drop)
