"""Stage 3 with the encoders trained through it (features.finetune_ppo_step; VERDICT r3 "missing" #6): ppo.update_minibatch hands
back d (policy loss + value loss) / d features, and the explicit schedule rollout (eval features) -> update (train features) ->
encoder backward -> AdamW reproduces the oracle's autograd over the composed chain (tencentpretrain/models/model.py:32-41 in front
of finetune/ppo.py:518-598)."""
import argparse

import pytest
import torch

from oracle import lr2ppo_oracle as O
from test_round3_gpu import _head_args, _one_layer_extractor

pytestmark = pytest.mark.gpu
REL = 2e-3          # the gradient bar of the encoder-backward tests


def _rel(a, b) -> float:
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def _heads(dev, ppo):
    Pa = O.seeded_params(O.head_param_spec("actor"), seed=7)
    Pc = O.seeded_params(O.head_param_spec("critic"), seed=8)
    Pr = O.seeded_params(O.head_param_spec("reward"), seed=23)
    with torch.no_grad():
        Pr["head.weight"] *= 40.0               # spread the rewards
    args = _head_args(dev)
    model = ppo.ActorCritic(args, None)
    model.actor.load_state_dict(Pa, strict=True)
    model.critic.load_state_dict(Pc, strict=True)
    reward = ppo.Reward(args, None)
    reward.load_state_dict(Pr, strict=True)
    return args, model.to(dev), reward.to(dev).eval(), Pa, Pc, Pr


def _oracle_update(Pa, Pc, text, img, rec_ref, seed, args):
    """policy loss + value loss of finetune/ppo.py:518-598 on (text, img) with the HIP path's dropout streams (actor: seed, critic:
    seed + 1) -> (policy loss, value loss); rec_ref = (old scores, rewards, old value, next_state) as constants."""
    s_old, r_old, v_old, ns = rec_ref
    T = text.shape[1]
    img_rep = img.unsqueeze(1).repeat(1, T, 1, 1)
    state = torch.arange(T).unsqueeze(0).repeat(text.shape[0], 1)
    s2 = O.actor_forward(Pa, text, img_rep, None, drop={"p": 0.1, "seed": seed, "site_base": 0}).view(text.shape[0], T)
    v2 = O.critic_forward(Pc, text, img_rep, state, drop={"p": 0.1, "seed": seed + 1, "site_base": 0})
    pl, vl, _ = O.ppo_update_math(s2, v2, s_old, r_old, v_old, ns, args.kl_div_loss_weight, args.entropy_weight, args.value_clip)
    return pl, vl


def test_update_minibatch_hands_back_the_feature_gradients(dev):
    """update_minibatch(..., input_grads=True): d (policy loss + value loss) / d text_emb and / d img_emb (the item's ONE set of image
    tokens: the sum over its tags and over both models) against the oracle's autograd with the same dropout masks; metrics and
    parameter updates are those of the plain call."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    args, model, reward, Pa, Pc, Pr = _heads(dev, ppo)
    g = torch.Generator().manual_seed(5)
    text, img = torch.randn(2, 2, 196, 768, generator=g) * 0.5, torch.randn(2, 16, 768, generator=g) * 0.5
    tgts = torch.randint(0, 3, (2, 2), generator=g)
    model.eval()
    rec = ppo.rollout_step(model, reward, text.to(dev), img.to(dev), tgts.to(dev))
    opt, copt, sch, csch = ppo.build_optimizer(args, model)          # lr 0 at the first step: weights stay
    model.train()
    runtime.set_dropout_seed(515)
    seed = runtime.peek_drop_seed()
    m, d_text, d_img = ppo.update_minibatch(args, model, opt, copt, rec, input_grads=True)
    assert d_text.shape == (2, 2, 196, 768) and d_img.shape == (2, 16, 768)
    tl, il = text.clone().requires_grad_(True), img.clone().requires_grad_(True)
    rec_ref = (rec[2].cpu(), rec[3].cpu(), rec[4].cpu(), rec[1].cpu())
    pl, vl = _oracle_update(Pa, Pc, tl, il, rec_ref, seed, args)
    (pl + vl).backward()
    assert abs(float(m[0]) - float(pl)) < 1e-3 * max(1.0, abs(float(pl))) and abs(float(m[1]) - float(vl)) < 1e-3 * max(1.0, abs(float(vl)))
    assert _rel(d_text, tl.grad) < REL, _rel(d_text, tl.grad)
    assert _rel(d_img, il.grad) < REL, _rel(d_img, il.grad)
    # the same call without input gradients: same metrics bit for bit
    runtime.set_dropout_seed(515)
    m2 = ppo.update_minibatch(args, model, opt, copt, rec)
    assert torch.equal(m, m2)


def test_finetune_ppo_step_trains_both_stacks_through_the_ppo_losses(dev):
    """finetune_ppo_step on 1-layer ViT-B/16 + 1-layer RoBERTa-base + the full-size heads, 2 items x 16 frames x 2 tags: the rollout
    record comes from EVAL-mode features; the update's losses and the encoder / embedding parameter gradients equal the oracle's
    autograd over embedding -> encoder -> (Actor, Critic) -> policy + value loss with the HIP path's dropout streams (2e-3); after
    the step every stack has moved."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    from lr2ppo_amd.finetune.features import build_encoder_optimizer, finetune_ppo_step, synthetic_raw_batch
    from test_round3_gpu import _oracle_chain          # noqa: F401  (same composition; restated below with two heads)
    from lr2ppo_amd import ops
    fx, pv, pt = _one_layer_extractor(dev)
    args, model, reward, Pa, Pc, Pr = _heads(dev, ppo)
    args.train_steps, args.batch_size = 20, 2
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    eopt, esch = build_encoder_optimizer(args, fx)
    sch.step(), csch.step(), esch.step()                               # leave lambda(0) = 0
    frames, ids, seg, tgts = synthetic_raw_batch(2, 2, generator=torch.Generator().manual_seed(35))
    before = {n: q.detach().clone() for n, q in fx.named_parameters() if q.numel() < 5_000_000}
    # what the step will see: eval-mode features for the rollout ...
    fx.eval(), model.eval()
    with torch.no_grad():
        t0, i0 = fx.extract(frames.to(dev), ids.to(dev), seg.to(dev))
        rec = ppo.rollout_step(model, reward, t0, i0, tgts.to(dev))
    rec_ref = (rec[2].cpu(), rec[3].cpu(), rec[4].cpu(), rec[1].cpu())
    runtime.set_dropout_seed(616)
    seed0 = runtime.peek_drop_seed()
    m = finetune_ppo_step(args, fx, model, reward, opt, copt, eopt, frames.to(dev), ids.to(dev), seg.to(dev), tgts.to(dev))
    assert torch.isfinite(m).all()
    # ... and the oracle chain in train mode: module calls draw seeds in the order image embedding, image encoder, text embedding,
    # text encoder (forward_train), actor, critic (update_minibatch)
    leaves = {}

    def req(d, tag):
        out = {}
        for k, v in d.items():
            out[k] = v.clone().requires_grad_(True)
            leaves[tag + k] = out[k]
        return out
    pvl, ptl = req(pv, "image."), req(pt, "text.")
    sub = lambda d, pre: {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}      # noqa: E731
    drop = lambda k: {"p": 0.1, "seed": seed0 + k, "site_base": 0}                          # noqa: E731
    B, n_img, T, L = 2, 16, 2, 196
    x = frames.float().div(255)
    x = ((x - torch.tensor(ops.CLIP_MEAN).view(1, 1, 3, 1, 1)) / torch.tensor(ops.CLIP_STD).view(1, 1, 3, 1, 1)).reshape(B * n_img, 3, 224, 224)
    vseg = torch.ones(B * n_img, 197, dtype=torch.long)
    e = O.vit_embedding(sub(pvl, "embedding."), x, 16, drop=drop(0))
    h = O.transformer_encoder(sub(pvl, "encoder."), e, vseg, 1, 12, True, drop=drop(1))
    img_emb = O.pooling_first(h, vseg).reshape(B, n_img, 768)
    s2 = seg.reshape(B * T, L)
    e = O.text_embedding(sub(ptl, "embedding."), ids.reshape(B * T, L), s2, drop=drop(2))
    text_emb = O.transformer_encoder(sub(ptl, "encoder."), e, s2, 1, 12, False, drop=drop(3)).reshape(B, T, L, 768)
    pl, vl = _oracle_update(Pa, Pc, text_emb, img_emb, rec_ref, seed0 + 4, args)
    (pl + vl).backward()
    assert abs(float(m[0]) - float(pl)) < 1e-3 * max(1.0, abs(float(pl))), (float(m[0]), float(pl))
    assert abs(float(m[1]) - float(vl)) < 1e-3 * max(1.0, abs(float(vl))), (float(m[1]), float(vl))
    G = {**{"image." + n: gq for n, gq in _named_grads(fx.image).items()}, **{"text." + n: gq for n, gq in _named_grads(fx.text).items()}}
    for name in ("text.encoder.transformer.0.feed_forward.linear_1.weight", "text.encoder.transformer.0.self_attn.linear_layers.2.weight",
                 "image.encoder.transformer.0.self_attn.linear_layers.0.weight", "image.encoder.transformer.0.feed_forward.linear_2.weight",
                 "image.embedding.patch.projection.weight", "text.embedding.pos.embedding.weight", "image.encoder.layer_norm.gamma"):
        r = _rel(G[name], leaves[name].grad)
        assert r < REL, (name, r)
    after = dict(fx.named_parameters())
    moved = {n: not torch.equal(before[n], after[n].detach()) for n in before}
    assert all(v for n, v in moved.items() if not n.endswith("linear_layers.1.bias")), [n for n, v in moved.items() if not v]


def _named_grads(stack):
    """{parameter name: gradient} of one stack: p.grad is bound to the module's flat gradient buffer (FeatureExtractor.bind_grads)."""
    return {n: q.grad for n, q in stack.named_parameters()}
