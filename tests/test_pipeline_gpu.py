"""The composed path of BASELINE config 2 / north_star: frames + token ids -> ViT-B/16 + RoBERTa-base -> (text_emb, img_emb)
-> LR2PPO heads (rollout + PPO update, stage-1 pointwise step), on a real MI355X against the CPU oracle chain
vit_embedding -> transformer_encoder -> pooling_first -> actor_forward."""
import argparse

import pytest
import torch

from oracle import lr2ppo_oracle as O

pytestmark = pytest.mark.gpu


def _head_args(dev, **over):
    d = dict(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True, kl_div_loss_weight=0.001,
             entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear", learning_rate=1e-3,
             critic_learning_rate=1e-3, train_steps=41, warmup=0.1, device=dev)
    d.update(over)
    return argparse.Namespace(**d)


def _extractor(dev):
    from lr2ppo_amd.finetune.features import FeatureExtractor
    fx = FeatureExtractor()
    pv = {**{"embedding." + k: v for k, v in O.seeded_params(O.vit_embedding_spec(768, 3, 16, 197), seed=61).items()},
          **{"encoder." + k: v for k, v in O.seeded_params(O.encoder_param_spec(12, 768, 3072, True), seed=62).items()}}
    pt = {**{"embedding." + k: v for k, v in O.seeded_params(O.text_embedding_spec(768, 50265, 514), seed=64).items()},
          **{"encoder." + k: v for k, v in O.seeded_params(O.encoder_param_spec(12, 768, 3072, False), seed=65).items()}}
    fx.image.load_state_dict(pv, strict=True)
    fx.text.load_state_dict(pt, strict=True)
    return fx.to(dev).eval(), pv, pt


def _oracle_features(pv, pt, frames, ids, seg):
    from lr2ppo_amd import ops
    B, n_img = frames.shape[:2]
    x = frames.float().div(255)
    x = (x - torch.tensor(ops.CLIP_MEAN).view(1, 1, 3, 1, 1)) / torch.tensor(ops.CLIP_STD).view(1, 1, 3, 1, 1)
    x = x.reshape(B * n_img, 3, 224, 224)
    pve = {k[len("embedding."):]: v for k, v in pv.items() if k.startswith("embedding.")}
    pvn = {k[len("encoder."):]: v for k, v in pv.items() if k.startswith("encoder.")}
    vseg = torch.ones(B * n_img, 197, dtype=torch.long)
    h = O.transformer_encoder(pvn, O.vit_embedding(pve, x, 16), vseg, 12, 12, True)
    img_emb = O.pooling_first(h, vseg).reshape(B, n_img, 768)
    T, L = ids.shape[1:]
    pte = {k[len("embedding."):]: v for k, v in pt.items() if k.startswith("embedding.")}
    ptn = {k[len("encoder."):]: v for k, v in pt.items() if k.startswith("encoder.")}
    s2 = seg.reshape(B * T, L)
    text_emb = O.transformer_encoder(ptn, O.text_embedding(pte, ids.reshape(B * T, L), s2), s2, 12, 12, False).reshape(B, T, L, 768)
    return text_emb, img_emb


def test_frames_and_ids_to_logits_match_the_oracle_chain(dev):
    """B = 2 items x 16 frames x 2 tags: features within 1e-3 of the oracle's ViT-B/16 / RoBERTa-base chain, Actor logits
    computed from the HIP features within 1e-3 of the oracle's logits computed from the oracle's features (north_star bar),
    Critic value likewise."""
    from lr2ppo_amd.finetune import ppo
    from lr2ppo_amd.finetune.features import synthetic_raw_batch
    fx, pv, pt = _extractor(dev)
    gen = torch.Generator().manual_seed(21)
    frames, ids, seg, tgts = synthetic_raw_batch(2, 2, generator=gen)
    text_emb, img_emb = fx.extract(frames.to(dev), ids.to(dev), seg.to(dev))
    assert text_emb.shape == (2, 2, 196, 768) and img_emb.shape == (2, 16, 768)
    with torch.no_grad():
        text_ref, img_ref = _oracle_features(pv, pt, frames, ids, seg)
    scale_t, scale_i = float(text_ref.abs().max()), float(img_ref.abs().max())
    assert (text_emb.cpu() - text_ref).abs().max().item() < 1e-3 * max(1.0, scale_t)
    assert (img_emb.cpu() - img_ref).abs().max().item() < 1e-3 * max(1.0, scale_i)
    Pa = O.seeded_params(O.head_param_spec("actor"), seed=7)
    Pc = O.seeded_params(O.head_param_spec("critic"), seed=8)
    model = ppo.ActorCritic(_head_args(dev), None)
    model.actor.load_state_dict(Pa, strict=True)
    model.critic.load_state_dict(Pc, strict=True)
    model = model.to(dev).eval()
    with torch.no_grad():
        loss, logits = model.actor(text_emb, img_emb, tgts.to(dev))
        state = torch.tensor([[1, 0], [0, 1]], device=dev)
        value = model.critic(text_emb, img_emb, None, state)
        img_rep = img_ref.unsqueeze(1).repeat(1, 2, 1, 1)
        loss_ref, logits_ref = O.actor_forward(Pa, text_ref, img_rep, tgts)
        value_ref = O.critic_forward(Pc, text_ref, img_rep, state.cpu())
    assert (logits.cpu() - logits_ref).abs().max().item() < 1e-3
    assert abs(float(loss) - float(loss_ref)) < 1e-3
    assert (value.cpu() - value_ref).abs().max().item() < 1e-3


def test_full_batch_pipeline_step_properties(dev):
    """B = 32 items (512 frames, 64 tag sequences) -> features -> rollout -> PPO update, the composed step bench.py times:
    features of an item do not depend on its batch-mates (vs a 2-item extraction), padded token positions do not influence
    the visible ones, the step is reproducible bit for bit from its seeds, and it moves the weights."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch
    torch.manual_seed(3)
    fx = FeatureExtractor()
    fx.init_normal()
    fx = fx.to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(5)
    frames, ids, seg, tgts = synthetic_raw_batch(32, 2, device=dev, generator=gen)
    text_emb, img_emb = fx.extract(frames, ids, seg)
    assert text_emb.shape == (32, 2, 196, 768) and img_emb.shape == (32, 16, 768)
    assert torch.isfinite(text_emb).all() and torch.isfinite(img_emb).all()
    sub = [5, 17]
    t2, i2 = fx.extract(frames[sub], ids[sub], seg[sub])
    assert (t2 - text_emb[sub]).abs().max().item() < 1e-4 * float(text_emb.abs().max())
    assert (i2 - img_emb[sub]).abs().max().item() < 1e-4 * float(img_emb.abs().max())
    # ids behind the visible prefix are padding: changing them must not change the visible positions' features
    ids2 = ids.clone()
    ids2[seg == 0] = 7
    t3, _ = fx.extract(frames[:2], ids2[:2], seg[:2])
    vis = seg[:2].bool()
    assert (t3[vis] - text_emb[:2][vis]).abs().max().item() < 1e-4 * float(text_emb.abs().max())

    def run():
        torch.manual_seed(11)
        args = _head_args(dev)
        model = ppo.ActorCritic(args, None)
        reward = ppo.Reward(args, None)
        for m in (model.actor, model.critic, reward):
            ppo._init_normal(m)
        model, rew = model.to(dev), reward.to(dev).eval()
        opt, copt, sch, csch = ppo.build_optimizer(args, model)
        sch.step(), csch.step()
        runtime.set_dropout_seed(99)
        model.eval()
        te, ie = fx.extract(frames, ids, seg)
        rec = ppo.rollout_step(model, rew, te, ie, tgts)
        model.train()
        w0 = model.actor.head.weight.detach().clone()
        out = ppo.train_model(args, model, opt, copt, sch, csch, [rec], 1)
        return out, model.actor.head.weight.detach().clone(), w0, model.critic.xitt[1][0].weight.detach().clone()

    a, wa, w0, ca = run()
    b, wb, _, cb = run()
    assert all(v == v for v in a) and a == b
    assert torch.equal(wa, wb) and torch.equal(ca, cb) and not torch.equal(wa, w0)


def test_stage1_pointwise_step_from_raw_inputs_at_reference_shape(dev):
    """BASELINE config 2 literally: ViT-B/16 + RoBERTa-base in front of finetune/pointwise.py's Classifier at batch 32 x 20
    tags (pointwise.sh:28): features -> train step; the loss equals the oracle's SmoothL1 on the step's own logits, the
    scheduler / optimizer advance, and a batch of 2 items of the same data gives the same per-item logits."""
    from lr2ppo_amd.finetune import pointwise
    from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch
    torch.manual_seed(4)
    fx = FeatureExtractor()
    fx.init_normal()
    fx = fx.to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(6)
    frames, ids, seg, tgts = synthetic_raw_batch(32, 20, device=dev, generator=gen)
    text_emb, img_emb = fx.extract(frames, ids, seg)
    assert text_emb.shape == (32, 20, 196, 768)
    args = _head_args(dev, train_steps=100, batch_size=32)
    model = pointwise.Classifier(args, None)
    from lr2ppo_amd.finetune import ppo
    ppo._init_normal(model)
    model = model.to(dev)
    model.eval()
    with torch.no_grad():
        logits_all = model(text_emb, img_emb, None).view(32, 20)
        logits_two = model(text_emb[3:5].contiguous(), img_emb[3:5].contiguous(), None).view(2, 20)
    assert (logits_all[3:5] - logits_two).abs().max().item() < 1e-4 * max(1.0, float(logits_all.abs().max()))
    want = O.smooth_l1(logits_all.cpu().view(-1), tgts.cpu().view(-1).float())
    with torch.no_grad():
        loss, _ = model(text_emb, img_emb, tgts)
    assert abs(float(loss) - float(want)) < 1e-5 * max(1.0, float(want))
    opt, sch = pointwise.build_optimizer(args, model)
    sch.step()
    w0 = model.head.weight.detach().clone()
    model.train()
    loss1 = pointwise.train_model(args, model, opt, sch, text_emb, img_emb, tgts)
    assert float(loss1) == float(loss1) and not torch.equal(w0, model.head.weight.detach())

