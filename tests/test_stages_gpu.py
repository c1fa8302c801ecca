"""SURVEY 8(f) rows 1-2 on the GPU: the stage-1 (pointwise) and stage-2 (pairwise reward) training steps of the HIP
path against fixtures frozen from the imported reference (tests/golden/stage{1,2}_step.npz) and against the oracle
with the dropout mask stream pinned."""
import argparse

import pytest
import torch

from conftest import load_golden
from oracle import lr2ppo_oracle as O

pytestmark = pytest.mark.gpu
ARGS = dict(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768)


def _args(dev, **kw):
    return argparse.Namespace(**ARGS, is_master=False, optimizer="adamw", scheduler="linear", learning_rate=1e-3,
                              train_steps=21, warmup=0.1, device=dev, **kw)


def _check_weights(g, named, step, tag):
    for key in [k for k in g if k.startswith(f"w{step}.")]:
        n = key[len(f"w{step}."):]
        got = named[n].detach().flatten()[g["idx." + n].to(named[n].device)].cpu()
        assert (got - g[key]).abs().max() < 2e-6, f"{tag}: {n} after step {step}"


@pytest.mark.parametrize("fuse", [True, False])
def test_stage1_train_model_matches_reference_golden(dev, fuse):
    from lr2ppo_amd.finetune import pointwise as pw
    g = load_golden("stage1_step.npz")
    bs, tags, steps = int(g["bs"]), int(g["tags"]), int(g["steps"])
    args = _args(dev, fuse_fc1_update=fuse)
    model = pw.Classifier(args, None)
    model.load_state_dict(O.seeded_params(O.head_param_spec("actor"), seed=17), strict=True)
    model = model.to(dev).eval()                                   # dropout off, as in the golden run
    opt, sch = pw.build_optimizer(args, model)
    named = dict(model.named_parameters())
    for step in range(steps):
        text, img, tgts = O.seeded_head_inputs(2000 + step, bs, tags)
        assert abs(opt.param_groups[0]["lr"] - float(g[f"lr_{step}"])) < 1e-12
        loss = pw.train_model(args, model, opt, sch, text.to(dev), img.to(dev), tgts.to(dev))
        ref = float(g[f"loss_{step}"])       # the third step's loss is O(100): lr 4.8e-4 on 519 M parameters
        assert loss.dim() == 0 and abs(float(loss) - ref) < 2e-5 * max(1.0, abs(ref)), step
        _check_weights(g, named, step, "stage1")
    text, img, _ = O.seeded_head_inputs(2100, bs, tags)
    with torch.no_grad():
        logits = model(text.to(dev), img.to(dev), None).cpu()
    ref = g["eval_logits"].view(-1)
    assert (logits.view(-1) - ref).abs().max() < 1e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("fuse", [True, False])
def test_stage2_train_model_matches_reference_golden(dev, fuse):
    from lr2ppo_amd.finetune import reward_pair_dataloader as rp
    g = load_golden("stage2_step.npz")
    bs, tags, steps = int(g["bs"]), int(g["tags"]), int(g["steps"])
    args = _args(dev, fuse_fc1_update=fuse)
    args.mode = "cls"                                              # what reward_pair_dataloader.sh passes; never read
    model = rp.Classifier(args, None)
    model.load_state_dict(O.seeded_params(O.head_param_spec("reward"), seed=19), strict=True)
    model = model.to(dev).eval()
    opt, sch = rp.build_optimizer(args, model)
    named = dict(model.named_parameters())
    for step in range(steps):
        text, img, tgts = O.seeded_head_inputs(3000 + step, bs, tags)
        chosen, reject = g[f"chosen_{step}"].to(dev), g[f"reject_{step}"].to(dev)
        assert abs(opt.param_groups[0]["lr"] - float(g[f"lr_{step}"])) < 1e-12
        loss, acc = rp.train_model(args, model, opt, sch, text.to(dev), img.to(dev), tgts.to(dev), chosen, reject)
        ref = float(g[f"loss_{step}"])
        assert abs(float(loss) - ref) < 2e-5 * max(1.0, abs(ref)), step
        assert float(acc) == float(g[f"acc_{step}"]), step
        _check_weights(g, named, step, "stage2")
    # the two-forward API of the reference gives the same scores as the fused [chosen ; reject] batch
    text, img, tgts = O.seeded_head_inputs(3100, bs, tags)
    chosen, reject = g["chosen_0"].to(dev), g["reject_0"].to(dev)
    with torch.no_grad():
        c = model(text.to(dev), img.to(dev), tgts.to(dev), chosen)
        r = model(text.to(dev), img.to(dev), tgts.to(dev), reject)
        both = model.engine_forward(torch.cat([text, text]).to(dev), torch.cat([img, img]).to(dev),
                                    torch.cat([chosen, reject]), save=False)
    assert torch.allclose(torch.cat([c, r]), both.view(-1), atol=2e-5)


def test_stage2_train_step_with_dropout_matches_oracle(dev):
    """Train mode (dropout 0.1 at every XiT site, trunk and tail) with the mask stream pinned: loss, accuracy and
    parameter gradients of one step against the oracle's autograd over the same [chosen ; reject] batch."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import reward_pair_dataloader as rp
    bs, tags = 2, 2
    text, img, _ = O.seeded_head_inputs(77, bs, tags)
    chosen = torch.tensor([[0, 1, 0, 1], [1, 0, 0, 1]])
    reject = torch.tensor([[0, 1, 1, 0], [1, 0, 1, 0]])
    P = O.seeded_params(O.head_param_spec("reward"), seed=23)
    with torch.no_grad():                       # spread the scores so that some hinges are active and some are not
        P["head.weight"] *= 40.0
    args = _args(dev, fuse_fc1_update=False)
    model = rp.Classifier(args, None)
    model.load_state_dict(P, strict=True)
    model = model.to(dev).train()
    opt, sch = rp.build_optimizer(args, model)  # lr 0 at the first step: weights stay, gradients are what we read
    runtime.set_dropout_seed(4242, calls=3)
    seed = runtime.peek_drop_seed()
    loss, acc = rp.train_model(args, model, opt, sch, text.to(dev), img.to(dev), torch.zeros(bs, tags), chosen.to(dev),
                               reject.to(dev))
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_loss, ref_acc, _ = O.stage2_loss(Pg, (text, img, chosen, reject), drop={"p": 0.1, "seed": seed, "site_base": 0})
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss.detach())) < 1e-3 * max(1.0, abs(float(ref_loss.detach())))
    assert float(acc) == float(ref_acc)
    G = model.grad_buffers()
    for n in ["text_proj.fc1.weight", "img_proj.fc2.bias", "xit.0.0.0.fn.1.queries.weight", "xit.1.0.weight",
              "out_layer.fc1.weight", "out_layer.fc2.weight", "pos_emb.weight", "xitt.0.0.0.fn.1.keys.weight",
              "xitt.0.0.1.fn.1.3.weight", "xitt.1.0.bias", "head.weight", "head.bias"]:
        ref_g = Pg[n].grad
        err = (G[n].cpu().view_as(ref_g) - ref_g).abs().max().item()
        assert err < 1e-6 + 2e-3 * ref_g.abs().max().item(), f"grad {n}: {err} vs scale {ref_g.abs().max().item()}"


def test_pair_hinge_kernel(dev):
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(5)
    for bs in (1, 7, 64, 1500):
        s = (torch.randn(2 * bs, generator=g) * 2).to(dev)
        out, ds = torch.empty(2, device=dev), torch.empty(2 * bs, device=dev)
        ops.pair_hinge(s, out, ds, bs=bs, margin=1.0)
        sc = s.cpu().double().requires_grad_(True)
        loss, acc = O.pair_hinge(sc[:bs], sc[bs:])
        loss.backward()
        assert abs(float(out[0]) - float(loss)) < 1e-5 and abs(float(out[1]) - float(acc)) < 1e-6
        assert torch.allclose(ds.cpu().double(), sc.grad, atol=1e-7)


def test_trad_classifier_matches_reference_golden(dev):
    """BASELINE configs[0] (finetune/pointwise_trad.py, 2 queries x 20 documents, 768-d features): three train steps and an
    inference pass of the seq-len-1 head against the imported reference."""
    from lr2ppo_amd.finetune import pointwise_trad as pt
    g = load_golden("trad_step.npz")
    steps = int(g["steps"])
    args = argparse.Namespace(mode="reg", labels_num=3, optimizer="adamw", scheduler="linear", learning_rate=1e-3, train_steps=21,
                              warmup=0.1)
    model = pt.Classifier(args, None)
    model.load_state_dict(O.seeded_params(O.trad_param_spec(), seed=27), strict=True)
    model = model.to(dev).eval()
    opt, sch = pt.build_optimizer(args, model)
    named = dict(model.named_parameters())
    for step in range(steps):
        assert abs(opt.param_groups[0]["lr"] - float(g[f"lr_{step}"])) < 1e-12
        loss = pt.train_model(args, model, opt, sch, g[f"feats_{step}"].to(dev), None, g[f"tgts_{step}"].to(dev))
        ref = float(g[f"loss_{step}"])
        assert abs(float(loss) - ref) < 2e-5 * max(1.0, abs(ref)), step
        _check_weights(g, named, step, "trad")
    with torch.no_grad():
        logits = model(g["feats_0"].to(dev), None, None).cpu()
    ref = g["eval_logits"]
    assert logits.shape == ref.shape and (logits - ref).abs().max() < 1e-3 * max(1.0, float(ref.abs().max()))


def test_trad_classifier_dropout_and_autograd_match_oracle(dev):
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import pointwise_trad as pt
    g = load_golden("trad_step.npz")
    P = O.seeded_params(O.trad_param_spec(), seed=27)
    model = pt.Classifier(argparse.Namespace(mode="reg", labels_num=3), None)
    model.load_state_dict(P, strict=True)
    model = model.to(dev).train()
    runtime.set_dropout_seed(909, calls=1)
    seed = runtime.peek_drop_seed()
    feats, tgts = g["feats_1"], g["tgts_1"]
    loss, logits = model(feats.to(dev), None, tgts.to(dev))          # drop-in autograd path
    loss.backward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_loss, ref_logits = O.trad_forward(Pg, feats, tgts, drop={"p": 0.1, "seed": seed, "site_base": 0})
    ref_loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-4 and (logits.cpu() - ref_logits.detach()).abs().max() < 1e-3
    for n, p in model.named_parameters():
        ref = Pg[n].grad
        err = (p.grad.cpu() - ref).abs().max().item()
        assert err < 1e-6 + 2e-3 * ref.abs().max().item(), f"grad {n}: {err} vs {ref.abs().max().item()}"


def test_stage1_cls_mode_train_step_and_evaluate_match_oracle(dev):
    """finetune/pointwise.py with --mode cls (:228-232, :342-345): one train_model step at lr 0 (first scheduler step) --
    NLL loss and gradients against the oracle's autograd with the dropout masks pinned --, then evaluate's class-weighted raw
    logits.  (The oracle's cls head is pinned to the reference by tests/golden/cls_step.npz.)"""
    from lr2ppo_amd import ops, runtime
    from lr2ppo_amd.finetune import pointwise as pw
    bs, tags = 2, 3
    P = O.seeded_params(O.head_param_spec("actor", n_out=3), seed=51)
    with torch.no_grad():
        P["head.weight"] *= 30.0
    text, img, _ = O.seeded_head_inputs(5100, bs, tags)
    tgts = torch.tensor([[0, 2, 1], [1, 1, 0]])
    args = _args(dev, fuse_fc1_update=False)
    args.mode = "cls"
    model = pw.Classifier(args, None)
    model.load_state_dict(P, strict=True)
    model = model.to(dev).train()
    assert model.n_out == 3
    opt, sch = pw.build_optimizer(args, model)
    runtime.set_dropout_seed(777, calls=2)
    seed = runtime.peek_drop_seed()
    loss = pw.train_model(args, model, opt, sch, text.to(dev), img.to(dev), tgts.to(dev))
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_loss, ref_logits = O.actor_forward_cls(Pg, text, img, tgts, drop={"p": 0.1, "seed": seed, "site_base": 0})
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss.detach())) < 1e-4 * max(1.0, abs(float(ref_loss.detach())))
    G = model.grad_buffers()
    for n in ["text_proj.fc1.weight", "img_proj.fc2.bias", "xit.0.0.0.fn.1.queries.weight", "out_layer.fc1.weight",
              "out_layer.fc2.weight", "head.weight", "head.bias"]:
        ref_g = Pg[n].grad
        err = (G[n].cpu().view_as(ref_g) - ref_g).abs().max().item()
        assert err < 1e-6 + 2e-3 * ref_g.abs().max().item(), f"grad {n}: {err} vs scale {ref_g.abs().max().item()}"
    # evaluate's score (pointwise.py:342-345): 0 * z0 + 1 * z1 + 2 * z2 on the raw logits, no softmax
    model.eval()
    with torch.no_grad():
        logits = model.engine_forward(text.to(dev), img.to(dev), save=False)
        score = ops.cls_scores(logits, None, torch.empty(bs * tags, device=dev), rows=bs * tags, C=3, softmax=False).cpu()
        want = O.cls_action_scores(O.actor_forward_cls(P, text, img), bs, tags, softmax=False).view(-1)
    assert (score - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))


def test_ppo_trad_two_cycles_match_reference_golden(dev):
    """finetune/ppo_trad.py (stage 3 at sequence length 1, 3 queries x 2 documents): rollout tensors of four
    minibatches, the 10 metrics of two train_model cycles (lr 0, then one warm-up step in) and sampled post-step weights of
    actor and critic against the imported reference; the modules' state_dict keys are the reference's."""
    from lr2ppo_amd.finetune import ppo_trad as pt
    g = load_golden("ppo_trad_step.npz")
    bs, tags = int(g["bs"]), int(g["tags"])
    args = argparse.Namespace(mode="reg", labels_num=3, is_master=False, kl_div_loss_weight=0.001, entropy_weight=0.001,
                              value_clip=0.5, optimizer="adamw", scheduler="linear", learning_rate=1e-3, critic_learning_rate=1e-3,
                              train_steps=41, warmup=0.1, device=dev)
    model = pt.ActorCritic(args, None)
    reward = pt.Reward(args, None)
    for mod, kind, seed in ((model.actor, "actor", 37), (model.critic, "critic", 38), (reward, "reward", 39)):
        spec = O.trad_head_param_spec(kind)
        assert [n for n, _ in mod.named_parameters()] == [n for n, _ in spec]
        mod.load_state_dict(O.seeded_params(spec, seed=seed), strict=True)
    model, reward = model.to(dev).eval(), reward.to(dev).eval()
    opt, copt, sch, csch = pt.build_optimizer(args, model)
    named = dict(model.named_parameters())
    for cycle in range(2):
        lrs = g[f"lr_{cycle}"]
        assert abs(opt.param_groups[0]["lr"] - float(lrs[0])) < 1e-12 and abs(copt.param_groups[0]["lr"] - float(lrs[1])) < 1e-12
        memories = []
        for mb in range(2):
            k = f"c{cycle}_mb{mb}_"
            rec = pt.rollout_step(model, reward, g[k + "text"].to(dev), None, g[k + "tgts"].to(dev))
            for got, key in ((rec[2], "scores"), (rec[4], "value"), (rec[3], "reward")):
                assert (got.cpu() - g[k + key]).abs().max() < 1e-4, (cycle, mb, key)
            assert torch.equal(rec[1].cpu(), g[k + "next_state"])
            memories.append(rec)
        out = pt.train_model(args, model, opt, copt, sch, csch, memories, 1)
        for i, (a, b) in enumerate(zip(out, g[f"metrics_{cycle}"].tolist())):
            assert abs(a - b) < 1e-4, f"cycle {cycle} metric {i}: {a} vs {b}"
        for key in [k for k in g if k.startswith(f"w{cycle}.")]:
            n = key[len(f"w{cycle}."):]
            w = named[n].detach().flatten()[g["idx." + n].to(dev)].cpu()
            assert (w - g[key]).abs().max() < 2e-6, f"weights {n} after cycle {cycle}"
    assert out[2] > 0.0          # KL becomes non-zero only after the first real update


def test_ppo_trad_train_mode_gradients_match_oracle(dev):
    """Dropout on: critic value and every sampled gradient of the seq-len-1 critic (trunk + pos_emb + second XiT + head)
    against the oracle's autograd with the masks pinned; evaluate() over SyntheticLTR runs and returns an NDCG in [0, 1]."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo_trad as pt
    from torch.utils.data import DataLoader
    bs, tags = 3, 3
    args = argparse.Namespace(mode="reg", labels_num=3, is_master=True, device=dev)
    spec = O.trad_head_param_spec("critic")
    P = O.seeded_params(spec, seed=71)
    with torch.no_grad():
        P["head.weight"] *= 30.0
    critic = pt.Critic(args, None)
    critic.load_state_dict(P, strict=True)
    critic = critic.to(dev).train()
    g = torch.Generator().manual_seed(72)
    text = torch.randn(bs, tags, 768, generator=g)
    index = torch.tensor([[2, 0, 1], [1, 1, 0], [0, 2, 2]])
    runtime.set_dropout_seed(4343, calls=1)
    seed = runtime.peek_drop_seed()
    value = critic.engine_forward(text.to(dev), None, index.to(dev), save=True)
    w = torch.tensor([0.7, -1.1, 0.4])
    critic.engine_backward(w.to(dev))
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = O.trad_critic_forward(Pg, text, index, drop={"p": 0.1, "seed": seed, "site_base": 0})
    (ref * w).sum().backward()
    assert (value.cpu() - ref.detach()).abs().max() < 1e-3 * max(1.0, float(ref.detach().abs().max()))
    G = critic.grad_buffers()
    for n in ["pos_emb.weight", "xit.0.0.0.fn.1.queries.weight", "xit.1.0.weight", "xitt.0.0.0.fn.1.keys.weight",
              "xitt.0.0.1.fn.1.3.weight", "out_layer.fc1.weight", "out_layer.fc2.bias", "head.weight", "head.bias"]:
        ref_g = Pg[n].grad
        err = (G[n].cpu().view_as(ref_g) - ref_g).abs().max().item()
        assert err < 1e-6 + 2e-3 * ref_g.abs().max().item(), f"grad {n}: {err} vs scale {ref_g.abs().max().item()}"
    model = pt.ActorCritic(args, None).to(dev)
    args.model = model
    ndcg = pt.evaluate(args, DataLoader(pt.SyntheticLTR(6, docs=20), batch_size=1))
    assert 0.0 <= float(ndcg) <= 1.0


def test_reward_trad_three_steps_match_reference_golden(dev):
    """finetune/reward_trad.py (stage 2 at sequence length 1, hinge margin 0.01): loss, accuracy, lr and sampled weights of
    three train_model steps against the imported reference; evaluate() over SyntheticTradPairs returns an accuracy."""
    from torch.utils.data import DataLoader
    from lr2ppo_amd.finetune import reward_trad as rt
    g = load_golden("reward_trad_step.npz")
    steps = int(g["steps"])
    args = argparse.Namespace(mode="reg", labels_num=3, is_master=True, optimizer="adamw", scheduler="linear", learning_rate=1e-3,
                              train_steps=21, warmup=0.1, device=dev)
    model = rt.Classifier(args, None)
    P = O.seeded_params(O.trad_head_param_spec("reward"), seed=43)
    P["head.weight"] = P["head.weight"] * 25.0
    model.load_state_dict(P, strict=True)
    model = model.to(dev).eval()
    opt, sch = rt.build_optimizer(args, model)
    named = dict(model.named_parameters())
    for step in range(steps):
        assert abs(opt.param_groups[0]["lr"] - float(g[f"lr_{step}"])) < 1e-12
        loss, acc = rt.train_model(args, model, opt, sch, g[f"feats_{step}"].to(dev), None, None, g[f"chosen_{step}"].to(dev),
                                   g[f"reject_{step}"].to(dev))
        ref = float(g[f"loss_{step}"])
        assert abs(float(loss) - ref) < 1e-4 * max(1.0, abs(ref)), (step, float(loss), ref)
        assert abs(float(acc) - float(g[f"acc_{step}"])) < 1e-6, step
        _check_weights(g, named, step, "reward_trad")
    acc = rt.evaluate(args, model, DataLoader(rt.SyntheticTradPairs(8, docs=6), batch_size=4))
    assert 0.0 <= acc <= 1.0


def test_pointwise_2data_trad_matches_reference_golden(dev):
    """finetune/pointwise_2data_trad.py (BASELINE configs[0]'s 136-dim MLP ranker; 46-dim for MQ2008): four train steps
    alternating the two feature widths and an inference pass per width against the imported reference -- including the rule
    that the projection a batch does not use is left untouched by AdamW (no moments, no weight decay)."""
    from lr2ppo_amd.finetune import pointwise_2data_trad as p2
    g = load_golden("trad2_step.npz")
    steps = int(g["steps"])
    args = argparse.Namespace(mode="reg", labels_num=3, optimizer="adamw", scheduler="linear", learning_rate=1e-3, train_steps=21,
                              warmup=0.1)
    model = p2.Classifier(args, None)
    spec = O.trad2_param_spec()
    assert [(n, tuple(p.shape)) for n, p in model.named_parameters()] == spec
    model.load_state_dict(O.seeded_params(spec, seed=47), strict=True)
    model = model.to(dev).eval()
    opt, sch = p2.build_optimizer(args, model)
    named = dict(model.named_parameters())
    for step in range(steps):
        assert abs(opt.param_groups[0]["lr"] - float(g[f"lr_{step}"])) < 1e-12
        loss = p2.train_model(args, model, opt, sch, g[f"feats_{step}"].to(dev), None, g[f"tgts_{step}"].to(dev))
        ref = float(g[f"loss_{step}"])
        assert abs(float(loss) - ref) < 2e-5 * max(1.0, abs(ref)), (step, float(loss), ref)
        _check_weights(g, named, step, "trad2")
    with torch.no_grad():
        for width, k in ((136, "feats_0"), (46, "feats_1")):
            logits = model(g[k].to(dev), None, None).cpu()
            ref = g[f"eval_logits_{width}"]
            assert logits.shape == ref.shape and (logits - ref).abs().max() < 1e-3 * max(1.0, float(ref.abs().max())), width


def test_pointwise_2data_trad_dropout_gradients_match_oracle(dev):
    """Train mode (dropout on, masks pinned): loss and gradients of the used projection, the XiT block and the head against
    the oracle's autograd, for both feature widths; the unused projection's .grad is None."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import pointwise_2data_trad as p2
    args = argparse.Namespace(mode="reg", labels_num=3, optimizer="adamw", scheduler="linear", learning_rate=1e-3, train_steps=21,
                              warmup=0.1)
    spec = O.trad2_param_spec()
    P = O.seeded_params(spec, seed=49)
    model = p2.Classifier(args, None)
    model.load_state_dict(P, strict=True)
    model = model.to(dev).train()
    opt, sch = p2.build_optimizer(args, model)                       # lr 0 at the first step: the weights stay
    gen = torch.Generator().manual_seed(50)
    for width, used, unused in ((136, "text_proj3", "text_proj"), (46, "text_proj", "text_proj3")):
        feats = torch.randn(3, 7, width, generator=gen)
        tgts = torch.randint(0, 3, (3, 7), generator=gen).float()
        runtime.set_dropout_seed(5151 + width, calls=1)
        seed = runtime.peek_drop_seed()
        opt2, sch2 = p2.build_optimizer(args, model)
        loss = p2.train_model(args, model, opt2, sch2, feats.to(dev), None, tgts.to(dev))
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        ref_loss, _ = O.trad2_forward(Pg, feats, tgts, drop={"p": 0.1, "seed": seed, "site_base": 0})
        ref_loss.backward()
        assert abs(float(loss) - float(ref_loss.detach())) < 1e-4 * max(1.0, abs(float(ref_loss.detach())))
        named = dict(model.named_parameters())
        assert named[f"{unused}.fc1.weight"].grad is None and Pg[f"{unused}.fc1.weight"].grad is None
        for n in [f"{used}.fc1.weight", f"{used}.fc1.bias", f"{used}.fc2.weight", "xit.0.0.0.fn.1.keys.weight", "xit.1.0.weight",
                  "out_layer.fc1.weight", "head.weight"]:
            ref_g = Pg[n].grad
            err = (named[n].grad.cpu() - ref_g).abs().max().item()
            assert err < 1e-6 + 2e-3 * ref_g.abs().max().item(), f"width {width} grad {n}: {err} vs {ref_g.abs().max().item()}"


def test_pointwise_2data_trad_autograd_dropin_leaves_unused_projection_without_gradient(dev):
    """loss.backward() through the nn.Module path: the projection the batch did not use keeps .grad = None (as upstream, where
    it is not part of the graph), the used one and the head receive gradients equal to the engine path's."""
    from lr2ppo_amd.finetune import pointwise_2data_trad as p2
    args = argparse.Namespace(mode="reg", labels_num=3)
    model = p2.Classifier(args, None)
    model.load_state_dict(O.seeded_params(O.trad2_param_spec(), seed=53), strict=True)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(54)
    feats, tgts = torch.randn(2, 5, 136, generator=g).to(dev), torch.randint(0, 3, (2, 5), generator=g).float().to(dev)
    loss, logits = model(feats, None, tgts)
    loss.backward()
    named = dict(model.named_parameters())
    assert named["text_proj.fc1.weight"].grad is None and named["text_proj.fc2.bias"].grad is None
    assert named["text_proj3.fc1.weight"].grad is not None and named["text_proj3.fc1.weight"].grad.abs().sum() > 0
    Pg = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in named.items()}
    ref_loss, _ = O.trad2_forward(Pg, feats.cpu(), tgts.cpu())
    ref_loss.backward()
    for n in ("text_proj3.fc1.weight", "xit.1.0.weight", "head.weight"):
        ref_g = Pg[n].grad
        assert (named[n].grad.cpu() - ref_g).abs().max().item() < 1e-6 + 2e-3 * ref_g.abs().max().item(), n
