"""The PPO step captured in a HIP graph (finetune/ppo.py::GraphedPPOStep) gives the bits of the eager step: dropout seeds and
learning rates are read from device memory (lr2_epilogue.drop_seed_dev / adam_lr_dev, lr2_layernorm_bwd, lr2_adamw_multi,
lr2_step_scalars_store), so one captured graph follows the host's dropout counter and the schedulers."""
import argparse
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(dev, mode="reg"):
    return argparse.Namespace(mode=mode, labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw",
                              scheduler="linear", learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=41, warmup=0.1,
                              device=dev)


def _build(ppo, args, dev, state=None):
    model, reward = ppo.ActorCritic(args, None), ppo.Reward(args, None)
    if state is None:
        torch.manual_seed(5)
        for m in (model, reward):
            ppo._init_normal(m)
    else:
        model.load_state_dict(state[0])
        reward.load_state_dict(state[1])
    model, reward = model.to(dev), reward.to(dev).eval()
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    model.actor.bind_grads(), model.critic.bind_grads()
    for _ in range(3):                  # leave the lr-0 first cycle
        sch.step(), csch.step()
    return model, reward, opt, copt, sch, csch


def test_device_seed_and_device_lr_give_the_by_value_bits(dev):
    """Kernel level: a dropout GEMM epilogue / LayerNorm backward with the seed split between the argument and device memory, and
    an AdamW launch with the rate in device memory, against the by-value calls."""
    from lr2ppo_amd import ops
    g = torch.Generator(device=dev).manual_seed(1)
    M, N, K = 256, 512, 128
    a, b = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    sc = ops.StepScalars(dev)
    slot = sc.new_lr()
    sc.store((77 << 24) + 5, [3e-4])
    torch.cuda.synchronize()
    assert int(sc.seed.item()) == (77 << 24) + 5 and float(sc.lr_tensor(slot).item()) == float(torch.tensor(3e-4, dtype=torch.float32))
    o1, o2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(a, b, o1, M, N, K, drop=ops.Drop(0.1, (77 << 24) + 5 + 2, 4))
    ops.gemm(a, b, o2, M, N, K, drop=ops.Drop(0.1, 2, 4, seed_dev=sc.seed))
    assert torch.equal(o1, o2) and float((o1 == 0).float().mean()) > 0.05
    # LayerNorm backward with masked planes
    rows, D = 300, 768
    dy, x = torch.randn(rows, D, device=dev, generator=g), torch.randn(rows, D, device=dev, generator=g)
    gamma = torch.randn(D, device=dev, generator=g)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = (var + 1e-6).rsqrt()
    outs = []
    for drop in (ops.Drop(0.1, (77 << 24) + 5 + 1, 9), ops.Drop(0.1, 1, 9, seed_dev=sc.seed)):
        dx, pl = torch.empty(rows, D, device=dev), ops.Planes.empty(rows, D, dev)
        dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
        part = torch.empty(ops.LN_BWD_BLOCKS * 2 * D, device=dev)
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, part, dg, db, rows=rows, D=D, dx_planes=pl, drop=drop)
        outs.append((dx, pl.to_float(), dg, db))
    for u, v in zip(*outs):
        assert torch.equal(u, v)
    assert float((outs[0][1] == 0).float().mean()) > 0.05


@pytest.mark.parametrize("mode", ["cls"])       # 'cls' = 'reg' + the class-logit chain (cls_scores_bwd) inside the captured step
def test_graphed_ppo_step_equals_eager_bits_and_follows_the_scheduler(dev, mode):
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    args = _args(dev, mode)
    A = _build(ppo, args, dev)
    state = ({k: v.detach().cpu().clone() for k, v in A[0].state_dict().items()},
             {k: v.detach().cpu().clone() for k, v in A[1].state_dict().items()})
    B = _build(ppo, args, dev, state)
    gen = torch.Generator().manual_seed(23)
    batches = [(torch.randn(4, 2, 196, 768, generator=gen).to(dev), torch.randn(4, 16, 768, generator=gen).to(dev),
                torch.randint(0, 3, (4, 2), generator=gen).to(dev)) for _ in range(6)]

    def eager(text, img, tgts):
        model, reward, opt, copt, sch, csch = A
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tgts)
        model.train()
        m = ppo.update_minibatch(args, model, opt, copt, rec)
        sch.step(), csch.step()
        return m.clone()

    runtime.set_dropout_seed(99)
    ref = [eager(*b) for b in batches]
    end_eager = runtime.peek_drop_seed()
    torch.cuda.synchronize()

    runtime.set_dropout_seed(99)
    model, reward, opt, copt, sch, csch = B
    step = ppo.GraphedPPOStep(args, model, reward, opt, copt)
    got = []
    for b in batches:
        got.append(step(*b).clone())
        sch.step(), csch.step()
    torch.cuda.synchronize()
    assert step.graph is not None and step.draws >= 2
    assert runtime.peek_drop_seed() == end_eager
    for i, (x, y) in enumerate(zip(got, ref)):
        assert torch.equal(x, y), f"metrics of step {i} differ: {x.tolist()} vs {y.tolist()}"
    lrs = {g["lr"] for g in opt.param_groups}
    assert len(lrs) == 1 and 0 < lrs.pop() < 1e-3            # the schedule moved while the graph was replayed
    for (n, p), (_, q) in zip(B[0].named_parameters(), A[0].named_parameters()):
        assert torch.equal(p, q), f"parameter {n} differs after 6 steps"
    for oa, ob in ((A[2], B[2]), (A[3], B[3])):
        for ga, gb in zip(oa.param_groups, ob.param_groups):
            for pa, pb in zip(ga["params"], gb["params"]):
                sa, sb = oa.state[pa], ob.state[pb]
                assert sa["step"] == sb["step"]
                assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])

    # host cost of one replayed step on an empty queue, beside the eager step's
    def host_ms(fn, n=5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(*batches[i % len(batches)])
        dt = (time.perf_counter() - t0) / n * 1e3
        torch.cuda.synchronize()
        return dt
    t_graph = host_ms(lambda *b: step(step.text, step.img, step.tgts))
    t_eager = host_ms(eager)
    print(f"host enqueue per PPO step: graph {t_graph:.3f} ms, eager {t_eager:.3f} ms")
    assert t_graph < t_eager
    step.release()                     # by-value rates again on the optimizers the graph used
    model.eval()
    rec = ppo.rollout_step(model, reward, *batches[0])
    model.train()
    assert torch.isfinite(ppo.update_minibatch(args, model, opt, copt, rec)).all()


def test_graphed_step_of_the_sequence_length_1_heads(dev):
    """finetune/ppo_trad.py's models (no image features, no 2-GB matrix, every gradient through the multi-tensor AdamW) through the
    same GraphedPPOStep: bits of the eager rollout + update over 5 steps with a moving schedule."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo_trad as pt
    from oracle import lr2ppo_oracle as O
    args = argparse.Namespace(mode="reg", labels_num=3, is_master=False, kl_div_loss_weight=0.001, entropy_weight=0.001,
                              value_clip=0.5, optimizer="adamw", scheduler="linear", learning_rate=1e-3, critic_learning_rate=1e-3,
                              train_steps=41, warmup=0.1, device=dev)

    def build():
        model, reward = pt.ActorCritic(args, None), pt.Reward(args, None)
        for mod, kind, seed in ((model.actor, "actor", 37), (model.critic, "critic", 38), (reward, "reward", 39)):
            mod.load_state_dict(O.seeded_params(O.trad_head_param_spec(kind), seed=seed), strict=True)
        model, reward = model.to(dev), reward.to(dev).eval()
        opt, copt, sch, csch = pt.build_optimizer(args, model)
        model.actor.bind_grads(), model.critic.bind_grads()
        for _ in range(3):
            sch.step(), csch.step()
        return model, reward, opt, copt, sch, csch

    gen = torch.Generator().manual_seed(5)
    batches = [(torch.randn(6, 2, 768, generator=gen).to(dev), torch.randint(0, 3, (6, 2), generator=gen).to(dev)) for _ in range(5)]
    runtime.set_dropout_seed(7)
    model, reward, opt, copt, sch, csch = A = build()
    ref = []
    for text, tgts in batches:
        model.eval()
        rec = pt.rollout_step(model, reward, text, None, tgts)
        model.train()
        ref.append(pt.update_minibatch(args, model, opt, copt, rec).clone())
        sch.step(), csch.step()
    runtime.set_dropout_seed(7)
    model, reward, opt, copt, sch, csch = B = build()
    step = ppo_mod().GraphedPPOStep(args, model, reward, opt, copt)
    got = []
    for text, tgts in batches:
        got.append(step(text, None, tgts).clone())
        sch.step(), csch.step()
    torch.cuda.synchronize()
    assert step.graph is not None
    for i, (x, y) in enumerate(zip(got, ref)):
        assert torch.equal(x, y), f"metrics of step {i} differ"
    for (n, p), (_, q) in zip(B[0].named_parameters(), A[0].named_parameters()):
        assert torch.equal(p, q), f"parameter {n} differs"


def ppo_mod():
    from lr2ppo_amd.finetune import ppo
    return ppo
