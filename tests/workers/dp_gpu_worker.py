"""2-rank rehearsal of the data-parallel PPO update on ONE GPU (gloo; both ranks use cuda:0).  Launched by
tests/test_dp_gpu.py through torch.distributed.run.  Each rank rolls out and updates on its own batch; afterwards the
replicas must be bit-identical (same averaged gradients everywhere) and must differ from a replica trained without
the exchange."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def _rank_case(r, dev):
    """Data and upstream gradient of rank r (both ranks can rebuild either rank's case)."""
    g = torch.Generator().manual_seed(900 + r)
    text = torch.randn(2, 2, 196, 768, generator=g).to(dev)
    img = torch.randn(2, 16, 768, generator=g).to(dev)
    return text, img, torch.randn(4, generator=g).to(dev), torch.randn(2, generator=g).to(dev)


def check_exchanged_gradient_is_the_rank_mean(args, model, dp, rank, world, dev):
    """The data-parallel gradient (factor all-gather + K = N*world wgrad with alpha = 1/world for out_layer.fc1.weight, one
    all-reduce for the rest) against the MEAN of the two ranks' gradients, each computed independently with no exchange at
    all (world-1 path) on BOTH ranks' data -- distinct data per rank, so a wrong alpha, a hi/lo plane mix-up, a rank-order
    or bucket-split bug cannot cancel.  Then the fused form: exp_avg of out_layer.fc1.weight after one fused AdamW step
    must be (1 - beta1) x that mean gradient."""
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    from lr2ppo_amd.tencentpretrain.utils.optimizers import AdamW
    local = ppo._DataParallel()
    local.world = 1                                           # no exchange: plain local backward
    model.train()
    for head, is_critic in ((model.actor, False), (model.critic, True)):
        state = torch.tensor([[1, 0], [0, 1]], device=dev)
        per_rank = []
        for r in range(world):
            text, img, wa, wc = _rank_case(r, dev)
            runtime.set_dropout_seed(4000 + r)
            if is_critic:
                head.engine_forward(text, img, state, save=True)
                head.engine_backward(wc, local)
            else:
                head.engine_forward(text, img, save=True)
                head.engine_backward(wa, local)
            per_rank.append(head._flat_grad.clone())
        want = torch.stack(per_rank).mean(0)
        text, img, wa, wc = _rank_case(rank, dev)
        runtime.set_dropout_seed(4000 + rank)
        if is_critic:
            head.engine_forward(text, img, state, save=True)
            head.engine_backward(wc, dp)
        else:
            head.engine_forward(text, img, save=True)
            head.engine_backward(wa, dp)
        dp.reduce(head)
        got = head._flat_grad
        for name, gbuf in head.grad_buffers().items():
            off = gbuf.data_ptr() - got.data_ptr()
            w = want[off // 4: off // 4 + gbuf.numel()].view_as(gbuf)
            scale = float(w.abs().max())
            err = float((gbuf - w).abs().max())
            assert err <= 1e-9 + 2e-5 * scale, f"rank {rank}: DP gradient of {name} is not the rank mean: {err} vs scale {scale}"
        # the two ranks' local gradients really differ (else the mean proves nothing)
        assert float((per_rank[0] - per_rank[1]).abs().max()) > 1e-3 * float(want.abs().max())
        # fused out_layer.fc1 update under DP: exp_avg == (1 - beta1) * mean gradient, bit-identical on both ranks
        fc1 = head.out_layer.fc1.weight
        opt = AdamW([{"params": [fc1], "weight_decay": 0.01}], lr=1e-4, correct_bias=False)
        saved = fc1.detach().clone()
        runtime.set_dropout_seed(4000 + rank)
        if is_critic:
            head.engine_forward(text, img, state, save=True)
            head.engine_backward(wc, dp, fc1_update=opt.external_update(fc1))
        else:
            head.engine_forward(text, img, save=True)
            head.engine_backward(wa, dp, fc1_update=opt.external_update(fc1))
        m = opt.state[fc1]["exp_avg"]
        gfc1 = head.grad_buffers()["out_layer.fc1.weight"]
        wfc1 = want[(gfc1.data_ptr() - got.data_ptr()) // 4:][:gfc1.numel()].view_as(gfc1)
        err = float((m - 0.1 * wfc1).abs().max())
        assert err <= 2e-6 * float(wfc1.abs().max()) + 1e-12, f"fused DP update: exp_avg off by {err}"
        both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, torch.stack([m.double().sum(), fc1.detach().double().sum()]).cpu())
        assert torch.equal(both[0], both[1]), "fused DP update differs between ranks"
        with torch.no_grad():
            fc1.copy_(saved)
    if rank == 0:
        print("DP_GRADIENT_IS_RANK_MEAN_OK")


def check_global_rank_loss(args, dp, rank, world, dev):
    """RankLoss across ranks: with the three statistics all-reduced, this rank's d(loss)/d(scores) equals `world` x the
    gradient of the single-rank loss over the concatenated batch, restricted to its own items."""
    from lr2ppo_amd import ops
    B, T = 3, 2
    g = torch.Generator().manual_seed(77)
    n = B * world
    scores, old = torch.randn(n, T, generator=g) * 0.02, torch.randn(n, T, generator=g) * 0.02
    rewards, oldv, val = torch.randn(n, generator=g), torch.randn(n, generator=g), torch.randn(n, generator=g)
    ns = torch.stack([torch.tensor([0, 1, 0, 1]) if i % 2 else torch.tensor([0, 1, 1, 0]) for i in range(n)])
    kw = dict(kl_w=0.001, ent_w=0.001, value_clip=0.5)

    def run(sl, nb, **extra):
        scal, per = torch.empty(4, device=dev), torch.empty(4, nb, device=dev)
        ds, dv = torch.empty(nb, T, device=dev), torch.empty(nb, device=dev)
        ops.ppo_loss(scores[sl].contiguous().to(dev), old[sl].contiguous().to(dev), rewards[sl].contiguous().to(dev),
                     oldv[sl].contiguous().to(dev), val[sl].contiguous().to(dev), ns[sl].contiguous().to(dev), scal, per, ds, dv,
                     B=nb, T=T, **kw, **extra)
        return scal, ds, dv

    big_scal, big_ds, big_dv = run(slice(0, n), n)
    mine = slice(rank * B, (rank + 1) * B)
    stats = torch.empty(3, device=dev)
    ops.ppo_loss(scores[mine].contiguous().to(dev), old[mine].contiguous().to(dev), rewards[mine].contiguous().to(dev),
                 oldv[mine].contiguous().to(dev), val[mine].contiguous().to(dev), ns[mine].contiguous().to(dev), None, None, None,
                 None, B=B, T=T, **kw, stats_out=stats)
    dist.all_reduce(stats)
    scal, ds, dv = run(mine, B, global_stats=stats, world=world)
    assert float((ds - world * big_ds[mine]).abs().max()) <= 1e-6 * float(big_ds.abs().max()) + 1e-10
    assert float((dv - world * big_dv[mine]).abs().max()) <= 1e-6 * float(big_dv.abs().max()) + 1e-10
    assert abs(float(scal[2]) - float(big_scal[2])) <= 1e-6 * abs(float(big_scal[2])) + 1e-10       # the same global R
    pol = scal[0].clone()
    dist.all_reduce(pol)
    assert abs(float(pol) / world - float(big_scal[0])) <= 1e-5 * abs(float(big_scal[0])) + 1e-9    # rank mean = global loss
    if rank == 0:
        print("GLOBAL_RANK_LOSS_OK")


def main():
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=rank == 0,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw",
                              scheduler="linear", learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=100, warmup=0.1,
                              device=dev, fuse_fc1_update=bool(int(os.environ.get("LR2_TEST_FUSE", "1"))))
    torch.manual_seed(7)
    model = ppo.ActorCritic(args, None)
    reward = ppo.Reward(args, None)
    with torch.no_grad():
        for p in list(model.parameters()) + list(reward.parameters()):
            p.normal_(0, 0.02)
    model, reward = model.to(dev), reward.to(dev).eval()
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    for _ in range(5):
        sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(100 + rank)
    dp = ppo._DataParallel()
    assert dp.world == world == 2
    check_exchanged_gradient_is_the_rank_mean(args, model, dp, rank, world, dev)
    check_global_rank_loss(args, dp, rank, world, dev)
    g = torch.Generator().manual_seed(500 + rank)
    w0 = model.actor.out_layer.fc1.weight[:4, :64].clone()
    for step in range(2):
        text = torch.randn(2, 2, 196, 768, generator=g).to(dev)
        img = torch.randn(2, 16, 768, generator=g).to(dev)
        tgts = torch.randint(0, 3, (2, 2), generator=g).to(dev)
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tgts)
        model.train()
        m = ppo.update_minibatch(args, model, opt, copt, rec, dp)
        assert torch.isfinite(m).all()
    # replicas identical: compare checksums of every parameter across the two ranks
    sums = torch.stack([p.detach().double().sum() for p in model.parameters()]).cpu()
    absmax = torch.stack([p.detach().abs().max().double() for p in model.parameters()]).cpu()
    both = [torch.zeros_like(sums) for _ in range(world)]
    dist.all_gather(both, sums)
    both2 = [torch.zeros_like(absmax) for _ in range(world)]
    dist.all_gather(both2, absmax)
    assert torch.equal(both[0], both[1]) and torch.equal(both2[0], both2[1]), "replicas diverged"
    assert not torch.equal(w0, model.actor.out_layer.fc1.weight[:4, :64]), "out_layer.fc1.weight was not updated"
    if rank == 0:
        print("DP_REHEARSAL_OK", float(m[0]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
