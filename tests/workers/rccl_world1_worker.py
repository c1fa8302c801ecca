"""Fresh child process: RCCL ("nccl") with ONE rank on cuda:0, the data-parallel exchange path FORCED (LR2_DP_FORCE=1).
Every collective of the N > 1 PPO step executes on RCCL's own stream -- all_gather_into_tensor of the out_layer.fc1 factor planes
on uint8 views at non-zero storage offsets, the tail all-reduce of each model's flat gradient buffer, the 3-float RankLoss statistics
all-reduce, the packed metric all-reduce -- and is the identity at world 1, so two rollout + update steps must leave bit-identical
weights, optimizer state and metrics to the plain single-rank step.  Both stream schedules, fused and unfused fc1 update."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


_INIT = {}      # the seeded initial weights, kept on the device: every run starts from the same bits without paying the 1.5 B-element
                # host initialisation + upload again (6 runs: ~30 s of a 50-s test)


def run(force: bool, streams: str, fuse: bool, dev):
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    os.environ["LR2_DP_FORCE"] = "1" if force else "0"
    os.environ["LR2_PPO_STREAMS"] = streams
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                              learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=41, warmup=0.1, device=dev,
                              fuse_fc1_update=fuse)
    if _INIT:
        with torch.device("meta"):              # no host allocation / default initialisation of 1.5 B elements: the weights are loaded below
            model, reward = ppo.ActorCritic(args, None), ppo.Reward(args, None)
    else:
        model, reward = ppo.ActorCritic(args, None), ppo.Reward(args, None)
    if not _INIT:
        torch.manual_seed(5)
        for m in (model, reward):
            ppo._init_normal(m)
        model, reward = model.to(dev), reward.to(dev).eval()
        _INIT["model"] = {k: v.detach().clone() for k, v in model.state_dict().items()}
        _INIT["reward"] = {k: v.detach().clone() for k, v in reward.state_dict().items()}
    else:
        model, reward = model.to_empty(device=dev), reward.to_empty(device=dev).eval()
        model.load_state_dict(_INIT["model"], strict=True)
        reward.load_state_dict(_INIT["reward"], strict=True)
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(99)
    dp = ppo._DataParallel()
    assert dp.active == force and dp.world == 1 and dp.backend == "nccl"
    gen = torch.Generator().manual_seed(23)
    mets = []
    for _ in range(2):
        text, img = torch.randn(3, 2, 196, 768, generator=gen).to(dev), torch.randn(3, 16, 768, generator=gen).to(dev)
        tg = torch.randint(0, 3, (3, 2), generator=gen).to(dev)
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tg)
        model.train()
        mets.append(ppo.update_minibatch(args, model, opt, copt, rec, dp).clone())
    torch.cuda.synchronize()
    state = {n: p.detach().clone() for n, p in model.named_parameters() if p.numel() < 3_000_000}
    # the 2 GB matrices and their first moments: a strided sample + a checksum
    for tag, head, o in (("actor", model.actor, opt), ("critic", model.critic, copt)):
        w = head.out_layer.fc1.weight
        state[tag + ".fc1.sample"] = w.detach().view(-1)[::4099].clone()
        state[tag + ".fc1.sum"] = w.detach().double().sum().view(1)
        state[tag + ".fc1.m.sample"] = o.state[w]["exp_avg"].view(-1)[::4099].clone()
    state["metrics"] = torch.stack(mets)
    del model, reward, opt, copt
    torch.cuda.empty_cache()
    return state


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("MASTER_PORT", "29633"), rank=0, world_size=1)
    probe = torch.ones(4, device=dev)
    dist.all_reduce(probe)                                   # RCCL executes
    torch.cuda.synchronize()
    assert float(probe.sum()) == 4.0
    print("RCCL_INIT_OK", flush=True)
    for fuse in (True, False):
        base = run(False, "0", fuse, dev)
        for streams in ("0", "1"):
            got = run(True, streams, fuse, dev)
            bad = [k for k in base if not torch.equal(base[k], got[k])]
            assert not bad, f"forced DP exchange (streams={streams}, fuse={fuse}) differs from the plain step in {bad[:6]}"
            print(f"RCCL_WORLD1_EXCHANGE_BITEQUAL fuse={int(fuse)} streams={streams}", flush=True)
    dist.destroy_process_group()
    print("RCCL_WORLD1_OK", flush=True)


if __name__ == "__main__":
    main()
