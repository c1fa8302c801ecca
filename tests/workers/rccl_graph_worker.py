"""Fresh child process: the data-parallel PPO step captured in a HIP graph WITH its RCCL collectives.  One rank on cuda:0, the exchange
path forced (LR2_DP_FORCE=1), one-stream schedule (the default of world > 1): three steps through ppo.GraphedPPOStep (eager, capture +
replay, replay) against the same three steps run eagerly from identical state -- metrics, weights and first moments bit for bit."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def run(graphed: bool, dev):
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                              learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=41, warmup=0.1, device=dev, fuse_fc1_update=True)
    torch.manual_seed(5)
    model, reward = ppo.ActorCritic(args, None), ppo.Reward(args, None)
    for m in (model, reward):
        ppo._init_normal(m)
    model, reward = model.to(dev), reward.to(dev).eval()
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(123)
    dp = ppo._DataParallel()
    assert dp.active and dp.world == 1 and dp.backend == "nccl"
    gen = torch.Generator().manual_seed(29)
    data = [(torch.randn(3, 2, 196, 768, generator=gen).to(dev), torch.randn(3, 16, 768, generator=gen).to(dev),
             torch.randint(0, 3, (3, 2), generator=gen).to(dev)) for _ in range(3)]
    mets = []
    step = ppo.GraphedPPOStep(args, model, reward, opt, copt) if graphed else None
    for i, (text, img, tg) in enumerate(data):
        if graphed:
            mets.append(step(text, img, tg).clone())
        else:
            model.eval()
            rec = ppo.rollout_step(model, reward, text, img, tg)
            model.train()
            mets.append(ppo.update_minibatch(args, model, opt, copt, rec, dp).clone())
        sch.step(), csch.step()                       # a moving schedule: the replayed steps read their rates from device memory
    torch.cuda.synchronize()
    if graphed:
        assert step.graph is not None
        step.release()
    state = {n: p.detach().clone() for n, p in model.named_parameters() if p.numel() < 3_000_000}
    for tag, head, o in (("actor", model.actor, opt), ("critic", model.critic, copt)):
        w = head.out_layer.fc1.weight
        state[tag + ".fc1.sample"] = w.detach().view(-1)[::4099].clone()
        state[tag + ".fc1.m.sample"] = o.state[w]["exp_avg"].view(-1)[::4099].clone()
    state["metrics"] = torch.stack(mets)
    del model, reward, opt, copt, step
    torch.cuda.empty_cache()
    return state


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["LR2_DP_FORCE"] = "1"
    os.environ["LR2_PPO_STREAMS"] = "0"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ.get("MASTER_PORT", "29641"), rank=0, world_size=1)
    probe = torch.ones(4, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    print("RCCL_INIT_OK", flush=True)
    base = run(False, dev)
    print("EAGER_DP_DONE", flush=True)
    got = run(True, dev)
    bad = [k for k in base if not torch.equal(base[k], got[k])]
    assert not bad, f"graphed data-parallel step differs from the eager one in {bad[:6]}"
    print("RCCL_GRAPH_BITEQUAL", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
