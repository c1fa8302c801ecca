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
