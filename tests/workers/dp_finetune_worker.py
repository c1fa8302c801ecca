"""2-rank rehearsal (gloo; both ranks on cuda:0) of the end-to-end fine-tune steps under data parallelism: each rank runs
features.finetune_ppo_step and features.finetune_pointwise_step on ITS OWN batch; the encoder / embedding gradients are averaged over the
ranks (one all-reduce per flat buffer, the text tower's issued under the image tower's backward), the heads' by ppo._DataParallel.
Afterwards every replica must hold the same bits (encoders, embeddings, heads) and they must have moved.  Launched by
tests/test_dp_gpu.py through torch.distributed.run."""
import argparse
import os
import sys
import warnings

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def digest(mods):
    parts = []
    for m in mods:
        for _, q in sorted(m.named_parameters(), key=lambda kv: kv[0]):
            parts.append(q.detach().double().sum().view(1))
            parts.append(q.detach().double().abs().sum().view(1))
    return torch.cat(parts).cpu()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import pointwise, ppo
    from lr2ppo_amd.finetune.features import (TEXT_CONFIG, VIT_CONFIG, FeatureExtractor, build_encoder_optimizer, encoder_args,
                                              finetune_pointwise_step, finetune_ppo_step, synthetic_raw_batch)
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=rank == 0,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                              learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=20, warmup=0.1, device=dev, batch_size=2,
                              fuse_fc1_update=True)
    torch.manual_seed(5)                                          # identical replicas
    fx = FeatureExtractor(encoder_args(VIT_CONFIG, layers_num=1), encoder_args(TEXT_CONFIG, layers_num=1))
    model, reward, cls = ppo.ActorCritic(args, None), ppo.Reward(args, None), pointwise.Classifier(args, None)
    for m in (fx, model, reward, cls):
        ppo._init_normal(m)
    with torch.no_grad():
        reward.head.weight.mul_(40.0)
    fx, model, reward, cls = fx.to(dev), model.to(dev), reward.to(dev).eval(), cls.to(dev)
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    popt, psch = pointwise.build_optimizer(args, cls)
    eopt, esch = build_encoder_optimizer(args, fx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in (sch, csch, psch, esch):
            s.step(), s.step()                                    # past lambda(0) = 0
    d0 = digest([fx, model, cls])
    same0 = [torch.zeros_like(d0) for _ in range(world)]
    dist.all_gather(same0, d0)
    assert torch.equal(same0[0], same0[1]), "replicas differ before the first step"
    runtime.set_dropout_seed(300 + rank)                          # per-rank dropout streams (seed + rank, as ppo.py:754)
    frames, ids, seg, tgts = synthetic_raw_batch(2, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(70 + rank))
    m = finetune_ppo_step(args, fx, model, reward, opt, copt, eopt, frames, ids, seg, tgts)
    assert torch.isfinite(m).all()
    cls.train(), fx.train()
    loss = finetune_pointwise_step(args, fx, cls, popt, psch, eopt, esch, frames, ids, seg, tgts)
    assert torch.isfinite(loss)
    fx.text.embedding.check_ids()
    torch.cuda.synchronize()
    d1 = digest([fx, model, cls])
    both = [torch.zeros_like(d1) for _ in range(world)]
    dist.all_gather(both, d1)
    assert torch.equal(both[0], both[1]), "replicas differ after the data-parallel fine-tune steps"
    assert not torch.equal(d0, d1), "nothing moved"
    # the ranks' batches really differ (else identical replicas prove nothing)
    fsum = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(fsum, frames.double().sum().view(1).cpu())
    assert float(fsum[0]) != float(fsum[1])
    dist.barrier()
    if rank == 0:
        print("DP_FINETUNE_REPLICAS_IDENTICAL_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
