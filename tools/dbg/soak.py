"""Soak: N composed PPO steps (frames + ids -> encoders -> rollout + update) and M end-to-end fine-tune steps in one process; prints
device memory (allocated / reserved) and the metrics' finiteness at intervals -- no growth after the first steps is the pass criterion."""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lr2ppo_amd import runtime  # noqa: E402
from lr2ppo_amd.finetune import pointwise, ppo  # noqa: E402
from lr2ppo_amd.finetune.features import FeatureExtractor, build_encoder_optimizer, finetune_pointwise_step, synthetic_raw_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--finetune", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                              learning_rate=1e-4, critic_learning_rate=1e-4, train_steps=1000, warmup=0.1, device=dev, batch_size=32)
    torch.manual_seed(7)
    model, reward = ppo.ActorCritic(args, None).to(dev), ppo.Reward(args, None).to(dev).eval()
    with torch.no_grad():
        for p in list(model.parameters()) + list(reward.parameters()):
            p.normal_(0, 0.02)
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(20):
            sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(5)
    fx = FeatureExtractor()
    fx.init_normal()
    fx = fx.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(1)
    dp = ppo._DataParallel()
    mem = []
    for i in range(a.steps):
        frames, ids, seg, tg = synthetic_raw_batch(32, 2, device=dev, generator=g)          # NEW tensors every step
        text, img = fx.extract(frames, ids, seg, check_ids=False)
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tg)
        model.train()
        m = ppo.update_minibatch(args, model, opt, copt, rec, dp)
        if i % 10 == 9 or i == a.steps - 1:
            torch.cuda.synchronize()
            mem.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
            print(f"composed step {i + 1}: finite {bool(torch.isfinite(m).all())}  allocated {mem[-1][0]} MiB  reserved {mem[-1][1]} MiB", flush=True)
    fx.text.embedding.check_ids()
    assert mem[-1][1] <= mem[1][1] * 1.02 + 64, "reserved memory keeps growing in the composed loop"
    pm = pointwise.Classifier(args, None).to(dev).train()
    with torch.no_grad():
        for p in pm.parameters():
            p.normal_(0, 0.02)
    popt, psch = pointwise.build_optimizer(args, pm)
    eopt, esch = build_encoder_optimizer(args, fx)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(20):
            psch.step(), esch.step()
    fx.train()
    mem2 = []
    for i in range(a.finetune):
        frames, ids, seg, tg = synthetic_raw_batch(32, 4, device=dev, generator=g)
        loss = finetune_pointwise_step(args, fx, pm, popt, psch, eopt, esch, frames, ids, seg, tg)
        if i % 5 == 4 or i == a.finetune - 1:
            torch.cuda.synchronize()
            mem2.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
            print(f"fine-tune step {i + 1}: loss {float(loss):.4f}  allocated {mem2[-1][0]} MiB  reserved {mem2[-1][1]} MiB", flush=True)
    fx.text.embedding.check_ids()
    assert mem2[-1][1] <= mem2[1][1] * 1.02 + 64, "reserved memory keeps growing in the fine-tune loop"
    print("SOAK_OK")


if __name__ == "__main__":
    main()
