"""Two-stream vs one-stream PPO schedule over many steps from identical state: every metric and a weight digest must match bit for bit."""
import argparse, copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from lr2ppo_amd import runtime
from lr2ppo_amd.finetune import ppo

steps = int(os.environ.get("STEPS", "40"))
dev = torch.device("cuda:0")
args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=False,
                          kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                          learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=1000, warmup=0.1, device=dev, fuse_fc1_update=True)
g = torch.Generator(device=dev).manual_seed(5)
data = [(torch.randn(32, 2, 196, 768, device=dev, generator=g), torch.randn(32, 16, 768, device=dev, generator=g),
         torch.randint(0, 3, (32, 2), device=dev, generator=g)) for _ in range(4)]
out = {}
for mode in ("1", "0"):
    os.environ["LR2_PPO_STREAMS"] = mode
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    model = ppo.ActorCritic(args, None).to(dev)
    reward = ppo.Reward(args, None).to(dev).eval()
    with torch.no_grad():
        for p in list(model.parameters()) + list(reward.parameters()):
            p.normal_(0, 0.02)
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(20):
            sch.step(), csch.step()
    model.actor.bind_grads(), model.critic.bind_grads()
    runtime.set_dropout_seed(99)
    ms = []
    for i in range(steps):
        text, img, tg = data[i % 4]
        model.eval()
        rec = ppo.rollout_step(model, reward, text, img, tg)
        model.train()
        ms.append(ppo.update_minibatch(args, model, opt, copt, rec))
    torch.cuda.synchronize()
    digest = [p.detach().double().sum().item() for p in model.parameters()]
    out[mode] = (torch.stack(ms).cpu(), digest)
    del model, reward, opt, copt
    torch.cuda.empty_cache()
same_m = torch.equal(out["1"][0], out["0"][0])
same_w = out["1"][1] == out["0"][1]
print("steps", steps, "metrics identical:", same_m, "weight digests identical:", same_w)
sys.exit(0 if (same_m and same_w) else 1)
