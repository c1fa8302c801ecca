import sys, os, argparse, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lr2ppo_amd import ops, runtime
from lr2ppo_amd.finetune import ppo
dev = torch.device("cuda:0")
margs = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                           kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw",
                           scheduler="linear", learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=1000, warmup=0.1,
                           device=dev, fuse_fc1_update=True)
torch.manual_seed(7)
model = ppo.ActorCritic(margs, None).to(dev)
reward = ppo.Reward(margs, None).to(dev).eval()
with torch.no_grad():
    for p in list(model.parameters()) + list(reward.parameters()):
        p.normal_(0, 0.02)
opt, copt, sch, csch = ppo.build_optimizer(margs, model)
for _ in range(20):
    sch.step(), csch.step()
model.actor.bind_grads(), model.critic.bind_grads()
dp = ppo._DataParallel()
g = torch.Generator(device=dev).manual_seed(1)
data = (torch.randn(32, 2, 196, 768, device=dev, generator=g), torch.randn(32, 16, 768, device=dev, generator=g),
        torch.randint(0, 3, (32, 2), device=dev, generator=g))
def step():
    model.eval(); rec = ppo.rollout_step(model, reward, *data); model.train()
    return ppo.update_minibatch(margs, model, opt, copt, rec, dp)
for _ in range(3): step()
torch.cuda.synchronize()
for n in (1, 2, 5, 10, 20, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c0 = time.thread_time()
    for _ in range(n): step()
    t_host = time.perf_counter() - t0; c_host = time.thread_time() - c0
    torch.cuda.synchronize()
    print(f"n={n}: host wall {t_host / n * 1e3:.2f} ms/step, host cpu {c_host / n * 1e3:.2f} ms/step, total {(time.perf_counter() - t0) / n * 1e3:.2f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
