"""Host cost and device time of the PPO head step: eager vs GraphedPPOStep, by component (bench shape 32 x 2 by default)."""
import argparse
import time
import warnings

import torch

from lr2ppo_amd import ops, runtime
from lr2ppo_amd.finetune import ppo

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--tags", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
margs = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                           kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                           learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=1000, warmup=0.1, device=dev)
torch.manual_seed(7)
model = ppo.ActorCritic(margs, None).to(dev)
reward = ppo.Reward(margs, None).to(dev).eval()
with torch.no_grad():
    for p in list(model.parameters()) + list(reward.parameters()):
        p.normal_(0, 0.02)
opt, copt, sch, csch = ppo.build_optimizer(margs, model)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(20):
        sch.step(), csch.step()
model.actor.bind_grads(), model.critic.bind_grads()
g = torch.Generator(device=dev).manual_seed(1000)
data = [(torch.randn(a.batch, a.tags, 196, 768, device=dev, generator=g), torch.randn(a.batch, 16, 768, device=dev, generator=g),
         torch.randint(0, 3, (a.batch, a.tags), device=dev, generator=g)) for _ in range(4)]


def eager(text, img, tgts):
    model.eval()
    rec = ppo.rollout_step(model, reward, text, img, tgts)
    model.train()
    return ppo.update_minibatch(margs, model, opt, copt, rec)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(*data[i % 4])
    host = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return host, (time.perf_counter() - t0) / n * 1e3


for i in range(5):
    eager(*data[i % 4])
h3, _ = timed(eager, 3)
_, e_ms = timed(eager, a.steps)
print(f"eager : host enqueue {h3:.3f} ms (3 steps, empty queue); {e_ms:.3f} ms per step over {a.steps}")

step = ppo.GraphedPPOStep(margs, model, reward, opt, copt)
for i in range(3):
    step(*data[i % 4])
h3, _ = timed(step, 3)
_, g_ms = timed(step, a.steps)
print(f"graph : host enqueue {h3:.3f} ms (3 steps, copies included); {g_ms:.3f} ms per step over {a.steps}")
h3s, _ = timed(lambda *b: step(step.text, step.img, step.tgts), 3)
print(f"graph : host enqueue {h3s:.3f} ms with the inputs already in the static buffers")
# components
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    step._store_scalars()
t1 = time.perf_counter()
for _ in range(N):
    step.graph.replay()
t2 = time.perf_counter()
for _ in range(N):
    opt.count_replayed_step(), copt.count_replayed_step(), runtime.advance(step.draws)
t3 = time.perf_counter()
torch.cuda.synchronize()
print(f"graph components (ms): store {1e3 * (t1 - t0) / N:.3f}, replay {1e3 * (t2 - t1) / N:.3f}, book-keeping {1e3 * (t3 - t2) / N:.3f}")
