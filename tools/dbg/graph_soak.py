"""Long equality soak of ppo.GraphedPPOStep: N PPO steps (bench shape 32 x 2, schedulers stepping every step, dropout on in the update)
eagerly on one model set and as graph replays on an identical one; every metric vector and the final parameters must be equal bit for
bit.  usage: python tools/dbg/graph_soak.py [--steps 200]"""
import argparse
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lr2ppo_amd import runtime  # noqa: E402
from lr2ppo_amd.finetune import ppo  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
margs = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=True,
                           kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                           learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=2 * a.steps, warmup=0.1, device=dev)


def build():
    torch.manual_seed(11)
    model, reward = ppo.ActorCritic(margs, None).to(dev), ppo.Reward(margs, None).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(12)
    with torch.no_grad():
        for p in list(model.parameters()) + list(reward.parameters()):
            p.normal_(0, 0.02, generator=g)
    opt, copt, sch, csch = ppo.build_optimizer(margs, model)
    model.actor.bind_grads(), model.critic.bind_grads()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sch.step(), csch.step()
    return model, reward, opt, copt, sch, csch


g = torch.Generator(device=dev).manual_seed(13)
data = [(torch.randn(32, 2, 196, 768, device=dev, generator=g), torch.randn(32, 16, 768, device=dev, generator=g),
         torch.randint(0, 3, (32, 2), device=dev, generator=g)) for _ in range(4)]
runtime.set_dropout_seed(5)
model, reward, opt, copt, sch, csch = A = build()
ref = []
for i in range(a.steps):
    model.eval()
    rec = ppo.rollout_step(model, reward, *data[i % 4])
    model.train()
    ref.append(ppo.update_minibatch(margs, model, opt, copt, rec).clone())
    sch.step(), csch.step()
runtime.set_dropout_seed(5)
model, reward, opt, copt, sch, csch = B = build()
step = ppo.GraphedPPOStep(margs, model, reward, opt, copt)
bad = 0
for i in range(a.steps):
    m = step(*data[i % 4]).clone()
    sch.step(), csch.step()
    if not torch.equal(m, ref[i]):
        bad += 1
        if bad < 4:
            print("step", i, "metrics differ", m.tolist(), ref[i].tolist())
torch.cuda.synchronize()
diff = [n for (n, p), (_, q) in zip(B[0].named_parameters(), A[0].named_parameters()) if not torch.equal(p, q)]
print(f"{a.steps} steps: {bad} metric vectors differ, {len(diff)} parameters differ; finite {bool(torch.isfinite(ref[-1]).all())}")
print("GRAPH_SOAK_OK" if not bad and not diff else "GRAPH_SOAK_FAILED")
