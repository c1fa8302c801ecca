"""Per-signature HIP-event profile of one BASELINE configs[1] step (features + stage-1 train step at 32 x 20 tags)."""
import argparse, os, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from lr2ppo_amd import ops
from lr2ppo_amd.finetune import pointwise
from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch

dev = torch.device("cuda:0")
args = argparse.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=False,
                          optimizer="adamw", scheduler="linear", learning_rate=1e-3, train_steps=1000, warmup=0.1, device=dev,
                          batch_size=32, fuse_fc1_update=True)
torch.manual_seed(9)
fx = FeatureExtractor(); fx.init_normal(); fx = fx.to(dev).eval()
model = pointwise.Classifier(args, None).to(dev)
with torch.no_grad():
    for p in model.parameters():
        p.normal_(0, 0.02)
opt, sch = pointwise.build_optimizer(args, model)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for _ in range(20):
        sch.step()
model.train()
g = torch.Generator(device=dev).manual_seed(1)
raw = synthetic_raw_batch(32, 20, device=dev, generator=g)
text, img = fx.extract(*raw[:3], check_ids=False)
for _ in range(2):
    pointwise.train_model(args, model, opt, sch, text, img, raw[3])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    pointwise.train_model(args, model, opt, sch, text, img, raw[3])
e1.record(); torch.cuda.synchronize()
print("head train step (features given):", round(e0.elapsed_time(e1) / 3, 3), "ms")
ops.profile_start()
pointwise.train_model(args, model, opt, sch, text, img, raw[3])
prof = ops.profile_stop()
tot = sum(v["ms"] for v in prof.values())
print("instrumented kernels:", round(tot, 3), "ms")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:22]:
    tf = v["flops"] / (v["ms"] / v["n"]) / 1e9 if v["flops"] else 0.0
    print(f"  {k:44s} n={v['n']:3d} {v['ms']:8.3f} ms  avg {v['ms'] / v['n'] * 1e3:8.1f} us  {tf:6.1f} TF/s")
