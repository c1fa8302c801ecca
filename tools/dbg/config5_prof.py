"""The config-5 feature extraction (ViT-L/14 + projection + RoBERTa-base, every encoder projection an MX-FP8 product) a few times, for a
kernel trace:  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 tools/dbg/config5_prof.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd.finetune.features import VIT_L14_CONFIG, FeatureExtractor, encoder_args, synthetic_raw_batch  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
fx = FeatureExtractor(encoder_args(VIT_L14_CONFIG), precision=os.environ.get("PRECISION", "mxfp8"))
fx.init_normal()
fx = fx.to(dev).eval()
frames, ids, seg, _ = synthetic_raw_batch(32, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
for _ in range(2):
    fx.extract(frames, ids, seg, check_ids=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 3
e0.record()
for _ in range(n):
    fx.extract(frames, ids, seg, check_ids=False)
e1.record()
torch.cuda.synchronize()
print(f"config-5 extraction ({fx.precision}): {e0.elapsed_time(e1) / n:.2f} ms")
