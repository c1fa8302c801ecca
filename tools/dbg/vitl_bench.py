"""ViT-L/14 (BASELINE config 5's encoder swap: hidden 1024, 24 layers, 16 heads, 257 tokens) forward over 512 frames, and the
ViT-B/16 + RoBERTa-base pair, at GEMM passes 3 (fp32-grade, parity mode) and 1 (plain bf16 operands: misses the parity bar)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from lr2ppo_amd import ops
from lr2ppo_amd.finetune.features import EncoderStack, encoder_args

dev = torch.device("cuda:0")
cfg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lr2ppo_amd", "configs", "vit_large_14_224.json")
a = encoder_args(cfg)
stack = EncoderStack(a, 10)
with torch.no_grad():
    for n, p in stack.named_parameters():
        if "gamma" not in n and "beta" not in n:
            p.normal_(0, 0.02)
stack = stack.to(dev).eval()
B = int(os.environ.get("FRAMES", "512"))
img = torch.randn(B, 3, 224, 224, device=dev)
seg = torch.ones(B, 257, dtype=torch.long, device=dev)
E, F, L, layers = 1024, 4096, 257, 24
fl = layers * (2.0 * B * L * E * (4 * E + 2 * F) + 4.0 * B * 16 * L * L * 64) + 2.0 * B * 256 * 640 * E
for passes in (3, 1):
    ops.set_gemm_passes(passes)
    with torch.no_grad():
        for _ in range(2):
            stack(img, seg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            stack(img, seg)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"model": "ViT-L/14", "frames": B, "passes": passes, "ms": round(ms, 2), "algorithmic_tflop": round(fl / 1e12, 1),
                      "tflops": round(fl / ms / 1e9, 1)}), flush=True)
ops.set_gemm_passes(3)
