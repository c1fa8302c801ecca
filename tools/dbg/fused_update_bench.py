"""The fused out_layer.fc1 update (TN weight-gradient GEMM with the AdamW step in its epilogue) as a function of the contraction
length K = 64 * world (data parallel: the factors of all ranks are concatenated) -- per-launch time, HBM GB/s, MFMA TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from lr2ppo_amd import ops

dev = torch.device("cuda:0")
F, W = 3072, 162816
p = torch.randn(F, W, device=dev) * 0.02
m = torch.zeros_like(p)
v = torch.zeros_like(p)
g = torch.Generator(device=dev).manual_seed(0)
for K in [int(k) for k in os.environ.get("KS", "64,128,256,512,640").split(",")]:
    dz = ops.split_planes(torch.randn(K, F, device=dev, generator=g) * 0.01, ops.Planes.empty(K, F, dev))
    fl = ops.split_planes(torch.randn(K, W, device=dev, generator=g), ops.Planes.empty(K, W, dev))
    adam = ops.AdamArgs(p, m, v, 1e-4, 0.9, 0.999, 1e-6, 0.01)
    bm, sp = ops.choose_tiling(F, W, K, True, True)
    for _ in range(2):
        ops.gemm(dz, fl, None, F, W, K, trans_a=True, trans_b=True, lda=F, ldb=W, splits=1, block_m=bm, adam=adam)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        ops.gemm(dz, fl, None, F, W, K, trans_a=True, trans_b=True, lda=F, ldb=W, splits=1, block_m=bm, adam=adam)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"K={K:4d}: {ms:7.3f} ms  {24.0 * F * W / ms / 1e6:6.0f} GB/s (p, m, v)  {2.0 * F * W * K / ms / 1e9:6.1f} TFLOP/s", flush=True)
