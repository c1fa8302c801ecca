"""MX-FP8 product and quantiser at the encoders' shapes beside the split-bf16 x 3 product (us per launch, TFLOP/s of 2MNK)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (M, N, K) in ((100864, 3072, 768), (100864, 768, 3072), (100864, 2304, 768), (100864, 768, 768), (131584, 4096, 1024), (131584, 1024, 4096),
                  (12544, 3072, 768)):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    bias = torch.randn(N, device=dev, generator=g) * 0.02
    out = torch.empty(M, N, device=dev)
    xm, wm = ops.quant_mxfp8(x), ops.quant_mxfp8(w)
    t8 = timed(lambda: ops.gemm_mxfp8(xm, wm, out, bias=bias, act=1))
    tq = timed(lambda: ops.quant_mxfp8(x, xm))
    xp, wp = ops.split_planes(x, ops.Planes.empty(M, K, dev)), ops.split_planes(w, ops.Planes.empty(N, K, dev))
    t3 = timed(lambda: ops.gemm(xp, wp, out, M, N, K, bias=bias, act=1, block_m=256, splits=1))
    fl = 2.0 * M * N * K
    print(f"M {M:6d} N {N:4d} K {K:4d}: mxfp8 {t8:8.1f} us ({fl / t8 / 1e6:7.1f} TFLOP/s) + quantise A {tq:6.1f} us ({8.0 * M * K / 1.25 / tq / 1e6:5.2f} TB/s) | "
          f"split-bf16 x3 {t3:8.1f} us ({fl / t3 / 1e6:6.1f} TFLOP/s fp32-equivalent) -> x{t3 / (t8 + tq):.2f}", flush=True)
