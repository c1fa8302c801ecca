"""Per-K-step / per-tile cost of the MX-FP8 ring kernel: K sweep at N = 1024 (M = 131584), run once per LR2_MX_ABLATE setting
(0 = product, 1 = no scale traffic, 2 = no operand DMA, 4 = no matrix instructions; read once per process)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, N = 131584, 1024
res = []
for K in (1024, 2048, 4096):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    xm, wm = ops.quant_mxfp8(x), ops.quant_mxfp8(w)
    out = ops.Mx8.empty(M, N, dev)
    for _ in range(3):
        ops.gemm_mxfp8(xm, wm, None, out_mx=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.gemm_mxfp8(xm, wm, None, out_mx=out)
    e.record()
    torch.cuda.synchronize()
    res.append((K, s.elapsed_time(e) / 10 * 1e3))
    del x, w, xm, wm, out
rounds = (M + 255) // 256 * (N // 256) / 256
b = (res[2][1] - res[0][1]) / rounds / ((4096 - 1024) / 128)
a = res[0][1] / rounds - 8 * b
print(f"LR2_MX_ABLATE={os.environ.get('LR2_MX_ABLATE', '0')}: " + ", ".join(f"K {k}: {t:7.1f} us" for k, t in res)
      + f" -> {b:5.2f} us per K step, {a:5.1f} us per tile outside the loop (MX-FP8 output, no bias)")
