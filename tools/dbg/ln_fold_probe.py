"""Probe for DESIGN.md 9.7 (LayerNorm folded into the consuming product): what the PRODUCER side of the fold would cost today.
At the ViT shapes (M = 100 864 rows, N = 768) it times the residual product writing fp32 only (what runs now), the same product
writing fp32 AND hi / lo planes from one epilogue (what the fold needs from it -- the planes LayerNorm writes today), and the
LayerNorm pass the fold removes.  The fold pays off by (LayerNorm) - (extra epilogue time) - (a ~5-us statistics finish) per LayerNorm.
    python tools/dbg/ln_fold_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, E = 100864, 768


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


h = torch.randn(M, E, device=dev, generator=g)
gamma, beta = torch.randn(E, device=dev, generator=g), torch.randn(E, device=dev, generator=g)
x_p = ops.Planes.empty(M, E, dev)
t_ln = timed(lambda: ops.layernorm_fwd(h, gamma, beta, None, rows=M, D=E, eps=1e-6, mode=1, out_planes=x_p))
print(f"layernorm_fwd [{M}, {E}] -> planes: {t_ln:7.1f} us", flush=True)
for K in (768, 3072):
    a = ops.split_planes(torch.randn(M, K, device=dev, generator=g), ops.Planes.empty(M, K, dev))
    w = ops.split_planes(torch.randn(E, K, device=dev, generator=g) * 0.02, ops.Planes.empty(E, K, dev))
    bias = torch.randn(E, device=dev, generator=g)
    out, out2, pl = torch.empty(M, E, device=dev), torch.empty(M, E, device=dev), ops.Planes.empty(M, E, dev)
    t0 = timed(lambda: ops.gemm(a, w, out, M, E, K, bias=bias, resid=h, block_m=256, splits=1))
    t1 = timed(lambda: ops.gemm(a, w, out2, M, E, K, bias=bias, resid=h, out_planes=pl, block_m=256, splits=1))
    same = torch.equal(out, out2)
    ref = ops.split_planes(out, ops.Planes.empty(M, E, dev))
    planes_ok = torch.equal(ref.buf, pl.buf)
    print(f"K = {K:4d}: residual product -> fp32 {t0:7.1f} us;  -> fp32 + planes {t1:7.1f} us  (+{t1 - t0:5.1f} us; fp32 equal {same}, planes == split_planes(fp32) {planes_ok});"
          f"  net of one folded LayerNorm: {t_ln - (t1 - t0) - 5.0:6.1f} us", flush=True)
