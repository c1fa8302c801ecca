"""The 192-row tile of the 256 x 256 NT kernel (csrc/gemm256.hip, MIH = 3): fp64 parity and us per launch for launches of less than one
round, against the kernels the dispatcher would otherwise pick.  LR2_GEMM_192=0 in the environment: the 256-row tile.
    python tools/dbg/gemm192_ab.py"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

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


print("LR2_GEMM_192 =", os.environ.get("LR2_GEMM_192", "1"))
for (M, N, K) in ((12544, 768, 3072), (12544, 768, 768), (12500, 768, 1536), (6272, 768, 3072), (9000, 1000, 512), (12544, 1024, 1024)):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.05
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    resid = torch.randn(M, N, device=dev, generator=g)
    xp, wp = ops.split_planes(x, ops.Planes.empty(M, K, dev)), ops.split_planes(w, ops.Planes.empty(N, K, dev))
    out = torch.full((M, N), float("nan"), device=dev)
    pl = ops.Planes.empty(M, N, dev)
    ops.gemm(xp, wp, out, M, N, K, bias=bias, resid=resid, block_m=256, splits=1)
    ref = x.double() @ w.double().t() + bias.double() + resid.double()
    err = float((out.double() - ref).abs().max())
    tol = 6e-5 * math.sqrt(K) * 0.05 * 4 + 5e-5 * float(ref.abs().max())
    ops.gemm(xp, wp, None, M, N, K, bias=bias, act=1, out_planes=pl, block_m=256, splits=1)
    o2 = torch.empty(M, N, device=dev)
    ops.gemm(xp, wp, o2, M, N, K, bias=bias, act=1, block_m=256, splits=1)
    same = torch.equal(pl.buf, ops.split_planes(o2, ops.Planes.empty(M, N, dev)).buf)
    t256 = timed(lambda: ops.gemm(xp, wp, out, M, N, K, bias=bias, resid=resid, block_m=256, splits=1))
    t128 = timed(lambda: ops.gemm(xp, wp, out, M, N, K, bias=bias, resid=resid, block_m=128, splits=1))
    t64 = timed(lambda: ops.gemm(xp, wp, out, M, N, K, bias=bias, resid=resid, block_m=64, splits=1))
    print(f"M {M} N {N} K {K}: max |err| {err:.2e} (tol {tol:.2e}) finite {bool(torch.isfinite(out).all())} planes == split(fp32) {same}; "
          f"block_m = 256 request {t256:7.1f} us, 128-row tiles {t128:7.1f}, 64-row tiles {t64:7.1f}", flush=True)
