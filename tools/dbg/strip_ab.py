"""A/B of the XCD-aware tile order's strip width (LR2_GEMM_STRIP, read per call) on the encoders' big products, one process:
us per launch for strips of 2 / 4 / 6 / 8 (default) / 12 / 16 tiles along N.  --nt: also LR2_GEMM256_VARIANT 2 / 3 (nt cache policy on
the A / B operand's LDS-DMA loads; needs profiles/experiments/r04_gemm256_nt_loads.diff.txt applied) x strips 4 / 8 / 12; --once: one launch per setting in a fixed order (for a --pmc FETCH_SIZE pass)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=12):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (M, N, K) in ((100864, 3072, 768), (100864, 2304, 768), (100864, 768, 3072), (100864, 768, 768), (131584, 4096, 1024)):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    bias = torch.randn(N, device=dev, generator=g) * 0.02
    xp, wp = ops.split_planes(x, ops.Planes.empty(M, K, dev)), ops.split_planes(w, ops.Planes.empty(N, K, dev))
    out = ops.Planes.empty(M, N, dev)
    res = {}
    once = "--once" in sys.argv                       # one launch per setting, in a fixed order: for a counter pass
    for rounds in range(1 if once else 2):
        for variant in ((0, 2, 3) if "--nt" in sys.argv else (0,)):          # LR2_GEMM256_VARIANT 2 / 3: nt loads of A / B
            os.environ["LR2_GEMM256_VARIANT"] = str(variant)
            for strip in ((4, 8, 12) if "--nt" in sys.argv else (8, 2, 4, 6, 12, 16, 8)):
                os.environ["LR2_GEMM_STRIP"] = str(strip)
                fn = lambda: ops.gemm(xp, wp, None, M, N, K, bias=bias, act=1, out_planes=out, block_m=256, splits=1)  # noqa: E731
                if once:
                    fn()
                    torch.cuda.synchronize()
                    print(f"LAUNCH M {M} N {N} K {K} variant {variant} strip {strip}", flush=True)
                else:
                    res.setdefault((variant, strip), []).append(timed(fn))
    if not once:
        print(f"M {M} N {N} K {K}: " + ", ".join(f"v{k[0]} strip {k[1]}: {min(v):7.1f}" for k, v in sorted(res.items())), flush=True)
    del x, w, xp, wp, out
