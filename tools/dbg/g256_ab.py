"""A/B of gemm256 schedule variants inside ONE process (interleaved rounds, same device): LR2_GEMM256_VARIANT is read per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lr2ppo_amd import ops
dev = torch.device("cuda:0")
variants = [v for v in os.environ.get("AB", "0,1").split(",")]
shapes = [(100864, 768, 768), (100864, 2304, 768), (100864, 768, 3072), (8192, 8192, 4096)]
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=dev, generator=g); b = torch.randn(N, K, device=dev, generator=g)
    ap = ops.split_planes(a, ops.Planes.empty(M, K, dev)); bp = ops.split_planes(b, ops.Planes.empty(N, K, dev))
    out = torch.empty(M, N, device=dev)
    res = {v: [] for v in variants}
    for rnd in range(5):
        for v in variants:
            os.environ["LR2_GEMM256_VARIANT"] = v
            for _ in range(2):
                ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / 10)
    line = f"M={M} N={N} K={K}: "
    for v in variants:
        ms = sorted(res[v])
        line += f" v{v}: med {ms[len(ms)//2]*1e3:.1f} us min {ms[0]*1e3:.1f} us ({2.0*M*N*K/ms[len(ms)//2]/1e9:.0f} TF) |"
    print(line, flush=True)
    del a, b, ap, bp, out
