"""Does a 256 x 256 tile of the K = 768 FFN-1 product take longer when more CUs run one at the same time?  One partial / full round of
the chip (tiles = 12 ... 256) and two full rounds; us per launch (HIP events over 50 launches, so ~2 us of launch gap are inside)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N, K = 3072, 768
g = torch.Generator(device=dev).manual_seed(0)
w = ops.split_planes(torch.randn(N, K, device=dev, generator=g) * 0.02, ops.Planes.empty(N, K, dev))
bias = torch.randn(N, device=dev, generator=g) * 0.02
print("ablate:", os.environ.get("LR2_GEMM_ABLATE", "0"))
for rows in (1, 2, 5, 10, 16, 21, 42, 64, 394):
    M = rows * 256
    a = ops.split_planes(torch.randn(M, K, device=dev, generator=g), ops.Planes.empty(M, K, dev))
    out_p = ops.Planes.empty(M, N, dev)
    for act in (0, 1):
        fn = lambda: ops.gemm(a, w, None, M, N, K, bias=bias, act=act, out_planes=out_p, block_m=256, splits=1)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 50 * 1e3
        tiles = rows * 12
        print(f"  tiles {tiles:5d} ({tiles / 256:5.2f} rounds) act {act}: {us:8.1f} us per launch, {us / max(1, -(-tiles // 256)):7.1f} us per round")
