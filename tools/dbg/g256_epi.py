import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lr2ppo_amd import ops
dev = torch.device("cuda:0")
K, N = 768, 768
for M in (256, 2048, 8192, 21760, 65536):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev)
    ap = ops.split_planes(a, ops.Planes.empty(M, K, dev)); bp = ops.split_planes(b, ops.Planes.empty(N, K, dev))
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 50 * 1e3
    tiles = ((M + 255) // 256) * 3
    print(f"M={M} tiles={tiles} rounds={tiles/256:.2f}: {us:.1f} us/launch", flush=True)
