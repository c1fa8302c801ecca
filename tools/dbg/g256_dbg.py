import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lr2ppo_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(256, 256, 32), (256, 256, 96), (256, 256, 64)]:
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    b = torch.randint(-2, 3, (N, K), generator=g).float()
    ref = a @ b.t()
    ap = ops.split_planes(a.to(dev), ops.Planes.empty(M, K, dev))
    bp = ops.split_planes(b.to(dev), ops.Planes.empty(N, K, dev))
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1)
    o = out.cpu()
    bad = (o != ref)
    print(M, N, K, "bad", int(bad.sum()), "nan", int(torch.isnan(o).sum()))
    if bad.any():
        rows = bad.any(1).nonzero().view(-1)
        cols = bad.any(0).nonzero().view(-1)
        print(" bad rows", rows[:8].tolist(), "...", rows[-4:].tolist(), len(rows))
        print(" bad cols", cols[:8].tolist(), "...", cols[-4:].tolist(), len(cols))
        # per 16x16 tile bad map for the first wave tile
        tm = bad.view(16, 16, 16, 16).any(3).any(1)
        print(tm.int())
        i, j = bad.nonzero()[0].tolist()
        print(" first bad", i, j, float(o[i, j]), float(ref[i, j]))
        # which partial sums? compare with per-k-tile contributions
        for kt in range(K // 32):
            part = a[i, kt*32:(kt+1)*32] @ b[j, kt*32:(kt+1)*32]
            print("  ktile", kt, float(part))
