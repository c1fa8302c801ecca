import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LR2_GEMM_ABLATE"] = "32"
import torch
from lr2ppo_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(65536, 768, 768), (65536, 768, 3072), (8192, 8192, 4096)]:
    a = torch.randn(M, K, device=dev, generator=g); b = torch.randn(N, K, device=dev, generator=g)
    ap = ops.split_planes(a, ops.Planes.empty(M, K, dev)); bp = ops.split_planes(b, ops.Planes.empty(N, K, dev))
    out = torch.empty(M, N, device=dev)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    ws = torch.zeros(4 * tiles + 16, device=dev)
    import lr2ppo_amd.ops as O
    for _ in range(30):      # sustained load so the clock settles
        ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1, splitk_ws=ws)
    torch.cuda.synchronize()
    d = ws.view(torch.int64)[: 2 * tiles].view(tiles, 2).cpu().double()
    cyc, real = d[:, 0], d[:, 1]
    nt = K // 32
    ghz = (cyc / (real * 10.0)).median().item()     # cycles per ns
    print(f"M={M} N={N} K={K}: clock {ghz:.3f} GHz; cycles per K step median {float((cyc / nt).median()):.0f} (ideal 3072) "
          f"min {float((cyc / nt).min()):.0f} max {float((cyc / nt).max()):.0f}; us per K step {float((real / nt).median()) / 100:.3f}", flush=True)
