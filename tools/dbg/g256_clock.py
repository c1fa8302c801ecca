import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["LR2_GEMM_ABLATE"] = "32"
import torch
from lr2ppo_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for rep in range(2):
  for (M, N, K) in [(100864, 768, 768), (100864, 3072, 768), (100864, 768, 3072), (8192, 8192, 4096)]:
    a = torch.randn(M, K, device=dev, generator=g); b = torch.randn(N, K, device=dev, generator=g)
    ap = ops.split_planes(a, ops.Planes.empty(M, K, dev)); bp = ops.split_planes(b, ops.Planes.empty(N, K, dev))
    out = torch.empty(M, N, device=dev)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    ws = torch.zeros(4 * tiles + 16, device=dev)
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.0:      # >= 2 s of back-to-back launches so the clock settles
        for _ in range(20):
            ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1, splitk_ws=ws)
        torch.cuda.synchronize(); n += 20
    wall = (time.time() - t0) / n
    d = ws.view(torch.int64)[: 2 * tiles].view(tiles, 2).cpu().double()
    cyc, real = d[:, 0], d[:, 1]
    nt = K // 32
    ghz = (cyc / (real * 10.0)).median().item()
    print(f"M={M} N={N} K={K}: clock {ghz:.3f} GHz; cycles/Kstep {float((cyc / nt).median()):.0f}; us/Kstep {float((real / nt).median()) / 100:.3f}; "
          f"launch {wall*1e6:.0f} us = {2.0*M*N*K/wall/1e12:.0f} TF; main-loop share {float(real.median())/100*((tiles+255)//256)/ (wall*1e6):.2f}", flush=True)
    del a, b, ap, bp, out, ws
