#!/usr/bin/env python3
"""Micro-benchmark of lr2_gemm on the shapes of the LR2PPO head (run on the GPU box).
usage: python tools/gemm_bench.py [--passes 3] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

SHAPES = [  # (form, M, N, K) as ops.gemm sees them
    ("NT", 12544, 3072, 768), ("NT", 12544, 768, 3072), ("NT", 12544, 768, 768), ("NN", 12544, 768, 3072),
    ("NN", 12544, 3072, 768), ("TN", 3072, 768, 12544), ("TN", 768, 3072, 12544), ("TN", 768, 768, 12544),
    ("NT", 64, 3072, 162816), ("NN", 64, 162816, 3072), ("TN", 3072, 162816, 64), ("NT", 1024, 3072, 768),
    ("NT", 4096, 4096, 4096),
    # dual-encoder forward at batch 32 (M = 32 * 197)
    ("NT", 6304, 2304, 768), ("NT", 6304, 768, 768), ("NT", 6304, 3072, 768), ("NT", 6304, 768, 3072),
    # NT vs NN (forward on W or on W^T) at the encoder's shapes
    ("NN", 6304, 2304, 768), ("NN", 6304, 3072, 768), ("NT", 100864, 2304, 768), ("NN", 100864, 2304, 768),
    ("NT", 100864, 3072, 768), ("NN", 100864, 3072, 768),
    # small GEMMs of the PPO step (image tokens, tail, out_layer.fc2): where split-K + its reduce launch compete with one pass
    ("NT", 1024, 768, 768), ("NN", 1024, 3072, 768), ("NT", 1024, 768, 3072), ("NT", 64, 768, 3072), ("NT", 256, 768, 768),
    ("TN", 768, 768, 1024), ("TN", 3072, 768, 1024), ("NN", 1024, 768, 3072),
    # 31..: ViT-B/16 forward over 512 frames (M = 512 * 197) and the pointwise head at 20 tags (M = 640 * 196)
    ("NT", 100864, 768, 768), ("NT", 100864, 768, 3072), ("NT", 125440, 3072, 768), ("NT", 125440, 768, 3072),
    ("NT", 8192, 8192, 8192),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=int, nargs="*", default=None, help="indices into SHAPES")
    ap.add_argument("--planes", action="store_true", help="operands as pre-split bf16 hi/lo planes (LDS-DMA path)")
    ap.add_argument("--bm", type=int, default=None, help="override block_m")
    ap.add_argument("--splits", type=int, default=None, help="override split-K factor")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for si, (form, M, N, K) in enumerate(SHAPES):
        if a.only is not None and si not in a.only:
            continue
        ta, tb = form == "TN", form in ("NN", "TN")
        A = torch.randn((K, M) if ta else (M, K), device=dev, generator=g)
        B = torch.randn((K, N) if tb else (N, K), device=dev, generator=g)
        out = torch.empty(M, N, device=dev)
        if a.planes:
            big_b = (M == 64 and K > 100000) or (N > 100000 and form == "NN")   # out_layer.fc1 weight: stays fp32
            Ap = ops.Planes.empty(*A.shape, dev)
            ops.split_planes(A, Ap)
            A = Ap
            if not big_b:
                Bp = ops.Planes.empty(*B.shape, dev)
                ops.split_planes(B, Bp)
                B = Bp
        bm, sp = ops.choose_tiling(M, N, K, ta, tb)
        bm = a.bm or bm
        sp = a.splits or sp
        ws = torch.empty(max(1, sp) * M * N, device=dev) if sp > 1 else None
        for _ in range(3):
            ops.gemm(A, B, out, M, N, K, trans_a=ta, trans_b=tb, splitk_ws=ws, passes=a.passes, block_m=bm, splits=sp)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            ops.gemm(A, B, out, M, N, K, trans_a=ta, trans_b=tb, splitk_ws=ws, passes=a.passes, block_m=bm, splits=sp)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        tf = 2.0 * M * N * K / ms / 1e9
        gbs = 4.0 * (M * K + N * K + M * N) / ms / 1e6
        print(f"{form} M={M:6d} N={N:6d} K={K:6d} bm={bm} splits={sp:2d}: {ms:8.4f} ms  {tf:7.1f} TFLOP/s  {gbs:7.0f} GB/s(alg)", flush=True)
        del A, B, out, ws


if __name__ == "__main__":
    main()
