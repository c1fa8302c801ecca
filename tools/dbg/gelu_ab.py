"""A/B of a GEMM epilogue variant: the ViT FFN-1 product (M = 100864, N = 3072, K = 768, planes in / planes out) with no activation,
with GELU, with GELU + saved pre-activation, and the FFN-2 input gradient with GELU'.  Run once per library build
(LR2_AB_LIB=/path/to/other/liblr2ppo_hip.so selects another build of the same ABI); prints us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lr2ppo_amd import _native, ops  # noqa: E402

if os.environ.get("LR2_AB_LIB"):            # a switch of THIS tool: another build of the same ABI, loaded explicitly
    _native.use_library(os.environ["LR2_AB_LIB"])


def main():
    dev = torch.device("cuda:0")
    M, N, K = 100864, 3072, 768
    g = torch.Generator(device=dev).manual_seed(0)
    a = ops.split_planes(torch.randn(M, K, device=dev, generator=g), ops.Planes.empty(M, K, dev))
    w = ops.split_planes(torch.randn(N, K, device=dev, generator=g) * 0.02, ops.Planes.empty(N, K, dev))
    bias = torch.randn(N, device=dev, generator=g) * 0.02
    out_p, z = ops.Planes.empty(M, N, dev), torch.empty(M, N, device=dev)
    dy = ops.split_planes(torch.randn(M, K, device=dev, generator=g), ops.Planes.empty(M, K, dev))
    cases = {
        "no activation": lambda: ops.gemm(a, w, None, M, N, K, bias=bias, out_planes=out_p, block_m=256, splits=1),
        "GELU": lambda: ops.gemm(a, w, None, M, N, K, bias=bias, act=1, out_planes=out_p, block_m=256, splits=1),
        "GELU + z": lambda: ops.gemm(a, w, None, M, N, K, bias=bias, act=1, out_z=z, out_planes=out_p, block_m=256, splits=1),
        "GELU' (dgrad)": lambda: ops.gemm(dy, w, None, M, N, K, act=2, aux_z=z, out_planes=out_p, block_m=256, splits=1),
    }
    print("library:", os.environ.get("LR2_AB_LIB", "in-tree"))
    for rep in range(2):
        for name, fn in cases.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            print(f"  pass {rep}  {name:16s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
