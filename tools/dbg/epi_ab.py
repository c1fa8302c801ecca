"""A/B of the GEMM epilogue between two library builds of one ABI (LR2_AB_LIB=/path/to/other/liblr2ppo_hip.so; run once per build):
the encoder's and the head's token products on the 256 x 256 kernel with each epilogue form; us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lr2ppo_amd import _native, ops  # noqa: E402

if os.environ.get("LR2_AB_LIB"):            # a switch of THIS tool: another build of the same ABI, loaded explicitly
    _native.use_library(os.environ["LR2_AB_LIB"])


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)

    def planes(r, c, s=1.0):
        return ops.split_planes(torch.randn(r, c, device=dev, generator=g) * s, ops.Planes.empty(r, c, dev))

    print("library:", os.environ.get("LR2_AB_LIB", "in-tree"))
    for M in (100864, 12544):
        x768, x3072 = planes(M, 768), planes(M, 3072)
        w_qkv, w_proj, w_f1, w_f2 = planes(2304, 768, 0.02), planes(768, 768, 0.02), planes(3072, 768, 0.02), planes(768, 3072, 0.02)
        b768, b2304, b3072 = (torch.randn(n, device=dev, generator=g) * 0.02 for n in (768, 2304, 3072))
        resid = torch.randn(M, 768, device=dev, generator=g)
        o768, o2304 = torch.empty(M, 768, device=dev), ops.Planes.empty(M, 2304, dev)
        p3072, z3072 = ops.Planes.empty(M, 3072, dev), torch.randn(M, 3072, device=dev, generator=g)
        p768 = ops.Planes.empty(M, 768, dev)
        cases = {
            "QKV  bias -> planes": lambda: ops.gemm(x768, w_qkv, None, M, 2304, 768, bias=b2304, out_planes=o2304, block_m=256, splits=1),
            "proj bias + resid -> fp32": lambda: ops.gemm(x768, w_proj, o768, M, 768, 768, bias=b768, resid=resid, block_m=256, splits=1),
            "FFN1 bias + GELU -> planes": lambda: ops.gemm(x768, w_f1, None, M, 3072, 768, bias=b3072, act=1, out_planes=p3072, block_m=256, splits=1),
            "FFN2 bias + resid -> fp32": lambda: ops.gemm(x3072, w_f2, o768, M, 768, 3072, bias=b768, resid=resid, block_m=256, splits=1),
            "FFN2 dgrad GELU' -> planes": lambda: ops.gemm(x768, w_f1, None, M, 3072, 768, act=2, aux_z=z3072, out_planes=p3072, block_m=256, splits=1),
            "proj dgrad accumulate": lambda: ops.gemm(x768, w_proj, o768, M, 768, 768, accumulate=True, block_m=256, splits=1),
            "FFN1 bias + GELU + z -> planes (train)": lambda: ops.gemm(x768, w_f1, None, M, 3072, 768, bias=b3072, act=1, out_z=z3072, out_planes=p3072, block_m=256, splits=1),
            "proj bias + dropout + resid (train)": lambda: ops.gemm(x768, w_proj, o768, M, 768, 768, bias=b768, resid=resid, drop=ops.Drop(0.1, 77, 3), block_m=256, splits=1),
            "FFN1 dgrad plain -> fp32": lambda: ops.gemm(x3072, w_f2, o768, M, 768, 3072, block_m=256, splits=1),
            "FFN2 bias + resid -> fp32 + planes": lambda: ops.gemm(x3072, w_f2, o768, M, 768, 3072, bias=b768, resid=resid, out_planes=p768, block_m=256, splits=1),
        }
        for name, fn in cases.items():
            best = 1e9
            for rep in range(3):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    fn()
                e.record()
                torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 20 * 1e3)
            outs = (o768, o2304.buf, p3072.buf, p768.buf)          # bit-level checksum of everything a case may have written
            chk = sum(int(t.view(torch.int16).to(torch.int64).sum()) for t in outs) & 0xFFFFFFFFFFFF
            print(f"  M {M:6d}  {name:36s} {best:8.1f} us   checksum {chk:012x}")


if __name__ == "__main__":
    main()
