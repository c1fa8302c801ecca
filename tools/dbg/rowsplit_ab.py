#!/usr/bin/env python3
"""A/B in one process: an NT product of planes at M = 12 544 (the heads / RoBERTa at 2 tags: 0.6-2.3 rounds of 256 x 256 tiles) as ONE
launch (today's ops.choose_tiling) against a ROW SPLIT -- the first M1 rows (whole rounds of the 256 x 256 kernel) + the remaining rows
on the 128- / 64-row kernels (2-3 workgroups per CU, short tiles).  usage: python tools/dbg/rowsplit_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def planes(x):
    return ops.split_planes(x, ops.Planes.empty(x.shape[0], x.shape[1], dev))


def rows(p, r0, r1):
    return ops.Planes(p.buf[r0 * p.cols:], r1 - r0, p.cols, lo_off=p.lo_off)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    g = torch.Generator(device=dev).manual_seed(0)
    ws = torch.empty(8 * 12544 * 3072, device=dev)
    for (M, N, K) in [(12544, 3072, 768), (12544, 2304, 768), (12544, 768, 768), (12544, 768, 3072), (12544, 1024, 1024), (25088, 768, 3072),
                      (25088, 3072, 768), (6272, 3072, 768)]:
        A = planes(torch.randn(M, K, device=dev, generator=g))
        Wf = torch.randn(N, K, device=dev, generator=g)
        W = planes(Wf)
        Wt = ops.split_planes_t(Wf, ops.Planes.empty(K, N, dev))
        bias = torch.randn(N, device=dev, generator=g)
        out = ops.Planes.empty(M, N, dev)
        ref = ops.Planes.empty(M, N, dev)

        def one(a, o, m, nn):
            if nn:
                bm, sp = ops.choose_tiling(m, N, K, False, True)
                ops.gemm(a, Wt, None, m, N, K, trans_b=True, ldb=N, bias=bias, act=1, out_planes=o, block_m=bm, splits=sp, splitk_ws=ws)
            else:
                bm, sp = ops.choose_tiling(m, N, K, False, False)
                ops.gemm(a, W, None, m, N, K, bias=bias, act=1, out_planes=o, block_m=bm, splits=sp, splitk_ws=ws)

        res = {}
        res["NT default"] = timeit(lambda: one(A, ref, M, False))
        res["NN default"] = timeit(lambda: one(A, out, M, True))
        res["NT 256 forced"] = timeit(lambda: ops.gemm(A, W, None, M, N, K, bias=bias, act=1, out_planes=out, block_m=256, splits=1))
        tn = (N + 255) // 256
        tiles_m = (M + 255) // 256
        for rounds in range(1, (tiles_m * tn) // 256 + 1):
            tr = (rounds * 256) // tn                    # tile rows that fit `rounds` rounds
            for tr_ in (tr, tr - 1):
                M1 = tr_ * 256
                if M1 <= 0 or M1 >= M:
                    continue
                for nn in (False, True):
                    def split(M1=M1, nn=nn):
                        ops.gemm(rows(A, 0, M1), W, None, M1, N, K, bias=bias, act=1, out_planes=rows(out, 0, M1), block_m=256, splits=1)
                        one(rows(A, M1, M), rows(out, M1, M), M - M1, nn)
                    res[f"split M1={M1} ({tr_ * tn} tiles) + {'NN' if nn else 'NT'} tail {M - M1}"] = timeit(split)
        best = min(res.values())
        print(f"--- M={M} N={N} K={K} ({tiles_m * tn} tiles of 256x256)")
        for k, v in res.items():
            print(f"   {k:55s} {v:8.1f} us {'  <-- best' if v == best else ''}", flush=True)


if __name__ == "__main__":
    main()
