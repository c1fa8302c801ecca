"""The GEMM epilogue's fast forms (gemm_common.h::epilogue_fast, taken by slabs that lie inside the matrix) against the general
per-element path, BIT FOR BIT: the same random products -- every fast form, tiles inside and across the matrix edge, 128- and 256-row
kernels, NT and NN -- run in two child processes, one with LR2_GEMM_ABLATE=256 (fast forms off); each prints a hash of everything
the product wrote (the general path itself is checked against fp64 by tools/dbg/fuzz_kernels.py and tests/test_kernels_gpu.py).  usage: python tools/dbg/fuzz_epilogue.py [--n 300] [--seed 0]"""
import argparse
import hashlib
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FORMS = [  # (act, z, drop, resid, out, planes)
    (0, 0, 0, 0, 0, 1), (1, 0, 0, 0, 0, 1), (0, 0, 0, 1, 1, 0), (1, 1, 0, 0, 0, 1), (2, 0, 0, 0, 0, 1), (0, 0, 1, 1, 1, 0),
    (0, 0, 0, 0, 1, 0), (0, 0, 0, 1, 1, 1), (0, 0, 1, 0, 0, 1), (1, 1, 1, 0, 0, 1), (2, 0, 1, 0, 0, 1),
    (1, 0, 0, 0, 1, 0), (0, 0, 1, 0, 1, 1),          # not instantiated: must fall through to the general path
]


def child(n, seed):
    import numpy as np
    import torch
    from lr2ppo_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)

    def planes(x):
        return ops.split_planes(x.to(dev).contiguous(), ops.Planes.empty(x.shape[0], x.shape[1], dev))

    bad = 0
    for it in range(n):
        act, z, drop, resid, want_out, want_pl = FORMS[int(rng.integers(0, len(FORMS)))]
        nn = bool(rng.integers(0, 2))
        bm = [128, 256][int(rng.integers(0, 2))] if not nn else 128
        M = int(rng.integers(1, 1100))
        N = int(rng.integers(1, 200)) * 8
        K = int(rng.integers(1, 9)) * 64
        A = torch.randn(M, K, generator=g)
        B = torch.randn((K, N) if nn else (N, K), generator=g) * 0.1
        bias = torch.randn(N, generator=g) if rng.integers(0, 4) else None
        res = torch.randn(M, N, generator=g) if resid else None
        aux = torch.randn(M, N, generator=g) if act == 2 else None
        out = torch.full((M, N), float("nan"), device=dev) if want_out else None
        zt = torch.full((M, N), float("nan"), device=dev) if z else None
        pl = ops.Planes.empty(M, N, dev) if want_pl else None
        if pl is not None:
            pl.buf.fill_(0x7FC0)
        dr = ops.Drop(0.1, seed=9000 + it, site=5) if drop else None
        ops.gemm(planes(A), planes(B), out, M, N, K, trans_b=nn, bias=None if bias is None else bias.to(dev),
                 resid=None if res is None else res.to(dev), aux_z=None if aux is None else aux.to(dev), act=act, out_z=zt,
                 out_planes=pl, drop=dr, block_m=bm, splits=1, alpha=1.0 if rng.integers(0, 2) else 0.5)
        torch.cuda.synchronize()
        h = hashlib.sha1()
        for t in (out, zt, None if pl is None else pl.buf):
            if t is not None:
                h.update(t.cpu().numpy().tobytes())
        print(f"case {it} form {(act, z, drop, resid, want_out, want_pl)} {'NN' if nn else 'NT'} bm {bm} M {M} N {N} K {K} {h.hexdigest()}",
              flush=True)
    print("done", n, "bad", bad, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a.n, a.seed)
    outs = []
    for ablate in ("0", "256"):
        env = dict(os.environ, LR2_GEMM_ABLATE=ablate, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--n", str(a.n), "--seed", str(a.seed)], env=env,
                           capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-4000:])
            raise SystemExit(f"child with LR2_GEMM_ABLATE={ablate} failed")
        outs.append([l for l in r.stdout.splitlines() if l.startswith("case ")])
    diff = [(x, y) for x, y in zip(*outs) if x != y]
    for x, y in diff[:20]:
        print("MISMATCH\n  fast   ", x, "\n  general", y)
    print(f"{len(outs[0])} cases, {len(diff)} differ between the fast forms and the general path")
    raise SystemExit(1 if diff or len(outs[0]) != a.n else 0)


if __name__ == "__main__":
    main()
