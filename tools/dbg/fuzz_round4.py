"""Randomised sweep of round 4's kernels against fp64 (run on the GPU box; seeds fixed):
  * lr2_gemm_mxfp8 on BOTH product kernels (the 256 x 256 LDS-DMA ring and the 128 x 128 LDS kernel; both scale-staging forms of the
    ring) with every output combination: fp32 (+ bias) (+ GELU) (+ residual), bf16 hi / lo planes, ONE bf16 plane, MX-FP8;
  * lr2_self_attn_fwd_bf16 (the fp8 mode's attention) for random (batch, heads, L <= 288, key masks);
  * the row split of large NT products of planes (lr2_gemm_row_split_plan) with the epilogues that may split.
    python tools/dbg/fuzz_round4.py [--n 40] [--seed 0]"""
import argparse
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lr2ppo_amd import _native, ops  # noqa: E402
from oracle import lr2ppo_oracle as O  # noqa: E402
from oracle.cpu_threads import fit_torch_threads  # noqa: E402

fit_torch_threads()          # one torch thread per usable core (cgroup quota): the fp64 references run on the host

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
g = torch.Generator().manual_seed(a.seed)
bad = 0


def planes(x):
    return ops.split_planes(x.to(dev).contiguous(), ops.Planes.empty(x.shape[0], x.shape[1], dev))


def fail(msg):
    global bad
    bad += 1
    print("MISMATCH", msg, flush=True)


# ---- MX-FP8 products ----
n_ring = 0
for it in range(a.n):
    ring = bool(rng.integers(0, 3))                            # two thirds on the ring kernel
    if ring:
        # just over one full round of 256 x 256 tiles (>= 80 % filled: the ring kernel's condition), ragged last tile row
        N = int(rng.integers(2, 17)) * 128
        tn = (N + 255) // 256
        rounds = int(rng.integers(1, 3))
        tm = (256 * rounds) // tn - int(rng.integers(0, 3))    # tm * tn in (0.8, 1] x 256 x rounds tiles
        while tm * tn < 0.8 * 256 * rounds or tm * tn < 256:
            tm += 1
        M = tm * 256 - int(rng.integers(0, 256))
        K = int(rng.integers(1, 9)) * 128 if rng.integers(0, 2) else int(rng.integers(1, 5)) * 512      # 4-byte gathers / 16-byte chunks from 2048
    else:
        M, N, K = int(rng.integers(1, 3000)), int(rng.integers(1, 17)) * 128, int(rng.integers(1, 13)) * 128
    while M * N * K > 9e10:
        K -= 128 if K <= 1024 else 512
    x = torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
    w = torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g) * 0.1 if rng.integers(0, 2) else None
    act = int(rng.integers(0, 2))
    resid = torch.randn(M, N, generator=g) if rng.integers(0, 3) == 0 else None
    xm, wm = ops.quant_mxfp8(x.to(dev)), ops.quant_mxfp8(w.to(dev))
    out = torch.full((M, N), float("nan"), device=dev)
    kw = dict(bias=None if bias is None else bias.to(dev), act=act, resid=None if resid is None else resid.to(dev))
    t256 = ((M + 255) // 256) * ((N + 255) // 256)
    n_ring += int(t256 >= 256 and t256 * 100 >= 80 * (-(-t256 // 256)) * 256)
    try:
        ops.gemm_mxfp8(xm, wm, out, **kw)
        da, db = xm.to_float().double().cpu(), wm.to_float().double().cpu()
        pre = da @ db.t() + (bias.double() if bias is not None else 0.0)
        want = O.gelu_erf(pre) if act else pre
        if resid is not None:
            want = want + resid.double()
        bound = 2e-3 * (da.abs() @ db.abs().t()) + 1e-5
        err = (out.double().cpu() - want).abs()
        if not bool((err <= bound).all()) or not bool(torch.isfinite(out).all()):
            fail(f"mx product M {M} N {N} K {K} act {act} bias {bias is not None} resid {resid is not None}: excess {(err - bound).max().item():.3e}")
        mode = int(rng.integers(0, 4))
        if mode == 0:
            pl = ops.Planes.empty(M, N, dev)
            ops.gemm_mxfp8(xm, wm, None, out_planes=pl, **kw)
            ok = torch.equal(pl.buf, ops.split_planes(out, ops.Planes.empty(M, N, dev)).buf)
        elif mode == 1:
            ob = torch.empty(M * N, dtype=torch.bfloat16, device=dev)
            ops.gemm_mxfp8(xm, wm, None, out_bf16=ob, **kw)
            ok = torch.equal(ob.view(M, N), out.to(torch.bfloat16))
        elif mode == 2:
            mx = ops.Mx8.empty(M, N, dev)
            ops.gemm_mxfp8(xm, wm, None, out_mx=mx, **kw)
            wq = ops.quant_mxfp8(out)
            ok = torch.equal(mx.q, wq.q) and torch.equal(mx.s, wq.s)
        else:
            mx, o2 = ops.Mx8.empty(M, N, dev), torch.empty(M, N, device=dev)
            ops.gemm_mxfp8(xm, wm, o2, out_mx=mx, **kw)
            wq = ops.quant_mxfp8(out)
            ok = torch.equal(mx.q, wq.q) and torch.equal(mx.s, wq.s) and torch.equal(o2, out)
        if not ok:
            fail(f"mx product output mode {mode} M {M} N {N} K {K} act {act} resid {resid is not None}")
    except Exception as e:                                     # noqa: BLE001
        fail(f"mx product raised M {M} N {N} K {K}: {e!r}"[:200])
print(f"mx products: {a.n} cases ({n_ring} on the ring kernel)", flush=True)

# ---- the fp8 mode's attention ----
for it in range(a.n):
    batch, heads, L = int(rng.integers(1, 5)), int(rng.integers(1, 5)), int(rng.integers(1, 289))
    E = heads * 64
    qkv = torch.cat([torch.randn(batch * L, E, generator=g) * 0.5, torch.randn(batch * L, E, generator=g) * 0.5,
                     torch.randn(batch * L, E, generator=g)], dim=1).to(torch.bfloat16)
    seg = (torch.rand(batch, L, generator=g) > 0.3 * float(rng.random())).long()
    seg[:, 0] = 1
    mask = (1.0 - (seg > 0).double()).view(batch, 1, 1, L) * -10000.0
    qh, kh, vh = (t.double().reshape(batch, L, heads, 64).transpose(1, 2) for t in qkv.split(E, dim=1))
    ref = (torch.softmax(qh @ kh.transpose(-2, -1) / 8.0 + mask, dim=-1) @ vh).transpose(1, 2).reshape(batch * L, E)
    out = torch.full((batch * L, E), float("nan"), device=dev)
    mx = ops.Mx8.empty(batch * L, E, dev)
    try:
        ops.self_attn_fwd_bf16(qkv.to(dev), seg.to(dev).view(-1), batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out=out, out_mx=mx)
        err = (out.double().cpu() - ref).abs().max().item()
        wq = ops.quant_mxfp8(out)
        if not (err < 6e-3 * max(1.0, float(vh.abs().max()))) or not torch.equal(mx.q, wq.q) or not torch.equal(mx.s, wq.s):
            fail(f"bf16 attention batch {batch} heads {heads} L {L}: err {err:.3e}")
    except Exception as e:                                     # noqa: BLE001
        fail(f"bf16 attention raised batch {batch} heads {heads} L {L}: {e!r}"[:200])
print(f"bf16 attention: {a.n} cases", flush=True)

# ---- row split ----
lib = _native.lib()
done = 0
for it in range(4 * a.n):
    if done >= a.n:
        break
    N = int(rng.integers(2, 13)) * 256 - int(rng.integers(0, 64)) * 4
    K = int(rng.integers(1, 5)) * 64
    tn = (N + 255) // 256
    rounds = int(rng.integers(1, 4))
    tm = (rounds * 256 + int(rng.integers(1, 127))) // tn + 1
    M = tm * 256 - int(rng.integers(0, 255))
    r, t = ctypes.c_int(), ctypes.c_int()
    lib.lr2_gemm_row_split_plan(M, N, K, ctypes.byref(r), ctypes.byref(t))
    if r.value == 0 or M * N > 30_000_000:
        continue
    done += 1
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2
    ref = A.double() @ B.double().t()
    epi = int(rng.integers(0, 6))
    bias = torch.randn(N, generator=g) if epi in (1, 2) else None
    resid = torch.randn(M, N, generator=g) if epi == 2 else None
    aux = torch.randn(M, N, generator=g) if epi == 4 else None
    base = torch.randn(M, N, generator=g) if epi == 5 else None
    out = base.to(dev).clone() if base is not None else torch.full((M, N), float("nan"), device=dev)
    pl = ops.Planes.empty(M, N, dev) if epi == 3 else None
    c0 = (ctypes.c_uint64 * 3)()
    c1 = (ctypes.c_uint64 * 3)()
    lib.lr2_gemm_launch_counts(c0)
    ops.gemm(planes(A), planes(B), out, M, N, K, bias=None if bias is None else bias.to(dev), resid=None if resid is None else resid.to(dev),
             act=1 if epi == 1 else (2 if epi == 4 else 0), out_planes=pl, aux_z=None if aux is None else aux.to(dev),
             accumulate=base is not None, block_m=256, splits=1)
    lib.lr2_gemm_launch_counts(c1)
    want = ref + (bias.double() if bias is not None else 0.0)
    if epi == 1:
        want = O.gelu_erf(want)
    if resid is not None:
        want = want + resid.double()
    if aux is not None:
        zz = aux.double()
        want = want * (0.5 * (1.0 + torch.erf(zz / math.sqrt(2.0))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2.0 * math.pi))
    if base is not None:
        want = want + base.double()
    err = (out.double().cpu() - want).abs()
    tol = 6e-5 * math.sqrt(K) + 5e-5 * want.abs()
    if [c1[i] - c0[i] for i in range(3)] != [1, 0, 1] or not bool((err <= tol).all()):
        fail(f"row split M {M} N {N} K {K} epi {epi} rows_256 {r.value}: launches {[c1[i] - c0[i] for i in range(3)]}, worst {float((err - tol).max()):.3e}")
    elif pl is not None and not torch.equal(pl.buf, ops.split_planes(out, ops.Planes.empty(M, N, dev)).buf):
        fail(f"row split planes output M {M} N {N} K {K}")
print(f"row split: {done} cases", flush=True)
print("no mismatch" if bad == 0 else f"{bad} MISMATCHES", flush=True)
sys.exit(1 if bad else 0)
