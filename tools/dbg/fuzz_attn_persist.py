"""Randomised sweep of the persistent attention kernels (run on the GPU box; seeds fixed):
  * lr2_self_attn_fwd, >= one (sequence, head) pair per CU (persistent 16-wave kernel) against the one-pair kernel run chunk by chunk
    (fewer pairs than CUs per call) on the same rows: context and log-sum-exp bit for bit; with dropout only the first chunk (the mask
    is indexed by the absolute sequence number);
  * lr2_self_attn_bwd given o + lse (persistent streaming kernels) against the recomputing kernels on the same inputs: L2 distance per
    gradient < 2e-5 of the largest gradient's norm, every element written;
  * lr2_self_attn_fwd_bf16 (MX-FP8 mode), persistent against one-pair chunk by chunk: fp32 context and MX bytes / scales bit for bit.
    python tools/dbg/fuzz_attn_persist.py [--n 60] [--seed 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
g = torch.Generator(device=dev).manual_seed(a.seed)
bad = 0


def fail(msg):
    global bad
    bad += 1
    print("MISMATCH", msg, flush=True)


def rel(x, y):
    return float((x.double() - y.double()).norm() / y.double().norm().clamp_min(1e-30))


for it in range(a.n):
    heads = int(rng.integers(1, 13))
    L = int(rng.integers(1, 225))
    batch = int(-(-256 // heads) * rng.integers(1, 7) + rng.integers(0, 40))      # 1 .. 7 pairs per workgroup, ragged
    while batch * L * heads * 64 * 3 > 6e8:
        batch -= 8
    if batch * heads < 256:
        continue
    E = heads * 64
    p = 0.1 if rng.integers(0, 2) else 0.0
    x = torch.randn(batch * L, 3 * E, device=dev, generator=g) * float(rng.uniform(0.2, 1.2))
    qkv = ops.split_planes(x, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = (torch.rand(batch, L, device=dev, generator=g) > float(rng.uniform(0, 0.5))).long()
    seg[:, 0] = 1
    seg = seg.view(-1)
    dr = ops.Drop(p, int(rng.integers(1, 1 << 30)), int(rng.integers(0, 50))) if p > 0 else None
    fwd_p, bwd_p = ops.self_attn_plan(batch, heads, L)
    if not (fwd_p and bwd_p):
        fail(f"plan says not persistent for batch {batch} heads {heads} L {L}")
        continue
    o = torch.full((batch * L, E), float("nan"), device=dev)
    lse = torch.full((batch * heads * L,), float("nan"), device=dev)
    ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=dr)
    nb = max(1, 255 // heads)
    ok = bool(torch.isfinite(o).all()) and bool(torch.isfinite(lse).all())
    for c0 in range(0, batch if p == 0.0 else nb, nb):
        c1 = min(batch, c0 + nb)
        n = c1 - c0
        qs = ops.split_planes(x[c0 * L:c1 * L].contiguous(), ops.Planes.empty(n * L, 3 * E, dev))
        o2, l2 = torch.empty(n * L, E, device=dev), torch.empty(n * heads * L, device=dev)
        ops.self_attn_fwd(qs, seg[c0 * L:c1 * L].contiguous(), o2, batch=n, heads=heads, L=L, head_dim=64, scale=0.125, lse=l2, drop=dr)
        ok = ok and torch.equal(o[c0 * L:c1 * L], o2) and torch.equal(lse[c0 * heads * L:c1 * heads * L], l2)
    if not ok:
        fail(f"forward batch {batch} heads {heads} L {L} p {p}")
    # backward
    op = ops.split_planes(o, ops.Planes.empty(batch * L, E, dev))
    do = ops.split_planes(torch.randn(batch * L, E, device=dev, generator=g), ops.Planes.empty(batch * L, E, dev))
    d_new, d_old = ops.Planes.empty(batch * L, 3 * E, dev), ops.Planes.empty(batch * L, 3 * E, dev)
    d_new.buf.fill_(0x7fc0)
    w1, w2, w3 = (torch.empty(batch * heads * L, device=dev) for _ in range(3))
    ops.self_attn_bwd(qkv, do, seg, d_new, lse.clone(), w1, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr, o=op)
    ops.self_attn_bwd(qkv, do, seg, d_old, w2, w3, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr)
    gn, go = d_new.to_float(), d_old.to_float()
    # relative to the largest of the three gradients: at L = 1 (softmax of one key) dQ and dK are exactly 0 in the recomputing kernels
    # (D is the same sum as dP there) and ~1e-7 when D comes from the rounded forward output
    scale_ = max(float(go[:, i * E:(i + 1) * E].double().norm()) for i in range(3)) + 1e-30
    rr = [float((gn[:, i * E:(i + 1) * E].double() - go[:, i * E:(i + 1) * E].double()).norm()) / scale_ for i in range(3)]
    if not bool(torch.isfinite(gn).all()) or max(rr) > 2e-5:
        fail(f"backward batch {batch} heads {heads} L {L} p {p}: rel {rr}")
print(f"split-bf16 attention: {a.n} cases", flush=True)

for it in range(a.n):
    heads = int(rng.integers(1, 17))
    L = int(rng.integers(1, 289))
    batch = int(-(-256 // heads) * rng.integers(1, 5) + rng.integers(0, 40))
    E = heads * 64
    qkv = (torch.randn(batch * L, 3 * E, device=dev, generator=g) * float(rng.uniform(0.2, 1.0))).to(torch.bfloat16)
    seg = (torch.rand(batch, L, device=dev, generator=g) > float(rng.uniform(0, 0.5))).long()
    seg[:, 0] = 1
    seg = seg.view(-1)
    out, mx = torch.full((batch * L, E), float("nan"), device=dev), ops.Mx8.empty(batch * L, E, dev)
    ops.self_attn_fwd_bf16(qkv, seg, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out=out, out_mx=mx)
    nb = max(1, 255 // heads)
    ok = bool(torch.isfinite(out).all())
    for c0 in range(0, batch, nb):
        c1 = min(batch, c0 + nb)
        n = c1 - c0
        o2, m2 = torch.empty(n * L, E, device=dev), ops.Mx8.empty(n * L, E, dev)
        ops.self_attn_fwd_bf16(qkv[c0 * L:c1 * L].contiguous(), seg[c0 * L:c1 * L].contiguous(), batch=n, heads=heads, L=L, head_dim=64,
                               scale=0.125, out=o2, out_mx=m2)
        ok = ok and torch.equal(out[c0 * L:c1 * L], o2) and torch.equal(mx.q.view(-1)[c0 * L * E:c1 * L * E], m2.q.view(-1)) \
            and torch.equal(mx.s.view(-1)[c0 * L * E // 32:c1 * L * E // 32], m2.s.view(-1))
    if not ok:
        fail(f"mx attention batch {batch} heads {heads} L {L}")
print(f"MX-FP8 mode attention: {a.n} cases", flush=True)
print("no mismatch" if bad == 0 else f"{bad} MISMATCHES", flush=True)
sys.exit(1 if bad else 0)
