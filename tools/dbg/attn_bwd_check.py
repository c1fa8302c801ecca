"""The streaming / persistent attention backward (lr2_self_attn_bwd given the forward's output and log-sum-exp, ABI 19) against the
recomputing kernels (o = None) and against fp64 autograd, eval and train mode (dropout 0.1 with the forward's mask); then us per call at
the training shapes.    python tools/dbg/attn_bwd_check.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bad = 0


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


for (batch, heads, L, p) in () if "--time-only" in sys.argv else ((512, 12, 197, 0.0), (512, 12, 197, 0.1), (64, 12, 196, 0.1), (300, 4, 97, 0.1), (40, 8, 224, 0.0), (256, 2, 33, 0.1),
                             (37, 12, 130, 0.1)):
    E = heads * 64
    x = torch.randn(batch * L, 3 * E, device=dev, generator=g) * 0.7
    qkv = ops.split_planes(x, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = (torch.rand(batch, L, device=dev, generator=g) > 0.2).long()
    seg[:, 0] = 1
    seg = seg.view(-1)
    dr = ops.Drop(p, 4321, 3) if p > 0 else None
    o = ops.Planes.empty(batch * L, E, dev)
    lse = torch.full((batch * heads * L,), float("nan"), device=dev)
    ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=dr)
    do = ops.split_planes(torch.randn(batch * L, E, device=dev, generator=g), ops.Planes.empty(batch * L, E, dev))
    d_new, d_old = ops.Planes.empty(batch * L, 3 * E, dev), ops.Planes.empty(batch * L, 3 * E, dev)
    d_new.buf.fill_(0x7fc0)
    d_old.buf.fill_(0x7fc0)
    ws1, ws2, ws3 = (torch.empty(batch * heads * L, device=dev) for _ in range(3))
    ops.self_attn_bwd(qkv, do, seg, d_new, lse.clone(), ws1, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr, o=o)
    ops.self_attn_bwd(qkv, do, seg, d_old, ws2, ws3, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr)
    gn, go = d_new.to_float(), d_old.to_float()
    plan = ops.self_attn_plan(batch, heads, L)
    r_all = [rel(gn[:, i * E:(i + 1) * E], go[:, i * E:(i + 1) * E]) for i in range(3)]
    r_lse, r_d = rel(lse, ws2), rel(ws1, ws3)
    msg = f"batch {batch} heads {heads} L {L} p {p}: plan (fwd, bwd persistent) {plan}; new vs recomputing kernels: dQ {r_all[0]:.1e} dK {r_all[1]:.1e} dV {r_all[2]:.1e}, lse {r_lse:.1e}, D {r_d:.1e}"
    ok = max(r_all) < 2e-5 and r_lse < 1e-6 and r_d < 1e-4 and bool(torch.isfinite(gn).all())
    if p == 0.0:
        # fp64 autograd on the last 2 sequences
        sl = slice((batch - 2) * L, batch * L)
        xx = qkv.to_float()[sl].double().cpu().requires_grad_(True)
        q, k, v = (t.reshape(2, L, heads, 64).transpose(1, 2) for t in xx.split(E, dim=1))
        mask = (1.0 - (seg[sl].view(2, 1, 1, L) > 0).double().cpu()) * -10000.0
        out = (torch.softmax(q @ k.transpose(-2, -1) / 8.0 + mask, dim=-1) @ v).transpose(1, 2).reshape(2 * L, E)
        out.backward(do.to_float()[sl].double().cpu())
        r64 = rel(gn[sl].cpu(), xx.grad)
        msg += f"; vs fp64 autograd {r64:.1e}"
        ok = ok and r64 < 2e-5
    bad += 0 if ok else 1
    print(msg + ("  ok" if ok else "  MISMATCH"), flush=True)

for (batch, heads, L) in ((512, 12, 197), (64, 12, 196)):
    E = heads * 64
    qkv = ops.split_planes(torch.randn(batch * L, 3 * E, device=dev, generator=g) * 0.7, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = torch.ones(batch * L, dtype=torch.int64, device=dev)
    o, do, dqkv = ops.Planes.empty(batch * L, E, dev), ops.Planes.empty(batch * L, E, dev), ops.Planes.empty(batch * L, 3 * E, dev)
    ops.split_planes(torch.randn(batch * L, E, device=dev, generator=g), do)
    lse, ws = torch.empty(batch * heads * L, device=dev), torch.empty(batch * heads * L, device=dev)
    dr = ops.Drop(0.1, 99, 1)
    ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=dr)
    for name, kw in (("given o + lse", dict(o=o)), ("recomputing", dict())):
        ts = []
        for rep in range(3):
            for _ in range(2):
                ops.self_attn_bwd(qkv, do, seg, dqkv, lse, ws, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr, **kw)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.self_attn_bwd(qkv, do, seg, dqkv, lse, ws, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr, **kw)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10 * 1e3)
        print(f"batch {batch} heads {heads} L {L}, dropout 0.1, {name}: {min(ts):8.1f} us per backward", flush=True)
sys.exit(1 if bad else 0)
