"""The persistent attention forward (csrc/selfattn.hip::self_attn_persist_kernel, >= one (sequence, head) pair per CU) against the
one-pair kernel (fewer pairs than CUs) on the same sequences -- bit for bit -- and against fp64; then us per launch at the `value`
loop's shapes.  LR2_ATTN_PERSIST=0 in the environment times the one-pair kernel instead.
    python tools/dbg/attn_persist_check.py"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (batch, heads, L) in ((512, 12, 197), (64, 12, 196), (300, 4, 97), (40, 8, 224), (256, 2, 33), (37, 12, 130)):
    E = heads * 64
    x = torch.randn(batch * L, 3 * E, device=dev, generator=g) * 0.7
    qkv = ops.split_planes(x, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = (torch.rand(batch, L, device=dev, generator=g) > 0.2).long()
    seg[:, 0] = 1
    seg = seg.view(-1)
    o = torch.full((batch * L, E), float("nan"), device=dev)
    op = ops.Planes.empty(batch * L, E, dev)
    ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125)
    ops.self_attn_fwd(qkv, seg, op, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125)
    # the first sequences alone: fewer pairs than CUs -> the one-pair kernel
    nb = max(1, 200 // heads)
    qs = ops.split_planes(x[:nb * L].contiguous(), ops.Planes.empty(nb * L, 3 * E, dev))
    o2 = torch.full((nb * L, E), float("nan"), device=dev)
    ops.self_attn_fwd(qs, seg[:nb * L].contiguous(), o2, batch=nb, heads=heads, L=L, head_dim=64, scale=0.125)
    same = torch.equal(o[:nb * L], o2)
    pl = ops.split_planes(o, ops.Planes.empty(batch * L, E, dev))
    same_pl = torch.equal(pl.buf, op.buf)
    # fp64 on the last 3 sequences
    sl = slice((batch - 3) * L, batch * L)
    xx = (qkv.to_float() if hasattr(qkv, "to_float") else x)[sl].double().cpu()
    q, k, v = (t.reshape(3, L, heads, 64).transpose(1, 2) for t in xx.split(E, dim=1))
    mask = (1.0 - (seg[sl].view(3, 1, 1, L) > 0).double().cpu()) * -10000.0
    ref = (torch.softmax(q @ k.transpose(-2, -1) / 8.0 + mask, dim=-1) @ v).transpose(1, 2).reshape(3 * L, E)
    err = float((o[sl].double().cpu() - ref).abs().max())
    # train mode: probability dropout + the log-sum-exp the backward reads
    dr = ops.Drop(0.1, 1234, 7)
    od, lse = torch.empty_like(o), torch.full((batch * heads * L,), float("nan"), device=dev)
    od2, lse2 = torch.empty_like(o2), torch.full((nb * heads * L,), float("nan"), device=dev)
    ops.self_attn_fwd(qkv, seg, od, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=dr)
    ops.self_attn_fwd(qs, seg[:nb * L].contiguous(), od2, batch=nb, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse2, drop=dr)
    same_drop = torch.equal(od[:nb * L], od2) and torch.equal(lse[:nb * heads * L], lse2) and not torch.equal(od, o)
    same = same and same_drop
    ok = same and same_pl and err < 2e-5 and bool(torch.isfinite(o).all()) and bool(torch.isfinite(lse).all())
    bad += 0 if ok else 1
    print(f"batch {batch} heads {heads} L {L}: first {nb} sequences equal the one-pair kernel: {same}; planes output == split(fp32 output): {same_pl}; "
          f"max |err| vs fp64 {err:.2e}  {'ok' if ok else 'MISMATCH'}", flush=True)

for (batch, heads, L) in ((512, 12, 197), (64, 12, 196)):
    E = heads * 64
    qkv = ops.split_planes(torch.randn(batch * L, 3 * E, device=dev, generator=g) * 0.7, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = torch.ones(batch * L, dtype=torch.int64, device=dev)
    op = ops.Planes.empty(batch * L, E, dev)
    ts = []
    for rep in range(3):
        for _ in range(3):
            ops.self_attn_fwd(qkv, seg, op, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.self_attn_fwd(qkv, seg, op, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 20 * 1e3)
    print(f"batch {batch} heads {heads} L {L}: {min(ts):7.1f} us per launch (LR2_ATTN_PERSIST={os.environ.get('LR2_ATTN_PERSIST', '1')})", flush=True)
sys.exit(1 if bad else 0)
