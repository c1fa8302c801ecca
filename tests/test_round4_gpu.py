"""Round-4 kernel-level parity on a real MI355X: the ROW SPLIT of large NT products (whole rounds of 256 x 256 tiles + a tail on the
128- / 64-row kernels: csrc/gemm.hip::lr2_gemm)."""
import math
import os
import subprocess
import sys

import pytest
import torch

from conftest import REPO
from oracle import lr2ppo_oracle as O

pytestmark = pytest.mark.gpu


def _planes(ops, x, dev):
    return ops.split_planes(x.to(dev).contiguous(), ops.Planes.empty(x.shape[0], x.shape[1], dev))


@pytest.mark.parametrize("M,N,K", [(8448, 2048, 128), (8400, 2048, 64), (9000, 1796, 192), (12544, 3072, 64)])
def test_row_split_product_matches_fp64_with_every_epilogue(dev, M, N, K):
    """More than one round of 256 x 256 tiles with a last round less than half full: lr2_gemm sends the rows of the whole rounds to
    the 256 x 256 kernel and the rest to the general kernels (ragged M and N included).  Every output row against fp64 -- plain,
    bias + GELU with the kept pre-activation and a planes output, residual, accumulate, GELU' -- and the seam between the two
    launches (row M1) checked explicitly; a fused dropout mask keeps the product on one launch (its element index is relative to
    the launch's first row) and still matches the oracle's mask."""
    from lr2ppo_amd import ops
    from test_kernels_gpu import _close
    tn = (N + 255) // 256
    tiles = ((M + 255) // 256) * tn
    assert tiles > 256 and 0 < tiles % 256 < 128 and ops.use_gemm256(M, N, K)
    M1 = ((tiles // 256) * 256 // tn) * 256
    assert 0 < M1 < M
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a, w, bias = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.2, torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ap, wp = _planes(ops, a, dev), _planes(ops, w, dev)
    z_ref = a.double() @ w.double().t()
    atol = 6e-5 * math.sqrt(K)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ap, wp, out, M, N, K, block_m=256, splits=1)
    _close(out, z_ref, atol, 5e-5, "row split, plain")
    seam = slice(M1 - 2, M1 + 2)
    assert torch.isfinite(out[seam]).all() and (out[seam].double().cpu() - z_ref[seam]).abs().max() < atol + 5e-5 * z_ref[seam].abs().max()
    z, pl = torch.full((M, N), float("nan"), device=dev), ops.Planes.empty(M, N, dev)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), act=1, out_z=z, out_planes=pl, block_m=256, splits=1)
    _close(z, z_ref + bias.double(), atol, 5e-5, "z")
    _close(out, O.gelu_erf(z_ref + bias.double()), atol, 5e-5, "gelu")
    assert torch.equal(pl.buf, ops.split_planes(out, ops.Planes.empty(M, N, dev)).buf)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), resid=resid.to(dev), block_m=256, splits=1)
    _close(out, z_ref + bias.double() + resid.double(), atol, 5e-5, "resid")
    out.copy_(resid.to(dev))
    ops.gemm(ap, wp, out, M, N, K, accumulate=True, alpha=0.5, block_m=256, splits=1)
    _close(out, 0.5 * z_ref + resid.double(), atol, 5e-5, "accumulate")
    ops.gemm(ap, wp, out, M, N, K, act=2, aux_z=resid.to(dev), block_m=256, splits=1)
    x = resid.double()
    gp = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    _close(out, z_ref * gp, atol, 5e-5, "gelu'")
    drop = ops.Drop(0.1, seed=78, site=5)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), drop=drop, resid=resid.to(dev), block_m=256, splits=1)
    keep = torch.from_numpy(O.dropout_keep_mask(78, 5, M * N, 0.1)).view(M, N)
    _close(out, (z_ref + bias.double()) * keep.double() / 0.9 + resid.double(), atol, 5e-5, "dropout + resid (single launch)")


def _counts():
    import ctypes
    from lr2ppo_amd import _native
    c = (ctypes.c_uint64 * 3)()
    assert _native.lib().lr2_gemm_launch_counts(c) == 0
    return list(c)


def test_row_split_is_what_runs(dev):
    """lr2_gemm follows lr2_gemm_row_split_plan: one call = one launch of the 256 x 256 kernel (the plan's leading rows) + one of the
    general family (the rest); with a fused dropout mask, or a shape whose last round is well filled, one launch of the 256 x 256
    kernel alone.  The rows of the whole rounds carry the bits of the unsplit launch (same kernel, same tiles)."""
    import ctypes
    from lr2ppo_amd import _native, ops
    M, N, K = 8448, 2048, 128
    r, t = ctypes.c_int(), ctypes.c_int()
    assert _native.lib().lr2_gemm_row_split_plan(M, N, K, ctypes.byref(r), ctypes.byref(t)) == 0
    assert (r.value, t.value) == (8192, 64)
    g = torch.Generator().manual_seed(5)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    ap, wp = _planes(ops, a, dev), _planes(ops, w, dev)
    out, whole = torch.empty(M, N, device=dev), torch.empty(r.value, N, device=dev)
    c0 = _counts()
    ops.gemm(ap, wp, out, M, N, K, block_m=256, splits=1)
    c1 = _counts()
    assert [c1[i] - c0[i] for i in range(3)] == [1, 0, 1]
    ops.gemm(ops.Planes(ap.buf, r.value, K, lo_off=ap.lo_off), wp, whole, r.value, N, K, block_m=256, splits=1)     # 256 tiles: one round
    c2 = _counts()
    assert [c2[i] - c1[i] for i in range(3)] == [1, 0, 0]
    assert torch.equal(out[:r.value], whole)
    ops.gemm(ap, wp, out, M, N, K, block_m=256, splits=1, drop=ops.Drop(0.1, 3, 1))
    c3 = _counts()
    assert [c3[i] - c2[i] for i in range(3)] == [1, 0, 0]


# ---- persistent attention kernels (csrc/selfattn.hip: self_attn_persist_kernel, self_attn_bwd_{dq,dkv}_persist_kernel) ----
def _attn_case(dev, batch, heads, L, seed):
    from lr2ppo_amd import ops
    E = heads * 64
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(batch * L, 3 * E, generator=g) * 0.7).to(dev)
    qkv = ops.split_planes(x, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = (torch.rand(batch, L, generator=g) > 0.2).long()
    seg[:, 0] = 1
    return ops, E, qkv, seg.view(-1).to(dev), g, x


@pytest.mark.parametrize("batch,heads,L,p", [(300, 4, 97, 0.1), (40, 8, 224, 0.0), (256, 2, 33, 0.1), (64, 12, 196, 0.0)])
def test_persistent_attention_forward_equals_one_pair_kernel_and_fp64(dev, batch, heads, L, p):
    """At least one (sequence, head) pair per CU: lr2_self_attn_fwd runs the persistent 16-wave kernel (lr2_self_attn_plan says so);
    the first sequences alone (fewer pairs than CUs) run the one-pair kernel on the same rows -- context, planes output and
    log-sum-exp agree BIT FOR BIT (both call attn_phase_a / attn_phase_b; dropout masks are indexed by (sequence, head, query, key));
    eval-mode context against the fp64 softmax(Q K^T / 8 - 10000 pad) V of the reference (multi_headed_attn.py:61-74)."""
    ops, E, qkv, seg, g, x = _attn_case(dev, batch, heads, L, 31)
    nb = max(1, 200 // heads)
    assert ops.self_attn_plan(batch, heads, L)[0] and not ops.self_attn_plan(nb, heads, L)[0]
    dr = ops.Drop(p, 1234, 7) if p > 0 else None
    o, lse = torch.full((batch * L, E), float("nan"), device=dev), torch.full((batch * heads * L,), float("nan"), device=dev)
    ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=dr)
    op = ops.Planes.empty(batch * L, E, dev)
    ops.self_attn_fwd(qkv, seg, op, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr)
    assert torch.equal(op.buf, ops.split_planes(o, ops.Planes.empty(batch * L, E, dev)).buf)
    qs = ops.split_planes(x[:nb * L].contiguous(), ops.Planes.empty(nb * L, 3 * E, dev))
    o2, lse2 = torch.full((nb * L, E), float("nan"), device=dev), torch.full((nb * heads * L,), float("nan"), device=dev)
    ops.self_attn_fwd(qs, seg[:nb * L].contiguous(), o2, batch=nb, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse2, drop=dr)
    assert torch.equal(o[:nb * L], o2) and torch.equal(lse[:nb * heads * L], lse2)
    assert torch.isfinite(o).all() and torch.isfinite(lse).all()
    if p == 0.0:
        sl = slice((batch - 3) * L, batch * L)
        xx = qkv.to_float()[sl].double().cpu()
        q, k, v = (t.reshape(3, L, heads, 64).transpose(1, 2) for t in xx.split(E, dim=1))
        mask = (1.0 - (seg[sl].view(3, 1, 1, L) > 0).double().cpu()) * -10000.0
        ref = (torch.softmax(q @ k.transpose(-2, -1) / 8.0 + mask, dim=-1) @ v).transpose(1, 2).reshape(3 * L, E)
        assert (o[sl].double().cpu() - ref).abs().max() < 2e-5                   # measured 1.1e-6 - 2.8e-6


@pytest.mark.parametrize("batch,heads,L,p", [(300, 4, 97, 0.1), (40, 8, 224, 0.0), (256, 2, 33, 0.1), (64, 12, 196, 0.1)])
def test_streaming_attention_backward_matches_recomputing_kernels_and_fp64(dev, batch, heads, L, p):
    """lr2_self_attn_bwd given the forward's output planes and log-sum-exp (ABI 19: P = exp(S - lse), D = sum_d dO O, persistent 16-wave
    kernels with half-by-half LDS-DMA refill) against the recomputing kernels (o = None) on the same inputs with the forward's dropout
    mask -- relative L2 < 2e-5 per gradient (measured 1.9e-6 - 3.5e-6: D is summed over 64 head columns instead of L keys), D itself
    < 1e-4 -- and, in eval mode, against fp64 autograd of the reference expression (multi_headed_attn.py:61-74)."""
    ops, E, qkv, seg, g, _ = _attn_case(dev, batch, heads, L, 47)
    assert ops.self_attn_plan(batch, heads, L)[1]
    dr = ops.Drop(p, 4321, 3) if p > 0 else None
    o, lse = ops.Planes.empty(batch * L, E, dev), torch.full((batch * heads * L,), float("nan"), device=dev)
    ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=dr)
    do = ops.split_planes(torch.randn(batch * L, E, generator=g).to(dev), ops.Planes.empty(batch * L, E, dev))
    d_new, d_old = ops.Planes.empty(batch * L, 3 * E, dev), ops.Planes.empty(batch * L, 3 * E, dev)
    d_new.buf.fill_(0x7fc0)                                                     # bf16 NaN: every element must be written
    ws1, ws2, ws3 = (torch.empty(batch * heads * L, device=dev) for _ in range(3))
    lse_in = lse.clone()
    ops.self_attn_bwd(qkv, do, seg, d_new, lse_in, ws1, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr, o=o)
    assert torch.equal(lse_in, lse)                                             # an input: untouched
    ops.self_attn_bwd(qkv, do, seg, d_old, ws2, ws3, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=dr)
    gn, go = d_new.to_float(), d_old.to_float()
    assert torch.isfinite(gn).all()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())      # noqa: E731
    for i, name in enumerate(("dQ", "dK", "dV")):
        assert rel(gn[:, i * E:(i + 1) * E], go[:, i * E:(i + 1) * E]) < 2e-5, name
    assert rel(lse, ws2) < 1e-6 and rel(ws1, ws3) < 1e-4
    if p == 0.0:
        sl = slice((batch - 2) * L, batch * L)
        xx = qkv.to_float()[sl].double().cpu().requires_grad_(True)
        q, k, v = (t.reshape(2, L, heads, 64).transpose(1, 2) for t in xx.split(E, dim=1))
        mask = (1.0 - (seg[sl].view(2, 1, 1, L) > 0).double().cpu()) * -10000.0
        out = (torch.softmax(q @ k.transpose(-2, -1) / 8.0 + mask, dim=-1) @ v).transpose(1, 2).reshape(2 * L, E)
        out.backward(do.to_float()[sl].double().cpu())
        assert rel(gn[sl].cpu(), xx.grad) < 2e-5                                # measured 4.9e-6


@pytest.mark.parametrize("M,N,K", [(12544, 768, 768), (12500, 700, 1536), (6000, 512, 256), (200, 256, 64), (12544, 768, 3072)])
def test_192_row_tile_of_the_256_kernel_matches_fp64(dev, M, N, K):
    """Less than one round of 256-row tiles that a round of 192-row tiles fills better: the NT launcher runs gemm256_nt_kernel<0, 3>
    (wave tile 96 x 64; csrc/gemm256.hip).  Ragged M / N, every epilogue family against fp64, planes output == split(fp32 output)."""
    from lr2ppo_amd import ops
    from test_kernels_gpu import _close
    assert ((M + 255) // 256) * ((N + 255) // 256) < ((M + 191) // 192) * ((N + 255) // 256) <= 256      # the launcher's condition
    g = torch.Generator().manual_seed(M + K)
    a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    bias, resid = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ap, wp = _planes(ops, a, dev), _planes(ops, w, dev)
    z_ref = a.double() @ w.double().t()
    atol = 6e-5 * math.sqrt(K)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ap, wp, out, M, N, K, block_m=256, splits=1)
    _close(out, z_ref, atol, 5e-5, "plain")
    pl = ops.Planes.empty(M, N, dev)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), act=1, out_planes=pl, block_m=256, splits=1)
    _close(out, O.gelu_erf(z_ref + bias.double()), atol, 5e-5, "gelu")
    assert torch.equal(pl.buf, ops.split_planes(out, ops.Planes.empty(M, N, dev)).buf)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), resid=resid.to(dev), block_m=256, splits=1)
    _close(out, z_ref + bias.double() + resid.double(), atol, 5e-5, "resid")
    out.copy_(resid.to(dev))
    ops.gemm(ap, wp, out, M, N, K, accumulate=True, alpha=0.5, block_m=256, splits=1)
    _close(out, 0.5 * z_ref + resid.double(), atol, 5e-5, "accumulate")
