"""Per-kernel parity on a real MI355X: every C-ABI entry point against the CPU oracle / fp64 torch.

Tolerances: split-bf16 (passes=3) GEMMs carry ~17 mantissa bits per operand, so a K-term dot product of
O(1) terms is off by ~6e-6*sqrt(K) rms -> atol 6e-5*sqrt(K) (10 sigma); everything else is plain fp32
arithmetic -> 1e-5.  Index / mask work is checked bit-exact.
"""
import math

import numpy as np
import pytest
import torch

from oracle import lr2ppo_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from lr2ppo_amd import ops as _ops
    return _ops


def _rand(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale)


def _close(got, ref, atol, rtol, what=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = err > bound
    assert not bad.any(), f"{what}: max err {err.max().item():.3e} (ref scale {ref.abs().max().item():.3e}), {int(bad.sum())} bad"


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K,bm,splits", [(200, 256, 192, 128, 1), (50, 128, 2560, 64, 5), (128, 384, 64, 128, 1),
                                              (333, 128, 768, 128, 3), (64, 256, 1024, 64, 1)])
def test_gemm_nt_forward(ops, dev, M, N, K, bm, splits):
    g = torch.Generator().manual_seed(M * 7 + K)
    a, b = _rand(g, M, K), _rand(g, N, K)
    ref = a.double() @ b.double().t()
    out = torch.full((M, N), float("nan"), device=dev)
    ws = torch.empty(max(1, splits) * M * N, device=dev)
    ops.gemm(a.to(dev), b.to(dev), out, M, N, K, splits=splits, block_m=bm, splitk_ws=ws, passes=3)
    _close(out, ref, atol=6e-5 * math.sqrt(K), rtol=5e-5, what="x3")
    out1 = torch.empty((M, N), device=dev)
    ops.gemm(a.to(dev), b.to(dev), out1, M, N, K, splits=splits, block_m=bm, splitk_ws=ws, passes=1)
    ref1 = a.bfloat16().double() @ b.bfloat16().double().t()
    _close(out1, ref1, atol=1e-4 * math.sqrt(K), rtol=1e-4, what="x1 vs bf16-rounded inputs")


@pytest.mark.parametrize("M,N,K,bm", [(130, 256, 192, 128), (64, 1280, 128, 64), (300, 128, 64, 128)])
def test_gemm_nn_dgrad(ops, dev, M, N, K, bm):
    """dx[M,N] = dy[M,K] @ W[K,N]  (W is an nn.Linear weight [out=K, in=N])."""
    g = torch.Generator().manual_seed(M + N + K)
    a, w = _rand(g, M, K), _rand(g, K, N)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a.to(dev), w.to(dev), out, M, N, K, trans_b=True, block_m=bm, splits=1)
    _close(out, a.double() @ w.double(), atol=6e-5 * math.sqrt(K), rtol=5e-5, what="NN")


@pytest.mark.parametrize("M,N,K,bm", [(256, 384, 300, 128), (128, 128, 64, 128), (64, 256, 37, 64), (384, 128, 1000, 128)])
def test_gemm_tn_wgrad_ragged_contraction(ops, dev, M, N, K, bm):
    """dW[M,N] = dy[K,M]^T @ x[K,N]; K (token count) is ragged -> zero filled by the buffer range check."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    a, b = _rand(g, K, M), _rand(g, K, N)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a.to(dev), b.to(dev), out, M, N, K, trans_a=True, trans_b=True, block_m=bm, splits=1)
    _close(out, a.double().t() @ b.double(), atol=6e-5 * math.sqrt(K), rtol=5e-5, what="TN")


def test_gemm_tn_splitk(ops, dev):
    g = torch.Generator().manual_seed(5)
    K, M, N = 1500, 128, 256
    a, b = _rand(g, K, M), _rand(g, K, N)
    out = torch.empty((M, N), device=dev)
    ws = torch.empty(4 * M * N, device=dev)
    ops.gemm(a.to(dev), b.to(dev), out, M, N, K, trans_a=True, trans_b=True, splits=4, splitk_ws=ws)
    _close(out, a.double().t() @ b.double(), atol=6e-5 * math.sqrt(K), rtol=5e-5)


def test_gemm_epilogues(ops, dev):
    g = torch.Generator().manual_seed(17)
    M, N, K = 150, 256, 128
    a, w, bias, resid = _rand(g, M, K), _rand(g, N, K, scale=0.1), _rand(g, N), _rand(g, M, N)
    ad, wd, bd, rd = a.to(dev), w.to(dev), bias.to(dev), resid.to(dev)
    z_ref = a.double() @ w.double().t() + bias.double()
    # bias + GELU with saved pre-activation
    out, z = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(ad, wd, out, M, N, K, bias=bd, act=1, out_z=z)
    _close(z, z_ref, 6e-5, 5e-5, "z")
    _close(out, O.gelu_erf(z_ref), 6e-5, 5e-5, "gelu")
    # bias + dropout + residual   (XiT sites 0/2: x = dropout(linear) + res)
    drop = ops.Drop(0.1, seed=1234, site=7)
    ops.gemm(ad, wd, out, M, N, K, bias=bd, drop=drop, resid=rd)
    keep = torch.from_numpy(O.dropout_keep_mask(1234, 7, M * N, 0.1)).view(M, N)
    _close(out, z_ref * keep.double() / 0.9 + resid.double(), 6e-5, 5e-5, "dropout+resid")
    assert 0.85 < keep.float().mean() < 0.95
    # GELU then dropout (FFN site 1)
    ops.gemm(ad, wd, out, M, N, K, bias=bd, act=1, drop=drop)
    _close(out, O.gelu_erf(z_ref) * keep.double() / 0.9, 6e-5, 5e-5, "gelu+dropout")
    # backward through dropout + GELU: dz = (acc * mask/(1-p)) * gelu'(z)
    zt = z_ref.clone().requires_grad_(True)
    O.gelu_erf(zt).sum().backward()
    ops.gemm(ad, wd, out, M, N, K, act=2, aux_z=z, drop=drop)
    _close(out, (a.double() @ w.double().t()) * keep.double() / 0.9 * zt.grad, 6e-5, 5e-5, "dgelu")
    # accumulate + alpha
    base = _rand(g, M, N)
    acc = base.to(dev).clone()
    ops.gemm(ad, wd, acc, M, N, K, accumulate=True, alpha=0.5)
    _close(acc, base.double() + 0.5 * (a.double() @ w.double().t()), 6e-5, 5e-5, "accumulate")


def test_gemm_rejects_bad_shapes(ops, dev):
    from lr2ppo_amd._native import NativeError
    a = torch.zeros(8, 100, device=dev)
    b = torch.zeros(128, 100, device=dev)
    with pytest.raises(NativeError):
        ops.gemm(a, b, torch.zeros(8, 128, device=dev), 8, 128, 100)       # K % 64 != 0
    with pytest.raises(NativeError):
        ops.gemm(torch.zeros(8, 66, device=dev), torch.zeros(128, 66, device=dev), torch.zeros(8, 128, device=dev), 8, 128, 64, lda=66, ldb=66)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_ragged_n_and_m(ops, dev, ta, tb):
    g = torch.Generator().manual_seed(11)
    M, N, K = 72, 200, 64
    a = _rand(g, K, M) if ta else _rand(g, M, K)
    b = _rand(g, K, N) if tb else _rand(g, N, K)
    ref = (a.double().t() if ta else a.double()) @ (b.double() if tb else b.double().t())
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a.to(dev), b.to(dev), out, M, N, K, trans_a=ta, trans_b=tb, splits=1)
    _close(out, ref, atol=6e-5 * math.sqrt(K), rtol=5e-5, what=f"ragged ta={ta} tb={tb}")


# ------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,D", [(37, 768), (9, 64), (130, 1024)])
def test_layernorm_fwd_both_semantics(ops, dev, rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x, gam, bet = _rand(g, rows, D) * 2 + 0.5, _rand(g, D), _rand(g, D)
    out = torch.empty(rows, D, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), out, mean, rstd, rows=rows, D=D, eps=1e-5, mode=0)
    _close(out, O.layernorm_torch(x.double(), gam.double(), bet.double()), 1e-5, 1e-5, "torch LN")
    _close(mean, x.double().mean(-1), 1e-6, 1e-6)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), out, None, None, rows=rows, D=D, eps=1e-6, mode=1)
    _close(out, O.layernorm_tp(x.double(), gam.double(), bet.double()), 1e-5, 1e-5, "TP LN")


def test_layernorm_fwd_writes_into_concat_layout(ops, dev):
    g = torch.Generator().manual_seed(3)
    n, L, D, extra = 3, 5, 64, 2
    x, gam, bet = _rand(g, n * L, D), _rand(g, D), _rand(g, D)
    flat = torch.zeros(n, (L + extra) * D, device=dev)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), flat, rows=n * L, D=D, group=L, group_stride=(L + extra) * D)
    ref = O.layernorm_torch(x, gam, bet).view(n, L * D)
    _close(flat[:, : L * D], ref, 1e-5, 1e-5)
    assert torch.all(flat[:, L * D:] == 0)


@pytest.mark.parametrize("rows,D,drop_p", [(50, 768, 0.0), (1000, 768, 0.1), (7, 64, 0.0)])
def test_layernorm_bwd(ops, dev, rows, D, drop_p):
    g = torch.Generator().manual_seed(rows)
    x, gam, bet, dy, rg = _rand(g, rows, D), _rand(g, D), _rand(g, D), _rand(g, rows, D), _rand(g, rows, D)
    xt, gt, bt = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    O.layernorm_torch(xt, gt, bt).backward(dy.double())
    xd = x.to(dev)
    out = torch.empty(rows, D, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.layernorm_fwd(xd, gam.to(dev), bet.to(dev), out, mean, rstd, rows=rows, D=D)
    dx, dxm = torch.empty(rows, D, device=dev), ops.Planes.empty(rows, D, dev)
    dgam, dbet = torch.empty(D, device=dev), torch.empty(D, device=dev)
    partials = torch.empty(ops.LN_BWD_BLOCKS * 2 * D, device=dev)
    drop = ops.Drop(drop_p, 99, 4) if drop_p > 0 else None
    ops.layernorm_bwd(dy.to(dev), xd, gam.to(dev), mean, rstd, dx, partials, dgam, dbet, rows=rows, D=D,
                      resid_grad=rg.to(dev), dx_planes=dxm, drop=drop)
    ref_dx = xt.grad + rg.double()
    _close(dx, ref_dx, 2e-5, 2e-5, "dx")
    _close(dgam, gt.grad, 1e-4, 1e-4, "dgamma")
    _close(dbet, bt.grad, 1e-4, 1e-4, "dbeta")
    if drop_p > 0:
        keep = torch.from_numpy(O.dropout_keep_mask(99, 4, rows * D, drop_p)).view(rows, D).double()
        _close(dxm.to_float(), ref_dx * keep / (1 - drop_p), 1e-4, 3e-5, "masked dx planes")
    else:
        _close(dxm.to_float(), ref_dx, 1e-4, 3e-5, "dx planes (p=0)")


def test_colsum(ops, dev):
    g = torch.Generator().manual_seed(8)
    x = _rand(g, 1234, 3072)
    out = torch.empty(3072, device=dev)
    ops.colsum(x.to(dev), out, torch.empty(128 * 3072, device=dev), rows=1234, cols=3072)
    _close(out, x.double().sum(0), 2e-4, 1e-5)


# ------------------------------------------------------------------------------------- attention
def _xattn_ref(q, k, v, heads, scale):
    b, n, e = q.shape
    m = k.shape[1]
    d = e // heads
    qh = q.view(b, n, heads, d).permute(0, 2, 1, 3)
    kh = k.view(b, m, heads, d).permute(0, 2, 1, 3)
    vh = v.view(b, m, heads, d).permute(0, 2, 1, 3)
    att = torch.softmax(qh @ kh.transpose(-1, -2), dim=-1) * scale
    return (att @ vh).permute(0, 2, 1, 3).reshape(b, n, e)


@pytest.mark.parametrize("batch,heads,Lq,Lk,hd", [(3, 8, 196, 16, 96), (5, 8, 4, 4, 96), (2, 8, 7, 3, 8), (4, 8, 2, 2, 96)])
def test_xattn_fwd_bwd(ops, dev, batch, heads, Lq, Lk, hd):
    g = torch.Generator().manual_seed(batch * Lq)
    E = heads * hd
    q, k, v, do = _rand(g, batch, Lq, E, scale=0.5), _rand(g, batch, Lk, E, scale=0.5), _rand(g, batch, Lk, E), _rand(g, batch, Lq, E)
    scale = 1.0 / math.sqrt(E)
    qt, kt, vt = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = _xattn_ref(qt, kt, vt, heads, scale)
    ref.backward(do.double())
    qd, kd, vd = q.to(dev).view(-1, E), k.to(dev).view(-1, E), v.to(dev).view(-1, E)
    o = torch.empty(batch * Lq, E, device=dev)
    ops.xattn_fwd(qd, kd, vd, o, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=scale)
    _close(o, ref.view(-1, E), 1e-6, 1e-5, "xattn fwd")
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.xattn_bwd(qd, kd, vd, do.to(dev).view(-1, E), dq, dk, dv, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd,
                  post_scale=scale)
    _close(dq, qt.grad.view(-1, E), 1e-6, 1e-4, "dq")
    _close(dk, kt.grad.view(-1, E), 1e-5, 1e-4, "dk")
    _close(dv, vt.grad.view(-1, E), 1e-5, 1e-4, "dv")


@pytest.mark.parametrize("batch,heads,L", [(3, 12, 197), (2, 4, 9), (2, 12, 196), (2, 2, 64), (1, 3, 130), (2, 1, 256)])
@pytest.mark.parametrize("planes_out", [False, True])
def test_self_attn_fwd_with_key_mask(ops, dev, batch, heads, L, planes_out):
    """MFMA self-attention (S^T = K Q^T, P fragments reused from the accumulators) from a fused QKV planes matrix."""
    g = torch.Generator().manual_seed(L)
    E = heads * 64
    q, k, v = _rand(g, batch, L, E, scale=0.3), _rand(g, batch, L, E, scale=0.3), _rand(g, batch, L, E)
    seg = torch.ones(batch, L, dtype=torch.long)
    seg[-1, L // 3:] = 0
    mask = (1.0 - (seg > 0).double()).view(batch, 1, 1, L) * -10000.0
    qh, kh, vh = (t.double().view(batch, L, heads, 64).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-2, -1) / 8.0 + mask, dim=-1)
    ref = (p @ vh).transpose(1, 2).reshape(batch * L, E)
    qkv = _planes(ops, torch.cat([q, k, v], dim=-1).view(batch * L, 3 * E), dev)
    o = ops.Planes.empty(batch * L, E, dev) if planes_out else torch.full((batch * L, E), float("nan"), device=dev)
    ops.self_attn_fwd(qkv, seg.to(dev).view(-1), o, batch=batch, heads=heads, L=L, head_dim=64, scale=1.0 / 8.0)
    _close(o.to_float() if planes_out else o, ref, 2e-5, 2e-5, "self-attn")


@pytest.mark.parametrize("batch,heads,L,drop_p", [(2, 3, 197, 0.0), (1, 2, 64, 0.0), (2, 1, 130, 0.1), (1, 2, 256, 0.0),
                                                    (2, 2, 196, 0.1), (2, 2, 257, 0.0), (1, 2, 514, 0.1), (1, 3, 300, 0.1),
                                                    (1, 1, 449, 0.0)])
def test_self_attn_bwd_matches_autograd(ops, dev, batch, heads, L, drop_p):
    """dQ / dK / dV of the MFMA self-attention (two kernels, probabilities recomputed, dropout mask replayed) against
    fp64 autograd of the reference formula; the forward with dropout and its log-sum-exp output are checked on the way.
    L > 256 (ViT-L/14's 257 tokens, RoBERTa's max_seq_length 514): the key / query block loops of both kernels."""
    import numpy as np
    from oracle import lr2ppo_oracle as O
    g = torch.Generator().manual_seed(L + heads)
    E = heads * 64
    qkv = torch.cat([_rand(g, batch * L, E, scale=0.3), _rand(g, batch * L, E, scale=0.3), _rand(g, batch * L, E)], dim=1)
    do = _rand(g, batch * L, E)
    seg = torch.ones(batch, L, dtype=torch.long)
    seg[-1, (2 * L) // 3:] = 0
    mask = (1.0 - (seg > 0).double()).view(batch, 1, 1, L) * -10000.0
    seed, site = 99, 5
    mult = torch.ones(batch, heads, L, L, dtype=torch.float64)
    if drop_p > 0:
        keep = O.attention_keep_mask(seed, site, batch, heads, L, drop_p)
        mult = torch.from_numpy(np.asarray(keep, dtype=np.float64)).view(batch, heads, L, L) / (1.0 - drop_p)
    x = qkv.double().requires_grad_(True)
    qh, kh, vh = (t.reshape(batch, L, heads, 64).transpose(1, 2) for t in x.split(E, dim=1))
    sc = qh @ kh.transpose(-2, -1) / 8.0 + mask
    ref_o = ((torch.softmax(sc, dim=-1) * mult) @ vh).transpose(1, 2).reshape(batch * L, E)
    (ref_o * do.double()).sum().backward()
    drop = ops.Drop(drop_p, seed, site) if drop_p > 0 else None
    qkv_p, do_p = _planes(ops, qkv, dev), _planes(ops, do, dev)
    o = torch.empty(batch * L, E, device=dev)
    lse = torch.empty(batch * heads * L, device=dev)
    ops.self_attn_fwd(qkv_p, seg.to(dev).view(-1), o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=drop)
    _close(o, ref_o.detach(), 3e-5, 3e-5, "self-attn fwd (dropout)")
    _close(lse.view(batch, heads, L), torch.logsumexp(sc.detach(), dim=-1), 1e-5, 1e-5, "lse")
    dqkv = ops.Planes.empty(batch * L, 3 * E, dev)
    dqkv.buf.fill_(0x7FC0)       # bf16 NaN pattern: every element must be written
    ws1, ws2 = torch.empty(batch * heads * L, device=dev), torch.empty(batch * heads * L, device=dev)
    ops.self_attn_bwd(qkv_p, do_p, seg.to(dev).view(-1), dqkv, ws1, ws2, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125,
                      drop=drop)
    got = dqkv.to_float()
    for name, lo in (("dQ", 0), ("dK", E), ("dV", 2 * E)):
        ref = x.grad[:, lo:lo + E]
        _close(got[:, lo:lo + E], ref, 3e-5 * max(1.0, float(ref.abs().max())), 5e-5, name)
    # the backward recomputes the statistics itself (natural-log domain, or block by block for L > 256; the forward works in
    # the log2 domain): equal to rounding
    _close(ws1, lse.double().cpu(), 1e-5, 1e-5, "lse (backward)")


# ------------------------------------------------------------------------------------- small ops
def test_gather_copy_head_period(ops, dev):
    g = torch.Generator().manual_seed(2)
    B, T, R = 4, 3, 64
    src = _rand(g, B, T, R)
    idx = torch.tensor([[0, 1, 2, 1], [2, 2, 0, 1], [1, 0, 0, 2], [0, 1, 1, 0]])
    dst = torch.empty(B, 4, R, device=dev)
    ops.gather_rows(src.to(dev), idx.to(dev), dst, B=B, t_in=T, t_out=4, row_elems=R)
    assert torch.equal(dst.cpu(), src[torch.arange(B).view(B, 1), idx])
    # expanded (stride-0) image features: img[b] repeated over tags
    img = _rand(g, B, R)
    dst2 = torch.empty(B, 4, R, device=dev)
    ops.gather_rows(img.to(dev), idx.to(dev), dst2, B=B, t_in=T, t_out=4, row_elems=R, src_bstride=R, src_tstride=0)
    assert torch.equal(dst2.cpu(), img.unsqueeze(1).expand(B, 4, R))
    dsrc = torch.empty(B, T, R, device=dev)
    ops.gather_rows_bwd(dst, idx.to(dev), dsrc, B=B, t_in=T, t_out=4, row_elems=R)
    ref = torch.zeros(B, T, R)
    for b in range(B):
        for j in range(4):
            ref[b, idx[b, j]] += dst.cpu()[b, j]
    _close(dsrc, ref, 1e-6, 1e-6)
    # concat copy
    rows, D, group = 6, 8, 2
    x = _rand(g, rows, D)
    flat = torch.zeros(3, 5 * D, device=dev)
    ops.copy_rows(x.to(dev), flat, rows=rows, D=D, group=group, dst_gstride=5 * D, dst_off=3 * D)
    assert torch.equal(flat.cpu()[:, 3 * D:], x.view(3, 2 * D)) and torch.all(flat[:, :3 * D] == 0)
    # head fwd/bwd with last-position select
    Bn, Tn, Dn = 5, 4, 768
    xx, w, b_, dy = _rand(g, Bn * Tn, Dn), _rand(g, 1, Dn), _rand(g, 1), _rand(g, Bn)
    y = torch.empty(Bn, device=dev)
    ops.head_fwd(xx.to(dev), w.to(dev), b_.to(dev), y, rows=Bn, D=Dn, row_step=Tn, row_off=Tn - 1)
    sel = xx.view(Bn, Tn, Dn)[:, -1]
    _close(y, sel.double() @ w.double().view(-1) + b_.double(), 1e-5, 1e-5)
    dx, dw, db = torch.empty(Bn * Tn, Dn, device=dev), torch.empty(1, Dn, device=dev), torch.empty(1, device=dev)
    ops.head_bwd(xx.to(dev), w.to(dev), dy.to(dev), dx, dw, db, rows=Bn, D=Dn, row_step=Tn, row_off=Tn - 1)
    ref_dx = torch.zeros(Bn, Tn, Dn, dtype=torch.double)
    ref_dx[:, -1] = dy.double().view(-1, 1) * w.double()
    _close(dx, ref_dx.view(-1, Dn), 1e-6, 1e-6)
    _close(dw, (dy.double().view(-1, 1) * sel.double()).sum(0, keepdim=True), 1e-5, 1e-5)
    _close(db, dy.double().sum().view(1), 1e-6, 1e-6)
    # pos-emb add + grad
    table = _rand(g, 4, Dn)
    out = torch.empty(Bn * Tn, Dn, device=dev)
    ops.add_period_rows(xx.to(dev), table.to(dev), out, rows=Bn * Tn, D=Dn, period=Tn)
    _close(out, (xx.view(Bn, Tn, Dn) + table.unsqueeze(0)).view(-1, Dn), 1e-6, 1e-6)
    dt = torch.zeros(4, Dn, device=dev)
    ops.period_rows_grad(xx.to(dev), dt, rows=Bn * Tn, D=Dn, period=Tn)
    _close(dt, xx.view(Bn, Tn, Dn).double().sum(0), 1e-5, 1e-5)


@pytest.mark.parametrize("B,T,seed", [(32, 2, 1), (24, 2, 2), (7, 2, 3), (16, 4, 4), (1, 2, 5)])
def test_ppo_loss_and_gradients(ops, dev, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    scores = _rand(g, B, T, scale=0.3)
    old = scores + _rand(g, B, T, scale=0.05)
    rewards, old_value = _rand(g, B, scale=0.2), _rand(g, B, scale=0.2)
    value = old_value + _rand(g, B, scale=0.6)        # some |v - old| exceed the clip
    state = torch.stack([torch.randperm(T, generator=g) for _ in range(B)])
    nxt = torch.cat([torch.arange(2).unsqueeze(0).repeat(B, 1), state], dim=1)
    kl_w, ent_w, clip = 0.001, 0.001, 0.5
    st, vt = scores.clone().requires_grad_(True), value.clone().requires_grad_(True)
    loss, vloss, ex = O.ppo_update_math(st, vt, old, rewards, old_value, nxt, kl_w, ent_w, clip)
    loss.backward()
    vloss.backward()
    scal, per = torch.empty(4, device=dev), torch.empty(4, B, device=dev)
    ds, dv = torch.empty(B, T, device=dev), torch.empty(B, device=dev)
    ops.ppo_loss(scores.to(dev), old.to(dev), rewards.to(dev), old_value.to(dev), value.to(dev), nxt.to(dev), scal, per, ds,
                 dv, B=B, T=T, kl_w=kl_w, ent_w=ent_w, value_clip=clip)
    _close(scal[0], loss, 1e-7, 1e-5, "policy loss")
    _close(scal[1], vloss, 1e-7, 1e-5, "value loss")
    _close(scal[2], ex["rank_loss"], 1e-7, 1e-5, "rank loss")
    _close(per[0], ex["kl"], 1e-7, 1e-4, "kl")
    _close(per[1], ex["entropy"], 1e-6, 1e-5, "entropy")
    _close(per[2], ex["rewards"], 1e-7, 1e-5, "rewards")
    _close(per[3], ex["advantages"], 1e-7, 1e-5, "advantages")
    _close(ds, st.grad, 1e-8, 1e-4, "dscores")
    _close(dv, vt.grad, 1e-8, 1e-4, "dvalue")


def test_smooth_l1(ops, dev):
    g = torch.Generator().manual_seed(4)
    pred, tgt = _rand(g, 77), torch.randint(0, 3, (77,), generator=g).float()
    pt = pred.clone().requires_grad_(True)
    ref = O.smooth_l1(pt, tgt)
    ref.backward()
    loss, dp = torch.empty(1, device=dev), torch.empty(77, device=dev)
    ops.smooth_l1(pred.to(dev), tgt.to(dev), loss, dp, n=77)
    _close(loss, ref.view(1), 1e-6, 1e-6)
    _close(dp, pt.grad, 1e-8, 1e-5)


def test_adamw_multi_matches_reference_order(ops, dev):
    import ctypes as C
    from lr2ppo_amd import _native as nat
    g = torch.Generator().manual_seed(6)
    sizes, wds = [1000, 7, 262147], [0.01, 0.0, 0.01]
    ps = [_rand(g, n) for n in sizes]
    ms, vs = [torch.zeros(n) for n in sizes], [torch.zeros(n) for n in sizes]
    pd, md, vd = [p.to(dev) for p in ps], [m.to(dev) for m in ms], [v.to(dev) for v in vs]
    for step in range(3):
        gs = [_rand(g, n, scale=10.0 ** (-2 * step)) for n in sizes]
        gd = [x.to(dev) for x in gs]
        chunks = []
        for p, gr, m, v, wd in zip(pd, gd, md, vd, wds):
            n, off = p.numel(), 0
            while off < n:
                c = min(65536, n - off)
                chunks.append((p.data_ptr() + 4 * off, gr.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off, c, wd))
                off += c
        arr = (nat.AdamChunk * len(chunks))()
        for i, (a, b, c_, d, cnt, wd) in enumerate(chunks):
            arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].count, arr[i].weight_decay = a, b, c_, d, cnt, wd
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        ops.adamw_multi(table, len(chunks), 1e-2, 0.9, 0.999, 1e-6)
        torch.cuda.synchronize()
        for i in range(3):
            ps[i], ms[i], vs[i] = O.adamw_step(ps[i], gs[i], ms[i], vs[i], 1e-2, wds[i])
            _close(pd[i], ps[i], 1e-6, 1e-5, f"p step {step}")
            _close(md[i], ms[i], 1e-8, 1e-5, "m")
            _close(vd[i], vs[i], 1e-10, 1e-5, "v")


def test_embedding_frontends(ops, dev, golden):
    gd = golden("embeddings_small.npz")
    P = {k[len("vit_param."):]: v for k, v in gd.items() if k.startswith("vit_param.")}
    img = gd["vit_img"]
    B, Cc, H, W, ps, D = 2, 3, 32, 48, 8, 32
    Pn = (H // ps) * (W // ps)
    patches = torch.empty(B * Pn, Cc * ps * ps, device=dev)
    ops.patchify(img.to(dev), patches, B=B, Cc=Cc, H=H, W=W, ps=ps)
    ref_p = img.view(B, Cc, H // ps, ps, W // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * Pn, -1)
    assert torch.equal(patches.cpu(), ref_p)
    proj = (ref_p @ P["patch.projection.weight"].view(D, -1).t()).to(dev)
    out = torch.empty(B, Pn + 1, D, device=dev)
    ops.vit_assemble(proj, P["patch.cls_emb"].to(dev).view(-1), P["pos.embedding.weight"].to(dev), out, B=B, P=Pn, D=D)
    _close(out, gd["vit_out"], 1e-5, 1e-5, "vit embedding")
    Pt = {k[len("txt_param."):]: v for k, v in gd.items() if k.startswith("txt_param.")}
    src, seg = gd["txt_src"], gd["txt_seg"]
    rows, L = src.numel(), src.shape[1]
    e = torch.empty(rows, D, device=dev)
    ops.text_embed(src.to(dev).view(-1), seg.to(dev).view(-1), Pt["word.embedding.weight"].to(dev),
                   Pt["pos.embedding.weight"].to(dev), Pt["seg.embedding.weight"].to(dev), e, rows=rows, L=L, D=D)
    o2 = torch.empty(rows, D, device=dev)
    ops.layernorm_fwd(e, Pt["layer_norm.gamma"].to(dev), Pt["layer_norm.beta"].to(dev), o2, rows=rows, D=D, eps=1e-6, mode=1)
    _close(o2.view(3, L, D), gd["txt_out"], 1e-5, 1e-5, "text embedding")


# ----------------------------------------------------------------------------- planes (pre-split operands)
def _planes(ops, x, dev):
    pl = ops.Planes.empty(x.shape[0], x.shape[1], dev)
    ops.split_planes(x.to(dev).contiguous(), pl)
    return pl


def test_split_planes_roundtrip_is_17_bit(ops, dev):
    g = torch.Generator().manual_seed(1)
    x = _rand(g, 300, 256) * 3.0
    pl = _planes(ops, x, dev)
    back = pl.to_float().cpu()
    rel = ((back - x).abs() / x.abs().clamp(min=1e-30)).max().item()
    assert rel < 2.0 ** -15.9, rel
    hi = x.bfloat16()
    assert torch.equal(pl.buf[: 300 * 256].view(torch.bfloat16).cpu().view(300, 256), hi)


@pytest.mark.parametrize("form,M,N,K,bm,splits", [("NT", 200, 256, 192, 128, 1), ("NT", 50, 128, 2560, 64, 5),
                                                  ("NN", 130, 256, 192, 128, 1), ("NN", 64, 1280, 128, 64, 1),
                                                  ("TN", 256, 384, 300, 128, 1), ("TN", 64, 256, 37, 64, 1),
                                                  ("TN", 384, 128, 1000, 128, 3), ("NT", 333, 128, 768, 128, 3)])
def test_gemm_planes_operands_all_forms(ops, dev, form, M, N, K, bm, splits):
    """Both operands as LDS-DMA planes; ragged M / N and ragged contraction (TN) rely on the descriptor's zero fill.
    Without split-K the result must equal the fp32-operand kernel bit for bit (same accumulation order)."""
    g = torch.Generator().manual_seed(M + N + K)
    ta, tb = form == "TN", form in ("NN", "TN")
    a = _rand(g, K, M) if ta else _rand(g, M, K)
    b = _rand(g, K, N) if tb else _rand(g, N, K)
    ref = (a.double().t() if ta else a.double()) @ (b.double() if tb else b.double().t())
    ws = torch.empty(max(1, splits) * M * N, device=dev)
    out_p = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(_planes(ops, a, dev), _planes(ops, b, dev), out_p, M, N, K, trans_a=ta, trans_b=tb, block_m=bm, splits=splits,
             splitk_ws=ws)
    _close(out_p, ref, atol=6e-5 * math.sqrt(K), rtol=5e-5, what=f"planes {form}")
    out_f = torch.empty((M, N), device=dev)
    ops.gemm(a.to(dev), b.to(dev), out_f, M, N, K, trans_a=ta, trans_b=tb, block_m=bm, splits=splits, splitk_ws=ws)
    if splits == 1:
        assert torch.equal(out_p, out_f), f"planes vs fp32 operands differ: {(out_p - out_f).abs().max().item()}"
    else:  # the planes kernel may cut K into 32-deep tiles, so its split boundaries (summation order) can differ
        _close(out_p, out_f.double(), atol=2e-5 * math.sqrt(K), rtol=2e-5, what=f"planes vs fp32 {form}")


@pytest.mark.parametrize("form,M,N,K,bm", [("NT", 64, 256, 1024, 64), ("NN", 64, 1280, 128, 64), ("NT", 200, 256, 192, 128)])
def test_gemm_planes_a_fp32_b(ops, dev, form, M, N, K, bm):
    """The out_layer.fc1 shape class: small planes A against a streamed fp32 B."""
    g = torch.Generator().manual_seed(M * 5 + N + K)
    tb = form == "NN"
    a = _rand(g, M, K)
    b = _rand(g, K, N) if tb else _rand(g, N, K)
    ref = a.double() @ (b.double() if tb else b.double().t())
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(_planes(ops, a, dev), b.to(dev), out, M, N, K, trans_b=tb, block_m=bm, splits=1)
    _close(out, ref, atol=6e-5 * math.sqrt(K), rtol=5e-5, what="planes A, fp32 B")


def test_gemm_planes_output_and_colsum(ops, dev):
    g = torch.Generator().manual_seed(23)
    M, N, K = 150, 256, 128
    a, w, bias = _rand(g, M, K), _rand(g, N, K, scale=0.1), _rand(g, N)
    outp = ops.Planes.empty(M, N, dev)
    z = torch.empty(M, N, device=dev)
    ops.gemm(a.to(dev), w.to(dev), None, M, N, K, bias=bias.to(dev), act=1, out_z=z, out_planes=outp)
    ref = O.gelu_erf(a.double() @ w.double().t() + bias.double())
    _close(outp.to_float(), ref, 6e-5, 5e-5, "planes epilogue output")
    cs = torch.empty(N, device=dev)
    ops.colsum(outp, cs, torch.empty(128 * N, device=dev), rows=M, cols=N)
    _close(cs, ref.sum(0), 2e-3, 1e-4, "colsum of planes")


def test_planes_outputs_of_layernorm_attention_copy(ops, dev):
    g = torch.Generator().manual_seed(29)
    rows, D = 40, 768
    x, gam, bet = _rand(g, rows, D), _rand(g, D), _rand(g, D)
    pl = ops.Planes.empty(rows, D, dev)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), None, rows=rows, D=D, out_planes=pl)
    _close(pl.to_float(), O.layernorm_torch(x.double(), gam.double(), bet.double()), 1e-4, 3e-5, "LN planes")
    batch, heads, Lq, Lk, hd = 2, 8, 5, 3, 96
    E = heads * hd
    q, k, v, do = _rand(g, batch * Lq, E, scale=0.5), _rand(g, batch * Lk, E, scale=0.5), _rand(g, batch * Lk, E), _rand(g, batch * Lq, E)
    scale = 1.0 / math.sqrt(E)
    of, op_ = torch.empty(batch * Lq, E, device=dev), ops.Planes.empty(batch * Lq, E, dev)
    ops.xattn_fwd(q.to(dev), k.to(dev), v.to(dev), of, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=scale)
    ops.xattn_fwd(q.to(dev), k.to(dev), v.to(dev), op_, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=scale)
    _close(op_.to_float(), of, 1e-6, 3e-5, "attention planes out")
    dqf, dkf, dvf = (torch.empty(n, E, device=dev) for n in (batch * Lq, batch * Lk, batch * Lk))
    dqp, dkp, dvp = (ops.Planes.empty(n, E, dev) for n in (batch * Lq, batch * Lk, batch * Lk))
    for dq_, dk_, dv_ in ((dqf, dkf, dvf), (dqp, dkp, dvp)):
        ops.xattn_bwd(q.to(dev), k.to(dev), v.to(dev), do.to(dev), dq_, dk_, dv_, batch=batch, heads=heads, Lq=Lq, Lk=Lk,
                      head_dim=hd, post_scale=scale)
    for a_, b_ in ((dqp, dqf), (dkp, dkf), (dvp, dvf)):
        _close(a_.to_float(), b_, 1e-7, 3e-5, "attention bwd planes")
    src = _rand(g, 6, 8)
    flat = ops.Planes(torch.zeros(2 * 3 * 40, dtype=torch.int16, device=dev), 3, 40)
    ops.copy_rows(src.to(dev), flat, rows=6, D=8, group=2, dst_gstride=40, dst_off=24)
    got = flat.to_float().cpu()
    _close(got[:, 24:], src.view(3, 16), 1e-6, 3e-5, "copy_rows planes")
    assert torch.all(got[:, :24] == 0)


@pytest.mark.parametrize("M,N,K,splits", [(256, 384, 64, 1), (200, 264, 128, 1), (128, 128, 1000, 3)])
def test_gemm_fused_adamw_epilogue_equals_gemm_then_adamw(ops, dev, M, N, K, splits):
    """Weight-gradient form (TN) with the optimizer step in the epilogue: p, exp_avg, exp_avg_sq must come out with
    the same bits as lr2_gemm -> grad followed by lr2_adamw_multi (two steps, decay on)."""
    from lr2ppo_amd.tencentpretrain.utils.optimizers import AdamW
    g = torch.Generator().manual_seed(M + N + K)
    p0 = _rand(g, M, N).to(dev)
    ws = torch.empty(max(1, splits) * M * N, device=dev)
    pa, pb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa = AdamW([{"params": [pa], "weight_decay": 0.01}], lr=1e-3, correct_bias=False)
    ob = AdamW([{"params": [pb], "weight_decay": 0.01}], lr=1e-3, correct_bias=False)
    pa.grad = torch.zeros_like(pa)
    pb.grad = torch.full_like(pb, float("nan"))          # never read on the fused path
    for step in range(2):
        a, b = _rand(g, K, M), _rand(g, K, N)
        ap, bp = _planes(ops, a, dev), _planes(ops, b, dev)
        ops.gemm(ap, bp, pa.grad, M, N, K, trans_a=True, trans_b=True, splits=splits, splitk_ws=ws, alpha=0.5)
        oa.step()
        ops.gemm(ap, bp, None, M, N, K, trans_a=True, trans_b=True, splits=splits, splitk_ws=ws, alpha=0.5,
                 adam=ob.external_update(pb))
        ob.step()                                        # nothing left to do for pb; clears the hand-over
        assert torch.equal(pa.detach(), pb.detach()), f"step {step}: weights differ by {(pa - pb).abs().max().item()}"
        assert torch.equal(oa.state[pa]["exp_avg"], ob.state[pb]["exp_avg"])
        assert torch.equal(oa.state[pa]["exp_avg_sq"], ob.state[pb]["exp_avg_sq"])
    assert not torch.equal(pa.detach(), p0) and oa.state[pa]["step"] == ob.state[pb]["step"] == 2


def test_split_planes_t_and_nn_forward_equals_nt(ops, dev):
    """Transposed weight planes W^T: exact transpose of the plain split, and the forward GEMM on them (NN form) agrees with
    the NT form on the original layout."""
    g = torch.Generator().manual_seed(12)
    w = _rand(g, 300, 200)                                  # ragged against the 32 x 32 transpose tiles
    wt = ops.split_planes_t(w.to(dev), ops.Planes.empty(200, 300, dev))
    ref = _planes(ops, w, dev)
    n = 300 * 200
    assert torch.equal(wt.buf[:n].view(200, 300), ref.buf[:n].view(300, 200).t())
    assert torch.equal(wt.buf[wt.lo_off:wt.lo_off + n].view(200, 300), ref.buf[ref.lo_off:ref.lo_off + n].view(300, 200).t())
    assert wt.transposed and not ref.transposed
    M, N, K = 260, 384, 192
    x, w = _rand(g, M, K), _rand(g, N, K)
    xp = _planes(ops, x, dev)
    o_nt, o_nn = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ops.gemm(xp, _planes(ops, w, dev), o_nt, M, N, K)
    ops.gemm(xp, ops.split_planes_t(w.to(dev), ops.Planes.empty(K, N, dev)), o_nn, M, N, K, trans_b=True, ldb=N)
    _close(o_nn, x.double() @ w.double().t(), atol=6e-5 * math.sqrt(K), rtol=5e-5, what="NN on W^T")
    _close(o_nn, o_nt.double(), atol=1e-5 * math.sqrt(K), rtol=1e-5, what="NN vs NT")
