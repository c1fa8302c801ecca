"""MX-FP8 fast mode (csrc/fp8.hip, BASELINE.json configs[4] "fp8 MFMA"): the quantiser against the OCP MX v1.0 rule restated with torch
(bytes and scale bytes equal), the product against an fp64 product of the DEQUANTISED operands (the instruction's adder tree is not
an fp32 sum: bounded by 2e-3 of the sum of |terms|), fused bias / GELU / residual, ragged M.  Not a parity path: how far a product is
from the fp32 one is measured and printed, not gated."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_quant(x):
    R, K = x.shape
    xb = x.double().view(R, K // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(1e-300))) - 8
    e = torch.where(amax < 1.17549435e-38, torch.full_like(e, -127.0), e).clamp(-127, 127)
    q = (xb * torch.exp2(-e)).clamp(-448, 448).float().to(torch.float8_e4m3fn)
    return q.view(R, K).view(torch.uint8), (e + 127).to(torch.uint8).view(R, K // 32)


@pytest.mark.parametrize("R,K", [(64, 128), (300, 768), (17, 4096)])
def test_quantiser_matches_the_mx_rule(dev, R, K):
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(R + K)
    x = torch.randn(R, K, generator=g) * torch.exp(torch.randn(R, 1, generator=g) * 3)
    x[0, :32] = 0.0                                          # an all-zero block
    x[1, 5] = 1e30                                           # a huge outlier in a block
    m = ops.quant_mxfp8(x.to(dev))
    q_ref, s_ref = _ref_quant(x)
    assert torch.equal(m.s.view(R, K // 32).cpu(), s_ref)
    got = m.q.view(R, K).cpu()
    same = (got == q_ref) | ((got & 0x7F) == 0) & ((q_ref & 0x7F) == 0)           # +0 / -0
    assert bool(same.all()), f"{int((~same).sum())} of {R * K} bytes differ"
    assert torch.equal(m.to_float().cpu(), (q_ref.view(torch.float8_e4m3fn).float().view(R, K // 32, 32)
                                            * torch.exp2(s_ref.float() - 127).view(R, K // 32, 1)).view(R, K))


@pytest.mark.parametrize("M,N,K,act,with_resid", [(128, 128, 128, 0, False), (300, 256, 768, 1, True), (1000, 384, 3072, 0, True),
                                                   (64, 1024, 1024, 1, False)])
def test_product_matches_the_dequantised_operands(dev, M, N, K, act, with_resid):
    from lr2ppo_amd import ops
    from oracle import lr2ppo_oracle as O
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g) * 0.1
    resid = torch.randn(M, N, generator=g) if with_resid else None
    am, bm = ops.quant_mxfp8(a.to(dev)), ops.quant_mxfp8(b.to(dev))
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_mxfp8(am, bm, out, bias=bias.to(dev), resid=None if resid is None else resid.to(dev), act=act)
    da, db = am.to_float().double().cpu(), bm.to_float().double().cpu()
    pre = da @ db.t() + bias.double()
    want = O.gelu_erf(pre) if act else pre
    if resid is not None:
        want = want + resid.double()
    bound = 2e-3 * (da.abs() @ db.abs().t()) + 1e-5         # the instruction's adder tree, relative to the sum of |terms|
    err = (out.double().cpu() - want).abs()
    assert bool((err <= bound).all()), f"worst excess {(err - bound).max().item():.3e} (err {err.max().item():.3e})"
    exact = a.double() @ b.double().t()                                       # how far fp8 is from the fp32 product (printed)
    rel = ((da @ db.t()) - exact).norm() / exact.norm()
    print(f"M {M} N {N} K {K}: MX-FP8 product vs fp32 product, relative L2 error {rel.item():.3e}")
    assert rel < 0.06


def test_product_can_hand_its_result_on_as_mx_fp8(dev):
    """out_mx: the epilogue quantises the row it would have stored -- bytes and scales equal to lr2_quant_mxfp8 of the fp32 result."""
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 333, 512, 256
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g) * 0.1
    am, bm = ops.quant_mxfp8(a.to(dev)), ops.quant_mxfp8(b.to(dev))
    out = torch.empty(M, N, device=dev)
    mx = ops.Mx8.empty(M, N, dev)
    ops.gemm_mxfp8(am, bm, out, bias=bias.to(dev), act=1, out_mx=mx)
    want = ops.quant_mxfp8(out)
    assert torch.equal(mx.s, want.s) and torch.equal(mx.q, want.q)
    pl = ops.Planes.empty(M, N, dev)                     # ... and as bf16 hi / lo planes: the split of the same fp32 row
    ops.gemm_mxfp8(am, bm, None, bias=bias.to(dev), act=1, out_planes=pl)
    assert torch.equal(pl.buf, ops.split_planes(out, ops.Planes.empty(M, N, dev)).buf)
    only = ops.Mx8.empty(M, N, dev)
    ops.gemm_mxfp8(am, bm, None, bias=bias.to(dev), act=1, out_mx=only)
    assert torch.equal(only.s, want.s) and torch.equal(only.q, want.q)


@pytest.mark.parametrize("positioning", ["pre", "post"])
def test_encoder_forward_fp8_stays_near_the_default_forward(dev, positioning):
    """TransformerEncoder.forward_fp8 (projections as MX-FP8 products): same shapes, finite, and within the distance e4m3 allows of
    forward() -- measured and printed; the parity tests never use this mode."""
    import argparse
    from lr2ppo_amd.tencentpretrain.encoders import str2encoder
    from lr2ppo_amd.tencentpretrain.opts import finetune_opts, tokenizer_opts
    ap = argparse.ArgumentParser()
    finetune_opts(ap)
    tokenizer_opts(ap)
    d = vars(ap.parse_args([]))
    d.update(emb_size=256, feedforward_size=512, hidden_size=256, hidden_act="gelu", heads_num=4, layers_num=3, max_seq_length=64,
             dropout=0.1, embedding=["word", "pos", "seg"], encoder="transformer", mask="fully_visible", layernorm_positioning=positioning)
    args = argparse.Namespace(**d)
    torch.manual_seed(3)
    enc = str2encoder["transformer"](args)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "gamma" not in n and "beta" not in n:
                p.normal_(0, 0.05)
    enc = enc.to(dev).eval()
    emb = torch.randn(6, 50, 256, device=dev)
    seg = torch.ones(6, 50, dtype=torch.long, device=dev)
    seg[:, 40:] = 0
    with torch.no_grad():
        ref = enc(emb, seg)
        got = enc.forward_fp8(emb, seg)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    rel = float((got - ref).norm() / ref.norm())
    print(f"{positioning}-LN, 3 layers: forward_fp8 vs forward, relative L2 distance {rel:.3e}")
    assert rel < 0.15


@pytest.mark.parametrize("mode", [0, 1])
def test_layernorm_can_leave_as_mx_fp8(dev, mode):
    """lr2_layernorm_fwd_mxfp8 = lr2_layernorm_fwd followed by lr2_quant_mxfp8 of its fp32 result, byte for byte."""
    from lr2ppo_amd import ops
    g = torch.Generator(device=dev).manual_seed(4)
    rows, D = 301, 768
    x = torch.randn(rows, D, device=dev, generator=g) * 3
    gamma, beta = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
    ref32 = torch.empty(rows, D, device=dev)
    ops.layernorm_fwd(x, gamma, beta, ref32, rows=rows, D=D, eps=1e-6, mode=mode)
    want = ops.quant_mxfp8(ref32)
    got, out32 = ops.Mx8.empty(rows, D, dev), torch.empty(rows, D, device=dev)
    ops.layernorm_fwd_mxfp8(x, gamma, beta, got, out32, rows=rows, D=D, eps=1e-6, mode=mode)
    assert torch.equal(out32, ref32) and torch.equal(got.s, want.s) and torch.equal(got.q, want.q)


# ---- the 256 x 256 LDS-DMA ring (csrc/gemm256_mx.hip): products whose one-workgroup-per-CU rounds are well filled ------------------
@pytest.mark.parametrize("M,N,K,act,with_resid", [(8192, 2048, 128, 0, False), (8000, 2048, 256, 1, True), (16384, 1024, 384, 0, True),
                                                   (7937, 2176, 1024, 1, False), (8192, 2048, 512, 0, False), (8100, 2048, 1536, 1, True),
                                                   (8192, 2048, 2560, 0, False)])
def test_ring_product_matches_the_dequantised_operands(dev, M, N, K, act, with_resid):
    """>= 256 tiles of 256 x 256 -> lr2_gemm_mxfp8 takes the ring kernel: both scale-staging forms (K % 512 == 0: the scales of four
    K steps per 16-byte DMA -- 1, 2, 3 and 5 such chunks per row; else one 4-byte gather per step), every K-step parity, ragged M (a last
    tile row of 64 / 1 rows), N a multiple of 128 but not of 256 (a half-empty last tile column), fused bias / GELU / residual --
    against the fp64 product of the dequantised operands (bound of the instruction's adder tree) and against the 128 x 128 kernel
    (LR2_FP8_256=0 is read once per process, so the comparison value comes from a slice that stays on the small kernel)."""
    from lr2ppo_amd import ops
    from oracle import lr2ppo_oracle as O
    assert ((M + 255) // 256) * ((N + 255) // 256) >= 256
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    a *= torch.exp(torch.randn(M, 1, generator=g))                              # rows of very different scale: the scale bytes matter
    bias = torch.randn(N, generator=g) * 0.1
    resid = torch.randn(M, N, generator=g) if with_resid else None
    am, bm = ops.quant_mxfp8(a.to(dev)), ops.quant_mxfp8(b.to(dev))
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm_mxfp8(am, bm, out, bias=bias.to(dev), resid=None if resid is None else resid.to(dev), act=act)
    da, db = am.to_float().double().cpu(), bm.to_float().double().cpu()
    pre = da @ db.t() + bias.double()
    want = O.gelu_erf(pre) if act else pre
    if resid is not None:
        want = want + resid.double()
    bound = 2e-3 * (da.abs() @ db.abs().t()) + 1e-5
    err = (out.double().cpu() - want).abs()
    assert torch.isfinite(out).all()
    assert bool((err <= bound).all()), f"worst excess {(err - bound).max().item():.3e} (err {err.max().item():.3e})"
    # the first 1024 rows again as their own product (4 x 9 tiles: the 128 x 128 kernel): same operands, same instruction
    sub = torch.empty(1024, N, device=dev)
    am_s = ops.Mx8(am.q[:1024 * K], am.s[:1024 * (K // 32)], 1024, K)
    ops.gemm_mxfp8(am_s, bm, sub, bias=bias.to(dev), resid=None if resid is None else resid[:1024].to(dev).contiguous(), act=act)
    assert (sub - out[:1024]).abs().max().item() <= 2 * float(bound[:1024].max())


def test_ring_product_hands_its_result_on_as_planes_and_mx_fp8(dev):
    """out_planes / out_mx of the ring kernel: the split / the quantisation of the fp32 row it would have stored, byte for byte."""
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(10)
    M, N, K = 8100, 2048, 256
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias = torch.randn(N, generator=g) * 0.1
    am, bm = ops.quant_mxfp8(a.to(dev)), ops.quant_mxfp8(b.to(dev))
    out = torch.empty(M, N, device=dev)
    mx = ops.Mx8.empty(M, N, dev)
    ops.gemm_mxfp8(am, bm, out, bias=bias.to(dev), act=1, out_mx=mx)
    want = ops.quant_mxfp8(out)
    assert torch.equal(mx.s, want.s) and torch.equal(mx.q, want.q)
    pl = ops.Planes.empty(M, N, dev)
    ops.gemm_mxfp8(am, bm, None, bias=bias.to(dev), act=1, out_planes=pl)
    assert torch.equal(pl.buf, ops.split_planes(out, ops.Planes.empty(M, N, dev)).buf)
    only = ops.Mx8.empty(M, N, dev)
    ops.gemm_mxfp8(am, bm, None, bias=bias.to(dev), act=1, out_mx=only)
    assert torch.equal(only.s, want.s) and torch.equal(only.q, want.q)


def test_ring_product_is_exact_on_small_integers(dev):
    """Operands that e4m3 holds exactly (small integers, scales forced apart per row): every product and sum is exact, so the tile /
    wave / quadrant / plane index maps and the scale routing of the ring kernel are checked bit for bit against integer arithmetic
    (asymmetric B: a transposed output would not pass)."""
    from lr2ppo_amd import ops
    for M, N, K in ((8192, 2048, 256), (8192, 2048, 1024)):            # 4-byte scale gathers / 16-byte scale chunks (two per row)
        g = torch.Generator().manual_seed(2)
        a = torch.randint(-3, 4, (M, K), generator=g).float() * torch.exp2(torch.randint(-2, 3, (M, 1), generator=g).float())
        b = torch.randint(-3, 4, (N, K), generator=g).float() * torch.exp2((torch.arange(N) % 5).float().view(-1, 1) - 2)
        for blk in range(K // 32):                        # every 32-element K block of every row on its own scale
            a[:, 32 * blk:32 * blk + 32] *= 2.0 ** (blk % 3)
            b[:, 32 * blk:32 * blk + 32] *= 2.0 ** ((blk * 5) % 4)
        am, bm = ops.quant_mxfp8(a.to(dev)), ops.quant_mxfp8(b.to(dev))
        assert torch.equal(am.to_float().cpu(), a) and torch.equal(bm.to_float().cpu(), b)
        out = torch.empty(M, N, device=dev)
        ops.gemm_mxfp8(am, bm, out)
        assert torch.equal(out.cpu(), (a.double() @ b.double().t()).float()), (M, N, K)


# ---- the MX-FP8 mode's attention (csrc/selfattn_mx.hip): one bf16 plane per operand, single-pass products, MX-FP8 context ------------
@pytest.mark.parametrize("batch,heads,L", [(2, 16, 257), (3, 12, 197), (2, 4, 64), (1, 2, 288), (2, 3, 100), (1, 1, 17), (2, 2, 225)])
def test_bf16_attention_matches_fp64_on_its_own_operands(dev, batch, heads, L):
    """lr2_self_attn_fwd_bf16 against the fp64 formula evaluated on the SAME bf16-rounded Q, K, V (so the only differences are the
    bf16 rounding of the un-normalised probabilities, 2^-9 relative each, and fp32 accumulation): key mask on the last sequence, every
    key-tile count (4 / 8 / 14 / 18 tiles, ragged last tile), ViT-L/14's 257 and ViT-B/16's 197 tokens; the MX-FP8 output equals
    lr2_quant_mxfp8 of the fp32 output byte for byte."""
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(L + heads)
    E = heads * 64
    qkv = torch.cat([torch.randn(batch * L, E, generator=g) * 0.5, torch.randn(batch * L, E, generator=g) * 0.5,
                     torch.randn(batch * L, E, generator=g)], dim=1).to(torch.bfloat16)
    seg = torch.ones(batch, L, dtype=torch.long)
    seg[-1, (2 * L) // 3:] = 0
    mask = (1.0 - (seg > 0).double()).view(batch, 1, 1, L) * -10000.0
    qh, kh, vh = (t.double().reshape(batch, L, heads, 64).transpose(1, 2) for t in qkv.split(E, dim=1))
    sc = qh @ kh.transpose(-2, -1) / 8.0 + mask
    ref = (torch.softmax(sc, dim=-1) @ vh).transpose(1, 2).reshape(batch * L, E)
    out = torch.full((batch * L, E), float("nan"), device=dev)
    mx = ops.Mx8.empty(batch * L, E, dev)
    ops.self_attn_fwd_bf16(qkv.to(dev), seg.to(dev).view(-1), batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out=out, out_mx=mx)
    assert torch.isfinite(out).all()
    err = (out.double().cpu() - ref).abs().max().item()
    assert err < 6e-3 * float(vh.abs().max()), err
    assert (out.double().cpu() - ref).norm() / ref.norm() < 2e-3
    want = ops.quant_mxfp8(out)
    assert torch.equal(mx.s, want.s) and torch.equal(mx.q, want.q)
    only = ops.Mx8.empty(batch * L, E, dev)
    ops.self_attn_fwd_bf16(qkv.to(dev), seg.to(dev).view(-1), batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out_mx=only)
    assert torch.equal(only.s, want.s) and torch.equal(only.q, want.q)
    with pytest.raises(Exception):
        ops.self_attn_fwd_bf16(qkv.to(dev), seg.to(dev).view(-1), batch=batch, heads=heads, L=300, head_dim=64, scale=0.125, out=out)


@pytest.mark.parametrize("batch,heads,L", [(80, 4, 257), (300, 2, 197), (520, 1, 40)])
def test_persistent_bf16_attention_equals_the_one_pair_kernel(dev, batch, heads, L):
    """At least one (sequence, head) pair per CU: lr2_self_attn_fwd_bf16 runs the persistent 12-wave kernel (K / V by LDS-DMA from two mover
    waves under the compute); the first sequences alone (fewer pairs than CUs) run the one-pair kernel on the same rows: fp32 context
    and MX-FP8 bytes / scales agree BIT FOR BIT (same arithmetic in the same order)."""
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(3 * L + heads)
    E = heads * 64
    qkv = (torch.randn(batch * L, 3 * E, generator=g) * 0.6).to(torch.bfloat16).to(dev)
    seg = (torch.rand(batch, L, generator=g) > 0.15).long()
    seg[:, 0] = 1
    seg = seg.view(-1).to(dev)
    out, mx = torch.full((batch * L, E), float("nan"), device=dev), ops.Mx8.empty(batch * L, E, dev)
    ops.self_attn_fwd_bf16(qkv, seg, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out=out, out_mx=mx)
    nb = max(1, 120 // heads)
    out2, mx2 = torch.full((nb * L, E), float("nan"), device=dev), ops.Mx8.empty(nb * L, E, dev)
    ops.self_attn_fwd_bf16(qkv[:nb * L].contiguous(), seg[:nb * L].contiguous(), batch=nb, heads=heads, L=L, head_dim=64, scale=0.125, out=out2,
                           out_mx=mx2)
    assert torch.isfinite(out).all()
    assert torch.equal(out[:nb * L], out2)
    assert torch.equal(mx.q.view(-1)[:nb * L * E], mx2.q.view(-1)) and torch.equal(mx.s.view(-1)[:nb * L * E // 32], mx2.s.view(-1))
    want = ops.quant_mxfp8(out)
    assert torch.equal(mx.s, want.s) and torch.equal(mx.q, want.q)


def test_product_can_leave_as_one_bf16_plane(dev):
    """gemm_mxfp8(out_bf16=...): the round-to-nearest bf16 of the fp32 row it would have stored, on both product kernels."""
    from lr2ppo_amd import ops
    g = torch.Generator().manual_seed(12)
    for M, N, K in ((333, 512, 256), (8100, 2048, 256)):
        a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
        bias = torch.randn(N, generator=g) * 0.1
        am, bm = ops.quant_mxfp8(a.to(dev)), ops.quant_mxfp8(b.to(dev))
        out = torch.empty(M, N, device=dev)
        ops.gemm_mxfp8(am, bm, out, bias=bias.to(dev))
        ob = torch.empty(M * N, dtype=torch.bfloat16, device=dev)
        ops.gemm_mxfp8(am, bm, None, bias=bias.to(dev), out_bf16=ob)
        assert torch.equal(ob.view(M, N), out.to(torch.bfloat16))
