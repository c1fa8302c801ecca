"""Randomised shape sweep of the GEMM family and the self-attention kernels against fp64 (run on the GPU box; seeds fixed).
    python tools/dbg/fuzz_kernels.py [--n 150] [--seed 0]"""
import argparse, math, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from lr2ppo_amd import ops
from oracle import lr2ppo_oracle as O
from oracle.cpu_threads import fit_torch_threads  # noqa: E402

fit_torch_threads()          # one torch thread per usable core (cgroup quota): the fp64 references run on the host

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=150)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--encoders", type=int, default=0, help="random small TransformerEncoder configs: forward, first-token forward and backward vs the oracle")
ap.add_argument("--trad", type=int, default=0, help="random (queries, documents, index) cases of the sequence-length-1 heads (ppo_trad / pointwise_2data_trad)")
ap.add_argument("--head-grads", type=int, default=0, help="train-mode (dropout pinned) value + gradients of the full-size Critic for random (batch, tags, index) vs the oracle's autograd")
ap.add_argument("--heads", type=int, default=0, help="random (batch, tags, index) cases of the full-size Actor / Critic / Reward vs the oracle")
a = ap.parse_args()
dev = torch.device("cuda:0")
rng = np.random.default_rng(a.seed)
g = torch.Generator().manual_seed(a.seed)


def planes(x):
    return ops.split_planes(x.to(dev).contiguous(), ops.Planes.empty(x.shape[0], x.shape[1], dev))


bad = 0
# ---- GEMM: forms x operand kinds x tile choices x epilogues ----
for it in range(a.n):
    form = ["NT", "NN", "TN"][int(rng.integers(0, 3))]
    M = int(rng.integers(1, 1400))
    N = int(rng.integers(1, 300)) * 4
    K = int(rng.integers(1, 12)) * 64
    bm = [None, 64, 128, 256][int(rng.integers(0, 4))]
    ta, tb = form == "TN", form in ("NN", "TN")
    if ta:
        M = (M + 7) // 8 * 8            # A is [K, M]: 16-byte rows of the planes / fp32 operand
    if tb:
        N = (N + 7) // 8 * 8            # B is [K, N]
    if bm == 256 and form != "NT":
        bm = None
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((K, N) if tb else (N, K), generator=g) * 0.1
    ref = (A.t() if ta else A).double() @ (B if tb else B.t()).double()
    use_planes = bool(rng.integers(0, 2)) or bm == 256
    Ad, Bd = (planes(A), planes(B)) if use_planes else (A.to(dev), B.to(dev))
    epi = int(rng.integers(0, 7))
    bias = torch.randn(N, generator=g) if epi in (1, 2) else None
    resid = torch.randn(M, N, generator=g) if epi == 2 else None
    out = torch.full((M, N), float("nan"), device=dev)
    pl = ops.Planes.empty(M, N, dev) if epi == 3 else None
    aux_z = torch.randn(M, N, generator=g) if epi == 4 else None          # GELU' epilogue (dgrad through a GELU)
    base = torch.randn(M, N, generator=g) if epi == 5 else None           # accumulate into the output
    if base is not None:
        out = base.to(dev).clone()
    drop = ops.Drop(0.1, seed=4000 + it, site=2) if epi == 6 else None
    splits = None
    if bm is not None:
        splits = 1 if bm == 256 else ops.choose_tiling(M, N, K, ta, tb)[1]
    ws = torch.empty(64 * M * N, device=dev) if (splits or 0) != 1 and M * N < 4_000_000 else None
    kw = dict(trans_a=ta, trans_b=tb, bias=None if bias is None else bias.to(dev), resid=None if resid is None else resid.to(dev),
              act=1 if epi == 1 else (2 if epi == 4 else 0), out_planes=pl, block_m=bm, splits=splits, splitk_ws=ws,
              aux_z=None if aux_z is None else aux_z.to(dev), accumulate=base is not None, drop=drop)
    try:
        ops.gemm(Ad, Bd, out, M, N, K, **kw)
    except Exception as e:                                    # noqa: BLE001
        print("gemm raised", form, M, N, K, bm, splits, epi, repr(e)[:120], flush=True)
        bad += 1
        continue
    want = ref + (bias.double() if bias is not None else 0.0)
    if epi == 1:
        want = O.gelu_erf(want)
    if resid is not None:
        want = want + resid.double()
    if aux_z is not None:
        zz = aux_z.double()
        want = want * (0.5 * (1.0 + torch.erf(zz / math.sqrt(2.0))) + zz * torch.exp(-0.5 * zz * zz) / math.sqrt(2.0 * math.pi))
    if base is not None:
        want = want + base.double()
    if drop is not None:
        keep = torch.from_numpy(np.asarray(O.dropout_keep_mask(4000 + it, 2, M * N, 0.1), dtype=np.float64)).view(M, N)
        want = want * keep / 0.9
    err = (out.double().cpu() - want).abs().max().item()
    tol = 8e-5 * math.sqrt(K) * max(1.0, float(want.abs().max()) / 10)
    ok = err < tol and (pl is None or (pl.to_float().double().cpu() - want).abs().max().item() < 2 * tol)
    if not ok:
        bad += 1
        print("GEMM MISMATCH", form, M, N, K, "bm", bm, "splits", splits, "planes", use_planes, "epi", epi, "err", err, "tol", tol, flush=True)
print("gemm cases done", a.n, "bad", bad, flush=True)

# ---- TN form of the 256 x 256 kernel (long contractions, K-splits, ragged everything) with the fused column sums of A ----
for it in range(max(8, a.n // 6)):
    M = int(rng.integers(1, 140)) * 8
    N = int(rng.integers(1, 140)) * 8
    K = int(rng.integers(33, 9000))
    steps = (K + 31) // 32
    splits = int(rng.integers(1, max(2, min(40, steps))))
    with_cs = bool(rng.integers(0, 2)) and M % 4 == 0
    A = torch.randn(K, M, generator=g) + 0.2
    B = torch.randn(K, N, generator=g) * 0.1
    Ad, Bd = planes(A), planes(B)
    out = torch.full((M, N), float("nan"), device=dev)
    ws = torch.empty(splits * M * N, device=dev)
    db = torch.full((M,), float("nan"), device=dev) if with_cs else None
    cs_ws = torch.empty(max(128, splits * ((N + 255) // 256)) * M, device=dev) if with_cs else None
    try:
        ops.gemm(Ad, Bd, out, M, N, K, trans_a=True, trans_b=True, lda=M, ldb=N, block_m=256, splits=splits, splitk_ws=ws, colsum=db,
                 colsum_ws=cs_ws)
    except Exception as e:                                    # noqa: BLE001
        print("gemm256 TN raised", M, N, K, splits, repr(e)[:120], flush=True)
        bad += 1
        continue
    want = A.double().t() @ B.double()
    err = (out.double().cpu() - want).abs().max().item()
    tol = 8e-5 * math.sqrt(K) * max(1.0, float(want.abs().max()) / 10)
    ok = err < tol
    if with_cs:
        ref_cs = Ad.to_float().double().sum(0).cpu()
        ok = ok and (db.double().cpu() - ref_cs).abs().max().item() < 3e-5 * math.sqrt(K) * max(1.0, float(ref_cs.abs().max()) / 10)
    if not ok:
        bad += 1
        print("GEMM256 TN MISMATCH", M, N, K, "splits", splits, "colsum", with_cs, "err", err, "tol", tol, flush=True)
print("gemm256 TN cases done, bad", bad, flush=True)

# ---- self-attention forward + backward, first-token attention ----
for it in range(max(10, a.n // 5)):
    batch, heads = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    L = int(rng.integers(1, 620))
    drop_p = [0.0, 0.1][int(rng.integers(0, 2))]
    E = heads * 64
    qkv = torch.cat([torch.randn(batch * L, E, generator=g) * 0.3, torch.randn(batch * L, E, generator=g) * 0.3,
                     torch.randn(batch * L, E, generator=g)], dim=1)
    do = torch.randn(batch * L, E, generator=g)
    seg = torch.ones(batch, L, dtype=torch.long)
    cut = int(rng.integers(1, L + 1))
    seg[-1, cut:] = 0
    mask = (1.0 - (seg > 0).double()).view(batch, 1, 1, L) * -10000.0
    seed, site = 1234 + it, 3
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
    qkv_p, do_p = planes(qkv), planes(do)
    o = torch.full((batch * L, E), float("nan"), device=dev)
    ops.self_attn_fwd(qkv_p, seg.to(dev).view(-1), o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=drop)
    e_f = (o.double().cpu() - ref_o.detach()).abs().max().item()
    dqkv = ops.Planes.empty(batch * L, 3 * E, dev)
    dqkv.buf.fill_(0x7FC0)
    w1, w2 = torch.empty(batch * heads * L, device=dev), torch.empty(batch * heads * L, device=dev)
    ops.self_attn_bwd(qkv_p, do_p, seg.to(dev).view(-1), dqkv, w1, w2, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=drop)
    e_b = (dqkv.to_float().double().cpu() - x.grad).abs().max().item()
    sb = max(1.0, float(x.grad.abs().max()))
    q0 = qkv[:, :E].view(batch, L, E)[:, 0, :].contiguous()
    kv_p = planes(qkv[:, E:].contiguous())
    o0 = torch.full((batch, E), float("nan"), device=dev)
    ops.first_token_attn(q0.to(dev), kv_p, seg.to(dev).view(-1), o0, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125)
    ref0 = (torch.softmax(sc.detach(), dim=-1) @ vh.detach()).transpose(1, 2).reshape(batch, L, E)[:, 0, :]
    e_0 = (o0.double().cpu() - ref0).abs().max().item()
    if not (e_f < 5e-5 and e_b < 1e-4 * sb and e_0 < 5e-5):
        bad += 1
        print("ATTN MISMATCH batch", batch, "heads", heads, "L", L, "drop", drop_p, "cut", cut, "fwd", e_f, "bwd", e_b, "first", e_0, flush=True)
print("attention cases done; total bad", bad, flush=True)
# ---- LayerNorm forward / backward (both semantics), XiT attention, PPO loss ----
def close(a, b, atol, rtol):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return bool(((a - b).abs() <= atol + rtol * b.abs()).all())


for it in range(max(10, a.n // 5)):
    rows, D = int(rng.integers(1, 1500)), int(rng.integers(1, 33)) * 32
    mode = int(rng.integers(0, 2))
    drop_p = [0.0, 0.1][int(rng.integers(0, 2))]
    x, gam, bet = torch.randn(rows, D, generator=g) * 2 + 0.5, torch.randn(D, generator=g), torch.randn(D, generator=g)
    dy, rg = torch.randn(rows, D, generator=g), torch.randn(rows, D, generator=g)
    eps = 1e-5 if mode == 0 else 1e-6
    ln = O.layernorm_torch if mode == 0 else O.layernorm_tp
    xt, gt, bt = x.double().requires_grad_(True), gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    ref = ln(xt, gt, bt)
    ref.backward(dy.double())
    out, mean, rstd = torch.empty(rows, D, device=dev), torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    ops.layernorm_fwd(x.to(dev), gam.to(dev), bet.to(dev), out, mean, rstd, rows=rows, D=D, eps=eps, mode=mode)
    dx, dxm = torch.empty(rows, D, device=dev), ops.Planes.empty(rows, D, dev)
    dgam, dbet = torch.empty(D, device=dev), torch.empty(D, device=dev)
    drop = ops.Drop(drop_p, 99 + it, 4) if drop_p > 0 else None
    ops.layernorm_bwd(dy.to(dev), x.to(dev), gam.to(dev), mean, rstd, dx, torch.empty(ops.LN_BWD_BLOCKS * 2 * D, device=dev), dgam, dbet, rows=rows,
                      D=D, resid_grad=rg.to(dev), dx_planes=dxm, drop=drop, mode=mode, eps=eps)
    ref_dx = xt.grad + rg.double()
    keep = torch.ones(rows, D, dtype=torch.float64)
    if drop_p > 0:
        keep = torch.from_numpy(O.dropout_keep_mask(99 + it, 4, rows * D, drop_p)).view(rows, D).double() / (1 - drop_p)
    sc = max(1.0, float(ref_dx.abs().max()))
    ok = (close(out, ref, 2e-5, 2e-5) and close(dx, ref_dx, 3e-5 * sc, 3e-5) and close(dgam, gt.grad, 2e-4 * math.sqrt(rows), 1e-4)
          and close(dbet, bt.grad, 2e-4 * math.sqrt(rows), 1e-4) and close(dxm.to_float(), ref_dx * keep, 2e-4 * sc, 5e-5))
    if not ok:
        bad += 1
        print("LN MISMATCH rows", rows, "D", D, "mode", mode, "drop", drop_p, flush=True)
print("layernorm cases done; total bad", bad, flush=True)

for it in range(max(10, a.n // 5)):
    batch, heads = int(rng.integers(1, 6)), 8
    Lq, Lk, hd = int(rng.integers(1, 257)), int(rng.integers(1, 17)), int(rng.integers(1, 25)) * 4
    E = heads * hd
    q, k = torch.randn(batch, Lq, E, generator=g) * 0.5, torch.randn(batch, Lk, E, generator=g) * 0.5
    v, do = torch.randn(batch, Lk, E, generator=g), torch.randn(batch, Lq, E, generator=g)
    scale = 1.0 / math.sqrt(E)
    qt, kt, vt = (t.double().requires_grad_(True) for t in (q, k, v))
    qh = qt.view(batch, Lq, heads, hd).permute(0, 2, 1, 3)
    kh = kt.view(batch, Lk, heads, hd).permute(0, 2, 1, 3)
    vh = vt.view(batch, Lk, heads, hd).permute(0, 2, 1, 3)
    ref = ((torch.softmax(qh @ kh.transpose(-1, -2), dim=-1) * scale) @ vh).permute(0, 2, 1, 3).reshape(batch, Lq, E)
    ref.backward(do.double())
    qd, kd, vd = q.to(dev).view(-1, E), k.to(dev).view(-1, E), v.to(dev).view(-1, E)
    o = torch.empty(batch * Lq, E, device=dev)
    ops.xattn_fwd(qd, kd, vd, o, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=scale)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    ops.xattn_bwd(qd, kd, vd, do.to(dev).view(-1, E), dq, dk, dv, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=scale)
    ok = (close(o, ref.view(-1, E), 1e-6, 2e-5) and close(dq, qt.grad.view(-1, E), 2e-6, 2e-4) and close(dk, kt.grad.view(-1, E), 2e-5, 2e-4)
          and close(dv, vt.grad.view(-1, E), 2e-5, 2e-4))
    if not ok:
        bad += 1
        print("XATTN MISMATCH batch", batch, "Lq", Lq, "Lk", Lk, "hd", hd, flush=True)
print("xattn cases done; total bad", bad, flush=True)

for it in range(max(10, a.n // 5)):
    B, T = int(rng.integers(1, 65)), int(rng.integers(2, 5))
    scores = torch.randn(B, T, generator=g) * 0.3
    old = scores + torch.randn(B, T, generator=g) * 0.05
    rewards, old_value = torch.randn(B, generator=g) * 0.2, torch.randn(B, generator=g) * 0.2
    value = old_value + torch.randn(B, generator=g) * 0.6
    state = torch.stack([torch.randperm(T, generator=g) for _ in range(B)])
    nxt = torch.cat([torch.arange(2).unsqueeze(0).repeat(B, 1), state], dim=1)
    st, vt = scores.clone().requires_grad_(True), value.clone().requires_grad_(True)
    loss, vloss, ex = O.ppo_update_math(st, vt, old, rewards, old_value, nxt, 0.001, 0.001, 0.5)
    loss.backward()
    vloss.backward()
    scal, per = torch.empty(4, device=dev), torch.empty(4, B, device=dev)
    ds, dv = torch.empty(B, T, device=dev), torch.empty(B, device=dev)
    ops.ppo_loss(scores.to(dev), old.to(dev), rewards.to(dev), old_value.to(dev), value.to(dev), nxt.to(dev), scal, per, ds, dv, B=B, T=T,
                 kl_w=0.001, ent_w=0.001, value_clip=0.5)
    ok = (close(scal[0], loss, 2e-7, 2e-5) and close(scal[1], vloss, 2e-7, 2e-5) and close(scal[2], ex["rank_loss"], 2e-7, 2e-5)
          and close(per[3], ex["advantages"], 2e-7, 2e-5) and close(ds, st.grad, 2e-8, 2e-4) and close(dv, vt.grad, 2e-8, 2e-4))
    if not ok:
        bad += 1
        print("PPO LOSS MISMATCH B", B, "T", T, flush=True)
print("ppo loss cases done; total bad", bad, flush=True)
# ---- TransformerEncoder at random small configurations: inference forward, first-token forward, training backward ----
if a.encoders:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
    from test_encoder_gpu import _args, ROBERTA
    from lr2ppo_amd.tencentpretrain.encoders import str2encoder
    for it in range(a.encoders):
        heads = int(rng.integers(1, 5))
        hidden, ff = heads * 64, int(rng.integers(1, 5)) * 64 * heads
        layers, tag = int(rng.integers(1, 3)), ["pre", "post"][int(rng.integers(0, 2))]
        B, L = int(rng.integers(1, 4)), int(rng.integers(2, 330))
        ea = _args(**{**ROBERTA, "hidden_size": hidden, "emb_size": hidden, "feedforward_size": ff, "heads_num": heads,
                      "layers_num": layers, "layernorm_positioning": tag, "dropout": 0.0})
        enc = str2encoder["transformer"](ea)
        spec = [(n, tuple(p.shape)) for n, p in enc.named_parameters()]
        P = O.seeded_params(spec, seed=500 + it, std=0.08, skip_gamma_beta=False)
        enc.load_state_dict(P, strict=True)
        enc = enc.to(dev).eval()
        x = torch.randn(B, L, hidden, generator=g)
        seg = torch.ones(B, L, dtype=torch.long)
        seg[-1, int(rng.integers(1, L + 1)):] = 0
        w = torch.randn(B, L, hidden, generator=g)
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        xr = x.clone().requires_grad_(True)
        ref = O.transformer_encoder(Pg, xr, seg, layers, heads, tag == "pre")
        (ref * w).sum().backward()
        with torch.no_grad():
            got = enc(x.to(dev), seg.to(dev))
            first = enc.forward_first_token(x.to(dev), seg.to(dev))
        xd = x.to(dev).requires_grad_(True)
        for prm in enc.parameters():
            prm.grad = None
        (enc(xd, seg.to(dev)) * w.to(dev)).sum().backward()
        sc = max(1.0, float(ref.detach().abs().max()))
        e_f = (got.cpu() - ref.detach()).abs().max().item() / sc
        e_1 = (first.cpu() - ref.detach()[:, 0, :]).abs().max().item() / sc
        e_x = (xd.grad.cpu() - xr.grad).abs().max().item() / max(1.0, float(xr.grad.abs().max()))
        e_p = 0.0
        for n, prm in enc.named_parameters():
            if n.endswith("linear_layers.1.bias"):          # key-projection bias: analytically zero gradient (rounding noise on both sides)
                continue
            rg = Pg[n].grad
            e_p = max(e_p, (prm.grad.cpu() - rg).abs().max().item() / max(1.0, float(rg.abs().max())))
        if max(e_f, e_1) > 2e-4 or max(e_x, e_p) > 2e-3:
            bad += 1
            print("ENCODER MISMATCH", dict(heads=heads, ff=ff, layers=layers, ln=tag, B=B, L=L), "fwd", e_f, "first", e_1, "dx", e_x, "dparam", e_p, flush=True)
    print("encoder cases done; total bad", bad, flush=True)

# ---- sequence-length-1 heads (`_trad` twins): random query / document counts, index orders, both LETOR feature widths ----
if a.trad:
    import argparse as _ap2
    from lr2ppo_amd.finetune import ppo_trad, pointwise_2data_trad as p2
    targs = _ap2.Namespace(mode="reg", labels_num=3)
    Pa, Pc = O.seeded_params(O.trad_head_param_spec("actor"), seed=61), O.seeded_params(O.trad_head_param_spec("critic"), seed=62)
    P2 = O.seeded_params(O.trad2_param_spec(), seed=63)
    for Pd in (Pa, Pc, P2):
        Pd["head.weight"] = Pd["head.weight"] * 20.0
    actor, critic, two = ppo_trad.Actor(targs, None), ppo_trad.Critic(targs, None), p2.Classifier(targs, None)
    actor.load_state_dict(Pa, strict=True), critic.load_state_dict(Pc, strict=True), two.load_state_dict(P2, strict=True)
    actor, critic, two = actor.to(dev).eval(), critic.to(dev).eval(), two.to(dev).eval()
    for it in range(a.trad):
        bs, docs = int(rng.integers(1, 9)), int(rng.integers(1, 41))
        feats = torch.randn(bs, docs, 768, generator=g)
        idx = torch.from_numpy(rng.integers(0, docs, size=(bs, int(rng.integers(1, 5))))).long()
        width = [46, 136][int(rng.integers(0, 2))]
        raw = torch.randn(bs, docs, width, generator=g)
        with torch.no_grad():
            e1 = (actor(feats.to(dev), None, None).cpu().view(-1) - O.trad_actor_forward(Pa, feats)).abs().max().item()
            e2 = (critic(feats.to(dev), None, None, idx.to(dev)).cpu() - O.trad_critic_forward(Pc, feats, idx)).abs().max().item()
            e3 = (two(raw.to(dev), None, None).cpu().view(-1) - O.trad2_forward(P2, raw).view(-1)).abs().max().item()
        if max(e1, e2, e3) > 2e-4:
            bad += 1
            print("TRAD MISMATCH bs", bs, "docs", docs, "idx", idx.shape, "width", width, "errs", (e1, e2, e3), flush=True)
    print("trad cases done; total bad", bad, flush=True)

# ---- full-size Critic in train mode: value and gradients for random batch / tags / index (duplicates) vs oracle autograd ----
if a.head_grads:
    import argparse as _ap3
    from lr2ppo_amd import runtime
    from lr2ppo_amd.finetune import ppo as _ppo
    hargs = _ap3.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768)
    Pc = O.seeded_params(O.head_param_spec("critic"), seed=8)
    critic = _ppo.Critic(hargs, None)
    critic.load_state_dict(Pc, strict=True)
    critic = critic.to(dev).train()
    names = ["text_proj.fc1.weight", "img_proj.fc2.weight", "xit.0.0.0.fn.1.keys.weight", "xit.0.0.1.fn.1.0.weight", "xit.1.0.weight",
             "out_layer.fc1.weight", "out_layer.fc1.bias", "out_layer.fc2.weight", "pos_emb.weight", "xitt.0.0.0.fn.1.queries.weight",
             "xitt.0.0.1.fn.1.3.bias", "head.weight", "head.bias"]
    for it in range(a.head_grads):
        bs, tags = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        text, img, _ = O.seeded_head_inputs(8000 + it, bs, tags)
        index = torch.from_numpy(rng.integers(0, tags, size=(bs, int(rng.integers(1, 5))))).long()
        wv = torch.randn(bs, generator=g)
        runtime.set_dropout_seed(3000 + it, calls=1)
        seed = runtime.peek_drop_seed()
        value = critic.engine_forward(text.to(dev), img.to(dev), index.to(dev), save=True)
        critic.engine_backward(wv.to(dev))
        Pg = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
        ref_v = O.critic_forward(Pg, text, img, index, drop={"p": 0.1, "seed": seed, "site_base": 0})
        (ref_v * wv).sum().backward()
        G = critic.grad_buffers()
        e_v = (value.cpu() - ref_v.detach()).abs().max().item()
        e_g = max((G[n].cpu() - Pg[n].grad).abs().max().item() / max(1e-12, float(Pg[n].grad.abs().max())) for n in names
                  if float(Pg[n].grad.abs().max()) > 0)
        if e_v > 1e-4 or e_g > 3e-3:
            bad += 1
            print("HEAD GRAD MISMATCH bs", bs, "tags", tags, "index", index.tolist(), "value err", e_v, "rel grad err", e_g, flush=True)
    print("head gradient cases done; total bad", bad, flush=True)

# ---- full-size heads: random batch / tag counts / index orders (duplicates allowed) against the CPU oracle ----
if a.heads:
    import argparse as _ap
    from lr2ppo_amd.finetune import ppo
    hargs = _ap.Namespace(mode="reg", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768)
    models = {}
    for kind, cls, sd in (("actor", ppo.Actor, 7), ("critic", ppo.Critic, 8), ("reward", ppo.Reward, 9)):
        P = O.seeded_params(O.head_param_spec(kind), seed=sd)
        m = cls(hargs, None)
        m.load_state_dict(P, strict=True)
        models[kind] = (m.to(dev).eval(), P)
    for it in range(a.heads):
        bs, tags = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        text, img, _ = O.seeded_head_inputs(7000 + it, bs, tags)
        idx_c = torch.from_numpy(rng.integers(0, tags, size=(bs, int(rng.integers(1, 5))))).long()
        idx_r = torch.from_numpy(rng.integers(0, tags, size=(bs, 4))).long()
        with torch.no_grad():
            got_a = models["actor"][0](text.to(dev), img.to(dev), None).cpu()
            got_c = models["critic"][0](text.to(dev), img.to(dev), None, idx_c.to(dev)).cpu()
            got_r = models["reward"][0](text.to(dev), img.to(dev), None, idx_r.to(dev)).cpu()
            ref_a = O.actor_forward(models["actor"][1], text, img, None)
            ref_c = O.critic_forward(models["critic"][1], text, img, idx_c)
            ref_r = O.reward_forward(models["reward"][1], text, img, idx_r)
        errs = [(got_a.view(-1) - ref_a.view(-1)).abs().max().item(), (got_c - ref_c).abs().max().item(), (got_r - ref_r).abs().max().item()]
        if max(errs) > 1e-4:
            bad += 1
            print("HEAD MISMATCH bs", bs, "tags", tags, "idx_c", idx_c.tolist(), "errs", errs, flush=True)
    print("head cases done; total bad", bad, flush=True)
sys.exit(1 if bad else 0)
