"""Round-2 parity additions on a real MI355X: encoder backward at ViT-B/16 / RoBERTa-base width against the reference's
autograd, DualEmbedding / DualEncoder (incl. tie_weights) forward + backward against the reference, the layer-level
forwards, uint8 frame patchify, out-of-range ids, deterministic embedding backward, NDCG on the device."""
import json
import os

import pytest
import torch

from conftest import GOLD, load_golden
from oracle import lr2ppo_oracle as O
from test_encoder_gpu import ROBERTA, VIT, _args, _cmp, _err
from test_oracle_golden import dual_case, enc_bwd_wide_case, sq_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["post", "pre"])
def test_encoder_backward_at_production_width_matches_reference_autograd(dev, tag):
    """2 layers x 768 x 12 heads, L = 197 (pre-LN) and L = 196 with padding (post-LN): the NT = 14 attention-backward
    instantiation, LayerNorm backward at D = 768, dgrad / wgrad with GELU' at 768 / 3072 -- vs the reference (sampled)."""
    from lr2ppo_amd.tencentpretrain.encoders import str2encoder
    g = load_golden("encoder_bwd_wide.npz")
    P, emb, wout, seg = enc_bwd_wide_case(tag)
    cfg = dict(VIT if tag == "pre" else ROBERTA, layers_num=2, layernorm_positioning=tag)
    enc = str2encoder["transformer"](_args(**cfg))
    enc.load_state_dict(P, strict=True)
    enc = enc.to(dev).eval()
    e = emb.to(dev).requires_grad_(True)
    out = enc(e, seg.to(dev))
    (out * wout.to(dev)).sum().backward()
    idx = g[f"{tag}_idx"]
    _cmp(out.detach().flatten().cpu()[idx], g[f"{tag}_out"], "out")
    _cmp(e.grad.flatten().cpu()[idx], g[f"{tag}_demb"], "demb")
    assert sq_close(float((e.grad.double() ** 2).sum()), float(g[f"{tag}_demb_sq"]), 2e-3)
    for n, p in enc.named_parameters():
        ref = g[f"{tag}_grad.{n}"]
        _cmp(p.grad.flatten().cpu()[g[f"{tag}_gidx.{n}"]], ref, n)
        assert sq_close(float((p.grad.double() ** 2).sum()), float(g[f"{tag}_gsq.{n}"]), 2e-3), n


DUAL_STREAM_TEXT = {"embedding": ["word", "pos", "seg"], "encoder": "transformer", "remove_embedding_layernorm": False,
                    "layernorm_positioning": "post", "max_seq_length": 20, "layers_num": 1}
DUAL_STREAM_VIT = {"embedding": ["patch", "pos"], "encoder": "transformer", "remove_embedding_layernorm": True,
                   "layernorm_positioning": "pre", "max_seq_length": 25, "layers_num": 1, "image_height": 32, "image_width": 48,
                   "patch_size": 8, "channels_num": 3}


def _dual_modules(tag, dev, dropout=0.1):
    from lr2ppo_amd.tencentpretrain.embeddings import DualEmbedding
    from lr2ppo_amd.tencentpretrain.encoders import DualEncoder
    s0, s1, tie = (DUAL_STREAM_TEXT, DUAL_STREAM_VIT, False) if tag == "tv" else (DUAL_STREAM_TEXT, DUAL_STREAM_TEXT, True)
    a = _args(**{**ROBERTA, "emb_size": 128, "hidden_size": 128, "feedforward_size": 256, "heads_num": 2, "layers_num": 1,
                 "dropout": dropout, "embedding": ["dual"], "encoder": "dual", "stream_0": dict(s0), "stream_1": dict(s1),
                 "tie_weights": tie, "image_height": 32, "image_width": 48, "patch_size": 8, "channels_num": 3})
    emb, enc = DualEmbedding(a, 100), DualEncoder(a)
    keys = json.load(open(os.path.join(GOLD, "dual_keys.json")))[tag]
    assert [[n, list(p.shape)] for n, p in emb.named_parameters()] == keys["embedding"]       # the reference's names and order
    assert [[n, list(p.shape)] for n, p in enc.named_parameters()] == keys["encoder"]
    pe, pn, _, _, _ = dual_case(tag)
    emb.load_state_dict({k: pe[k if k in pe else k.replace("embedding_1.", "embedding_0.")] for k in emb.state_dict()}, strict=True)
    enc.load_state_dict({k: pn[k if k in pn else k.replace("encoder_1.", "encoder_0.")] for k in enc.state_dict()}, strict=True)
    return emb.to(dev), enc.to(dev)


@pytest.mark.parametrize("tag", ["tv", "tt"])
def test_dual_embedding_and_dual_encoder_match_reference(dev, tag):
    """A15: DualEmbedding (inner + stream LayerNorm, both differentiable) + DualEncoder, untied text/image and tied
    text/text: outputs and every parameter gradient against the reference's autograd (eval mode = dropout off)."""
    g = load_golden("dual.npz")
    emb, enc = _dual_modules(tag, dev)
    emb.eval(), enc.eval()
    if tag == "tt":
        assert emb.embedding_0 is emb.embedding_1 and enc.encoder_0 is enc.encoder_1
    src = (g[f"{tag}_src0"].to(dev), g[f"{tag}_src1"].to(dev))
    seg = (g[f"{tag}_seg0"].to(dev), g[f"{tag}_seg1"].to(dev))
    e = emb(src, seg)
    h = enc(e, seg)
    ((h[0] * g[f"{tag}_w0"].to(dev)).sum() + (h[1] * g[f"{tag}_w1"].to(dev)).sum()).backward()
    for i in range(2):
        _cmp(e[i], g[f"{tag}_e{i}"], f"e{i}")
        _cmp(h[i], g[f"{tag}_h{i}"], f"h{i}")
    for n, p in emb.named_parameters():
        assert p.grad is not None, f"{n} received no gradient"
        _cmp(p.grad, g[f"{tag}_egrad.{n}"], n)
    for n, p in enc.named_parameters():
        _cmp(p.grad, g[f"{tag}_ngrad.{n}"], n)


def test_dual_embedding_train_mode_drops_both_streams_twice(dev):
    """dual_embedding.py:51-52 applies self.dropout to both streams on top of each Embedding's own dropout: with p = 0.5 a
    train-mode output keeps about a quarter of its entries (text stream: the inner drop is applied before the stream
    LayerNorm, so only the outer one shows as zeros), and the masks are reproducible from the runtime's seed."""
    from lr2ppo_amd import runtime
    g = load_golden("dual.npz")
    emb, _ = _dual_modules("tv", dev, dropout=0.5)
    emb.train()
    src = (g["tv_src0"].to(dev), g["tv_src1"].to(dev))
    seg = (g["tv_seg0"].to(dev), g["tv_seg1"].to(dev))
    runtime.set_dropout_seed(77)
    with torch.no_grad():
        a0, a1 = emb(src, seg)
    runtime.set_dropout_seed(77)
    with torch.no_grad():
        b0, b1 = emb(src, seg)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    z0, z1 = float((a0 == 0).float().mean()), float((a1 == 0).float().mean())
    assert 0.4 < z0 < 0.6, z0                 # text: outer dropout only is visible after the stream LayerNorm
    assert 0.65 < z1 < 0.85, z1               # image stream (no LayerNorms): inner and outer masks compound, 1 - 0.5^2
    # and the gradient flows through both dropouts and the stream LayerNorm to the embedding tables
    runtime.set_dropout_seed(77)
    o0, o1 = emb(src, seg)
    (o0.sum() + o1.sum()).backward()
    assert emb.embedding_0.word.embedding.weight.grad.abs().sum() > 0
    assert emb.stream_0_layer_norm.gamma.grad.abs().sum() > 0
    assert emb.embedding_1.patch.projection.weight.grad.abs().sum() > 0


@pytest.mark.parametrize("tag", ["post", "pre"])
def test_transformer_layer_forward_is_the_encoder_layer(dev, tag):
    """layers/transformer.py:50-73: TransformerLayer.forward(hidden, mask) on its own == the matching layer of the oracle
    encoder (no final LayerNorm), forward and backward; MultiHeadedAttention / PositionwiseFeedForward forwards too."""
    from lr2ppo_amd.tencentpretrain.layers.transformer import TransformerLayer
    a = _args(**{**ROBERTA, "hidden_size": 128, "emb_size": 128, "feedforward_size": 256, "heads_num": 2, "layers_num": 1,
                 "layernorm_positioning": tag, "dropout": 0.0})
    layer = TransformerLayer(a)
    spec = [(n, tuple(p.shape)) for n, p in layer.named_parameters()]
    P = O.seeded_params(spec, seed=5, std=0.2, skip_gamma_beta=False)
    layer.load_state_dict(P, strict=True)
    layer = layer.to(dev).eval()
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(2, 40, 128, generator=gen)
    seg = torch.ones(2, 40, dtype=torch.long)
    seg[1, 25:] = 0
    mask = ((1.0 - (seg > 0).unsqueeze(1).repeat(1, 40, 1).unsqueeze(1).float()) * -10000.0)
    Pg = {"transformer.0." + k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    # the oracle encoder adds the stack's final LayerNorm for pre-LN: give it an identity one and undo nothing -- compare
    # against the layer body computed directly
    mask_o = mask
    if tag == "post":
        ref = O.transformer_encoder(Pg, xr, seg, 1, 2, False)
    else:
        t = "transformer.0"
        inter = O.layernorm_tp(xr, Pg[f"{t}.layer_norm_1.gamma"], Pg[f"{t}.layer_norm_1.beta"])
        hmid = xr + O.tp_attention(Pg, f"{t}.self_attn", inter, mask_o, 2)
        o = O.layernorm_tp(hmid, Pg[f"{t}.layer_norm_2.gamma"], Pg[f"{t}.layer_norm_2.beta"])
        ref = O.linear(Pg, f"{t}.feed_forward.linear_2", O.gelu_erf(O.linear(Pg, f"{t}.feed_forward.linear_1", o))) + hmid
    w = torch.randn(2, 40, 128, generator=gen)
    (ref * w).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    out, prev = layer(xd, mask.to(dev))
    assert prev is None
    (out * w.to(dev)).sum().backward()
    _cmp(out, ref.detach(), "layer out")
    _cmp(xd.grad, xr.grad, "layer dx")
    for n, p in layer.named_parameters():
        if n.endswith("linear_layers.1.bias"):     # key bias: analytically zero gradient (softmax shift invariance), noise only
            assert p.grad.abs().max().item() < 1e-4
            continue
        _cmp(p.grad, Pg["transformer.0." + n].grad, n)
    with torch.no_grad():
        xin = x.to(dev)
        got, _ = layer.self_attn(xin, xin, xin, mask.to(dev))
        want = O.tp_attention({k: v.detach() for k, v in Pg.items()}, "transformer.0.self_attn", x, mask, 2)
        _cmp(got, want, "self_attn")
        got = layer.feed_forward(xin)
        Pd = {k: v.detach() for k, v in Pg.items()}
        want = O.linear(Pd, "transformer.0.feed_forward.linear_2", O.gelu_erf(O.linear(Pd, "transformer.0.feed_forward.linear_1", x)))
        _cmp(got, want, "feed_forward")
    causal = torch.triu(torch.full((40, 40), -10000.0), 1).view(1, 1, 40, 40).repeat(2, 1, 1, 1)
    with pytest.raises(NotImplementedError):
        layer(xd, causal.to(dev))
    with pytest.raises(RuntimeError):
        layer.self_attn(xd, xd, xd, mask.to(dev))          # grad-enabled call of the inference-only forward: loud


def test_uint8_frames_patchify_matches_loader_normalisation(dev):
    """uint8 frames -> (x / 255 - mean) / std -> patch rows, fused (dataloader.py:559-561 + patch_embedding.py:27): the
    planes written by the kernel equal the split of the torch expression bit for bit, for ps = 16, 8 and 14 (padded K)."""
    from lr2ppo_amd import ops
    gen = torch.Generator().manual_seed(3)
    for ps, H, W in ((16, 64, 48), (8, 32, 48), (14, 28, 56)):
        frames = torch.randint(0, 256, (3, 3, H, W), generator=gen, dtype=torch.uint8)
        x = frames.float().div(255)
        mean, std = torch.tensor(ops.CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(ops.CLIP_STD).view(1, 3, 1, 1)
        x = (x - mean) / std
        P = (H // ps) * (W // ps)
        Kd = 3 * ps * ps
        Kp = (Kd + 63) // 64 * 64
        rows = x.view(3, 3, H // ps, ps, W // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(3 * P, Kd)
        want = torch.zeros(3 * P, Kp)
        want[:, :Kd] = rows
        pl = ops.Planes.empty(3 * P, Kp, dev)
        ops.patchify_planes(frames.to(dev), pl, B=3, Cc=3, H=H, W=W, ps=ps, mean=ops.CLIP_MEAN, std=ops.CLIP_STD)
        ref = ops.split_planes(want.to(dev), ops.Planes.empty(3 * P, Kp, dev))
        assert torch.equal(pl.buf, ref.buf), ps
        pl2 = ops.Planes.empty(3 * P, Kp, dev)                 # fp32 images take the same route
        ops.patchify_planes(x.contiguous().to(dev), pl2, B=3, Cc=3, H=H, W=W, ps=ps)
        assert torch.equal(pl2.buf, ref.buf), ps


def test_vit_embedding_accepts_uint8_frames(dev):
    from lr2ppo_amd import ops
    from lr2ppo_amd.tencentpretrain.embeddings import Embedding, str2embedding
    a = _args(**VIT)
    emb = Embedding(a)
    for n in a.embedding:
        emb.update(str2embedding[n](a, 10), n)
    emb.load_state_dict(O.seeded_params(O.vit_embedding_spec(768, 3, 16, 197), seed=61), strict=True)
    emb = emb.to(dev).eval()
    gen = torch.Generator().manual_seed(4)
    frames = torch.randint(0, 256, (2, 3, 224, 224), generator=gen, dtype=torch.uint8)
    x = (frames.float().div(255) - torch.tensor(ops.CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(ops.CLIP_STD).view(1, 3, 1, 1)
    seg = torch.ones(2, 197, dtype=torch.long, device=dev)
    assert torch.equal(emb(frames.to(dev), seg), emb(x.to(dev), seg))
    P = O.seeded_params(O.vit_embedding_spec(768, 3, 16, 197), seed=61)
    _cmp(emb(frames.to(dev), seg), O.vit_embedding(P, x, 16), "vit embedding of uint8 frames")


def test_out_of_range_ids_raise_like_nn_embedding(dev):
    from lr2ppo_amd.tencentpretrain.embeddings import Embedding, str2embedding
    a = _args(**{**ROBERTA, "emb_size": 32, "max_seq_length": 20, "dropout": 0.0})
    emb = Embedding(a)
    for n in a.embedding:
        emb.update(str2embedding[n](a, 100), n)
    emb = emb.to(dev).eval()
    src = torch.randint(0, 100, (2, 8), device=dev)
    seg = torch.ones(2, 8, dtype=torch.long, device=dev)
    emb(src, seg)
    bad = src.clone()
    bad[1, 3] = 100
    with pytest.raises(IndexError, match="token id"):
        emb(bad, seg)
    emb(src, seg)                                           # the error word was cleared
    bad_seg = seg.clone()
    bad_seg[0, 0] = 3
    with pytest.raises(IndexError, match="segment id"):
        emb(src, bad_seg)
    with pytest.raises(IndexError):
        emb(torch.zeros(1, 21, dtype=torch.long, device=dev), torch.ones(1, 21, dtype=torch.long, device=dev))   # > max_seq_length
    emb.defer_id_check = True
    emb(bad, seg)                                           # deferred: no raise here ...
    with pytest.raises(IndexError):
        emb.check_ids()                                     # ... but at the caller's synchronisation point


def test_text_embedding_backward_is_deterministic_and_exact(dev):
    """Word / segment table gradients without float atomics: identical bits run to run, equal to an index_add in fp64."""
    from lr2ppo_amd import ops
    gen = torch.Generator().manual_seed(9)
    rows, D, vocab = 1500, 768, 50
    dx = torch.randn(rows, D, generator=gen)
    ids = torch.randint(0, vocab, (rows,), generator=gen)
    ids[:400] = 7                                           # one very frequent token
    seg = torch.randint(0, 3, (rows,), generator=gen)
    outs = []
    for _ in range(3):
        dword, dseg = torch.zeros(vocab + 5, D, device=dev), torch.zeros(3, D, device=dev)
        ops.text_embed_bwd(dx.to(dev), ids.to(dev), seg.to(dev), dword, dseg, rows=rows, D=D)
        outs.append((dword.cpu(), dseg.cpu()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    want_w = torch.zeros(vocab + 5, D, dtype=torch.float64).index_add_(0, ids, dx.double())
    want_s = torch.zeros(3, D, dtype=torch.float64).index_add_(0, seg, dx.double())
    assert (outs[0][0].double() - want_w).abs().max() < 1e-4
    assert (outs[0][1].double() - want_s).abs().max() < 2e-3
    assert outs[0][0][vocab:].abs().max() == 0


def test_ndcg_on_device_matches_reference_vectors_bit_for_bit(dev):
    """f-4: lr2_ndcg against the fixtures frozen from the reference's AverageNDCGMeter (ndcg.npz), against the host meter on
    256 ragged synthetic items, and the all-zero-gold branch."""
    from lr2ppo_amd import ops
    from lr2ppo_amd.ndcg import AverageNDCGMeter
    g = load_golden("ndcg.npz")
    n = int(g["n_cases"])
    scores, gold, offs = [], [], [0]
    for c in range(n):
        scores.append(g[f"scores_{c}"].float().view(-1))
        gold.append(g[f"gold_{c}"].long().view(-1))
        offs.append(offs[-1] + scores[-1].numel())
    out = ops.ndcg(torch.cat(scores).to(dev), torch.cat(gold).to(dev), torch.tensor(offs, dtype=torch.int64, device=dev)).cpu()
    for c in range(n):
        assert torch.equal(out[c], g[f"ndcg_{c}"].float()), (c, out[c], g[f"ndcg_{c}"])
    gen = torch.Generator().manual_seed(12)
    meter = AverageNDCGMeter()
    scores, gold, offs, want = [], [], [0], []
    for i in range(256):
        T = int(torch.randint(1, 21, (1,), generator=gen))
        s = torch.randn(T, generator=gen)
        t = torch.randint(0, 3, (T,), generator=gen) if i % 17 else torch.zeros(T, dtype=torch.long)
        scores.append(s), gold.append(t), offs.append(offs[-1] + T)
        want.append(meter.return_ndcg_at_k_from_scores(s, t))
    out = ops.ndcg(torch.cat(scores).to(dev), torch.cat(gold).to(dev), torch.tensor(offs, dtype=torch.int64, device=dev)).cpu()
    assert (out - torch.stack(want)).abs().max() < 1e-6
    assert torch.equal(out[0], torch.ones(6))              # item 0 has all-zero gold: ideal DCG 0 -> NDCG := 1


# ---- the 256 x 256 ping-pong NT kernel (csrc/gemm256.hip) ---------------------------------------------------------------------
def _planes(ops, x, dev):
    return ops.split_planes(x.to(dev).contiguous(), ops.Planes.empty(x.shape[0], x.shape[1], dev))


@pytest.mark.parametrize("M,N,K", [(256, 256, 32), (256, 256, 64), (512, 768, 768), (700, 520, 96), (1000, 3072, 160), (130, 260, 3072)])
def test_gemm256_nt_matches_fp64(dev, M, N, K):
    """Every K-step parity (1, 2, odd, even counts), ragged M and N, several tiles per workgroup column: against fp64, and
    against the general kernel (same split-bf16 arithmetic, different summation order)."""
    import math
    from lr2ppo_amd import ops
    from test_kernels_gpu import _close
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    ref = a.double() @ b.double().t()
    ap, bp = _planes(ops, a, dev), _planes(ops, b, dev)
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(ap, bp, out, M, N, K, block_m=256, splits=1)
    _close(out, ref, atol=6e-5 * math.sqrt(K), rtol=5e-5, what="gemm256")
    if K % 64 == 0:                         # the general NT kernel steps K by 64
        out128 = torch.empty((M, N), device=dev)
        ops.gemm(ap, bp, out128, M, N, K, block_m=128, splits=1)
        _close(out, out128.double().cpu(), atol=2e-5 * math.sqrt(K), rtol=2e-5, what="gemm256 vs general kernel")


def test_gemm256_exact_on_integers_and_asymmetric(dev):
    """Small-integer operands make every product and sum exact in the split-bf16 arithmetic: the tile / wave / quadrant /
    fragment index maps are checked bit for bit (an asymmetric B catches a transposed output; distinct values per row and
    column catch a swapped half or plane)."""
    from lr2ppo_amd import ops
    M, N, K = 512, 512, 128
    g = torch.Generator().manual_seed(1)
    a = torch.randint(-3, 4, (M, K), generator=g).float() + torch.arange(M).float().view(-1, 1) % 5
    b = torch.randint(-3, 4, (N, K), generator=g).float() + (torch.arange(N).float().view(-1, 1) % 7) * 2
    ref = (a.double() @ b.double().t()).float()
    out = torch.empty(M, N, device=dev)
    ops.gemm(_planes(ops, a, dev), _planes(ops, b, dev), out, M, N, K, block_m=256, splits=1)
    assert torch.equal(out.cpu(), ref)


def test_gemm256_fused_epilogues(dev):
    """bias + GELU (+ saved z) + planes output, dropout + residual, strided planes output (the QKV / FFN epilogues of the
    encoder forward), on the 256 kernel."""
    from lr2ppo_amd import ops
    from test_kernels_gpu import _close
    g = torch.Generator().manual_seed(19)
    M, N, K = 600, 512, 256
    a, w, bias, resid = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g), \
        torch.randn(M, N, generator=g)
    ap, wp = _planes(ops, a, dev), _planes(ops, w, dev)
    z_ref = a.double() @ w.double().t() + bias.double()
    out, z, pl = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), ops.Planes.empty(M, N, dev)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), act=1, out_z=z, out_planes=pl, block_m=256, splits=1)
    _close(z, z_ref, 1e-4, 5e-5, "z")
    _close(out, O.gelu_erf(z_ref), 1e-4, 5e-5, "gelu")
    ref_pl = ops.split_planes(out, ops.Planes.empty(M, N, dev))
    assert torch.equal(pl.buf, ref_pl.buf)                                  # the planes output is the split of the fp32 one
    drop = ops.Drop(0.1, seed=77, site=3)
    ops.gemm(ap, wp, out, M, N, K, bias=bias.to(dev), drop=drop, resid=resid.to(dev), block_m=256, splits=1)
    keep = torch.from_numpy(O.dropout_keep_mask(77, 3, M * N, 0.1)).view(M, N)
    _close(out, z_ref * keep.double() / 0.9 + resid.double(), 1e-4, 5e-5, "dropout+resid")


# ---- self-attention beyond one LDS-resident key block (L > 256) --------------------------------------------------------------
@pytest.mark.parametrize("batch,heads,L,drop_p", [(2, 16, 257, 0.0), (1, 3, 514, 0.0), (2, 2, 300, 0.1), (1, 2, 449, 0.0),
                                                    (1, 1, 700, 0.0)])
def test_self_attn_forward_blocked_keys(dev, batch, heads, L, drop_p):
    """ViT-L/14 (257 tokens), RoBERTa's max_seq_length (514) and beyond: the key-block loop with running max / sum against
    the fp64 reference formula -- key mask on the last sequence, probability dropout with the global key index, fp32 and
    planes outputs, log-sum-exp."""
    import numpy as np
    from lr2ppo_amd import ops
    from test_kernels_gpu import _close
    g = torch.Generator().manual_seed(L + heads)
    E = heads * 64
    qkv = torch.cat([torch.randn(batch * L, E, generator=g) * 0.3, torch.randn(batch * L, E, generator=g) * 0.3,
                     torch.randn(batch * L, E, generator=g)], dim=1)
    seg = torch.ones(batch, L, dtype=torch.long)
    seg[-1, (2 * L) // 3:] = 0
    mask = (1.0 - (seg > 0).double()).view(batch, 1, 1, L) * -10000.0
    mult = torch.ones(batch, heads, L, L, dtype=torch.float64)
    seed, site = 42, 7
    if drop_p > 0:
        keep = O.attention_keep_mask(seed, site, batch, heads, L, drop_p)
        mult = torch.from_numpy(np.asarray(keep, dtype=np.float64)).view(batch, heads, L, L) / (1.0 - drop_p)
    qh, kh, vh = (t.double().reshape(batch, L, heads, 64).transpose(1, 2) for t in qkv.split(E, dim=1))
    sc = qh @ kh.transpose(-2, -1) / 8.0 + mask
    ref = ((torch.softmax(sc, dim=-1) * mult) @ vh).transpose(1, 2).reshape(batch * L, E)
    qkv_p = ops.split_planes(qkv.to(dev), ops.Planes.empty(batch * L, 3 * E, dev))
    o = torch.full((batch * L, E), float("nan"), device=dev)
    lse = torch.empty(batch * heads * L, device=dev)
    drop = ops.Drop(drop_p, seed, site) if drop_p > 0 else None
    ops.self_attn_fwd(qkv_p, seg.to(dev).view(-1), o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=drop)
    _close(o, ref, 3e-5, 3e-5, "blocked self-attn")
    _close(lse.view(batch, heads, L), torch.logsumexp(sc, dim=-1), 1e-5, 1e-5, "lse")
    op = ops.Planes.empty(batch * L, E, dev)
    ops.self_attn_fwd(qkv_p, seg.to(dev).view(-1), op, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, drop=drop)
    assert torch.equal(op.buf, ops.split_planes(o, ops.Planes.empty(batch * L, E, dev)).buf)


def test_vit_l14_config_forward_matches_oracle(dev):
    """BASELINE config 5 groundwork: the ViT-L/14 encoder swap (lr2ppo_amd/configs/vit_large_14_224.json: hidden 1024, 24
    layers, 16 heads, patch 14 -> 257 tokens, patch rows padded 588 -> 640 for whole K tiles): embedding + the first two
    layers against the oracle on the host."""
    import json
    import os
    from lr2ppo_amd.finetune.features import EncoderStack, encoder_args
    cfg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lr2ppo_amd", "configs", "vit_large_14_224.json")
    full = json.load(open(cfg))
    assert (full["hidden_size"], full["layers_num"], full["heads_num"], full["patch_size"], full["max_seq_length"]) == (1024, 24, 16, 14, 257)
    a = encoder_args(cfg, layers_num=2)
    stack = EncoderStack(a, 10)
    pe = O.seeded_params(O.vit_embedding_spec(1024, 3, 14, 257), seed=91)
    pn = O.seeded_params(O.encoder_param_spec(2, 1024, 4096, True), seed=92)
    stack.embedding.load_state_dict(pe, strict=True)
    stack.encoder.load_state_dict(pn, strict=True)
    stack = stack.to(dev).eval()
    gen = torch.Generator().manual_seed(93)
    img = torch.randn(2, 3, 224, 224, generator=gen)
    seg = torch.ones(2, 257, dtype=torch.long)
    with torch.no_grad():
        got = stack(img.to(dev), seg.to(dev))
        want = O.transformer_encoder(pn, O.vit_embedding(pe, img, 14), seg, 2, 16, True)
    assert got.shape == (2, 257, 1024)
    _cmp(got, want, "ViT-L/14, 2 layers")
    # training path at 257 tokens: the key / query block loops of the attention backward -- parameter gradients of both layers
    # against the oracle's autograd (eval mode: no dropout to pin)
    for prm in stack.parameters():
        prm.grad = None
    w = torch.randn(2, 257, 1024, generator=gen)
    (stack(img.to(dev), seg.to(dev)) * w.to(dev)).sum().backward()
    pg = {k: v.clone().requires_grad_(True) for k, v in pn.items()}
    (O.transformer_encoder(pg, O.vit_embedding(pe, img, 14), seg, 2, 16, True) * w).sum().backward()
    for name in ("transformer.0.self_attn.linear_layers.0.weight", "transformer.0.self_attn.linear_layers.1.weight",
                 "transformer.0.self_attn.linear_layers.2.bias", "transformer.1.self_attn.linear_layers.0.weight",
                 "transformer.1.feed_forward.linear_1.weight", "transformer.0.layer_norm_1.gamma"):
        got_g = dict(stack.encoder.named_parameters())[name].grad
        ref_g = pg[name].grad
        err = (got_g.cpu().double() - ref_g.double()).abs().max().item()
        assert err < 1e-3 * max(1.0, ref_g.abs().max().item()), (name, err, ref_g.abs().max().item())


# ---- mode = 'cls' (finetune/ppo.py:209-210,229-242,532-537,641-643,859-863) -----------------------------------------------------
def test_cls_mode_rollout_and_update_match_reference(dev):
    """The 3-way classification actor: logits, NLL loss, expected-label scores, the rollout record, the 10 metrics of one
    train_model cycle, sampled actor gradients and post-step weights against the reference's own run (cls_step.npz)."""
    import argparse
    from lr2ppo_amd.finetune import ppo
    g = load_golden("cls_step.npz")
    bs, tags = int(g["bs"]), int(g["tags"])
    args = argparse.Namespace(mode="cls", labels_num=3, seq_length=196, max_imgs=16, visual_feat_dim=768, is_master=False,
                              kl_div_loss_weight=0.001, entropy_weight=0.001, value_clip=0.5, optimizer="adamw", scheduler="linear",
                              learning_rate=1e-3, critic_learning_rate=1e-3, train_steps=41, warmup=0.1, device=dev,
                              fuse_fc1_update=False)
    model = ppo.ActorCritic(args, None)
    assert [(n, tuple(p.shape)) for n, p in model.actor.named_parameters()] == O.head_param_spec("actor", n_out=3)
    model.actor.load_state_dict(O.seeded_params(O.head_param_spec("actor", n_out=3), seed=17), strict=True)
    model.critic.load_state_dict(O.seeded_params(O.head_param_spec("critic"), seed=18), strict=True)
    reward = ppo.Reward(args, None)
    reward.load_state_dict(O.seeded_params(O.head_param_spec("reward"), seed=19), strict=True)
    model, reward = model.to(dev).eval(), reward.to(dev).eval()
    opt, copt, sch, csch = ppo.build_optimizer(args, model)
    sch.step(), csch.step()
    assert abs(opt.param_groups[0]["lr"] - float(g["lr"][0])) < 1e-12
    text, img, tgts = O.seeded_head_inputs(2000, bs, tags)
    with torch.no_grad():
        loss, logits = model.actor(text.to(dev), img.to(dev), tgts.to(dev))
    assert logits.shape == (bs * tags, 3)
    assert _err(logits, g["logits"]) < 1e-4 and abs(float(loss) - float(g["nll"])) < 1e-4
    rec = ppo.rollout_step(model, reward, text.to(dev), img.to(dev), tgts.to(dev))
    assert _err(rec[2], g["scores"]) < 1e-4 and _err(rec[4], g["value"]) < 1e-4 and _err(rec[3], g["reward"]) < 1e-4
    assert torch.equal(rec[1].cpu(), g["next_state"])
    # evaluate()'s raw-logit scores
    from lr2ppo_amd import ops
    ev = ops.cls_scores(logits, None, torch.empty(bs * tags, device=dev), rows=bs * tags, C=3, softmax=False)
    assert _err(ev, g["eval_scores"]) < 1e-4
    out = ppo.train_model(args, model, opt, copt, sch, csch, [rec], 1)
    for i, (a, b) in enumerate(zip(out, g["metrics"].tolist())):
        assert abs(a - b) < 1e-4, f"metric {i}: {a} vs {b}"
    named = dict(model.named_parameters())
    for key in [k for k in g if k.startswith("g.")]:
        n = key[2:]
        idx = g["idx." + n].to(dev)
        ref = g[key]
        got = named[n].grad.detach().flatten()[idx]
        assert _err(got, ref) < 1e-6 + 2e-3 * float(ref.abs().max()), f"grad {n}"
        assert _err(named[n].detach().flatten()[idx], g["w." + n]) < 2e-6, f"weights {n}"
    # NLL through autograd (the drop-in nn.Module path): gradient of the loss w.r.t. the head
    model.zero_grad()
    loss2, _ = model.actor(text.to(dev), img.to(dev), tgts.to(dev))
    loss2.backward()
    assert model.actor.head.weight.grad is not None and model.actor.head.weight.grad.abs().sum() > 0


# ---- last encoder layer for token 0 only ('first' pooling; TransformerEncoder.forward_first_token) --------------------------
@pytest.mark.parametrize("batch,heads,L", [(3, 2, 50), (2, 12, 197), (1, 1, 700)])
def test_first_token_attention_kernel(dev, batch, heads, L):
    """lr2_first_token_attn against the fp64 formula of multi_headed_attn.py:61-74 restricted to query 0 (key mask on the last
    sequence)."""
    import math
    from lr2ppo_amd import ops
    from test_kernels_gpu import _close
    g = torch.Generator().manual_seed(L + heads)
    E = heads * 64
    q = torch.randn(batch, E, generator=g)
    kv = torch.cat([torch.randn(batch * L, E, generator=g) * 0.5, torch.randn(batch * L, E, generator=g)], dim=1)
    seg = torch.ones(batch, L, dtype=torch.long)
    seg[-1, (2 * L) // 3:] = 0
    kv_p = _planes(ops, kv, dev)
    kvf = kv_p.to_float().double().cpu()                                           # what the kernel reads (hi + lo)
    k = kvf[:, :E].view(batch, L, heads, 64).permute(0, 2, 1, 3)
    v = kvf[:, E:].view(batch, L, heads, 64).permute(0, 2, 1, 3)
    s = torch.einsum("bhd,bhld->bhl", q.double().view(batch, heads, 64), k) / math.sqrt(64.0)
    s = s + (1.0 - (seg > 0).double()).view(batch, 1, L) * -10000.0
    ref = torch.einsum("bhl,bhld->bhd", torch.softmax(s, dim=-1), v).reshape(batch, E)
    o = torch.full((batch, E), float("nan"), device=dev)
    ops.first_token_attn(q.to(dev), kv_p, seg.view(-1).to(dev), o, batch=batch, heads=heads, L=L, head_dim=64,
                         scale=1.0 / math.sqrt(64.0))
    _close(o, ref, 2e-5, 2e-5, "first-token attention")


@pytest.mark.parametrize("tag,hidden,heads,L,layers", [("pre", 768, 12, 197, 2), ("post", 768, 12, 196, 2), ("pre", 128, 2, 50, 3),
                                                       ("post", 128, 2, 40, 1)])
def test_forward_first_token_is_row_zero_of_the_full_forward(dev, tag, hidden, heads, L, layers):
    """The pruned last layer (keys / values for every row; query, projection, feed-forward, LayerNorms for row 0) gives the
    full inference forward's row 0 -- both LayerNorm placements, key padding, ViT-B/16 and RoBERTa-base widths; with
    gradients enabled the call falls back to the full (differentiable) forward."""
    from lr2ppo_amd.tencentpretrain.encoders import str2encoder
    from test_encoder_gpu import _args, ROBERTA
    a = _args(**{**ROBERTA, "hidden_size": hidden, "emb_size": hidden, "feedforward_size": 4 * hidden, "heads_num": heads,
                 "layers_num": layers, "layernorm_positioning": tag, "dropout": 0.0})
    enc = str2encoder["transformer"](a)
    spec = [(n, tuple(p.shape)) for n, p in enc.named_parameters()]
    enc.load_state_dict(O.seeded_params(spec, seed=31, std=0.05, skip_gamma_beta=False), strict=True)
    enc = enc.to(dev).eval()
    gen = torch.Generator().manual_seed(32)
    B = 5
    x = torch.randn(B, L, hidden, generator=gen).to(dev)
    seg = torch.ones(B, L, dtype=torch.long)
    seg[1, L // 2:] = 0
    seg[4, L - 3:] = 0
    seg = seg.to(dev)
    with torch.no_grad():
        full = enc(x, seg)
        first = enc.forward_first_token(x, seg)
    assert first.shape == (B, hidden)
    err = (first - full[:, 0, :]).abs().max().item()
    assert err < 2e-5 * max(1.0, full[:, 0, :].abs().max().item()), err
    xg = x.clone().requires_grad_(True)
    out = enc.forward_first_token(xg, seg)                      # grad mode: full forward, sliced, differentiable
    out.sum().backward()
    assert xg.grad is not None and xg.grad.abs().sum() > 0
    assert (out.detach() - full[:, 0, :]).abs().max().item() < 2e-5 * max(1.0, full.abs().max().item())


def test_large_m_dgrad_through_transposed_weight_matches_nn_form(dev):
    """engine.linear_dgrad with w_f32: at M large enough for the 256 x 256 kernel the input gradient runs as dy @ (W^T)^T on a
    scratch transpose of the weight -- same products as the NN form on W (different summation order), with the GELU' epilogue
    (planes output), the plain fp32 output and the accumulate form."""
    from lr2ppo_amd import engine, ops
    g = torch.Generator().manual_seed(77)
    M, E, F = 77824, 768, 3072            # 304 x 3 tiles of 256 x 256 on the narrow output: 89 % of four rounds
    assert ops.use_gemm256(M, F, E) and ops.use_gemm256(M, E, F)
    ws = engine.Workspace(dev)
    w2 = (torch.randn(E, F, generator=g) * 0.05).to(dev)              # linear_2.weight [out = E, in = F]
    dy = torch.randn(M, E, generator=g).to(dev)
    z = torch.randn(M, F, generator=g).to(dev)
    dy_p, w2_p = _planes(ops, dy, dev), _planes(ops, w2, dev)
    a, b = ops.Planes.empty(M, F, dev), ops.Planes.empty(M, F, dev)
    engine.linear_dgrad(ws, dy_p, w2_p, None, M, F, E, act=2, aux_z=z, out_planes=a)
    engine.linear_dgrad(ws, dy_p, w2_p, None, M, F, E, act=2, aux_z=z, out_planes=b, w_f32=w2)
    ra, rb = a.to_float(), b.to_float()
    assert (ra - rb).abs().max().item() < 3e-5 * max(1.0, ra.abs().max().item())
    w1 = (torch.randn(F, E, generator=g) * 0.05).to(dev)              # linear_1.weight [out = F, in = E]
    dz_p, w1_p = _planes(ops, z, dev), _planes(ops, w1, dev)
    o1, o2 = torch.empty(M, E, device=dev), torch.empty(M, E, device=dev)
    engine.linear_dgrad(ws, dz_p, w1_p, o1, M, E, F)
    engine.linear_dgrad(ws, dz_p, w1_p, o2, M, E, F, w_f32=w1)
    assert (o1 - o2).abs().max().item() < 3e-5 * max(1.0, o1.abs().max().item())
    ref = z[:64].double() @ w1.double()
    assert (o2[:64].double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    base = torch.randn(M, E, generator=g).to(dev)
    o3 = base.clone()
    engine.linear_dgrad(ws, dz_p, w1_p, o3, M, E, F, accumulate=True, w_f32=w1)
    assert (o3 - (base + o2)).abs().max().item() < 1e-4 * max(1.0, o2.abs().max().item())
