"""Dual-encoder forward micro-benchmark (ViT-B/16 + RoBERTa-base, random weights, batch 32 of the repo's default shapes).
Prints ms per forward, algorithmic TFLOP/s (2MNK of the GEMMs + 4 L^2 d of attention) and a per-kernel-class breakdown
from HIP events.  python tools/encoder_bench.py [--batch 32] [--iters 10] [--passes 3]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lr2ppo_amd import ops  # noqa: E402
from lr2ppo_amd.tencentpretrain.encoders import str2encoder  # noqa: E402
from lr2ppo_amd.tencentpretrain.opts import finetune_opts, tokenizer_opts  # noqa: E402

VIT = dict(emb_size=768, feedforward_size=3072, hidden_size=768, hidden_act="gelu", heads_num=12, layers_num=12, dropout=0.1,
           max_seq_length=197, encoder="transformer", mask="fully_visible", layernorm_positioning="pre")
ROBERTA = dict(emb_size=768, feedforward_size=3072, hidden_size=768, hidden_act="gelu", heads_num=12, layers_num=12,
               max_seq_length=514, dropout=0.1, encoder="transformer", mask="fully_visible")


def _args(**over):
    p = argparse.ArgumentParser()
    finetune_opts(p)
    tokenizer_opts(p)
    d = vars(p.parse_args([]))
    d.update(over)
    return argparse.Namespace(**d)


def flops(B, L, E=768, F=3072, layers=12, heads=12, first_token_only=False):
    """Matrix flops of one encoder forward.  first_token_only: the last layer as TransformerEncoder.forward_first_token runs
    it (keys / values for every row; query, scores, projection and feed-forward for row 0 of each sequence)."""
    M = B * L
    gemm = 2.0 * M * E * (3 * E + E + 2 * F)
    attn = 4.0 * B * heads * L * L * (E // heads)
    if not first_token_only:
        return layers * (gemm + attn)
    last = 2.0 * M * E * (2 * E) + 2.0 * B * E * (E + E + 2 * F) + 4.0 * B * heads * L * (E // heads)
    return (layers - 1) * (gemm + attn) + last


def measure_forward(batch_items=32, tags=2, n_img=16, iters=3, passes=3, dev=None):
    """Dual-encoder forward at the PPO step's feature-extraction shapes: ViT-B/16 over batch_items*n_img frames
    [*, 197, 768], RoBERTa-base over batch_items*tags label sequences [*, 196, 768]; random N(0, 0.02) weights.
    -> {"ms", "algorithmic_tflop", "tflops", "mfma_issue_frac"} (MFMA issue fraction = passes * flops / time / 2.5 PF)."""
    dev = dev or torch.device("cuda:0")
    ops.set_gemm_passes(passes)
    total_ms, total_fl, parts = 0.0, 0.0, {}
    for name, cfg, B, L in (("vit-b/16", VIT, batch_items * n_img, 197), ("roberta-base", ROBERTA, batch_items * tags, 196)):
        enc = str2encoder["transformer"](_args(**cfg))
        with torch.no_grad():
            for n, p in enc.named_parameters():
                if "gamma" not in n and "beta" not in n:
                    p.normal_(0, 0.02)
        enc = enc.to(dev).eval()
        emb = torch.randn(B, L, 768, device=dev)
        seg = torch.ones(B, L, dtype=torch.int64, device=dev)
        with torch.no_grad():
            enc(emb, seg)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                enc(emb, seg)
            e.record()
            torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        fl = flops(B, L)
        parts[name] = {"batch": B, "seq": L, "ms": round(ms, 3), "tflops": round(fl / ms / 1e9, 1)}
        total_ms += ms
        total_fl += fl
        del enc, emb
        torch.cuda.empty_cache()
    return {"ms": round(total_ms, 3), "algorithmic_tflop": round(total_fl / 1e12, 2), "tflops": round(total_fl / total_ms / 1e9, 1),
            "mfma_issue_frac": round(passes * total_fl / total_ms / 1e9 / 2500.0, 4), "passes": passes, "parts": parts}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ppo-shapes", action="store_true", help="ViT over batch*16 frames, RoBERTa over batch*2 sequences")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--detail", action="store_true", help="per-signature launch averages")
    ap.add_argument("--train", action="store_true", help="time forward + backward (train mode, dropout 0.1) instead")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops.set_gemm_passes(a.passes)
    torch.manual_seed(0)
    if a.ppo_shapes:
        print(measure_forward(a.batch, iters=a.iters, passes=a.passes, dev=dev))
        return
    total_ms, total_fl = 0.0, 0.0
    for name, cfg, L in (("vit-b/16", VIT, 197), ("roberta-base", ROBERTA, 196)):
        enc = str2encoder["transformer"](_args(**cfg))
        with torch.no_grad():
            for n, p in enc.named_parameters():
                if "gamma" not in n and "beta" not in n:
                    p.normal_(0, 0.02)
        enc = enc.to(dev)
        enc = enc.train() if a.train else enc.eval()
        emb = torch.randn(a.batch, L, 768, device=dev, requires_grad=a.train)
        seg = torch.ones(a.batch, L, dtype=torch.int64, device=dev)
        dout = torch.randn(a.batch, L, 768, device=dev)

        def run():
            if a.train:
                enc(emb, seg).backward(dout)
            else:
                with torch.no_grad():
                    enc(emb, seg)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.iters):
            run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.iters
        fl = flops(a.batch, L) * (3.0 if a.train else 1.0)
        total_ms += ms
        total_fl += fl
        ops.profile_start()
        run()
        prof = ops.profile_stop()
        classes = {}
        for k, v in prof.items():
            c = k.split("_")[0] + ("_" + k.split("_")[1] if k.startswith("gemm") else "")
            classes[c] = classes.get(c, 0.0) + v["ms"]
        print(f"{name:13s} B={a.batch} L={L}: {ms:7.3f} ms/forward  {fl / ms / 1e9:7.1f} TFLOP/s (algorithmic)  "
              f"frac of 2.5 PF bf16 dense x{a.passes} passes: {a.passes * fl / ms / 1e9 / 2500:.3f}", flush=True)
        print("    events:", {k: round(v, 3) for k, v in sorted(classes.items(), key=lambda kv: -kv[1])}, flush=True)
        if a.detail:
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                tf = v["flops"] / (v["ms"] / v["n"]) / 1e9 if v["flops"] else 0.0
                print(f"      {k:40s} n={v['n']:3d} avg {v['ms'] / v['n'] * 1e3:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
    print(f"dual encoder: {total_ms:.3f} ms  {total_fl / total_ms / 1e9:.1f} TFLOP/s algorithmic, "
          f"MFMA issue fraction {a.passes * total_fl / total_ms / 1e9 / 2500:.3f}")


if __name__ == "__main__":
    main()
