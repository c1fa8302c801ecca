"""Dual-encoder forward + backward at the PPO step's shapes (ViT-B/16 over batch*16 frames, RoBERTa-base over batch*tags
sequences, train mode, every parameter gradient) -- the workload of bench.py's `dual_encoder_train`, alone, for
`rocprofv3 --kernel-trace --stats -- python3 tools/dbg/enc_train_prof.py`.  Prints ms per iteration and the HIP-event
ranking of the labelled signatures."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lr2ppo_amd import ops  # noqa: E402
from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch  # noqa: E402
import encoder_bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tags", type=int, default=2)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--detail", action="store_true")
    ap.add_argument("--each", action="store_true", help="print every iteration's duration (first process on a fresh box: how long until the chip is warm)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    fx = FeatureExtractor()
    fx.init_normal()
    fx = fx.to(dev).train()
    fx.bind_grads()
    g = torch.Generator(device=dev).manual_seed(1)
    frames, ids, seg, _ = synthetic_raw_batch(a.batch, a.tags, device=dev, generator=g)
    d_text = torch.randn(a.batch, a.tags, 196, 768, device=dev, generator=g) * 1e-3
    d_img = torch.randn(a.batch, 16, 768, device=dev, generator=g) * 1e-3

    def it():
        t, i, ctx = fx.forward_train(frames, ids, seg)
        fx.backward_train(ctx, d_text, d_img)

    if a.each:
        import time
        t00 = time.perf_counter()
        for k in range(a.iters):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it()
            torch.cuda.synchronize()
            print(f"  iteration {k}: {(time.perf_counter() - t0) * 1e3:7.1f} ms   (t = {time.perf_counter() - t00:5.1f} s)", flush=True)
    it()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        it()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    patch = 2.0 * a.batch * 16 * 196 * 768 * 768
    fl = 3.0 * (encoder_bench.flops(a.batch * 16, 197) + encoder_bench.flops(a.batch * a.tags, 196) + patch) - patch
    print(f"dual encoder train: {ms:.2f} ms  {fl / 1e12:.1f} TFLOP  {fl / ms / 1e9:.1f} TFLOP/s  mfma issue frac {3 * fl / ms / 1e9 / 2500:.3f}")
    if a.detail:
        ops.profile_start()
        it()
        prof = ops.profile_stop()
        tot = sum(v["ms"] for v in prof.values())
        print(f"labelled signatures: {tot:.2f} ms of {ms:.2f}")
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:30]:
            tf = v["flops"] / (v["ms"] / v["n"]) / 1e9 if v["flops"] else 0.0
            print(f"  {k:44s} n={v['n']:3d} tot {v['ms']:8.3f} ms avg {v['ms'] / v['n'] * 1e3:8.1f} us  {tf:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
