"""CPU emulation of what operand precision of the matrix products does to the path's outputs (no GPU): the oracle chain
uint8 frames + token ids -> ViT-B/16 + RoBERTa-base (12 + 12 layers) -> Actor logits / Critic value, with every `@` of the oracle
rounding its operands first (products and sums stay fp32 -- what an MFMA with fp32 accumulate does):
    fp32 (reference) | bf16 x 1 | f16 x 1 | f16, left operand exact (= 2 passes: a_hi b_hi + a_lo b_hi)
    python tools/dbg/precision_emulation.py [--items 2] [--layers 12]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from oracle import lr2ppo_oracle as O  # noqa: E402
from oracle.cpu_threads import fit_torch_threads  # noqa: E402

fit_torch_threads()
ap = argparse.ArgumentParser()
ap.add_argument("--items", type=int, default=2)
ap.add_argument("--layers", type=int, default=12)
ap.add_argument("--frames", type=int, default=16)
a = ap.parse_args()

_mm = torch.Tensor.__matmul__
MODE = {"m": "fp32"}


def _round(x, dt):
    return x.to(dt).float()


def _patched(x, y):
    m = MODE["m"]
    if m == "fp32" or not x.is_floating_point():
        return _mm(x, y)
    if m == "bf16":
        return _mm(_round(x, torch.bfloat16), _round(y, torch.bfloat16))
    if m == "f16":
        return _mm(_round(x, torch.float16), _round(y, torch.float16))
    if m == "f16_left_exact":
        return _mm(x, _round(y, torch.float16))
    raise ValueError(m)


torch.Tensor.__matmul__ = _patched

g = torch.Generator().manual_seed(7)
B, T, L, F = a.items, 2, 196, a.frames
frames = torch.randint(0, 256, (B, F, 3, 224, 224), generator=g, dtype=torch.uint8)
ids = torch.randint(5, 50265, (B, T, L), generator=g)
seg = (torch.arange(L).view(1, 1, L) < torch.randint(4, L + 1, (B, T, 1), generator=g)).long()
pv = {**{"embedding." + k: v for k, v in O.seeded_params(O.vit_embedding_spec(768, 3, 16, 197), seed=11).items()},
      **{"encoder." + k: v for k, v in O.seeded_params(O.encoder_param_spec(a.layers, 768, 3072, True), seed=12).items()}}
pt = {**{"embedding." + k: v for k, v in O.seeded_params(O.text_embedding_spec(768, 50265, 514), seed=13).items()},
      **{"encoder." + k: v for k, v in O.seeded_params(O.encoder_param_spec(a.layers, 768, 3072, False), seed=14).items()}}
PA = O.seeded_params(O.head_param_spec("actor", max_imgs=F), seed=15)
PC = O.seeded_params(O.head_param_spec("critic", max_imgs=F), seed=16)
tg = torch.randint(0, 3, (B, T), generator=g)
index = torch.tensor([[0, 1]] * B)

res = {}
with torch.no_grad():
    for mode in ("fp32", "bf16", "f16", "f16_left_exact"):
        MODE["m"] = mode
        text, img = O.feature_chain(pv, pt, frames, ids, seg, patch=16, vit_layers=a.layers, vit_heads=12, text_layers=a.layers)
        img_t = img.unsqueeze(1).repeat(1, T, 1, 1)
        _, logits = O.actor_forward(PA, text, img_t, tg)
        value = O.critic_forward(PC, text, img_t, index)
        res[mode] = (text, img, logits.flatten(), value.flatten())
        if mode != "fp32":
            r = res["fp32"]
            rel = lambda x, y: float((x - y).norm() / y.norm())      # noqa: E731
            print(f"{mode:>16}: features rel L2 text {rel(text, r[0]):.2e} img {rel(img, r[1]):.2e} | max |d logit| "
                  f"{float((res[mode][2] - r[2]).abs().max()):.2e} (|logit| max {float(r[2].abs().max()):.3f}) | max |d value| "
                  f"{float((res[mode][3] - r[3]).abs().max()):.2e}", flush=True)
    # heads alone on the fp32 features (the head-only metric's situation)
    text, img = res["fp32"][0], res["fp32"][1]
    img_t = img.unsqueeze(1).repeat(1, T, 1, 1)
    for mode in ("bf16", "f16", "f16_left_exact"):
        MODE["m"] = mode
        _, logits = O.actor_forward(PA, text, img_t, tg)
        print(f"{mode:>16}: heads alone on fp32 features: max |d logit| {float((logits.flatten() - res['fp32'][2]).abs().max()):.2e}", flush=True)
