"""Randomised shape sweep of the GEMM family (forms x operand kinds x tile choices x fused epilogues) and of the self-attention
forward / backward / first-token kernels, LayerNorm, XiT attention, the PPO loss and six random small TransformerEncoder
configurations against fp64 formulas / the oracle: tools/dbg/fuzz_kernels.py with a fixed seed, as a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_gemm_and_attention_shapes_match_fp64(dev):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "dbg", "fuzz_kernels.py"), "--n", "120", "--seed", "11", "--encoders", "6"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bad 0" in r.stdout


def test_randomised_round4_kernels_match_fp64(dev):
    """tools/dbg/fuzz_round4.py with a fixed seed: MX-FP8 products on both product kernels and both scale-staging forms with every
    output combination, the fp8 mode's bf16 attention for random (batch, heads, L <= 288, masks), row-split products with the epilogues
    that may split (and the launch counters saying the split happened)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "dbg", "fuzz_round4.py"), "--n", "10", "--seed", "21"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "no mismatch" in r.stdout


def test_fixed_seed_sweep_of_the_persistent_attention_kernels():
    """tools/dbg/fuzz_attn_persist.py with a fixed seed: random (batch, heads, L <= 224, masks, dropout) with 1-7 (sequence, head) pairs per
    workgroup -- persistent forward == the one-pair kernel chunk by chunk (bit for bit), streaming backward within 2e-5 of the recomputing
    kernels, every element written; the MX-FP8 mode's persistent attention (L <= 288) == its one-pair kernel, bytes and scales."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "dbg", "fuzz_attn_persist.py"), "--n", "12", "--seed", "5"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "no mismatch" in r.stdout
