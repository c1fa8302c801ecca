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


def test_row_split_is_what_runs_and_the_switch_turns_it_off(dev):
    """The row split changes WHICH kernel computes the tail rows, so the two settings differ in the last bits there and nowhere
    else: rows of the whole rounds are bit-identical with LR2_GEMM_ROWSPLIT=0, the tail rows agree to rounding (child processes:
    the switch is read once per process)."""
    code = r'''
import sys, torch, hashlib
sys.path.insert(0, %r)
from lr2ppo_amd import ops
dev = torch.device("cuda:0")
M, N, K = 8448, 2048, 128
g = torch.Generator().manual_seed(5)
a, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
ap = ops.split_planes(a.to(dev), ops.Planes.empty(M, K, dev)); wp = ops.split_planes(w.to(dev), ops.Planes.empty(N, K, dev))
out = torch.empty(M, N, device=dev)
ops.gemm(ap, wp, out, M, N, K, block_m=256, splits=1)
o = out.cpu()
print(hashlib.sha1(o[:8192].numpy().tobytes()).hexdigest(), hashlib.sha1(o[8192:].numpy().tobytes()).hexdigest(), float(o[8192:].double().abs().sum()))
''' % REPO
    res = {}
    for flag in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, LR2_GEMM_ROWSPLIT=flag))
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = r.stdout.split()
    assert res["1"][0] == res["0"][0]                       # the whole rounds: the same kernel, the same bits
    assert res["1"][1] != res["0"][1]                       # the tail: another kernel (the split really happened)
    assert abs(float(res["1"][2]) - float(res["0"][2])) < 1e-4 * float(res["0"][2])
