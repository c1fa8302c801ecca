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
