"""Explicit forward/backward schedules of the LR2PPO head on the gfx950 kernels.

The reference builds this graph implicitly through nn.Module calls and autograd
(finetune/ppo.py:214-232,265-297; finetune/xit.py).  Here the schedule is explicit: static shapes,
activations in a grow-only workspace (288 GB of HBM: nothing is re-allocated on the steady-state path),
one kernel launch per fused group, gradients written straight into persistent fp32 grad buffers.
The same schedule is used by the autograd wrapper (drop-in nn.Module path) and by train_model/bench.

Everything is fp32 in HBM; GEMMs run as split-bf16 on MFMA (ops.gemm).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops

SEQ_LEN = 196   # hard-coded in the reference (finetune/ppo.py:219-220)
XIT_HEADS = 8   # finetune/xit.py:114
DROP_P = 0.1    # finetune/xit.py:26-28


class Workspace:
    """Named grow-only fp32 device buffers."""

    def __init__(self, device):
        self.device = device
        self._bufs: Dict[str, torch.Tensor] = {}

    def vec(self, name: str, numel: int) -> torch.Tensor:
        b = self._bufs.get(name)
        if b is None or b.numel() < numel:
            b = torch.empty(max(numel, 4), dtype=torch.float32, device=self.device)
            self._bufs[name] = b
        return b[:numel]

    def mat(self, name: str, rows: int, cols: int) -> torch.Tensor:
        return self.vec(name, rows * cols).view(rows, cols)

    def bytes(self) -> int:
        return sum(b.numel() * 4 for b in self._bufs.values())

    def release(self):
        self._bufs.clear()


class DropCfg:
    """Train-time dropout of one XiT block: three sites (attention out, FFN hidden, FFN out)."""

    def __init__(self, p: float, seed: int, site_base: int):
        self.p, self.seed, self.site_base = p, seed, site_base

    def site(self, i: int) -> Optional[ops.Drop]:
        return ops.Drop(self.p, self.seed, self.site_base + i) if self.p > 0 else None


def _splitk_ws(ws: Workspace, M, N, K, trans_a=False):
    bm, sp = ops.choose_tiling(M, N, K, trans_a)
    return (ws.vec("splitk", sp * M * N), sp, bm) if sp > 1 else (None, 1, bm)


def linear_fwd(ws, x, w, b, out, M, N, K, **kw):
    """out[M,N] = x[M,K] @ w[N,K]^T + b (+ fused epilogue)."""
    skw, sp, bm = _splitk_ws(ws, M, N, K)
    return ops.gemm(x, w, out, M, N, K, bias=b, splitk_ws=skw, splits=sp, block_m=bm, **kw)


def linear_dgrad(ws, dy, w, out, M, N_in, N_out, **kw):
    """out[M,N_in] = dy[M,N_out] @ w[N_out,N_in]."""
    skw, sp, bm = _splitk_ws(ws, M, N_in, N_out)
    return ops.gemm(dy, w, out, M, N_in, N_out, trans_b=True, splitk_ws=skw, splits=sp, block_m=bm, **kw)


def linear_wgrad(ws, dy, x, dw, db, M, N_in, N_out):
    """dw[N_out,N_in] = dy[M,N_out]^T @ x[M,N_in];  db[N_out] = colsum(dy)."""
    skw, sp, bm = _splitk_ws(ws, N_out, N_in, M, trans_a=True)
    ops.gemm(dy, x, dw, N_out, N_in, M, trans_a=True, trans_b=True, lda=N_out, ldb=N_in, splitk_ws=skw, splits=sp,
             block_m=bm)
    if db is not None:
        nb = min(128, M)
        ops.colsum(dy, db, ws.vec("colsum_partials", nb * N_out), rows=M, cols=N_out, nblocks=nb)


def _ln_bwd(ws, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, **kw):
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, ws.vec("ln_partials", 256 * 2 * D), dgamma, dbeta, rows=rows, D=D, **kw)


# ---------------------------------------------------------------------------------------------
# XiT block (finetune/xit.py:23-42,71-74)
# ---------------------------------------------------------------------------------------------
class XitKeys:
    """state_dict key names of one XiT (the nn.Sequential nesting of finetune/xit.py:9-42)."""

    def __init__(self, prefix: str):
        a, f = f"{prefix}.0.0.0.fn", f"{prefix}.0.0.1.fn"
        self.ln_x_w, self.ln_x_b = f"{a}.0.ln_x.weight", f"{a}.0.ln_x.bias"
        self.ln_y_w, self.ln_y_b = f"{a}.0.ln_y.weight", f"{a}.0.ln_y.bias"
        self.k_w, self.k_b = f"{a}.1.keys.weight", f"{a}.1.keys.bias"
        self.q_w, self.q_b = f"{a}.1.queries.weight", f"{a}.1.queries.bias"
        self.v_w, self.v_b = f"{a}.1.values.weight", f"{a}.1.values.bias"
        self.p_w, self.p_b = f"{a}.1.projection.weight", f"{a}.1.projection.bias"
        self.ln2_w, self.ln2_b = f"{f}.0.weight", f"{f}.0.bias"
        self.f1_w, self.f1_b = f"{f}.1.0.weight", f"{f}.1.0.bias"
        self.f2_w, self.f2_b = f"{f}.1.3.weight", f"{f}.1.3.bias"
        self.lnf_w, self.lnf_b = f"{prefix}.1.0.weight", f"{prefix}.1.0.bias"


def xit_forward(ws: Workspace, tag: str, P: Dict[str, torch.Tensor], keys: XitKeys, x: torch.Tensor, y: torch.Tensor,
                batch: int, Lq: int, Lk: int, E: int, out: torch.Tensor, *, save: bool, drop: Optional[DropCfg] = None,
                out_group: int = 0, out_gstride: int = 0, heads: int = XIT_HEADS):
    """x: [batch*Lq, E] residual stream, y: [batch*Lk, E]; writes LN_final(block(x, y)) to `out`
    (row r at out + (r//out_group)*out_gstride + (r%out_group)*E when out_group>0)."""
    Mq, Mk, F = batch * Lq, batch * Lk, 4 * E
    hd = E // heads
    t = tag
    d0 = drop.site(0) if drop else None
    d1 = drop.site(1) if drop else None
    d2 = drop.site(2) if drop else None
    xn, yn = ws.mat(t + "xn", Mq, E), ws.mat(t + "yn", Mk, E)
    st = {n: ws.vec(t + n, r) for n, r in (("mx", Mq), ("rx", Mq), ("my", Mk), ("ry", Mk), ("m1", Mq), ("r1", Mq),
                                           ("mf", Mq), ("rf", Mq))}
    ops.layernorm_fwd(x, P[keys.ln_x_w], P[keys.ln_x_b], xn, st["mx"], st["rx"], rows=Mq, D=E)
    ops.layernorm_fwd(y, P[keys.ln_y_w], P[keys.ln_y_b], yn, st["my"], st["ry"], rows=Mk, D=E)
    q, k, v = ws.mat(t + "q", Mq, E), ws.mat(t + "k", Mk, E), ws.mat(t + "v", Mk, E)
    linear_fwd(ws, xn, P[keys.q_w], P[keys.q_b], q, Mq, E, E)
    linear_fwd(ws, yn, P[keys.k_w], P[keys.k_b], k, Mk, E, E)
    linear_fwd(ws, yn, P[keys.v_w], P[keys.v_b], v, Mk, E, E)
    o = ws.mat(t + "o", Mq, E)
    ops.xattn_fwd(q, k, v, o, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=1.0 / math.sqrt(E))
    x1 = ws.mat(t + "x1", Mq, E)
    linear_fwd(ws, o, P[keys.p_w], P[keys.p_b], x1, Mq, E, E, drop=d0, resid=x)
    x1n = ws.mat(t + "x1n", Mq, E)
    ops.layernorm_fwd(x1, P[keys.ln2_w], P[keys.ln2_b], x1n, st["m1"], st["r1"], rows=Mq, D=E)
    hf = ws.mat(t + "hf", Mq, F)
    zf = ws.mat(t + "zf", Mq, F) if save else None
    linear_fwd(ws, x1n, P[keys.f1_w], P[keys.f1_b], hf, Mq, F, E, act=1, out_z=zf, drop=d1)
    x2 = ws.mat(t + "x2", Mq, E)
    linear_fwd(ws, hf, P[keys.f2_w], P[keys.f2_b], x2, Mq, E, F, drop=d2, resid=x1)
    ops.layernorm_fwd(x2, P[keys.lnf_w], P[keys.lnf_b], out, st["mf"], st["rf"], rows=Mq, D=E, group=out_group,
                      group_stride=out_gstride)
    return out


def xit_backward(ws: Workspace, tag: str, P, G, keys: XitKeys, x, y, d_out, batch, Lq, Lk, E, dx_out, dy_out, *,
                 drop: Optional[DropCfg] = None, out_group=0, out_gstride=0, dy_extra=None, heads: int = XIT_HEADS,
                 same_xy: bool = False):
    """Backward of xit_forward.  d_out has the (out_group, out_gstride) row mapping of the forward output.
    dx_out <- dL/dx, dy_out <- dL/dy (+ dy_extra).  With same_xy (x is y, the `xitt` self-attention of
    finetune/ppo.py:290) only dx_out is produced and holds the sum.  Parameter grads go to G[name]."""
    Mq, Mk, F = batch * Lq, batch * Lk, 4 * E
    hd = E // heads
    t = tag
    d0 = drop.site(0) if drop else None
    d1 = drop.site(1) if drop else None
    d2 = drop.site(2) if drop else None
    g = lambda n: ws.mat(t + n, *_shape(n, Mq, Mk, E, F))  # noqa: E731
    st = lambda n, r: ws.vec(t + n, r)  # noqa: E731
    xn, yn, q, k, v, o = g("xn"), g("yn"), g("q"), g("k"), g("v"), g("o")
    x1, x1n, hf, zf, x2 = g("x1"), g("x1n"), g("hf"), g("zf"), g("x2")
    # final LN
    dx2 = ws.mat(t + "dx2", Mq, E)
    dx2m = ws.mat(t + "dxm", Mq, E) if d2 else None
    _ln_bwd(ws, d_out, x2, P[keys.lnf_w], st("mf", Mq), st("rf", Mq), dx2, G[keys.lnf_w], G[keys.lnf_b], Mq, E,
            group=out_group, group_stride=out_gstride, dx_masked=dx2m, drop=d2)
    dF2 = dx2m if d2 else dx2
    # FFN
    linear_wgrad(ws, dF2, hf, G[keys.f2_w], G[keys.f2_b], Mq, F, E)
    dzf = ws.mat(t + "dzf", Mq, F)
    linear_dgrad(ws, dF2, P[keys.f2_w], dzf, Mq, F, E, act=2, aux_z=zf, drop=d1)
    linear_wgrad(ws, dzf, x1n, G[keys.f1_w], G[keys.f1_b], Mq, E, F)
    dx1n = ws.mat(t + "dtmp", Mq, E)
    linear_dgrad(ws, dzf, P[keys.f1_w], dx1n, Mq, E, F)
    dx1 = ws.mat(t + "dx1", Mq, E)
    dx1m = ws.mat(t + "dxm", Mq, E) if d0 else None
    _ln_bwd(ws, dx1n, x1, P[keys.ln2_w], st("m1", Mq), st("r1", Mq), dx1, G[keys.ln2_w], G[keys.ln2_b], Mq, E,
            resid_grad=dx2, dx_masked=dx1m, drop=d0)
    dA = dx1m if d0 else dx1
    # attention
    linear_wgrad(ws, dA, o, G[keys.p_w], G[keys.p_b], Mq, E, E)
    do = ws.mat(t + "dtmp", Mq, E)
    linear_dgrad(ws, dA, P[keys.p_w], do, Mq, E, E)
    dq, dk, dv = ws.mat(t + "dq", Mq, E), ws.mat(t + "dk", Mk, E), ws.mat(t + "dv", Mk, E)
    ops.xattn_bwd(q, k, v, do, dq, dk, dv, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd,
                  post_scale=1.0 / math.sqrt(E))
    linear_wgrad(ws, dq, xn, G[keys.q_w], G[keys.q_b], Mq, E, E)
    linear_wgrad(ws, dk, yn, G[keys.k_w], G[keys.k_b], Mk, E, E)
    linear_wgrad(ws, dv, yn, G[keys.v_w], G[keys.v_b], Mk, E, E)
    dxn = ws.mat(t + "dtmp", Mq, E)
    linear_dgrad(ws, dq, P[keys.q_w], dxn, Mq, E, E)
    dyn = ws.mat(t + "dyn", Mk, E)
    linear_dgrad(ws, dk, P[keys.k_w], dyn, Mk, E, E)
    linear_dgrad(ws, dv, P[keys.v_w], dyn, Mk, E, E, accumulate=True)
    _ln_bwd(ws, dxn, x, P[keys.ln_x_w], st("mx", Mq), st("rx", Mq), dx_out, G[keys.ln_x_w], G[keys.ln_x_b], Mq, E,
            resid_grad=dx1)
    if same_xy:
        dsum = ws.mat(t + "dsum", Mq, E)
        _ln_bwd(ws, dyn, y, P[keys.ln_y_w], st("my", Mk), st("ry", Mk), dsum, G[keys.ln_y_w], G[keys.ln_y_b], Mk, E,
                resid_grad=dx_out)
        dx_out.copy_(dsum)
    else:
        _ln_bwd(ws, dyn, y, P[keys.ln_y_w], st("my", Mk), st("ry", Mk), dy_out, G[keys.ln_y_w], G[keys.ln_y_b], Mk, E,
                resid_grad=dy_extra)


def _shape(n, Mq, Mk, E, F):
    return {"xn": (Mq, E), "yn": (Mk, E), "q": (Mq, E), "k": (Mk, E), "v": (Mk, E), "o": (Mq, E), "x1": (Mq, E),
            "x1n": (Mq, E), "hf": (Mq, F), "zf": (Mq, F), "x2": (Mq, E)}[n]


# ---------------------------------------------------------------------------------------------
# Shared trunk of Actor / Critic / Reward  (finetune/ppo.py:215-227 == :273-285 == :326-338)
# ---------------------------------------------------------------------------------------------
XIT = XitKeys("xit")
XITT = XitKeys("xitt")


def _img_shared(img_emb: torch.Tensor) -> bool:
    """True when the image tokens are shared by all tags of an item: [bs, n_img, E] or a stride-0 expand of it
    (the reference materialises the repeat at finetune/ppo.py:831; identical rows give identical results)."""
    return img_emb.dim() == 3 or (img_emb.dim() == 4 and img_emb.stride(1) == 0)


def trunk_forward(ws: Workspace, P, text: torch.Tensor, img: torch.Tensor, bs: int, tags: int, n_img: int, E: int, *,
                  save: bool, drop: Optional[DropCfg] = None, img_shared: bool = False) -> torch.Tensor:
    """text: [bs*tags*196, E] fp32; img: [bs*tags*n_img, E] (or [bs*n_img, E] when img_shared).
    Returns g2 [bs*tags, E] = out_layer(concat(xit(text_proj, img_proj), img_proj))."""
    N = bs * tags
    Mt, F = N * SEQ_LEN, 4 * E
    Mi_src = (bs if img_shared else N) * n_img
    Mi = N * n_img
    h1 = ws.mat("h1", Mt, F)
    z1 = ws.mat("z1", Mt, F) if save else None
    linear_fwd(ws, text, P["text_proj.fc1.weight"], P["text_proj.fc1.bias"], h1, Mt, F, E, act=1, out_z=z1)
    tf = ws.mat("tf", Mt, E)
    linear_fwd(ws, h1, P["text_proj.fc2.weight"], P["text_proj.fc2.bias"], tf, Mt, E, F)
    hi = ws.mat("hi", Mi_src, F)
    zi = ws.mat("zi", Mi_src, F) if save else None
    linear_fwd(ws, img, P["img_proj.fc1.weight"], P["img_proj.fc1.bias"], hi, Mi_src, F, E, act=1, out_z=zi)
    imf_src = ws.mat("imf_src", Mi_src, E)
    linear_fwd(ws, hi, P["img_proj.fc2.weight"], P["img_proj.fc2.bias"], imf_src, Mi_src, E, F)
    if img_shared and tags > 1:
        imf = ws.mat("imf", Mi, E)       # replicate the projected image tokens over tags (cheap: [N*n_img, E])
        ops.gather_rows(imf_src, None, imf.view(bs, tags, n_img * E), B=bs, t_in=1, t_out=tags, row_elems=n_img * E,
                        src_bstride=n_img * E, src_tstride=0)
    else:
        imf = imf_src
    Wflat = (SEQ_LEN + n_img) * E
    flat = ws.mat("flat", N, Wflat)
    ops.copy_rows(imf, flat, rows=Mi, D=E, group=n_img, dst_gstride=Wflat, dst_off=SEQ_LEN * E)
    xit_forward(ws, "xit.", P, XIT, tf, imf, N, SEQ_LEN, n_img, E, flat, save=save, drop=drop, out_group=SEQ_LEN,
                out_gstride=Wflat)
    g1 = ws.mat("g1", N, F)
    zo = ws.mat("zo", N, F) if save else None
    linear_fwd(ws, flat, P["out_layer.fc1.weight"], P["out_layer.fc1.bias"], g1, N, F, Wflat, act=1, out_z=zo)
    g2 = ws.mat("g2", N, E)
    linear_fwd(ws, g1, P["out_layer.fc2.weight"], P["out_layer.fc2.bias"], g2, N, E, F)
    return g2


def trunk_backward(ws: Workspace, P, G, text, img, dg2, bs, tags, n_img, E, *, drop: Optional[DropCfg] = None,
                   img_shared: bool = False):
    """Backward of trunk_forward(save=True); fills G[...] for every trunk parameter (inputs get no gradient:
    text/img embeddings are data, finetune/ppo.py:827-835)."""
    N = bs * tags
    Mt, F = N * SEQ_LEN, 4 * E
    Mi_src = (bs if img_shared else N) * n_img
    Mi = N * n_img
    Wflat = (SEQ_LEN + n_img) * E
    h1, z1, tf = ws.mat("h1", Mt, F), ws.mat("z1", Mt, F), ws.mat("tf", Mt, E)
    hi, zi = ws.mat("hi", Mi_src, F), ws.mat("zi", Mi_src, F)
    imf = ws.mat("imf", Mi, E) if (img_shared and tags > 1) else ws.mat("imf_src", Mi_src, E)
    flat, g1, zo = ws.mat("flat", N, Wflat), ws.mat("g1", N, F), ws.mat("zo", N, F)
    # out_layer
    linear_wgrad(ws, dg2, g1, G["out_layer.fc2.weight"], G["out_layer.fc2.bias"], N, F, E)
    dzo = ws.mat("dzo", N, F)
    linear_dgrad(ws, dg2, P["out_layer.fc2.weight"], dzo, N, F, E, act=2, aux_z=zo)
    linear_wgrad(ws, dzo, flat, G["out_layer.fc1.weight"], G["out_layer.fc1.bias"], N, Wflat, F)
    dflat = ws.mat("dflat", N, Wflat)
    linear_dgrad(ws, dzo, P["out_layer.fc1.weight"], dflat, N, Wflat, F)
    # image part of the concat -> dense [Mi, E] gradient
    dimf_cat = ws.mat("dimf_cat", Mi, E)
    ops.gather_rows(dflat[:, SEQ_LEN * E:], None, dimf_cat.view(N, 1, n_img * E), B=N, t_in=1, t_out=1,
                    row_elems=n_img * E, src_bstride=Wflat, src_tstride=0)
    dtf, dimf = ws.mat("dtf", Mt, E), ws.mat("dimf", Mi, E)
    xit_backward(ws, "xit.", P, G, XIT, tf, imf, dflat, N, SEQ_LEN, n_img, E, dtf, dimf, drop=drop, out_group=SEQ_LEN,
                 out_gstride=Wflat, dy_extra=dimf_cat)
    # text_proj
    linear_wgrad(ws, dtf, h1, G["text_proj.fc2.weight"], G["text_proj.fc2.bias"], Mt, F, E)
    dz1 = ws.mat("dz1", Mt, F)
    linear_dgrad(ws, dtf, P["text_proj.fc2.weight"], dz1, Mt, F, E, act=2, aux_z=z1)
    linear_wgrad(ws, dz1, text, G["text_proj.fc1.weight"], G["text_proj.fc1.bias"], Mt, E, F)
    # img_proj
    if img_shared and tags > 1:
        dimf_src = ws.mat("dimf_src", Mi_src, E)   # sum the per-tag gradients of the shared image tokens
        ops.gather_rows_bwd(dimf.view(bs, tags, n_img * E), torch.zeros(bs, tags, dtype=torch.int64, device=ws.device),
                            dimf_src.view(bs, 1, n_img * E), B=bs, t_in=1, t_out=tags, row_elems=n_img * E)
    else:
        dimf_src = dimf
    linear_wgrad(ws, dimf_src, hi, G["img_proj.fc2.weight"], G["img_proj.fc2.bias"], Mi_src, F, E)
    dzi = ws.mat("dzi", Mi_src, F)
    linear_dgrad(ws, dimf_src, P["img_proj.fc2.weight"], dzi, Mi_src, F, E, act=2, aux_z=zi)
    linear_wgrad(ws, dzi, img, G["img_proj.fc1.weight"], G["img_proj.fc1.bias"], Mi_src, E, F)
