"""Explicit forward/backward schedules of the LR2PPO head on the gfx950 kernels.

The reference builds this graph implicitly through nn.Module calls and autograd
(finetune/ppo.py:214-232,265-297; finetune/xit.py).  Here the schedule is explicit: static shapes,
activations in a grow-only workspace (288 GB of HBM: nothing is re-allocated on the steady-state path),
one kernel launch per fused group, gradients written straight into persistent fp32 grad buffers.
The same schedule is used by the autograd wrapper (drop-in nn.Module path) and by train_model/bench.

Numerics: everything the reference keeps in fp32 is fp32 here (parameters, optimizer state, residual stream,
pre-activations, attention inputs, gradients).  Tensors whose ONLY consumers are GEMMs are stored as bf16 hi/lo
"planes" (ops.Planes: x = hi + lo, same bytes as fp32) by the kernel that produces them -- LayerNorm outputs, GELU
outputs, attention outputs, masked gradients, and a per-forward split of the (small) token-GEMM weights -- so the
split-bf16 GEMMs stream their operands by LDS-DMA with no conversion work.  The 2 GB out_layer.fc1 weight is read as
fp32 and split inside the GEMM (it is streamed exactly once per pass; a split copy would double optimizer traffic).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import _native, ops
from .ops import Planes

SEQ_LEN = 196   # hard-coded in the reference (finetune/ppo.py:219-220)
XIT_HEADS = 8   # finetune/xit.py:114
DROP_P = 0.1    # finetune/xit.py:26-28


class Workspace:
    """Named grow-only device buffers (fp32 matrices and bf16-planes matrices)."""

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

    def planes(self, name: str, rows: int, cols: int) -> Planes:
        """[2][rows][cols] bf16 (hi plane, lo plane) -- the same bytes as mat(name, rows, cols)."""
        key = "pl:" + name
        b = self._bufs.get(key)
        need = 2 * rows * cols
        if b is None or b.numel() < need:
            b = torch.empty(max(need, 8), dtype=torch.int16, device=self.device)
            self._bufs[key] = b
        return Planes(b, rows, cols)

    def bytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self._bufs.values())

    def release(self):
        self._bufs.clear()


class Arena:
    """ONE device allocation carved into fp32 / bf16-planes views: the activations a training forward keeps for its backward
    (freed as a whole when the last view dies).  256-byte aligned pieces; a request beyond the reserved size falls back to
    its own allocation, so a wrong size estimate costs speed, never correctness."""

    def __init__(self, device, nbytes: int):
        self.device = device
        self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        self.off = 0

    def _take(self, nbytes: int) -> torch.Tensor:
        start = (self.off + 255) & ~255
        if start + nbytes > self.buf.numel():
            return torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.off = start + nbytes
        return self.buf[start:start + nbytes]

    def vec(self, numel: int) -> torch.Tensor:
        return self._take(4 * numel).view(torch.float32)

    def mat(self, rows: int, cols: int) -> torch.Tensor:
        return self.vec(rows * cols).view(rows, cols)

    def planes(self, rows: int, cols: int) -> Planes:
        return Planes(self._take(4 * rows * cols).view(torch.int16), rows, cols)


class DropCfg:
    """Train-time dropout of one XiT block: three sites (attention out, FFN hidden, FFN out)."""

    def __init__(self, p: float, seed: int, site_base: int, seed_dev: Optional[torch.Tensor] = None):
        self.p, self.seed, self.site_base, self.seed_dev = p, seed, site_base, seed_dev     # seed_dev: see ops.Drop

    def site(self, i: int) -> Optional[ops.Drop]:
        return ops.Drop(self.p, self.seed, self.site_base + i, self.seed_dev) if self.p > 0 else None

    def at(self, site_base: int) -> "DropCfg":
        """The same mask stream at another site base (the tail block beside the trunk's)."""
        return DropCfg(self.p, self.seed, site_base, self.seed_dev)


class WeightPlanes:
    """bf16 hi/lo planes of a model's GEMM weights, re-split from the fp32 parameters by ONE kernel launch at the
    start of every forward (19 M parameters: ~30 us) -- no cache to invalidate when an optimizer, load_state_dict or
    the user changes the parameters."""

    CHUNK = 1 << 16

    T = "^T"     # key suffix of the transposed copy of a wide-output weight (forward runs NN on it)

    def __init__(self, named: Dict[str, torch.Tensor], names: List[str], transposed: Optional[List[str]] = None):
        import ctypes as C
        self._t = [(named[n], n) for n in (transposed or [])]
        dev = named[names[0]].device
        total = sum(2 * named[n].numel() for n in names)
        self.buf = torch.empty(total, dtype=torch.int16, device=dev)
        self.planes: Dict[str, Planes] = {}
        self._sig = tuple((n, named[n].data_ptr()) for n in names)
        rows, off = [], 0
        for n in names:
            w = named[n]
            if w.dim() != 2 or not w.is_contiguous() or w.numel() % 4:
                raise ValueError(f"weight {n} must be a contiguous 2-D tensor")
            k = w.numel()
            self.planes[n] = Planes(self.buf[off:off + 2 * k], w.shape[0], w.shape[1])
            hi_ptr = self.buf.data_ptr() + 2 * off
            o = 0
            while o < k:
                c = min(self.CHUNK, k - o)
                rows.append((w.data_ptr() + 4 * o, hi_ptr + 2 * o, k, c))
                o += c
            off += 2 * k
        arr = (_native.SplitChunk * len(rows))()
        for i, (src, dst, lo, cnt) in enumerate(rows):
            arr[i].src, arr[i].dst_hi, arr[i].lo_off, arr[i].count = src, dst, lo, cnt
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.n_chunks = len(rows)

    def matches(self, named: Dict[str, torch.Tensor]) -> bool:
        return all(n in named and named[n].data_ptr() == p for n, p in self._sig)

    def refresh(self):
        ops.split_planes_multi(self.table, self.n_chunks)
        for w, n in self._t:
            key = n + self.T
            if key not in self.planes:
                self.planes[key] = Planes.empty(w.shape[1], w.shape[0], w.device)
            ops.split_planes_t(w, self.planes[key])


_INPUT_PLANES: Dict[int, tuple] = {}


def input_planes(owner: torch.Tensor, x2d: torch.Tensor) -> Planes:
    """Planes of an external fp32 input (text / image features).  The rollout feeds the same batch tensor to actor,
    critic and reward: the split is done once per live tensor OBJECT and version (never per address: a freed batch's
    address is reused by the next one) and shared through a small LRU."""
    import weakref
    hit = _INPUT_PLANES.get(id(owner))
    if hit is not None and hit[0]() is owner and hit[1] == owner._version and hit[2].rows == x2d.shape[0]:
        return hit[2]
    pl = Planes.empty(x2d.shape[0], x2d.shape[1], x2d.device)
    ops.split_planes(x2d, pl)
    for k in [k for k, v in _INPUT_PLANES.items() if v[0]() is None]:
        del _INPUT_PLANES[k]
    if len(_INPUT_PLANES) >= 8:
        _INPUT_PLANES.pop(next(iter(_INPUT_PLANES)))
    _INPUT_PLANES[id(owner)] = (weakref.ref(owner), owner._version, pl)
    return pl


def _splitk_ws(ws: Workspace, M, N, K, trans_a=False, trans_b=False):
    bm, sp = ops.choose_tiling(M, N, K, trans_a, trans_b)
    return (ws.vec("splitk", sp * M * N), sp, bm) if sp > 1 else (None, 1, bm)


def linear_fwd(ws, x, w, b, out, M, N, K, **kw):
    """out[M,N] = x[M,K] @ w[N,K]^T + b (+ fused epilogue); x / w fp32 tensors or Planes.  A weight given as transposed
    planes W^T [K, N] (ops.split_planes_t) runs the NN form of the kernel, ~10 % faster than NT on wide outputs."""
    if isinstance(w, Planes) and w.transposed:
        skw, sp, bm = _splitk_ws(ws, M, N, K, trans_b=True)
        return ops.gemm(x, w, out, M, N, K, trans_b=True, ldb=N, bias=b, splitk_ws=skw, splits=sp, block_m=bm, **kw)
    skw, sp, bm = _splitk_ws(ws, M, N, K)
    return ops.gemm(x, w, out, M, N, K, bias=b, splitk_ws=skw, splits=sp, block_m=bm, **kw)


def linear_dgrad(ws, dy, w, out, M, N_in, N_out, *, w_f32: Optional[torch.Tensor] = None, **kw):
    """out[M,N_in] = dy[M,N_out] @ w[N_out,N_in].
    w_f32: the fp32 weight behind the planes `w`.  When the product is large enough for the 256 x 256 kernel (NT only) the
    weight is transposed into a scratch planes matrix (one ~6-us launch for a 3072 x 768 weight) and the product runs as
    dy @ (w^T)^T there: at M = 125 440 (stage 1 at 20 tags, encoder training at scale) 1.3 instead of 2.3 ms for the FFN
    input gradient."""
    if w_f32 is not None and isinstance(dy, Planes) and ops.use_gemm256(M, N_in, N_out):
        wt = ops.split_planes_t(w_f32, ws.planes("dgrad_wT", N_in, N_out))
        return ops.gemm(dy, wt, out, M, N_in, N_out, block_m=256, splits=1, **kw)
    skw, sp, bm = _splitk_ws(ws, M, N_in, N_out, trans_b=True)
    return ops.gemm(dy, w, out, M, N_in, N_out, trans_b=True, splitk_ws=skw, splits=sp, block_m=bm, **kw)


def linear_wgrad(ws, dy, x, dw, db, M, N_in, N_out):
    """dw[N_out,N_in] = dy[M,N_out]^T @ x[M,N_in];  db[N_out] = colsum(dy).  At thousands of token rows the product runs on the TN
    form of the 256 x 256 kernel (ops.choose_tiling) and the bias gradient comes out of the same launch -- the column sums of
    the dy fragments it stages anyway -- instead of a second pass over dy."""
    skw, sp, bm = _splitk_ws(ws, N_out, N_in, M, trans_a=True, trans_b=True)
    fused = db is not None and bm == 256 and isinstance(dy, Planes) and isinstance(x, Planes) and N_out % 4 == 0
    cs_ws = ws.vec("colsum_gemm", max(128, sp * ((N_in + 255) // 256)) * N_out) if fused else None
    ops.gemm(dy, x, dw, N_out, N_in, M, trans_a=True, trans_b=True, lda=N_out, ldb=N_in, splitk_ws=skw, splits=sp,
             block_m=bm, colsum=db if fused else None, colsum_ws=cs_ws)
    if db is not None and not fused:
        nb = min(512 if N_out <= 1024 else 256, M)     # row chunks: enough workgroups to fill 256 CUs at 1024 columns each
        ops.colsum(dy, db, ws.vec("colsum_partials", nb * N_out), rows=M, cols=N_out, nblocks=nb)


def _ln_bwd(ws, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, **kw):
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, ws.vec("ln_partials", ops.LN_BWD_BLOCKS * 2 * D), dgamma, dbeta, rows=rows, D=D, **kw)


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

    def gemm_weights(self) -> List[str]:
        return [self.q_w, self.k_w, self.v_w, self.p_w, self.f1_w, self.f2_w]


def xit_forward(ws: Workspace, tag: str, P, W: Dict[str, Planes], keys: XitKeys, x: torch.Tensor, y: torch.Tensor,
                batch: int, Lq: int, Lk: int, E: int, out, *, save: bool, drop: Optional[DropCfg] = None,
                out_group: int = 0, out_gstride: int = 0, heads: int = XIT_HEADS):
    """x: [batch*Lq, E] fp32 residual stream, y: [batch*Lk, E] fp32; W: weight planes.  Writes LN_final(block(x, y)) to
    `out` (fp32 tensor or Planes; row r at (r//out_group)*out_gstride + (r%out_group)*E when out_group>0)."""
    Mq, Mk, F = batch * Lq, batch * Lk, 4 * E
    hd = E // heads
    t = tag
    d0 = drop.site(0) if drop else None
    d1 = drop.site(1) if drop else None
    d2 = drop.site(2) if drop else None
    xn, yn = ws.planes(t + "xn", Mq, E), ws.planes(t + "yn", Mk, E)
    st = {n: ws.vec(t + n, r) for n, r in (("mx", Mq), ("rx", Mq), ("my", Mk), ("ry", Mk), ("m1", Mq), ("r1", Mq),
                                           ("mf", Mq), ("rf", Mq))}
    ops.layernorm_fwd(x, P[keys.ln_x_w], P[keys.ln_x_b], None, st["mx"], st["rx"], rows=Mq, D=E, out_planes=xn)
    ops.layernorm_fwd(y, P[keys.ln_y_w], P[keys.ln_y_b], None, st["my"], st["ry"], rows=Mk, D=E, out_planes=yn)
    q, k, v = ws.mat(t + "q", Mq, E), ws.mat(t + "k", Mk, E), ws.mat(t + "v", Mk, E)
    linear_fwd(ws, xn, W[keys.q_w], P[keys.q_b], q, Mq, E, E)
    linear_fwd(ws, yn, W[keys.k_w], P[keys.k_b], k, Mk, E, E)
    linear_fwd(ws, yn, W[keys.v_w], P[keys.v_b], v, Mk, E, E)
    o = ws.planes(t + "o", Mq, E)
    ops.xattn_fwd(q, k, v, o, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd, post_scale=1.0 / math.sqrt(E))
    x1 = ws.mat(t + "x1", Mq, E)
    linear_fwd(ws, o, W[keys.p_w], P[keys.p_b], x1, Mq, E, E, drop=d0, resid=x)
    x1n = ws.planes(t + "x1n", Mq, E)
    ops.layernorm_fwd(x1, P[keys.ln2_w], P[keys.ln2_b], None, st["m1"], st["r1"], rows=Mq, D=E, out_planes=x1n)
    hf = ws.planes(t + "hf", Mq, F)
    zf = ws.mat(t + "zf", Mq, F) if save else None
    linear_fwd(ws, x1n, fwd_weight(W, keys.f1_w, Mq), P[keys.f1_b], None, Mq, F, E, act=1, out_z=zf, drop=d1, out_planes=hf)
    x2 = ws.mat(t + "x2", Mq, E)
    linear_fwd(ws, hf, W[keys.f2_w], P[keys.f2_b], x2, Mq, E, F, drop=d2, resid=x1)
    out_pl = out if isinstance(out, Planes) else None
    ops.layernorm_fwd(x2, P[keys.lnf_w], P[keys.lnf_b], None if out_pl else out, st["mf"], st["rf"], rows=Mq, D=E,
                      group=out_group, group_stride=out_gstride, out_planes=out_pl)
    return out


def xit_backward(ws: Workspace, tag: str, P, W, G, keys: XitKeys, x, y, d_out, batch, Lq, Lk, E, dx_out, dy_out, *,
                 drop: Optional[DropCfg] = None, out_group=0, out_gstride=0, dy_extra=None, heads: int = XIT_HEADS,
                 same_xy: bool = False, dx_planes: Optional[Planes] = None, dy_planes: Optional[Planes] = None):
    """Backward of xit_forward.  d_out (fp32) has the (out_group, out_gstride) row mapping of the forward output.
    dx_out <- dL/dx, dy_out <- dL/dy (+ dy_extra), both fp32; dx_planes / dy_planes additionally receive them as planes
    (operands of the caller's next GEMMs).  With same_xy (x is y, the `xitt` self-attention of finetune/ppo.py:290) only
    dx_out is produced and holds the sum.  Parameter grads go to G[name]."""
    Mq, Mk, F = batch * Lq, batch * Lk, 4 * E
    hd = E // heads
    t = tag
    d0 = drop.site(0) if drop else None
    d1 = drop.site(1) if drop else None
    d2 = drop.site(2) if drop else None
    st = lambda n, r: ws.vec(t + n, r)  # noqa: E731
    xn, yn, o = ws.planes(t + "xn", Mq, E), ws.planes(t + "yn", Mk, E), ws.planes(t + "o", Mq, E)
    x1n, hf = ws.planes(t + "x1n", Mq, E), ws.planes(t + "hf", Mq, F)
    q, k, v = ws.mat(t + "q", Mq, E), ws.mat(t + "k", Mk, E), ws.mat(t + "v", Mk, E)
    x1, zf, x2 = ws.mat(t + "x1", Mq, E), ws.mat(t + "zf", Mq, F), ws.mat(t + "x2", Mq, E)
    # final LN
    dx2 = ws.mat(t + "dx2", Mq, E)
    dF2 = ws.planes(t + "dxm", Mq, E)
    _ln_bwd(ws, d_out, x2, P[keys.lnf_w], st("mf", Mq), st("rf", Mq), dx2, G[keys.lnf_w], G[keys.lnf_b], Mq, E,
            group=out_group, group_stride=out_gstride, dx_planes=dF2, drop=d2)
    # FFN
    linear_wgrad(ws, dF2, hf, G[keys.f2_w], G[keys.f2_b], Mq, F, E)
    dzf = ws.planes(t + "dzf", Mq, F)
    linear_dgrad(ws, dF2, W[keys.f2_w], None, Mq, F, E, act=2, aux_z=zf, drop=d1, out_planes=dzf, w_f32=P[keys.f2_w])
    linear_wgrad(ws, dzf, x1n, G[keys.f1_w], G[keys.f1_b], Mq, E, F)
    dx1n = ws.mat(t + "dtmp", Mq, E)
    linear_dgrad(ws, dzf, W[keys.f1_w], dx1n, Mq, E, F, w_f32=P[keys.f1_w])
    dx1 = ws.mat(t + "dx1", Mq, E)
    dA = ws.planes(t + "dxm", Mq, E)
    _ln_bwd(ws, dx1n, x1, P[keys.ln2_w], st("m1", Mq), st("r1", Mq), dx1, G[keys.ln2_w], G[keys.ln2_b], Mq, E,
            resid_grad=dx2, dx_planes=dA, drop=d0)
    # attention
    linear_wgrad(ws, dA, o, G[keys.p_w], G[keys.p_b], Mq, E, E)
    do = ws.mat(t + "dtmp", Mq, E)
    linear_dgrad(ws, dA, W[keys.p_w], do, Mq, E, E, w_f32=P[keys.p_w])
    dq, dk, dv = ws.planes(t + "dq", Mq, E), ws.planes(t + "dk", Mk, E), ws.planes(t + "dv", Mk, E)
    ops.xattn_bwd(q, k, v, do, dq, dk, dv, batch=batch, heads=heads, Lq=Lq, Lk=Lk, head_dim=hd,
                  post_scale=1.0 / math.sqrt(E))
    linear_wgrad(ws, dq, xn, G[keys.q_w], G[keys.q_b], Mq, E, E)
    linear_wgrad(ws, dk, yn, G[keys.k_w], G[keys.k_b], Mk, E, E)
    linear_wgrad(ws, dv, yn, G[keys.v_w], G[keys.v_b], Mk, E, E)
    dxn = ws.mat(t + "dtmp", Mq, E)
    linear_dgrad(ws, dq, W[keys.q_w], dxn, Mq, E, E, w_f32=P[keys.q_w])
    dyn = ws.mat(t + "dyn", Mk, E)
    linear_dgrad(ws, dk, W[keys.k_w], dyn, Mk, E, E)
    linear_dgrad(ws, dv, W[keys.v_w], dyn, Mk, E, E, accumulate=True)
    if same_xy:
        dpart = ws.mat(t + "dpart", Mq, E)
        _ln_bwd(ws, dxn, x, P[keys.ln_x_w], st("mx", Mq), st("rx", Mq), dpart, G[keys.ln_x_w], G[keys.ln_x_b], Mq, E,
                resid_grad=dx1)
        _ln_bwd(ws, dyn, y, P[keys.ln_y_w], st("my", Mk), st("ry", Mk), dx_out, G[keys.ln_y_w], G[keys.ln_y_b], Mk, E,
                resid_grad=dpart, dx_planes=dx_planes)
    else:
        _ln_bwd(ws, dxn, x, P[keys.ln_x_w], st("mx", Mq), st("rx", Mq), dx_out, G[keys.ln_x_w], G[keys.ln_x_b], Mq, E,
                resid_grad=dx1, dx_planes=dx_planes)
        _ln_bwd(ws, dyn, y, P[keys.ln_y_w], st("my", Mk), st("ry", Mk), dy_out, G[keys.ln_y_w], G[keys.ln_y_b], Mk, E,
                resid_grad=dy_extra, dx_planes=dy_planes)


# ---------------------------------------------------------------------------------------------
# Shared trunk of Actor / Critic / Reward  (finetune/ppo.py:215-227 == :273-285 == :326-338)
# ---------------------------------------------------------------------------------------------
XIT = XitKeys("xit")
XITT = XitKeys("xitt")
TRUNK_GEMM_WEIGHTS = ["text_proj.fc1.weight", "text_proj.fc2.weight", "img_proj.fc1.weight", "img_proj.fc2.weight",
                      "out_layer.fc2.weight"] + XIT.gemm_weights()
FC1 = "out_layer.fc1.weight"   # 2 GB: stays fp32, split inside the GEMM
# [3072, 768] weights: a transposed planes copy W^T serves the forward (NN form); dgrad / wgrad keep the original layout
TRUNK_T_WEIGHTS = ["text_proj.fc1.weight", "img_proj.fc1.weight", XIT.f1_w]


def fwd_weight(W, name, M: Optional[int] = None):
    """The planes a forward GEMM should use for weight `name`: its transposed copy (NN form) when the model keeps one --
    unless the product is large enough for the 256 x 256 NT kernel, which takes the weight in its own [out, in] layout."""
    w = W[name]
    if M is not None and ops.use_gemm256(M, w.rows, w.cols):
        return w
    return W.get(name + WeightPlanes.T, w)


def _img_shared(img_emb: torch.Tensor) -> bool:
    """True when the image tokens are shared by all tags of an item: [bs, n_img, E] or a stride-0 expand of it
    (the reference materialises the repeat at finetune/ppo.py:831; identical rows give identical results)."""
    return img_emb.dim() == 3 or (img_emb.dim() == 4 and img_emb.stride(1) == 0)


def trunk_forward(ws: Workspace, P, W, text, img, bs: int, tags: int, n_img: int, E: int, *, save: bool,
                  drop: Optional[DropCfg] = None, img_shared: bool = False) -> torch.Tensor:
    """text: Planes [bs*tags*196, E]; img: Planes [bs*tags*n_img, E] (or [bs*n_img, E] when img_shared).
    Returns g2 [bs*tags, E] (fp32) = out_layer(concat(xit(text_proj, img_proj), img_proj))."""
    N = bs * tags
    Mt, F = N * SEQ_LEN, 4 * E
    Mi_src = (bs if img_shared else N) * n_img
    Mi = N * n_img
    h1 = ws.planes("h1", Mt, F)
    z1 = ws.mat("z1", Mt, F) if save else None
    linear_fwd(ws, text, fwd_weight(W, "text_proj.fc1.weight", Mt), P["text_proj.fc1.bias"], None, Mt, F, E, act=1, out_z=z1,
               out_planes=h1)
    tf = ws.mat("tf", Mt, E)
    linear_fwd(ws, h1, W["text_proj.fc2.weight"], P["text_proj.fc2.bias"], tf, Mt, E, F)
    hi = ws.planes("hi", Mi_src, F)
    zi = ws.mat("zi", Mi_src, F) if save else None
    linear_fwd(ws, img, fwd_weight(W, "img_proj.fc1.weight", Mi_src), P["img_proj.fc1.bias"], None, Mi_src, F, E, act=1, out_z=zi,
               out_planes=hi)
    imf_src = ws.mat("imf_src", Mi_src, E)
    linear_fwd(ws, hi, W["img_proj.fc2.weight"], P["img_proj.fc2.bias"], imf_src, Mi_src, E, F)
    if img_shared and tags > 1:
        imf = ws.mat("imf", Mi, E)       # replicate the projected image tokens over tags (cheap: [N*n_img, E])
        ops.gather_rows(imf_src, None, imf.view(bs, tags, n_img * E), B=bs, t_in=1, t_out=tags, row_elems=n_img * E,
                        src_bstride=n_img * E, src_tstride=0)
    else:
        imf = imf_src
    Wflat = (SEQ_LEN + n_img) * E
    flat = ws.planes("flat", N, Wflat)
    ops.copy_rows(imf, flat, rows=Mi, D=E, group=n_img, dst_gstride=Wflat, dst_off=SEQ_LEN * E)
    xit_forward(ws, "xit.", P, W, XIT, tf, imf, N, SEQ_LEN, n_img, E, flat, save=save, drop=drop, out_group=SEQ_LEN,
                out_gstride=Wflat)
    g1 = ws.planes("g1", N, F)
    zo = ws.mat("zo", N, F) if save else None
    linear_fwd(ws, flat, P[FC1], P["out_layer.fc1.bias"], None, N, F, Wflat, act=1, out_z=zo, out_planes=g1)
    g2 = ws.mat("g2", N, E)
    linear_fwd(ws, g1, W["out_layer.fc2.weight"], P["out_layer.fc2.bias"], g2, N, E, F)
    return g2


def trunk_backward(ws: Workspace, P, W, G, text, img, dg2, bs, tags, n_img, E, *, drop: Optional[DropCfg] = None,
                   img_shared: bool = False, dp=None, fc1_update=None, fc1_early: bool = False, input_grads: bool = False):
    """Backward of trunk_forward(save=True); fills G[...] for every trunk parameter.  dg2: fp32 [bs*tags, E].
    input_grads: also return (d text [bs*tags*196, E], d img [Mi_src, E]) as fresh fp32 tensors -- the gradients the
    reference's autograd hands to whatever produced text_emb / img_emb (encoders being fine-tuned behind the head:
    finetune/ppo.py:214-232 is plain autograd, tencentpretrain/models/model.py:32-41).  Two NN GEMMs on operands the
    backward already holds (dz1 x text_proj.fc1.weight, dzi x img_proj.fc1.weight); the PPO loop feeds pre-extracted
    features (finetune/ppo.py:827-835) and leaves it off.
    fc1_update (ops.AdamArgs): apply the optimizer step of out_layer.fc1.weight inside its weight-gradient GEMM, issued
    after the last reader of the old weight (the input-gradient GEMM); G[out_layer.fc1.weight] is then left untouched.
    fc1_early: issue that fused update right behind the input-gradient GEMM instead of at the very end -- same operands, same
    bits; a model whose backward runs on a second stream uses it so that its 12-GB HBM-bound pass falls beside the other
    model's MFMA-bound token GEMMs instead of beside the other model's own 12-GB pass (single-rank only: with data
    parallelism the update waits for the factor all-gather and stays last)."""
    N = bs * tags
    Mt, F = N * SEQ_LEN, 4 * E
    Mi_src = (bs if img_shared else N) * n_img
    Mi = N * n_img
    Wflat = (SEQ_LEN + n_img) * E
    h1, z1, tf = ws.planes("h1", Mt, F), ws.mat("z1", Mt, F), ws.mat("tf", Mt, E)
    hi, zi = ws.planes("hi", Mi_src, F), ws.mat("zi", Mi_src, F)
    imf = ws.mat("imf", Mi, E) if (img_shared and tags > 1) else ws.mat("imf_src", Mi_src, E)
    flat, g1, zo = ws.planes("flat", N, Wflat), ws.planes("g1", N, F), ws.mat("zo", N, F)
    # out_layer
    dg2p = ws.planes("dg2p", N, E)
    ops.split_planes(dg2, dg2p)
    linear_wgrad(ws, dg2p, g1, G["out_layer.fc2.weight"], G["out_layer.fc2.bias"], N, F, E)
    dzo = ws.planes("dzo", N, F)
    linear_dgrad(ws, dg2p, W["out_layer.fc2.weight"], None, N, F, E, act=2, aux_z=zo, out_planes=dzo)
    if dp is not None and getattr(dp, "active", dp.world > 1):
        # Data parallel: dW_fc1 = sum over ranks of dzo_r^T flat_r is a rank-(N*world) product of two thin factors.
        # All-gather the factors (42 MB per rank) instead of all-reducing the 2 GB product; the gathers run on the
        # communication stream while the rest of backward proceeds, the K = N*world wgrad GEMM is issued last.
        nb = min(128, N)
        ops.colsum(dzo, G["out_layer.fc1.bias"], ws.vec("colsum_partials", nb * F), rows=N, cols=F, nblocks=nb)
        fc1_pending = (dp.gather_planes_start(dzo, ws, "dzo_all"), dp.gather_planes_start(flat, ws, "flat_all"))
    else:
        fc1_pending = None
        if fc1_update is None:
            linear_wgrad(ws, dzo, flat, G[FC1], G["out_layer.fc1.bias"], N, Wflat, F)
        else:
            nb = min(128, N)
            ops.colsum(dzo, G["out_layer.fc1.bias"], ws.vec("colsum_partials", nb * F), rows=N, cols=F, nblocks=nb)
    dflat = ws.mat("dflat", N, Wflat)
    linear_dgrad(ws, dzo, P[FC1], dflat, N, Wflat, F)

    def fused_fc1_update(dzo_all, flat_all, Kall, alpha):
        skw, sp, bm = _splitk_ws(ws, F, Wflat, Kall, trans_a=True, trans_b=True)
        ops.gemm(dzo_all, flat_all, None if fc1_update is not None else G[FC1], F, Wflat, Kall, trans_a=True, trans_b=True,
                 lda=F, ldb=Wflat, splitk_ws=skw, splits=sp, block_m=bm, alpha=alpha, adam=fc1_update)

    early = fc1_early and fc1_update is not None and fc1_pending is None
    if early:
        fused_fc1_update(dzo, flat, N, 1.0)
    # image part of the concat -> dense [Mi, E] gradient
    dimf_cat = ws.mat("dimf_cat", Mi, E)
    ops.gather_rows(dflat[:, SEQ_LEN * E:], None, dimf_cat.view(N, 1, n_img * E), B=N, t_in=1, t_out=1,
                    row_elems=n_img * E, src_bstride=Wflat, src_tstride=0)
    dtf, dimf = ws.mat("dtf", Mt, E), ws.mat("dimf", Mi, E)
    dtf_p = ws.planes("dtf_p", Mt, E)
    shared = img_shared and tags > 1
    dimf_p = None if shared else ws.planes("dimf_p", Mi, E)
    xit_backward(ws, "xit.", P, W, G, XIT, tf, imf, dflat, N, SEQ_LEN, n_img, E, dtf, dimf, drop=drop, out_group=SEQ_LEN,
                 out_gstride=Wflat, dy_extra=dimf_cat, dx_planes=dtf_p, dy_planes=dimf_p)
    # text_proj
    linear_wgrad(ws, dtf_p, h1, G["text_proj.fc2.weight"], G["text_proj.fc2.bias"], Mt, F, E)
    dz1 = ws.planes("dz1", Mt, F)
    linear_dgrad(ws, dtf_p, W["text_proj.fc2.weight"], None, Mt, F, E, act=2, aux_z=z1, out_planes=dz1,
                 w_f32=P["text_proj.fc2.weight"])
    linear_wgrad(ws, dz1, text, G["text_proj.fc1.weight"], G["text_proj.fc1.bias"], Mt, E, F)
    d_text = d_img = None
    if input_grads:
        d_text = torch.empty(Mt, E, dtype=torch.float32, device=ws.device)
        linear_dgrad(ws, dz1, W["text_proj.fc1.weight"], d_text, Mt, E, F, w_f32=P["text_proj.fc1.weight"])
    # img_proj
    if shared:
        dimf_src = ws.mat("dimf_src", Mi_src, E)   # sum the per-tag gradients of the shared image tokens
        ops.gather_rows_bwd(dimf.view(bs, tags, n_img * E), torch.zeros(bs, tags, dtype=torch.int64, device=ws.device),
                            dimf_src.view(bs, 1, n_img * E), B=bs, t_in=1, t_out=tags, row_elems=n_img * E)
        dimf_p = ws.planes("dimf_p", Mi_src, E)
        ops.split_planes(dimf_src, dimf_p)
    linear_wgrad(ws, dimf_p, hi, G["img_proj.fc2.weight"], G["img_proj.fc2.bias"], Mi_src, F, E)
    dzi = ws.planes("dzi", Mi_src, F)
    linear_dgrad(ws, dimf_p, W["img_proj.fc2.weight"], None, Mi_src, F, E, act=2, aux_z=zi, out_planes=dzi)
    linear_wgrad(ws, dzi, img, G["img_proj.fc1.weight"], G["img_proj.fc1.bias"], Mi_src, E, F)
    if input_grads:
        d_img = torch.empty(Mi_src, E, dtype=torch.float32, device=ws.device)
        linear_dgrad(ws, dzi, W["img_proj.fc1.weight"], d_img, Mi_src, E, F, w_f32=P["img_proj.fc1.weight"])
    if (fc1_pending is not None or fc1_update is not None) and not early:
        if fc1_pending is not None:
            dzo_all, flat_all = dp.gather_planes_finish(fc1_pending[0]), dp.gather_planes_finish(fc1_pending[1])
            fused_fc1_update(dzo_all, flat_all, N * dp.world, 1.0 / dp.world)            # already the rank average
        else:
            fused_fc1_update(dzo, flat, N, 1.0)
    return d_text, d_img


# ---------------------------------------------------------------------------------------------
# The `_trad` trunk: the head at sequence length 1 (finetune/pointwise_trad.py:146-157, ppo_trad.py:160-171)
# ---------------------------------------------------------------------------------------------
TRAD_FC1, TRAD_FC2 = "out_layer.fc1.weight", "out_layer.fc2.weight"


def trad_trunk_forward(ws: Workspace, P, W, x0: torch.Tensor, N: int, E: int, *, save: bool, drop: Optional[DropCfg] = None):
    """x0 [N, E]: one pre-projected feature per document, used as both streams of the XiT block, concatenated behind the
    block's output, through out_layer = Mlp(2E, 4E, E) -> g2 [N, E] (workspace buffer "g2")."""
    F = 4 * E
    cat = ws.planes("cat", N, 2 * E)                         # [XiT(x0, x0) | x0]
    ops.copy_rows(x0, cat, rows=N, D=E, group=1, dst_gstride=2 * E, dst_off=E)
    xit_forward(ws, "xit.", P, W, XIT, x0, x0, N, 1, 1, E, cat, save=save, drop=drop, out_group=1, out_gstride=2 * E)
    g1 = ws.planes("g1", N, F)
    zo = ws.mat("zo", N, F) if save else None
    linear_fwd(ws, cat, fwd_weight(W, TRAD_FC1), P["out_layer.fc1.bias"], None, N, F, 2 * E, act=1, out_z=zo, out_planes=g1)
    g2 = ws.mat("g2", N, E)
    linear_fwd(ws, g1, W[TRAD_FC2], P["out_layer.fc2.bias"], g2, N, E, F)
    return g2


def trad_trunk_backward(ws: Workspace, P, W, G, x0: torch.Tensor, dg2: torch.Tensor, N: int, E: int, *,
                        drop: Optional[DropCfg] = None, want_dx: bool = False):
    """Backward of trad_trunk_forward(save=True): fills G[...] for xit.* and out_layer.*.  The feature is data in
    pointwise_trad / ppo_trad (no input gradient); want_dx (pointwise_2data_trad: the feature comes out of a projection MLP)
    returns dL/dx0 [N, E] = the block's two input gradients + the concat's share."""
    F = 4 * E
    cat, g1, zo = ws.planes("cat", N, 2 * E), ws.planes("g1", N, F), ws.mat("zo", N, F)
    dg2p = ops.split_planes(dg2, ws.planes("dg2p", N, E))
    linear_wgrad(ws, dg2p, g1, G[TRAD_FC2], G["out_layer.fc2.bias"], N, F, E)
    dzo = ws.planes("dzo", N, F)
    linear_dgrad(ws, dg2p, W[TRAD_FC2], None, N, F, E, act=2, aux_z=zo, out_planes=dzo)
    linear_wgrad(ws, dzo, cat, G[TRAD_FC1], G["out_layer.fc1.bias"], N, 2 * E, F)
    dcat = ws.mat("dcat", N, 2 * E)
    linear_dgrad(ws, dzo, W[TRAD_FC1], dcat, N, 2 * E, F)
    # only the block's parameters need gradients; d(out) / d(block output) = dcat[:, :E]
    dx0 = ws.mat("dx0", N, E)
    xit_backward(ws, "xit.", P, W, G, XIT, x0, x0, dcat, N, 1, 1, E, dx0, None, drop=drop, out_group=1,
                 out_gstride=2 * E, same_xy=True)
    if want_dx:
        dx0.add_(dcat[:, E:])
        return dx0
    return None


def feature_proj_forward(ws: Workspace, P, prefix: str, x: torch.Tensor, N: int, Kin: int, E: int, *, save: bool):
    """Mlp(Kin, 4E, E) on raw LETOR features (finetune/pointwise_2data_trad.py:135-136,147-150: 46-d MQ2008 / 136-d MSLR rows)
    -> [N, E] fp32 (workspace buffer).  Kin is not a multiple of the GEMM's K tile: x and fc1.weight are zero-padded to
    Kp = ceil(Kin / 64) * 64 columns (padding contributes exact zeros to every product)."""
    F, Kp = 4 * E, -(-Kin // 64) * 64
    t = "fp:" + prefix
    xpad, w1pad = ws.mat(t + "x", N, Kp), ws.mat(t + "w1", F, Kp)
    xpad.zero_(), w1pad.zero_()
    xpad[:, :Kin].copy_(x.reshape(N, Kin))
    w1pad[:, :Kin].copy_(P[prefix + ".fc1.weight"])
    x_p = ops.split_planes(xpad, ws.planes(t + "x", N, Kp))
    w1_p = ops.split_planes(w1pad, ws.planes(t + "w1", F, Kp))
    w2_p = ops.split_planes(P[prefix + ".fc2.weight"], ws.planes(t + "w2", E, F))
    h_p = ws.planes(t + "h", N, F)
    z = ws.mat(t + "z", N, F) if save else None
    linear_fwd(ws, x_p, w1_p, P[prefix + ".fc1.bias"], None, N, F, Kp, act=1, out_z=z, out_planes=h_p)
    out = ws.mat(t + "out", N, E)
    linear_fwd(ws, h_p, w2_p, P[prefix + ".fc2.bias"], out, N, E, F)
    return out


def feature_proj_backward(ws: Workspace, P, G, prefix: str, dout: torch.Tensor, N: int, Kin: int, E: int):
    """Backward of feature_proj_forward(save=True): parameter gradients of prefix.fc1 / fc2 (the raw features are data)."""
    F, Kp = 4 * E, -(-Kin // 64) * 64
    t = "fp:" + prefix
    x_p, w2_p, h_p, z = ws.planes(t + "x", N, Kp), ws.planes(t + "w2", E, F), ws.planes(t + "h", N, F), ws.mat(t + "z", N, F)
    dout_p = ops.split_planes(dout, ws.planes(t + "dout", N, E))
    linear_wgrad(ws, dout_p, h_p, G[prefix + ".fc2.weight"], G[prefix + ".fc2.bias"], N, F, E)
    dz_p = ws.planes(t + "dz", N, F)
    linear_dgrad(ws, dout_p, w2_p, None, N, F, E, act=2, aux_z=z, out_planes=dz_p)
    dw1 = ws.mat(t + "dw1", F, Kp)
    linear_wgrad(ws, dz_p, x_p, dw1, G[prefix + ".fc1.bias"], N, Kp, F)
    G[prefix + ".fc1.weight"].copy_(dw1[:, :Kin])
