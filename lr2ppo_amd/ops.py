"""Thin torch-tensor wrappers over the C ABI (include/lr2ppo_hip.h).

PyTorch is used for device memory and streams only: every wrapper passes raw device pointers and the
current HIP stream to the native library.  All tensors are fp32, contiguous, on a HIP device.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Optional

import torch

from . import _native as _nat

# GEMM precision: 3 = split-bf16 (fp32-grade, parity mode, default); 1 = single bf16 pass.
_PASSES = 3


def set_gemm_passes(p: int):
    global _PASSES
    if p not in (1, 3):
        raise ValueError("passes must be 1 or 3")
    _PASSES = p


def get_gemm_passes() -> int:
    return _PASSES


# ---- optional per-launch timing (bench.py): HIP events on the launch stream around selected entry points ----
_PROF = None
_PROF_ONLY = None

# Kernels of this package write parameters through raw pointers (the optimizer), which torch's tensor version counters do not see:
# every parameter such a kernel updates gets its own write counter bumped (mark_params_written), and caches of derived data (weight
# planes) compare it next to p._version.  Per parameter, not global: the PPO heads' optimizer steps must not invalidate the frozen
# encoders' weight planes (round 2's single global epoch re-split both stacks' weights on every step of the composed loop).
def mark_params_written(params):
    for p in params:
        p._lr2_writes = getattr(p, "_lr2_writes", 0) + 1


def param_write_count(p) -> int:
    return getattr(p, "_lr2_writes", 0)


def profile_start(only=None):
    """Start collecting (start, end) HIP events per kernel signature; see profile_stop().  `only`: a collection of
    signatures to restrict the collection to (two event records per launch cost ~25 us of host time; with ~120 timed
    launches per PPO step that alone can make the step host-bound)."""
    global _PROF, _PROF_ONLY
    _PROF = {}
    _PROF_ONLY = None if only is None else frozenset(only)


def profile_stop():
    """-> {key: {"ms": total, "n": launches, "flops": per-launch, "bytes": per-launch}} (synchronises)."""
    global _PROF
    prof, _PROF = _PROF, None
    torch.cuda.synchronize()
    out = {}
    for key, rec in (prof or {}).items():
        ms = sum(s.elapsed_time(e) for s, e in rec["ev"])
        out[key] = {"ms": ms, "n": len(rec["ev"]), "flops": rec["flops"], "bytes": rec["bytes"]}
    return out


class _Timed:
    __slots__ = ("key", "flops", "bytes", "s")

    def __init__(self, key, flops=0, nbytes=0):
        self.key, self.flops, self.bytes = key, flops, nbytes

    def __enter__(self):
        self.s = None
        if _PROF is not None and (_PROF_ONLY is None or self.key in _PROF_ONLY):
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if self.s is not None and _PROF is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            rec = _PROF.setdefault(self.key, {"ev": [], "flops": self.flops, "bytes": self.bytes})
            rec["ev"].append((self.s, e))
        return False


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    # torch.cuda.current_stream() builds a Stream object through several Python layers (~8 us, 300 launches per PPO
    # step); the raw handle of the same stream comes straight from the C++ side in ~0.3 us
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if t.dtype != torch.float32 or not t.is_cuda:
            raise TypeError(f"expected a float32 HIP tensor, got {t.dtype} on {t.device}")


class Drop:
    """Dropout spec for one site: p, seed, site id (see csrc/common.h::dropout_keep).  seed_dev: int64[1] device tensor whose
    value the kernels ADD to `seed` at run time (HIP-graph capture of a training step, StepScalars)."""
    __slots__ = ("p", "seed", "site", "seed_dev")

    def __init__(self, p: float, seed: int, site: int, seed_dev: Optional[torch.Tensor] = None):
        self.p, self.seed, self.site, self.seed_dev = float(p), int(seed), int(site), seed_dev

    def seed_dev_ptr(self):
        return self.seed_dev.data_ptr() if self.seed_dev is not None else None


class StepScalars:
    """The per-step scalars of a captured training step, in device memory (lr2_step_scalars_store): the dropout seed and up to
    14 learning rates.  store() is one small launch whose kernel arguments carry the values."""
    MAX_LRS = 14

    def __init__(self, device):
        self.buf = torch.zeros(8, dtype=torch.int64, device=device)
        self.seed = self.buf[0:1]
        self._lr_view = self.buf.view(torch.float32)
        self._lrs = (C.c_float * self.MAX_LRS)()
        self.n_lrs = 0

    def new_lr(self) -> int:
        if self.n_lrs >= self.MAX_LRS:
            raise ValueError("StepScalars holds at most 14 learning rates")
        self.n_lrs += 1
        return self.n_lrs - 1

    def lr_tensor(self, slot: int) -> torch.Tensor:
        return self._lr_view[2 + slot:3 + slot]

    def store(self, seed: int, lrs):
        for i, v in enumerate(lrs):
            self._lrs[i] = v
        _nat.check(_nat.lib().lr2_step_scalars_store(self.buf.data_ptr(), int(seed) & 0xFFFFFFFFFFFFFFFF, self._lrs, len(lrs),
                                                     _stream()), "lr2_step_scalars_store")


class Planes:
    """A matrix stored as two bf16 planes [hi | lo] (x = hi + lo) -- the operand format of the split-bf16 GEMM.
    Same bytes as the fp32 [rows, cols] tensor it replaces.  `lo_off` = elements from the hi plane to the lo plane."""
    __slots__ = ("buf", "rows", "cols", "lo_off", "transposed")

    def __init__(self, buf: torch.Tensor, rows: int, cols: int, lo_off: Optional[int] = None):
        self.transposed = False      # True: this is W^T [in, out] of an nn.Linear weight (see split_planes_t)
        if buf.dtype != torch.int16 or not buf.is_cuda:
            raise TypeError("Planes storage must be an int16 HIP tensor")
        self.buf, self.rows, self.cols = buf, rows, cols
        self.lo_off = rows * cols if lo_off is None else lo_off
        if buf.numel() < self.lo_off + rows * cols:
            raise ValueError("Planes storage too small")

    @staticmethod
    def empty(rows: int, cols: int, device) -> "Planes":
        return Planes(torch.empty(2 * rows * cols, dtype=torch.int16, device=device), rows, cols)

    def data_ptr(self) -> int:
        return self.buf.data_ptr()

    def plane_bytes(self) -> int:
        return self.rows * self.cols * 2

    def to_float(self) -> torch.Tensor:
        """hi + lo as fp32 (tests / debugging only)."""
        n = self.rows * self.cols
        hi = self.buf[:n].view(torch.bfloat16).float()
        lo = self.buf[self.lo_off:self.lo_off + n].view(torch.bfloat16).float()
        return (hi + lo).view(self.rows, self.cols)


def split_planes(src: torch.Tensor, dst: Planes):
    _chk_f32(src)
    n = src.numel()
    _nat.check(_nat.lib().lr2_split_planes(src.data_ptr(), dst.data_ptr(), dst.lo_off, n, _stream()), "lr2_split_planes")
    return dst


def split_planes_t(src: torch.Tensor, dst: Planes):
    """dst [C, R] planes = transpose of the fp32 matrix src [R, C]; marks dst as a transposed weight."""
    _chk_f32(src)
    R, C = src.shape
    if dst.rows != C or dst.cols != R or not src.is_contiguous():
        raise ValueError("split_planes_t: dst must be [C, R] planes of a contiguous [R, C] matrix")
    _nat.check(_nat.lib().lr2_split_planes_t(src.data_ptr(), dst.data_ptr(), dst.lo_off, R, C, _stream()), "lr2_split_planes_t")
    dst.transposed = True
    return dst


def dropout_planes(src: torch.Tensor, dst: Planes, drop: Optional[Drop]):
    """dst = planes of dropout_mask(src) / (1 - p) (drop None / p = 0: plain split)."""
    _chk_f32(src)
    p, seed, site = (drop.p, drop.seed, drop.site) if drop is not None else (0.0, 0, 0)
    _nat.check(_nat.lib().lr2_dropout_planes(src.data_ptr(), dst.data_ptr(), dst.lo_off, src.numel(), p, seed, site, _stream()),
               "lr2_dropout_planes")
    return dst


def dropout_apply(src: torch.Tensor, dst: torch.Tensor, drop: Optional[Drop]):
    """dst = dropout_mask(src) / (1 - p) (fp32; drop None / p = 0: copy)."""
    _chk_f32(src, dst)
    if drop is None or drop.p <= 0.0:
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src)
        return dst
    _nat.check(_nat.lib().lr2_dropout_apply(src.data_ptr(), dst.data_ptr(), src.numel(), drop.p, drop.seed, drop.site, _stream()),
               "lr2_dropout_apply")
    return dst


def text_embed_bwd(dx, src, seg, dword, dseg, *, rows, D):
    """dword[t] = sum of dx rows whose token is t (rows of tokens PRESENT in src are overwritten, all others left as they are:
    zero the table first; a caller accumulating into a live table must add the result itself); dseg[s] = sum of dx rows with
    segment s.  Deterministic: rows are grouped by token with a stable device sort, summed in row order inside pieces of 256
    sorted positions and the pieces of a long run (the padding id) in piece order (no float atomics)."""
    _chk_f32(dx, dword, dseg)
    if src.dtype != torch.int64 or seg.dtype != torch.int64:
        raise TypeError("src / seg must be int64")
    sorted_ids, order = torch.sort(src.view(-1), stable=True)
    rpb, wseg = 64, 256                                          # LR2_TEXT_EMBED_BWD_ROWS_PER_BLOCK / _WORD_SEG
    n_seg = dseg.shape[0]
    partials = torch.empty(((rows + rpb - 1) // rpb) * n_seg * D, dtype=torch.float32, device=dx.device)
    wpart = torch.empty(2 * ((rows + wseg - 1) // wseg) * D, dtype=torch.float32, device=dx.device)
    _nat.check(_nat.lib().lr2_text_embed_bwd(dx.data_ptr(), sorted_ids.data_ptr(), order.data_ptr(), seg.data_ptr(),
                                             dword.data_ptr(), dseg.data_ptr(), partials.data_ptr(), wpart.data_ptr(), rows, D,
                                             dword.shape[0], n_seg, _stream()), "lr2_text_embed_bwd")


def split_planes_multi(table_dev: torch.Tensor, n_chunks: int):
    _nat.check(_nat.lib().lr2_split_planes_multi(table_dev.data_ptr(), n_chunks, _stream()), "lr2_split_planes_multi")


class AdamArgs:
    """AdamW step fused into a weight-gradient GEMM (lr2_epilogue.adam_*): the GEMM result is the gradient of `p` and is
    consumed on the fly -- m, v, p are updated exactly as lr2_adamw_multi would, the gradient never reaches HBM."""
    __slots__ = ("p", "m", "v", "lr", "beta1", "beta2", "eps", "weight_decay", "lr_dev")

    def __init__(self, p, m, v, lr, beta1, beta2, eps, weight_decay, lr_dev: Optional[torch.Tensor] = None):
        self.lr_dev = lr_dev        # float32[1] device tensor read by the kernel in place of lr (captured steps)
        _chk_f32(p, m, v)
        if not (p.is_contiguous() and m.is_contiguous() and v.is_contiguous() and p.shape == m.shape == v.shape):
            raise ValueError("AdamArgs: p, m, v must be contiguous and of one shape")
        self.p, self.m, self.v = p, m, v
        self.lr, self.beta1, self.beta2, self.eps, self.weight_decay = lr, beta1, beta2, eps, weight_decay


def use_gemm256(M: int, N: int, K: int) -> bool:
    return _use_gemm256(M, N, K, _PASSES)


@functools.lru_cache(maxsize=4096)
def _use_gemm256(M: int, N: int, K: int, passes: int) -> bool:
    """True when an NT product of planes should run on the 256 x 256 ping-pong kernel (csrc/gemm256.hip): whole 32-deep K
    steps and enough 256 x 256 tiles that the last round of one-workgroup-per-CU rounds is well filled.  Measured on
    MI355X (tools/gemm_bench.py --planes): M = 100864 (ViT-B/16 over 512 frames) 355-385 TFLOP/s at K = 768 and ~460 at
    K = 3072 against 310-325 / 355 for the 128-row kernels; at M = 12544 (0.6-2.3 rounds) the two are level."""
    if K % 32 or passes != 3:
        return False
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    rounds = -(-tiles // 256)
    if tiles >= 256 and tiles >= 0.85 * rounds * 256:
        return True
    # round 4 (tools/dbg/rowsplit_ab.py): more than one round with a last round LESS than half full -- lr2_gemm then sends the rows of
    # the whole rounds to the 256 x 256 kernel and the remaining rows to the 128- / 64-row kernels (row split: two launches, no
    # reduction): M = 12544, N = 3072, K = 768 (588 tiles) 166 us against 199 (NN on 128-row tiles) / 184 (three rounds of 256 x 256)
    if 256 < tiles < 5 * 256 and 0 < tiles % 256 < 128 and K % 64 == 0 and os.environ.get("LR2_GEMM_ROWSPLIT", "1") != "0":
        return True
    # one partial round, long contraction (M = 12544, N = 768, K = 3072: 147 tiles): each CU runs one tile at the 256 x 256 kernel's
    # main-loop rate and the epilogue is amortised over 96 K steps -- 182 us against 197 (64-row tiles) / 212 (128-row), round 3;
    # round 4: also three quarters of a round at K >= 1024 (M = 12544, N = 1024, K = 1024: 196 tiles, 77 us against 86)
    if (128 <= tiles <= 256 and K >= 2304) or (192 <= tiles <= 256 and K >= 1024):
        return True
    # round 4, late: the kernel's 192-row tile (gemm256.hip, MIH = 3; the launcher takes it when a round of 192-row tiles fills CUs a
    # round of 256-row tiles leaves idle): M = 12544, N = 768: 147 -> 198 workgroups -- K = 768: 50 us against 56 (64-row tiles),
    # K = 1536: 80 against 94 (tools/dbg/gemm192_ab.py)
    t192 = ((M + 191) // 192) * ((N + 255) // 256)
    return tiles < 256 and 192 <= t192 <= 256 and t192 > tiles and K >= 768 and os.environ.get("LR2_GEMM_192", "1") != "0"


@functools.lru_cache(maxsize=4096)
def _gemm256_tn_splits(M: int, N: int, K: int, passes: int) -> int:
    """K-split count when the TN product C[M, N] = A[K, M]^T B[K, N] of planes should run on the TN form of the 256 x 256 kernel
    (csrc/gemm256.hip::gemm256_tn_kernel), else 0.  Weight gradients at thousands of token rows: few output tiles (768 x 3072 =
    36), a long contraction; tiles x splits fill ONE round of the 256 CUs, each workgroup runs >= 16 K steps of 32 rows."""
    if passes != 3 or K < 4096 or os.environ.get("LR2_GEMM_256_TN", "1") == "0":
        return 0
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles > 128 or M * N < 0.85 * tiles * 65536:
        return 0
    steps = (K + 31) // 32
    return max(1, min(256 // tiles, steps // 16))


@functools.lru_cache(maxsize=4096)
def choose_tiling(M: int, N: int, K: int, trans_a: bool, trans_b: bool = False):
    """(block_m, splits).  256 CUs hold 2 (BM=128) or 3 (BM=64) workgroups each; pick the split-K factor that minimises
    rounds x (K-tiles per workgroup + fixed prologue/epilogue cost) + the cost of writing/reading the partial slabs, so
    that skinny GEMMs fill the chip without wave-quantisation tails (576 workgroups on 512 slots = 2 rounds)."""
    if not trans_a and not trans_b and use_gemm256(M, N, K):
        return 256, 1        # honoured for planes x planes operands only (lr2_gemm falls back to 128-row tiles otherwise)
    if trans_a and trans_b:
        sp256 = _gemm256_tn_splits(M, N, K, _PASSES)
        if sp256:
            return 256, sp256    # planes x planes only, as above
    bm = 64 if (M <= 64 and not trans_a) else 128
    tiles = ((M + bm - 1) // bm) * ((N + 127) // 128)
    if bm == 128 and not trans_a and not trans_b and tiles < 1536:
        # NT with fewer than three rounds of 128-row tiles (512 resident workgroups): when the last round is poorly filled
        # it costs more than the lower intensity of 64-row tiles.  Measured with tools/gemm_bench.py: M=12544 N=768
        # (588 tiles) +9..15 %, M=6304 N=768 K=768 (300) +17 %, M=6304 N=3072 (1200) +17 %; full rounds (4096^3: 1024
        # tiles) and M=12544 N=3072 (2352) are 10-25 % faster with 128-row tiles.
        last = tiles % 512
        if (tiles > 512 and 0 < last <= 128) or (256 < tiles <= 512 and K < 1536):
            return 64, 1
    k_tiles = (K + 63) // 64
    slots = 768 if bm == 64 else 512
    if bm == 128 and not trans_a and tiles < 512 and K <= 1024 and ((M + 63) // 64) * ((N + 127) // 128) >= 64:
        # small short-K GEMMs (image tokens: M = 1024, K = 768): one pass of 64-row tiles beats split-K + its reduce launch
        # (NT 1024x768x768: 16.6 vs 19.0 us, NN 1024x3072x768: 21.9 vs 28.7 us)
        return 64, 1
    if tiles >= slots or k_tiles < 8:
        return bm, 1
    slab_cost = M * N * 6.4e-7            # one fp32 slab written + read, in units of one K-tile step (~2.5 us)
    best, best_cost = 1, None
    for s in range(1, min(k_tiles // 2, 64) + 1):
        per = -(-k_tiles // s)
        s_eff = -(-k_tiles // per)
        rounds = -(-(tiles * s_eff) // slots)
        cost = rounds * (per + 3.0) + (s_eff * slab_cost if s_eff > 1 else 0.0)
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = s_eff, cost
    return bm, best


def _LIB_GEMM(*args):
    """Bound on first use (the library is loaded lazily so that importing ops never needs the GPU)."""
    global _LIB_GEMM
    _LIB_GEMM = _nat.lib().lr2_gemm
    return _LIB_GEMM(*args)


def gemm(a, b, out: Optional[torch.Tensor], M: int, N: int, K: int, *, trans_a=False, trans_b=False,
         lda: Optional[int] = None, ldb: Optional[int] = None, ld_out: Optional[int] = None,
         bias: Optional[torch.Tensor] = None, act: int = 0, out_z: Optional[torch.Tensor] = None,
         aux_z: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, drop: Optional[Drop] = None,
         accumulate: bool = False, alpha: float = 1.0, out_planes: Optional[Planes] = None,
         splitk_ws: Optional[torch.Tensor] = None, splits: Optional[int] = None, block_m: Optional[int] = None,
         passes: Optional[int] = None, adam: Optional[AdamArgs] = None, colsum: Optional[torch.Tensor] = None,
         colsum_ws: Optional[torch.Tensor] = None):
    """out[M,N] = op(a) @ op(b) with the fused epilogue; a / b are fp32 tensors or Planes; the result goes to `out`
    (fp32) and/or `out_planes`, or -- with `adam` -- straight into the AdamW update of adam.p[M,N].
    See lr2_gemm in include/lr2ppo_hip.h."""
    a_pl, b_pl = isinstance(a, Planes), isinstance(b, Planes)
    _chk_f32(None if a_pl else a, None if b_pl else b, out, bias, out_z, aux_z, resid)
    if lda is None:
        lda = M if trans_a else K
    if ldb is None:
        ldb = N if trans_b else K
    if ld_out is None:
        ld_out = N
    if block_m is None or splits is None:
        bm, sp = choose_tiling(M, N, K, trans_a, trans_b)
    if block_m is not None:
        bm = block_m
    if splits is not None:
        sp = splits
    if sp > 1:
        need = sp * M * N
        if splitk_ws is None or splitk_ws.numel() < need:
            raise ValueError(f"split-K workspace too small: need {need} floats")
    e = _nat.Epilogue()
    e.bias, e.resid, e.aux_z, e.out, e.out_z = _ptr(bias), _ptr(resid), _ptr(aux_z), _ptr(out), _ptr(out_z)
    e.ld_resid, e.ld_aux, e.ld_out, e.ld_z = _ld(resid, N), _ld(aux_z, N), ld_out, _ld(out_z, N)
    if out_planes is not None:
        e.out_hi, e.out_lo_off, e.ld_planes = out_planes.data_ptr(), out_planes.lo_off, out_planes.cols
    e.act, e.accumulate, e.alpha = act, 1 if accumulate else 0, alpha
    if drop is not None and drop.p > 0.0:
        e.drop_p, e.drop_seed, e.drop_site, e.drop_seed_dev = drop.p, drop.seed, drop.site, drop.seed_dev_ptr()
    if adam is not None:
        if out is not None or out_planes is not None or adam.p.numel() != M * N or ld_out != N:
            raise ValueError("gemm(adam=...): the result is consumed by the update; p must be [M, N] and out/out_planes None")
        e.adam_p, e.adam_m, e.adam_v = adam.p.data_ptr(), adam.m.data_ptr(), adam.v.data_ptr()
        e.adam_lr, e.adam_beta1, e.adam_beta2 = adam.lr, adam.beta1, adam.beta2
        e.adam_eps, e.adam_weight_decay = adam.eps, adam.weight_decay
        e.adam_lr_dev = adam.lr_dev.data_ptr() if adam.lr_dev is not None else None
    if colsum is not None:
        # weight-gradient form: colsum[m] = sum_k A[k, m] (the bias gradient) from the same launch (lr2_epilogue.colsum)
        _chk_f32(colsum, colsum_ws)
        if not (trans_a and trans_b) or colsum.numel() != M or colsum_ws is None \
                or colsum_ws.numel() < max(128, sp * ((N + 255) // 256)) * M:
            raise ValueError("gemm(colsum=...): TN form only; colsum [M], colsum_ws >= max(128, splits * ceil(N / 256)) * M floats")
        e.colsum, e.colsum_ws = colsum.data_ptr(), colsum_ws.data_ptr()
    a_bytes = a.plane_bytes() if a_pl else a.numel() * 4
    b_bytes = b.plane_bytes() if b_pl else b.numel() * 4
    args = (a.data_ptr(), b.data_ptr(), M, N, K, lda, ldb, 1 if trans_a else 0, 1 if trans_b else 0, a_bytes, b_bytes,
            1 if a_pl else 0, a.lo_off * 2 if a_pl else 0, 1 if b_pl else 0, b.lo_off * 2 if b_pl else 0, C.byref(e),
            _ptr(splitk_ws), sp, bm, passes or _PASSES, _stream())
    if _PROF is None:               # the steady-state path: no label formatting, no event objects
        rc = _LIB_GEMM(*args)
    else:
        form = "TN" if trans_a else ("NN" if trans_b else "NT")
        src = ("p" if a_pl else "f") + ("p" if b_pl else "f")
        label = f"gemm_{form}_{src}_M{M}_N{N}_K{K}" + ("_adamw" if adam is not None else "")
        alg_bytes = 4.0 * (M * K + N * K) + (24.0 if adam is not None else 4.0) * M * N
        with _Timed(label, 2.0 * M * N * K, alg_bytes):
            rc = _LIB_GEMM(*args)
    if rc:
        _nat.check(rc, f"lr2_gemm(M={M},N={N},K={K},ta={trans_a},tb={trans_b},planes a={a_pl} b={b_pl})")
    return out if out is not None else out_planes


def _ld(t: Optional[torch.Tensor], default: int) -> int:
    return default if t is None else t.shape[-1]


def layernorm_fwd(x, gamma, beta, out, mean=None, rstd=None, *, rows, D, eps=1e-5, mode=0, group=0, group_stride=0,
                  out_planes: Optional[Planes] = None):
    """out (fp32) and/or out_planes receive LN(x); both use the (group, group_stride) row mapping."""
    _chk_f32(x, gamma, beta, out, mean, rstd)
    with _Timed(f"lnfwd_R{rows}_D{D}", 0.0, 8.0 * rows * D):
        rc = _nat.lib().lr2_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(out),
                                          out_planes.data_ptr() if out_planes is not None else None,
                                          out_planes.lo_off if out_planes is not None else 0, _ptr(mean), _ptr(rstd), rows, D,
                                          eps, mode, group, group_stride, _stream())
    _nat.check(rc, "lr2_layernorm_fwd")
    return out if out is not None else out_planes


# workgroups of the LayerNorm backward (each leaves one partial row of d gamma / d beta): 4 per CU.  With one per CU (round 2) the
# kernel kept 24 KB of loads in flight per CU and ran at 3.4 TB/s on the encoders' [100864, 768] rows (rocprofv3, profiles/r03_*).
LN_BWD_BLOCKS = 512       # workgroups of 4 rows-at-a-time waves: 2 per CU measured best at the encoder shapes (tools/dbg/ln_bwd_time.py)


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, partials, dgamma, dbeta, *, rows, D, group=0, group_stride=0,
                  resid_grad=None, dx_planes: Optional[Planes] = None, drop: Optional[Drop] = None, nblocks=LN_BWD_BLOCKS, mode=0,
                  eps=1e-5):
    """dx (fp32) = LN'(dy) + resid_grad; dx_planes = planes of dropout_mask(dx)/(1-p) (p = 0: of dx).
    mode / eps: the forward's (0 = nn.LayerNorm, 1 = TencentPretrain LayerNorm)."""
    _chk_f32(dy, x, gamma, mean, rstd, dx, partials, dgamma, dbeta, resid_grad)
    nb = min(nblocks, (rows + 3) // 4)
    if partials.numel() < nb * 2 * D:
        raise ValueError("layernorm_bwd partials workspace too small")
    p, seed, site = (drop.p, drop.seed, drop.site) if drop is not None else (0.0, 0, 0)
    L = _nat.lib()
    rc = L.lr2_layernorm_bwd(dy.data_ptr(), group, group_stride, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                             rstd.data_ptr(), _ptr(resid_grad), _ptr(dx),
                             dx_planes.data_ptr() if dx_planes is not None else None,
                             dx_planes.lo_off if dx_planes is not None else 0, p, seed, site,
                             drop.seed_dev_ptr() if drop is not None else None, partials.data_ptr(), nb, rows, D, mode, eps, _stream())
    _nat.check(rc, "lr2_layernorm_bwd")
    if dbeta.data_ptr() == dgamma.data_ptr() + 4 * D:       # [d gamma | d beta] adjacent (flat gradient buffers): one finishing launch
        _nat.check(L.lr2_colsum_partials_finish(partials.data_ptr(), nb, 2 * D, 2 * D, dgamma.data_ptr(), 0, _stream()), "finish")
        return
    _nat.check(L.lr2_colsum_partials_finish(partials.data_ptr(), nb, D, 2 * D, dgamma.data_ptr(), 0, _stream()), "finish")
    _nat.check(L.lr2_colsum_partials_finish(partials.data_ptr() + 4 * D, nb, D, 2 * D, dbeta.data_ptr(), 0, _stream()),
               "finish")


def colsum(x, out, partials, *, rows, cols, ld=None, nblocks=128):
    """Column sums of an fp32 matrix or a Planes matrix -> fp32 [cols]."""
    is_pl = isinstance(x, Planes)
    _chk_f32(None if is_pl else x, out, partials)
    nb = min(nblocks, rows)
    if partials.numel() < nb * cols:
        raise ValueError("colsum partials workspace too small")
    rc = _nat.lib().lr2_colsum(x.data_ptr(), 1 if is_pl else 0, x.lo_off if is_pl else 0, rows, cols, ld or cols,
                               partials.data_ptr(), nb, out.data_ptr(), _stream())
    _nat.check(rc, "lr2_colsum")
    return out


def xattn_fwd(q, k, v, o, *, batch, heads, Lq, Lk, head_dim, post_scale):
    """o: fp32 tensor or Planes."""
    o_pl = isinstance(o, Planes)
    _chk_f32(q, k, v, None if o_pl else o)
    _nat.check(_nat.lib().lr2_xattn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1 if o_pl else 0,
                                        o.lo_off if o_pl else 0, batch, heads, Lq, Lk, head_dim, post_scale, _stream()),
               "lr2_xattn_fwd")
    return o


def xattn_bwd(q, k, v, do, dq, dk, dv, *, batch, heads, Lq, Lk, head_dim, post_scale):
    """dq / dk / dv: all fp32 tensors or all Planes."""
    pl = isinstance(dq, Planes)
    _chk_f32(q, k, v, do)
    if not pl:
        _chk_f32(dq, dk, dv)
    _nat.check(_nat.lib().lr2_xattn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), do.data_ptr(), dq.data_ptr(), dk.data_ptr(),
                                        dv.data_ptr(), 1 if pl else 0, dq.lo_off if pl else 0, dk.lo_off if pl else 0, batch,
                                        heads, Lq, Lk, head_dim, post_scale, _stream()), "lr2_xattn_bwd")


def _qkv_ptrs(qkv, E):
    base = qkv.data_ptr()
    return base, base + 2 * E, base + 4 * E


def self_attn_fwd(qkv: "Planes", seg, o, *, batch, heads, L, head_dim, scale, lse=None, drop: Optional[Drop] = None):
    """Encoder self-attention on the matrix cores.  qkv: Planes [batch*L, 3*heads*head_dim] = [Q | K | V] (what one fused
    QKV GEMM writes); seg: int64 [batch*L]; o: fp32 tensor or Planes [batch*L, heads*head_dim]; lse: optional fp32
    [batch*heads*L]; drop: dropout on the probabilities."""
    E = heads * head_dim
    if not isinstance(qkv, Planes) or qkv.cols != 3 * E or qkv.rows != batch * L:
        raise TypeError("self_attn_fwd: qkv must be a Planes matrix [batch*L, 3*heads*head_dim]")
    if seg.dtype != torch.int64:
        raise TypeError("seg must be int64")
    o_pl = isinstance(o, Planes)
    _chk_f32(None if o_pl else o, lse)
    q, k, v = _qkv_ptrs(qkv, E)
    p, seed, site = (drop.p, drop.seed, drop.site) if drop is not None else (0.0, 0, 0)
    with _Timed(f"selfattn_B{batch}_H{heads}_L{L}", 4.0 * batch * heads * L * L * head_dim, 16.0 * batch * L * E):
        _nat.check(_nat.lib().lr2_self_attn_fwd(q, k, v, qkv.lo_off, qkv.cols, seg.data_ptr(),
                                                None if o_pl else o.data_ptr(), o.data_ptr() if o_pl else None,
                                                o.lo_off if o_pl else 0, E, _ptr(lse), p, seed, site, batch, heads, L, head_dim,
                                                scale, _stream()), "lr2_self_attn_fwd")
    return o


def first_token_attn(q: torch.Tensor, kv: "Planes", seg, o: torch.Tensor, *, batch, heads, L, head_dim, scale):
    """Attention of query row 0 of every sequence against all keys (lr2_first_token_attn).  q: fp32 [batch, E] (projected);
    kv: Planes [batch*L, 2E] = [K | V]; seg: int64 [batch*L]; o: fp32 [batch, E]."""
    E = heads * head_dim
    if not isinstance(kv, Planes) or kv.cols != 2 * E or kv.rows != batch * L:
        raise TypeError("first_token_attn: kv must be a Planes matrix [batch*L, 2*heads*head_dim]")
    if seg.dtype != torch.int64:
        raise TypeError("seg must be int64")
    _chk_f32(q, o)
    if q.shape != (batch, E) or o.shape != (batch, E) or not q.is_contiguous() or not o.is_contiguous():
        raise ValueError("first_token_attn: q / o must be contiguous [batch, heads*head_dim]")
    _nat.check(_nat.lib().lr2_first_token_attn(q.data_ptr(), E, kv.data_ptr(), kv.data_ptr() + 2 * E, kv.lo_off, 2 * E,
                                               seg.data_ptr(), o.data_ptr(), E, batch, heads, L, head_dim, scale, _stream()),
               "lr2_first_token_attn")
    return o


def self_attn_bwd(qkv: "Planes", do: "Planes", seg, dqkv: "Planes", lse_ws, dsum_ws, *, batch, heads, L, head_dim, scale,
                  drop: Optional[Drop] = None, o: Optional["Planes"] = None):
    """dQKV (Planes [batch*L, 3E]) from QKV and dO (Planes [batch*L, E]); lse_ws / dsum_ws: fp32 [batch*heads*L].  Without `o` both are
    scratch (everything is recomputed).  With `o` = the forward's output planes [batch*L, E], lse_ws must hold the log-sum-exp
    self_attn_fwd(..., lse=lse_ws) wrote for the same inputs: the streaming / persistent kernels (lr2_self_attn_bwd, ABI 19)."""
    E = heads * head_dim
    for name, t, cols in (("qkv", qkv, 3 * E), ("do", do, E), ("dqkv", dqkv, 3 * E)) + ((("o", o, E),) if o is not None else ()):
        if not isinstance(t, Planes) or t.cols != cols or t.rows != batch * L:
            raise TypeError(f"self_attn_bwd: {name} must be a Planes matrix [batch*L, {cols}]")
    if seg.dtype != torch.int64:
        raise TypeError("seg must be int64")
    _chk_f32(lse_ws, dsum_ws)
    if lse_ws.numel() < batch * heads * L or dsum_ws.numel() < batch * heads * L:
        raise ValueError("self_attn_bwd: statistics workspaces too small")
    q, k, v = _qkv_ptrs(qkv, E)
    dq, dk, dv = _qkv_ptrs(dqkv, E)
    p, seed, site = (drop.p, drop.seed, drop.site) if drop is not None else (0.0, 0, 0)
    with _Timed(f"selfattnbwd_B{batch}_H{heads}_L{L}", 14.0 * batch * heads * L * L * head_dim, 32.0 * batch * L * E):
        _nat.check(_nat.lib().lr2_self_attn_bwd(q, k, v, qkv.lo_off, qkv.cols, do.data_ptr(), do.lo_off, do.cols, seg.data_ptr(),
                                                dq, dk, dv, dqkv.lo_off, dqkv.cols, o.data_ptr() if o is not None else None,
                                                o.lo_off if o is not None else 0, o.cols if o is not None else 0,
                                                lse_ws.data_ptr(), dsum_ws.data_ptr(), p, seed, site, batch, heads, L, head_dim, scale,
                                                _stream()), "lr2_self_attn_bwd")
    return dqkv


def self_attn_plan(batch: int, heads: int, L: int, ld: Optional[int] = None, ld_do: Optional[int] = None):
    """(forward persistent?, backward persistent?) for a call of this shape (lr2_self_attn_plan): which form of the kernels runs."""
    import ctypes
    E = heads * 64
    f, b = ctypes.c_int(), ctypes.c_int()
    _nat.check(_nat.lib().lr2_self_attn_plan(batch, heads, L, 3 * E if ld is None else ld, E if ld_do is None else ld_do,
                                             ctypes.byref(f), ctypes.byref(b)), "lr2_self_attn_plan")
    return bool(f.value), bool(b.value)


def gather_rows(src, index, dst, *, B, t_in, t_out, row_elems, src_bstride=None, src_tstride=None):
    _chk_f32(src, dst)
    if index is not None and index.dtype != torch.int64:
        raise TypeError("index must be int64")
    _nat.check(_nat.lib().lr2_gather_rows(src.data_ptr(), _ptr(index), dst.data_ptr(), B, t_in, t_out, row_elems,
                                    src_bstride if src_bstride is not None else t_in * row_elems,
                                    src_tstride if src_tstride is not None else row_elems, _stream()), "lr2_gather_rows")
    return dst


def gather_rows_bwd(ddst, index, dsrc, *, B, t_in, t_out, row_elems):
    _chk_f32(ddst, dsrc)
    _nat.check(_nat.lib().lr2_gather_rows_bwd(ddst.data_ptr(), _ptr(index), dsrc.data_ptr(), B, t_in, t_out, row_elems,
                                        _stream()), "lr2_gather_rows_bwd")
    return dsrc


def copy_rows(src, dst, *, rows, D, group, dst_gstride, dst_off):
    """dst: fp32 tensor or Planes (the concat buffer of finetune/ppo.py:224)."""
    pl = isinstance(dst, Planes)
    _chk_f32(src, None if pl else dst)
    _nat.check(_nat.lib().lr2_copy_rows(src.data_ptr(), dst.data_ptr(), 1 if pl else 0, dst.lo_off if pl else 0, rows, D,
                                        group, dst_gstride, dst_off, _stream()), "lr2_copy_rows")
    return dst


def head_fwd(x, w, b, y, *, rows, D, row_step=1, row_off=0):
    _chk_f32(x, w, b, y)
    _nat.check(_nat.lib().lr2_head_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, D, row_step, row_off,
                                 _stream()), "lr2_head_fwd")
    return y


def head_bwd(x, w, dy, dx, dw, db, *, rows, D, row_step=1, row_off=0, total_rows=None):
    _chk_f32(x, w, dy, dx, dw, db)
    _nat.check(_nat.lib().lr2_head_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dx), _ptr(dw), _ptr(db), rows, D,
                                 row_step, row_off, total_rows if total_rows is not None else rows * row_step, _stream()),
            "lr2_head_bwd")


def add_period_rows(x, table, out, *, rows, D, period):
    _chk_f32(x, table, out)
    _nat.check(_nat.lib().lr2_add_period_rows(x.data_ptr(), table.data_ptr(), out.data_ptr(), rows, D, period, _stream()),
            "lr2_add_period_rows")
    return out


def period_rows_grad(dy, dtable, *, rows, D, period):
    _chk_f32(dy, dtable)
    _nat.check(_nat.lib().lr2_period_rows_grad(dy.data_ptr(), dtable.data_ptr(), rows, D, period, _stream()),
            "lr2_period_rows_grad")
    return dtable


def ppo_loss(scores, old_scores, rewards, old_value, value, next_state, scalars, per_item, dscores, dvalue, *, B, T,
             kl_w, ent_w, value_clip, margin=0.01, adv_eps=-0.1, rank_len=2, stats_out=None, global_stats=None, world=1):
    """stats_out (fp32[3]): write only this rank's {hinge sum, positive count, sum |A|} (pass 1 of the data-parallel form);
    global_stats (fp32[3], all-reduced) + world: use the global RankLoss statistics (pass 2); see lr2_ppo_loss."""
    _chk_f32(scores, old_scores, rewards, old_value, value, scalars, per_item, dscores, dvalue, stats_out, global_stats)
    if next_state.dtype != torch.int64 or not next_state.is_contiguous():
        raise TypeError("next_state must be contiguous int64")
    _nat.check(_nat.lib().lr2_ppo_loss(scores.data_ptr(), old_scores.data_ptr(), rewards.data_ptr(), old_value.data_ptr(),
                                 value.data_ptr(), next_state.data_ptr(), next_state.shape[1], rank_len, B, T, kl_w, ent_w,
                                 value_clip, margin, adv_eps, _ptr(scalars), _ptr(per_item), _ptr(dscores), _ptr(dvalue),
                                 _ptr(stats_out), _ptr(global_stats), world, _stream()), "lr2_ppo_loss")


def cls_head_fwd(x, w, b, y, *, rows, D, C):
    _chk_f32(x, w, b, y)
    _nat.check(_nat.lib().lr2_cls_head_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, D, C, _stream()),
               "lr2_cls_head_fwd")
    return y


def cls_head_bwd(x, w, dy, dx, dw, db, *, rows, D, C):
    _chk_f32(x, w, dy, dx, dw, db)
    _nat.check(_nat.lib().lr2_cls_head_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), _ptr(dx), _ptr(dw), _ptr(db), rows, D, C,
                                           _stream()), "lr2_cls_head_bwd")


def cls_scores(logits, probs, scores, *, rows, C, softmax=True):
    """scores[r] = sum_k k * (softmax(logits[r]) if softmax else logits[r])_k; probs (optional) receives the distribution."""
    _chk_f32(logits, probs, scores)
    _nat.check(_nat.lib().lr2_cls_scores(logits.data_ptr(), _ptr(probs), scores.data_ptr(), rows, C, 1 if softmax else 0,
                                         _stream()), "lr2_cls_scores")
    return scores


def cls_scores_bwd(probs, scores, dscores, dlogits, *, rows, C):
    _chk_f32(probs, scores, dscores, dlogits)
    _nat.check(_nat.lib().lr2_cls_scores_bwd(probs.data_ptr(), scores.data_ptr(), dscores.data_ptr(), dlogits.data_ptr(), rows,
                                             C, _stream()), "lr2_cls_scores_bwd")
    return dlogits


def nll_loss(logits, tgts, loss, dlogits=None, *, rows, C):
    _chk_f32(logits, loss, dlogits)
    if tgts.dtype != torch.int64:
        raise TypeError("tgts must be int64")
    _nat.check(_nat.lib().lr2_nll_loss(logits.data_ptr(), tgts.data_ptr(), rows, C, loss.data_ptr(), _ptr(dlogits), _stream()),
               "lr2_nll_loss")
    return loss


def smooth_l1(pred, target, loss, dpred=None, *, n, beta=0.3):
    _chk_f32(pred, target, loss, dpred)
    _nat.check(_nat.lib().lr2_smooth_l1(pred.data_ptr(), target.data_ptr(), n, beta, loss.data_ptr(), _ptr(dpred), _stream()),
            "lr2_smooth_l1")
    return loss


def pair_hinge(scores, loss_acc, dscores=None, *, bs, margin=1.0):
    """scores: [2*bs] = chosen then reject; loss_acc: [2] (loss, accuracy); dscores: [2*bs] or None."""
    _chk_f32(scores, loss_acc, dscores)
    if scores.numel() != 2 * bs or loss_acc.numel() < 2 or (dscores is not None and dscores.numel() != 2 * bs):
        raise ValueError("pair_hinge: scores/dscores must hold 2*bs elements, loss_acc 2")
    _nat.check(_nat.lib().lr2_pair_hinge(scores.data_ptr(), bs, margin, loss_acc.data_ptr(), _ptr(dscores), _stream()),
               "lr2_pair_hinge")
    return loss_acc


def adamw_multi(table_dev: torch.Tensor, n_chunks: int, lr: float, beta1: float, beta2: float, eps: float,
                n_params: int = 0, lr_dev: Optional[torch.Tensor] = None):
    with _Timed(f"adamw_{n_params}", 0.0, 28.0 * n_params):
        _nat.check(_nat.lib().lr2_adamw_multi(table_dev.data_ptr(), n_chunks, lr, beta1, beta2, eps,
                                              lr_dev.data_ptr() if lr_dev is not None else None, _stream()), "lr2_adamw_multi")


class Mx8:
    """A matrix in MX-FP8 (OCP MX v1.0: e4m3fn bytes + one E8M0 scale byte per 32 consecutive elements of a row) -- the operand
    format of gemm_mxfp8, the inference-only fp8 fast mode (csrc/fp8.hip; NOT the parity path)."""
    __slots__ = ("q", "s", "rows", "cols")

    def __init__(self, q: torch.Tensor, s: torch.Tensor, rows: int, cols: int):
        if q.dtype != torch.uint8 or s.dtype != torch.uint8 or not q.is_cuda or q.numel() < rows * cols or s.numel() < rows * (cols // 32):
            raise TypeError("Mx8: uint8 HIP tensors [rows, cols] and [rows, cols / 32]")
        self.q, self.s, self.rows, self.cols = q, s, rows, cols

    @staticmethod
    def empty(rows: int, cols: int, device) -> "Mx8":
        return Mx8(torch.empty(rows * cols, dtype=torch.uint8, device=device), torch.empty(rows * (cols // 32), dtype=torch.uint8, device=device),
                   rows, cols)

    def to_float(self) -> torch.Tensor:
        """Dequantised fp32 matrix (tests / debugging)."""
        v = self.q[:self.rows * self.cols].view(torch.float8_e4m3fn).float().view(self.rows, self.cols // 32, 32)
        sc = torch.exp2(self.s[:self.rows * (self.cols // 32)].float() - 127.0).view(self.rows, self.cols // 32, 1)
        return (v * sc).view(self.rows, self.cols)


def quant_mxfp8(x: torch.Tensor, dst: Optional[Mx8] = None) -> Mx8:
    """x fp32 [rows, K] (rows may be strided, K % 32 == 0) -> Mx8 (lr2_quant_mxfp8)."""
    _chk_f32(x)
    if x.dim() != 2 or x.stride(1) != 1 or x.shape[1] % 32:
        raise ValueError("quant_mxfp8: a 2-D fp32 matrix with contiguous rows and K % 32 == 0")
    R, K = x.shape
    dst = dst or Mx8.empty(R, K, x.device)
    if dst.rows != R or dst.cols != K:
        raise ValueError("quant_mxfp8: destination shape")
    _nat.check(_nat.lib().lr2_quant_mxfp8(x.data_ptr(), x.stride(0), dst.q.data_ptr(), dst.s.data_ptr(), R, K, _stream()), "lr2_quant_mxfp8")
    return dst


def layernorm_fwd_mxfp8(x, gamma, beta, dst: Mx8, out: Optional[torch.Tensor] = None, *, rows, D, eps=1e-5, mode=0) -> Mx8:
    """LN(x) as MX-FP8 (and as fp32 when `out` is given) -- lr2_layernorm_fwd_mxfp8."""
    _chk_f32(x, gamma, beta, out)
    if dst.rows != rows or dst.cols != D:
        raise ValueError("layernorm_fwd_mxfp8: destination shape")
    with _Timed(f"lnfwd_mx_R{rows}_D{D}", 0.0, 5.0 * rows * D):
        _nat.check(_nat.lib().lr2_layernorm_fwd_mxfp8(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(out), dst.q.data_ptr(),
                                                      dst.s.data_ptr(), rows, D, eps, mode, _stream()), "lr2_layernorm_fwd_mxfp8")
    return dst


def gemm_mxfp8(a: Mx8, b: Mx8, out: Optional[torch.Tensor], *, bias=None, resid=None, act: int = 0, out_mx: Optional[Mx8] = None,
               out_planes: Optional[Planes] = None, out_bf16: Optional[torch.Tensor] = None):
    """out[M, N] fp32 = a . b^T (+ bias) (act 1: GELU) (+ resid) with a [M, K], b [N, K] in MX-FP8 (lr2_gemm_mxfp8).
    out_mx: the result also (out given) or only (out None) as MX-FP8 -- the next product's A operand.
    out_bf16: the result as ONE bf16 plane (an int16 / bfloat16 tensor of M * N elements): the operands of self_attn_fwd_bf16."""
    _chk_f32(out, bias, resid)
    M, N, K = a.rows, b.rows, a.cols
    if out_bf16 is not None:
        if out_planes is not None or out_bf16.element_size() != 2 or not out_bf16.is_cuda or out_bf16.numel() < M * N:
            raise ValueError("gemm_mxfp8: out_bf16 is a 2-byte HIP tensor of M * N elements, and excludes out_planes")
        out_planes = Planes.__new__(Planes)       # a single plane: lo_off = 0 (see include/lr2ppo_hip.h)
        out_planes.buf, out_planes.rows, out_planes.cols, out_planes.lo_off, out_planes.transposed = out_bf16, M, N, 0, False
    if b.cols != K or (out is None and out_mx is None and out_planes is None):
        raise ValueError("gemm_mxfp8: a [M, K], b [N, K], out [M, N] and / or out_mx / out_planes / out_bf16")
    if out_planes is not None and (out_planes.rows != M or out_planes.cols != N):
        raise ValueError("gemm_mxfp8: out_planes must be [M, N]")
    if out is not None and (out.dim() != 2 or out.shape[0] != M or out.shape[1] != N or out.stride(1) != 1):
        raise ValueError("gemm_mxfp8: out must be [M, N] with contiguous rows")
    if out_mx is not None and (out_mx.rows != M or out_mx.cols != N):
        raise ValueError("gemm_mxfp8: out_mx must be [M, N]")
    with _Timed(f"gemm_mxfp8_M{M}_N{N}_K{K}", 2.0 * M * N * K, float(M * K + N * K + 4 * M * N)):
        _nat.check(_nat.lib().lr2_gemm_mxfp8(a.q.data_ptr(), a.s.data_ptr(), b.q.data_ptr(), b.s.data_ptr(), _ptr(out),
                                             out.stride(0) if out is not None else N, _ptr(bias), _ptr(resid),
                                             resid.stride(0) if resid is not None else 0, act,
                                             out_mx.q.data_ptr() if out_mx is not None else None,
                                             out_mx.s.data_ptr() if out_mx is not None else None,
                                             out_planes.data_ptr() if out_planes is not None else None,
                                             out_planes.lo_off if out_planes is not None else 0,
                                             out_planes.cols if out_planes is not None else 0, M, N, K, _stream()), "lr2_gemm_mxfp8")
    return out if out is not None else (out_mx if out_mx is not None else out_planes)


def self_attn_fwd_bf16(qkv: torch.Tensor, seg, *, batch, heads, L, head_dim, scale, out: Optional[torch.Tensor] = None,
                       out_mx: Optional[Mx8] = None):
    """Encoder self-attention of the MX-FP8 mode (lr2_self_attn_fwd_bf16).  qkv: ONE bf16 plane [batch * L, 3E] = [Q | K | V] (a 2-byte
    HIP tensor: what gemm_mxfp8(out_bf16=...) writes); the context as fp32 `out` [batch * L, E] and / or as MX-FP8 `out_mx`."""
    E = heads * head_dim
    if qkv.element_size() != 2 or not qkv.is_cuda or qkv.numel() < batch * L * 3 * E:
        raise TypeError("self_attn_fwd_bf16: qkv is a 2-byte HIP tensor [batch * L, 3 * heads * head_dim]")
    if seg.dtype != torch.int64:
        raise TypeError("seg must be int64")
    _chk_f32(out)
    if out is None and out_mx is None:
        raise ValueError("self_attn_fwd_bf16: out and / or out_mx")
    if out_mx is not None and (out_mx.rows != batch * L or out_mx.cols != E):
        raise ValueError("self_attn_fwd_bf16: out_mx must be [batch * L, heads * head_dim]")
    base = qkv.data_ptr()
    with _Timed(f"selfattn_bf16_B{batch}_H{heads}_L{L}", 4.0 * batch * heads * L * L * head_dim, 7.0 * batch * L * E):
        _nat.check(_nat.lib().lr2_self_attn_fwd_bf16(base, base + 2 * E, base + 4 * E, 3 * E, seg.data_ptr(), _ptr(out),
                                                     out_mx.q.data_ptr() if out_mx is not None else None,
                                                     out_mx.s.data_ptr() if out_mx is not None else None, E, batch, heads, L, head_dim,
                                                     scale, _stream()), "lr2_self_attn_fwd_bf16")
    return out if out is not None else out_mx


def text_embed(src, seg, word, pos, seg_table, out, *, rows, L, D, err: Optional[torch.Tensor] = None):
    """err: optional int32[1] device word; bit 0 / bit 1 are set when a token / segment id is out of range."""
    _chk_f32(word, pos, seg_table, out)
    if src.dtype != torch.int64 or seg.dtype != torch.int64:
        raise TypeError("src / seg must be int64")
    if L > pos.shape[0]:
        raise IndexError(f"sequence length {L} exceeds the position table ({pos.shape[0]} rows)")
    _nat.check(_nat.lib().lr2_text_embed(src.data_ptr(), seg.data_ptr(), word.data_ptr(), pos.data_ptr(), seg_table.data_ptr(),
                                         out.data_ptr(), rows, L, D, word.shape[0], seg_table.shape[0], _ptr(err), _stream()),
               "lr2_text_embed")
    return out


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # tencentpretrain/utils/dataloader.py:561
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def patchify_planes(img, out: Planes, *, B, Cc, H, W, ps, mean=None, std=None):
    """Patch rows of `img` as planes [B*P, Cc*ps*ps].  img: fp32 [B,Cc,H,W], or uint8 frames (then normalised on the fly
    with mean / std per channel when both are given: (x / 255 - mean) / std)."""
    is_u8 = img.dtype == torch.uint8
    if not is_u8:
        _chk_f32(img)
    if not img.is_cuda or not img.is_contiguous():
        raise TypeError("patchify_planes: img must be a contiguous HIP tensor")
    P = (H // ps) * (W // ps)
    if out.rows != B * P or out.cols < Cc * ps * ps:
        raise ValueError("patchify_planes: out must be [B*P, >= C*ps*ps] planes (extra columns are zero-filled)")
    m3 = s3 = None
    if is_u8 and mean is not None and std is not None:
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    _nat.check(_nat.lib().lr2_patchify_planes(img.data_ptr(), 1 if is_u8 else 0, out.data_ptr(), out.lo_off, out.cols, B, Cc, H, W,
                                              ps, m3, s3, _stream()), "lr2_patchify_planes")
    return out


_NDCG_TABLES = {}


def ndcg(scores, gold, offsets, ks=(1, 3, 5, 10, 20, 100000000)):
    """-> fp32 [n_items, len(ks)] NDCG@k of ragged items (scores / gold flat, offsets int64 [n_items + 1]), on the device."""
    _chk_f32(scores)
    if gold.dtype != torch.int64 or offsets.dtype != torch.int64:
        raise TypeError("gold / offsets must be int64")
    dev = scores.device
    key = (str(dev), tuple(ks))
    if key not in _NDCG_TABLES:
        # discount table computed exactly as the reference does (torch.log2 of an int64 tensor on the host, ndcg.py:31)
        disc = torch.log2(torch.arange(64, dtype=torch.int64) + 2).to(dev)
        _NDCG_TABLES[key] = (disc, torch.tensor(list(ks), dtype=torch.int64, device=dev))
    disc, ks_t = _NDCG_TABLES[key]
    n_items = offsets.numel() - 1
    out = torch.empty(n_items, len(ks), dtype=torch.float32, device=dev)
    if n_items > 0:
        _nat.check(_nat.lib().lr2_ndcg(scores.data_ptr(), gold.data_ptr(), offsets.data_ptr(), disc.data_ptr(), ks_t.data_ptr(),
                                       len(ks), out.data_ptr(), n_items, _stream()), "lr2_ndcg")
    return out


def patchify(img, out, *, B, Cc, H, W, ps):
    _chk_f32(img, out)
    _nat.check(_nat.lib().lr2_patchify(img.data_ptr(), out.data_ptr(), B, Cc, H, W, ps, _stream()), "lr2_patchify")
    return out


def vit_assemble(proj, cls, pos, out, *, B, P, D):
    _chk_f32(proj, cls, pos, out)
    _nat.check(_nat.lib().lr2_vit_assemble(proj.data_ptr(), cls.data_ptr(), pos.data_ptr(), out.data_ptr(), B, P, D, _stream()),
            "lr2_vit_assemble")
    return out
