"""Build + load the gfx950 kernel library (C ABI declared in include/lr2ppo_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be bound,
importing any kernel raises.  The library is built in-tree (lr2ppo_amd/csrc/liblr2ppo_hip.so) so it
travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(CSRC, "liblr2ppo_hip.so")
SOURCES = ["gemm.hip", "gemm256.hip", "gemm256_mx.hip", "norm.hip", "attn.hip", "selfattn.hip", "selfattn_mx.hip", "misc.hip", "fp8.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"), os.path.join(CSRC, "fp8_common.h"), os.path.join(INCLUDE, "lr2ppo_hip.h")]

ABI_VERSION = 19     # == LR2_ABI_VERSION of include/lr2ppo_hip.h (tests assert the two agree)

_lock = threading.Lock()
_lib = None
_lib_override = None


def use_library(path: str):
    """Measurement scaffolding (tools/dbg/*_ab.py): load ANOTHER build of the same ABI for an A/B timing.  An explicit call made by
    the tool itself before the first kernel call -- the product loader reads no environment variable and loads the in-tree library."""
    global _lib_override
    if _lib is not None:
        raise RuntimeError("lr2ppo_amd: the kernel library is already loaded")
    _lib_override = path


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library (hipcc cross-compiles without a GPU)."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, "-c",
               os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode(errors='replace')}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB_PATH


class Epilogue(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("resid", C.c_void_p), ("aux_z", C.c_void_p), ("out", C.c_void_p),
                ("out_z", C.c_void_p), ("out_hi", C.c_void_p), ("out_lo_off", C.c_uint64), ("ld_planes", C.c_int32),
                ("ld_resid", C.c_int32), ("ld_aux", C.c_int32), ("ld_out", C.c_int32),
                ("ld_z", C.c_int32), ("act", C.c_int32), ("accumulate", C.c_int32), ("alpha", C.c_float),
                ("drop_p", C.c_float), ("drop_site", C.c_uint32), ("_pad", C.c_uint32), ("drop_seed", C.c_uint64),
                ("adam_p", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_lr", C.c_double),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("adam_weight_decay", C.c_double), ("colsum", C.c_void_p), ("colsum_ws", C.c_void_p), ("drop_seed_dev", C.c_void_p), ("adam_lr_dev", C.c_void_p)]


class SplitChunk(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst_hi", C.c_void_p), ("lo_off", C.c_uint64), ("count", C.c_uint64)]


class AdamChunk(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("count", C.c_uint64),
                ("weight_decay", C.c_float), ("_pad", C.c_float)]


_P, _I, _F, _U64, _U32 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32

# name -> argtypes; every symbol include/lr2ppo_hip.h declares must appear here (tests check both ways)
SIGNATURES = {
    "lr2_abi_version": [],
    "lr2_device_info": [C.c_char_p, _I],
    "lr2_gemm": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _U64, _U64, _I, _U64, _I, _U64, C.POINTER(Epilogue), _P, _I, _I, _I, _P],
    "lr2_gemm_row_split_plan": [_I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "lr2_gemm_launch_counts": [C.POINTER(C.c_uint64)],
    "lr2_gather_rows": [_P, _P, _P, _I, _I, _I, _U64, _U64, _U64, _P],
    "lr2_gather_rows_bwd": [_P, _P, _P, _I, _I, _I, _U64, _P],
    "lr2_copy_rows": [_P, _P, _I, _U64, _I, _I, _I, _U64, _U64, _P],
    "lr2_split_planes": [_P, _P, _U64, _U64, _P],
    "lr2_dropout_planes": [_P, _P, _U64, _U64, _F, _U64, _U32, _P],
    "lr2_dropout_apply": [_P, _P, _U64, _F, _U64, _U32, _P],
    "lr2_text_embed_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, C.c_int64, _I, _P],
    "lr2_split_planes_t": [_P, _P, _U64, _I, _I, _P],
    "lr2_split_planes_multi": [_P, _I, _P],
    "lr2_layernorm_fwd": [_P, _P, _P, _P, _P, _U64, _P, _P, _I, _I, _F, _I, _I, _U64, _P],
    "lr2_layernorm_bwd": [_P, _I, _U64, _P, _P, _P, _P, _P, _P, _P, _U64, _F, _U64, _U32, _P, _P, _I, _I, _I, _I, _F, _P],
    "lr2_colsum_partials_finish": [_P, _I, _I, _I, _P, _I, _P],
    "lr2_colsum": [_P, _I, _U64, _I, _I, _I, _P, _I, _P, _P],
    "lr2_xattn_fwd": [_P, _P, _P, _P, _I, _U64, _I, _I, _I, _I, _I, _F, _P],
    "lr2_xattn_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _U64, _U64, _I, _I, _I, _I, _I, _F, _P],
    "lr2_self_attn_fwd": [_P, _P, _P, _U64, _I, _P, _P, _P, _U64, _I, _P, _F, _U64, _U32, _I, _I, _I, _I, _F, _P],
    "lr2_first_token_attn": [_P, _I, _P, _P, _U64, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "lr2_self_attn_bwd": [_P, _P, _P, _U64, _I, _P, _U64, _I, _P, _P, _P, _P, _U64, _I, _P, _U64, _I, _P, _P, _F, _U64, _U32, _I,
                          _I, _I, _I, _F, _P],
    "lr2_self_attn_plan": [_I, _I, _I, _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "lr2_head_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lr2_head_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "lr2_add_period_rows": [_P, _P, _P, _I, _I, _I, _P],
    "lr2_period_rows_grad": [_P, _P, _I, _I, _I, _P],
    "lr2_ppo_loss": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P],
    "lr2_smooth_l1": [_P, _P, _I, _F, _P, _P, _P],
    "lr2_cls_head_fwd": [_P, _P, _P, _P, _I, _I, _I, _P],
    "lr2_cls_head_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "lr2_cls_scores": [_P, _P, _P, _I, _I, _I, _P],
    "lr2_cls_scores_bwd": [_P, _P, _P, _P, _I, _I, _P],
    "lr2_nll_loss": [_P, _P, _I, _I, _P, _P, _P],
    "lr2_pair_hinge": [_P, _I, _F, _P, _P, _P],
    "lr2_adamw_multi": [_P, _I, C.c_double, C.c_double, C.c_double, C.c_double, _P, _P],
    "lr2_step_scalars_store": [_P, _U64, _P, _I, _P],
    "lr2_quant_mxfp8": [_P, _I, _P, _P, _I, _I, _P],
    "lr2_layernorm_fwd_mxfp8": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P],
    "lr2_gemm_mxfp8": [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _P, _P, _U64, _I, _I, _I, _I, _P],
    "lr2_self_attn_fwd_bf16": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "lr2_text_embed": [_P, _P, _P, _P, _P, _P, _I, _I, _I, C.c_int64, _I, _P, _P],
    "lr2_patchify_planes": [_P, _I, _P, _U64, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P],
    "lr2_ndcg": [_P, _P, _P, _P, _P, _I, _P, _I, _P],
    "lr2_patchify": [_P, _P, _I, _I, _I, _I, _I, _P],
    "lr2_vit_assemble": [_P, _P, _P, _P, _I, _I, _I, _P],
}


def lib() -> C.CDLL:
    """The loaded kernel library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"lr2ppo_amd: native library {LIB_PATH} is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64 (SONAME libamdhip64.so.7).  Import torch FIRST so that the loader
        # resolves this library's libamdhip64.so.7 to that already-loaded runtime; loaded in the other order the
        # process would hold two HIP/HSA runtimes and the second one finds "no ROCm-capable device".
        import torch  # noqa: F401
        try:
            handle = C.CDLL(_lib_override or LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 not loadable
            raise RuntimeError(f"lr2ppo_amd: cannot load {LIB_PATH}: {e}") from e
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError(f"lr2ppo_amd: {LIB_PATH} does not export {name}") from e
            fn.argtypes = argtypes
            fn.restype = C.c_int
        if handle.lr2_abi_version() != ABI_VERSION:
            raise RuntimeError(f"lr2ppo_amd: ABI version mismatch: {LIB_PATH} reports {handle.lr2_abi_version()}, the python "
                               f"package needs {ABI_VERSION}; rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
        if _stale():
            # a library older than its sources with an unchanged ABI number would otherwise load silently
            import warnings
            warnings.warn(f"lr2ppo_amd: {LIB_PATH} is older than its sources; rebuild with __graft_entry__.build()", RuntimeWarning)
        _lib = handle
        return _lib


class NativeError(RuntimeError):
    pass


_ERR = {-1: "LR2_ERR_ARG (bad argument)", -2: "LR2_ERR_SHAPE (unsupported shape)", -3: "LR2_ERR_LAUNCH (HIP launch failed)"}


def check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what} failed: {_ERR.get(rc, rc)}")
