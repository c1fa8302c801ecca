"""A/B of two stand-alone builds of csrc/selfattn.hip (tools/dbg/micro/build/libsa_<name>.so): us per call of lr2_self_attn_bwd (given o + lse,
dropout 0.1) and lr2_self_attn_fwd at the training shape, alternating between the libraries inside one process.
    python tools/dbg/attn_lib_ab.py u1 u2"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from lr2ppo_amd import _native, ops  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith("-")]
dev = torch.device("cuda:0")
batch, heads, L = 512, 12, 197
E = heads * 64
g = torch.Generator(device=dev).manual_seed(0)
qkv = ops.split_planes(torch.randn(batch * L, 3 * E, device=dev, generator=g) * 0.5, ops.Planes.empty(batch * L, 3 * E, dev))
seg = torch.ones(batch * L, dtype=torch.int64, device=dev)
o, do, dqkv = ops.Planes.empty(batch * L, E, dev), ops.Planes.empty(batch * L, E, dev), ops.Planes.empty(batch * L, 3 * E, dev)
ops.split_planes(torch.randn(batch * L, E, device=dev, generator=g), do)
lse, dsum = torch.zeros(batch * heads * L, device=dev), torch.zeros(batch * heads * L, device=dev)
ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse, drop=ops.Drop(0.1, 7, 1))
q, k, v = ops._qkv_ptrs(qkv, E)
dq, dk, dv = ops._qkv_ptrs(dqkv, E)
libs = {}
for n in names:
    lib = C.CDLL(os.path.join(REPO, "tools", "dbg", "micro", "build", f"libsa_{n}.so"))
    lib.lr2_self_attn_bwd.argtypes, lib.lr2_self_attn_bwd.restype = _native.SIGNATURES["lr2_self_attn_bwd"], C.c_int
    lib.lr2_self_attn_fwd.argtypes, lib.lr2_self_attn_fwd.restype = _native.SIGNATURES["lr2_self_attn_fwd"], C.c_int
    libs[n] = lib
st = lambda: torch.cuda.current_stream().cuda_stream      # noqa: E731
res = {}
for rep in range(3):
    for n, lib in libs.items():
        calls = {
            "bwd": lambda: lib.lr2_self_attn_bwd(q, k, v, qkv.lo_off, qkv.cols, do.data_ptr(), do.lo_off, do.cols, seg.data_ptr(), dq, dk, dv,
                                                 dqkv.lo_off, dqkv.cols, o.data_ptr(), o.lo_off, o.cols, lse.data_ptr(), dsum.data_ptr(), 0.1, 7, 1,
                                                 batch, heads, L, 64, 0.125, st()),
            "fwd": lambda: lib.lr2_self_attn_fwd(q, k, v, qkv.lo_off, qkv.cols, seg.data_ptr(), None, o.data_ptr(), o.lo_off, E, None, 0.0, 0, 0,
                                                 batch, heads, L, 64, 0.125, st()),
            "fwd, dropout 0.1 + lse": lambda: lib.lr2_self_attn_fwd(q, k, v, qkv.lo_off, qkv.cols, seg.data_ptr(), None, o.data_ptr(), o.lo_off, E,
                                                                    lse.data_ptr(), 0.1, 7, 1, batch, heads, L, 64, 0.125, st()),
        }
        for what, call in calls.items():
            for _ in range(2):
                assert call() == 0
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                call()
            e.record()
            torch.cuda.synchronize()
            res.setdefault((n, what), []).append(s.elapsed_time(e) / 8 * 1e3)
for (n, what), ts in sorted(res.items()):
    print(f"{n:>6} {what}: {min(ts):8.1f} us", flush=True)
