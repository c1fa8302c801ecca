"""Where the encoder attention forward's time goes, by ablation: lr2ppo_amd/csrc/selfattn.hip is compiled once per LR2_SA_ABLATE
setting into tools/dbg/micro/build/libsa_<k>.so (stand-alone: the file needs nothing from the other sources) and
lr2_self_attn_fwd is timed at the `value` loop's shape (512 sequences x 12 heads x 197 tokens, planes output).
    python tools/dbg/attn_ablate.py --build      (here: hipcc cross-compiles)       python tools/dbg/attn_ablate.py     (GPU box)
Bits: 1 no K / V global loads, 2 no sub-tile work (staging only), 4 no softmax arithmetic, 8 no lo split of P, 16 no P V product,
32 no S product, 64 no output store.  Ablated builds give wrong results by design."""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tools", "dbg", "micro", "build")
BWD = "--bwd" in sys.argv      # the persistent backward: bits 1 (no LDS-DMA), 64 (no stores), 128 (no fragment loads of the compute waves)
SETTINGS = [0, 1, 64, 128, 129, 193] if BWD else [0, 1, 2, 3, 4, 8, 12, 16, 32, 48, 64, 65, 4 | 8 | 64, 1 | 4 | 8 | 64]
EXTRA = [a for a in sys.argv[1:] if a.startswith("-D")]


def build():
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for k in SETTINGS:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DLR2_SA_ABLATE={k}", *EXTRA, "-I",
               os.path.join(REPO, "include"), "-I", os.path.join(REPO, "lr2ppo_amd", "csrc"),
               os.path.join(REPO, "lr2ppo_amd", "csrc", "selfattn.hip"), "-o", os.path.join(OUT, f"libsa_{k}.so")]
        procs.append(subprocess.Popen(cmd))
        if len(procs) >= 4:
            for p in procs:
                assert p.wait() == 0
            procs = []
    for p in procs:
        assert p.wait() == 0
    print("built", len(SETTINGS), "variants in", OUT)


def main():
    import torch
    from lr2ppo_amd import _native, ops
    dev = torch.device("cuda:0")
    batch, heads, L = (64, 12, 196) if "--text" in sys.argv else (512, 12, 197)
    E = heads * 64
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = ops.split_planes(torch.randn(batch * L, 3 * E, device=dev, generator=g) * 0.5, ops.Planes.empty(batch * L, 3 * E, dev))
    seg = torch.ones(batch * L, dtype=torch.int64, device=dev)
    o = ops.Planes.empty(batch * L, E, dev)
    q, k, v = ops._qkv_ptrs(qkv, E)
    sig = _native.SIGNATURES["lr2_self_attn_fwd"]
    res = {}
    if BWD:
        do = ops.split_planes(torch.randn(batch * L, E, device=dev, generator=g), ops.Planes.empty(batch * L, E, dev))
        dqkv = ops.Planes.empty(batch * L, 3 * E, dev)
        lse, dsum = torch.zeros(batch * heads * L, device=dev), torch.zeros(batch * heads * L, device=dev)
        ops.self_attn_fwd(qkv, seg, o, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, lse=lse)
        dq, dk, dv = ops._qkv_ptrs(dqkv, E)
        sigb = _native.SIGNATURES["lr2_self_attn_bwd"]
        p_drop = 0.1 if "--drop" in sys.argv else 0.0
        for rep in range(2):
            for kset in SETTINGS:
                lib = C.CDLL(os.path.join(OUT, f"libsa_{kset}.so"))
                fn = lib.lr2_self_attn_bwd
                fn.argtypes, fn.restype = sigb, C.c_int
                call = lambda: fn(q, k, v, qkv.lo_off, qkv.cols, do.data_ptr(), do.lo_off, do.cols, seg.data_ptr(), dq, dk, dv,  # noqa: E731
                                  dqkv.lo_off, dqkv.cols, o.data_ptr(), o.lo_off, o.cols, lse.data_ptr(), dsum.data_ptr(), p_drop, 7, 1,
                                  batch, heads, L, 64, 0.125, torch.cuda.current_stream().cuda_stream)
                for _ in range(2):
                    assert call() == 0
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(6):
                    call()
                e.record()
                torch.cuda.synchronize()
                res.setdefault(kset, []).append(s.elapsed_time(e) / 6 * 1e3)
        for kset in SETTINGS:
            print(f"backward (dQ + dK/dV kernels), dropout {p_drop}, ablate {kset:3d}: {min(res[kset]):7.1f} us", flush=True)
        return
    for rep in range(2):
        for kset in SETTINGS:
            lib = C.CDLL(os.path.join(OUT, f"libsa_{kset}.so"))
            fn = lib.lr2_self_attn_fwd
            fn.argtypes, fn.restype = sig, C.c_int
            call = lambda: fn(q, k, v, qkv.lo_off, qkv.cols, seg.data_ptr(), None, o.data_ptr(), o.lo_off, E, None, 0.0, 0, 0,  # noqa: E731
                              batch, heads, L, 64, 0.125, torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                assert call() == 0
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                call()
            e.record()
            torch.cuda.synchronize()
            res.setdefault(kset, []).append(s.elapsed_time(e) / 10 * 1e3)
    for kset in SETTINGS:
        print(f"ablate {kset:3d}: {min(res[kset]):7.1f} us", flush=True)


if __name__ == "__main__":
    build() if "--build" in sys.argv else main()
