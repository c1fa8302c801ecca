"""us per call of lr2_layernorm_bwd at the encoder training shape ([100864, 768], TencentPretrain LayerNorm, residual gradient added,
fp32 + dropout-masked planes output) for a few grid sizes, and a hash of the outputs (a scheduling change must not move a bit).
    python tools/dbg/ln_bwd_time.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import _native, ops  # noqa: E402

if "--lib" in sys.argv:                      # another build of the same ABI (an A/B of the kernel's bits and speed)
    _native.use_library(sys.argv[sys.argv.index("--lib") + 1])

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, D = (12544 if "--text" in sys.argv else 100864), 768
x, dy, rg = (torch.randn(M, D, device=dev, generator=g) for _ in range(3))
gamma = torch.randn(D, device=dev, generator=g)
mean, std = x.mean(1), x.std(1)
rstd = 1.0 / (std + 1e-6)
dx, dxp = torch.empty(M, D, device=dev), ops.Planes.empty(M, D, dev)
gb = torch.empty(2 * D, device=dev)
dr = ops.Drop(0.1, 5, 2)
for nb in (256, 384, 512, 768, 1024, 2048):
    part = torch.empty(nb * 2 * D, device=dev)
    fn = lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, part, gb[:D], gb[D:], rows=M, D=D, resid_grad=rg, dx_planes=dxp, drop=dr,  # noqa: E731
                                   nblocks=nb, mode=1, eps=1e-6)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    h = hashlib.sha1(dx.cpu().numpy().tobytes() + dxp.buf.cpu().numpy().tobytes()).hexdigest()[:12]
    hg = hashlib.sha1(gb.cpu().numpy().tobytes()).hexdigest()[:12]
    t = s.elapsed_time(e) / 20 * 1e3
    print(f"blocks {nb:5d}: {t:7.1f} us per call ({20.0 * M * D / t / 1e6:.2f} TB/s of x, dy, residual gradient in, dx fp32 + planes out)  dx hash {h}  dgamma|dbeta hash {hg}", flush=True)
