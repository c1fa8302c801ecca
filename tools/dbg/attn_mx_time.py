"""us per launch of lr2_self_attn_fwd_bf16 (the MX-FP8 mode's attention) at ViT-L/14's shape (512 x 16 heads x 257 tokens, MX-FP8 output);
LR2_ATTN_PERSIST=0 in the environment times the one-pair kernel.    python tools/dbg/attn_mx_time.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from lr2ppo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (batch, heads, L) in ((512, 16, 257), (512, 12, 197)):
    E = heads * 64
    qkv = (torch.randn(batch * L, 3 * E, device=dev) * 0.6).to(torch.bfloat16)
    seg = torch.ones(batch * L, dtype=torch.int64, device=dev)
    mx = ops.Mx8.empty(batch * L, E, dev)
    ts = []
    for rep in range(3):
        for _ in range(3):
            ops.self_attn_fwd_bf16(qkv, seg, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out_mx=mx)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.self_attn_fwd_bf16(qkv, seg, batch=batch, heads=heads, L=L, head_dim=64, scale=0.125, out_mx=mx)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 20 * 1e3)
    print(f"batch {batch} heads {heads} L {L}: {min(ts):7.1f} us per launch (LR2_ATTN_PERSIST={os.environ.get('LR2_ATTN_PERSIST', '1')})", flush=True)
