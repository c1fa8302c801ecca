#!/usr/bin/env python3
"""Can HBM-bound work share the chip with the encoders' 256 x 256 GEMMs?  (the fused out_layer.fc1 update of step k has
no consumer until step k + 1's rollout, and the dual-encoder forward of step k + 1 leaves HBM idle for 55 ms.)
A 12-GB read-modify-write stream (tools/dbg/micro/coresident_probe.hip: no LDS, <= 40 VGPRs -- small enough to be resident beside an
8-wave GEMM workgroup) on a side stream, started right behind the first encoder launches: time of the extraction alone, of the
stream alone, and of both together.  usage: python tools/dbg/coresident_probe.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lr2ppo_amd.finetune.features import FeatureExtractor, synthetic_raw_batch  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "dbg", "micro", "libcoresident_probe.so"))
lib.probe_rmw_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p]


def main():
    torch.manual_seed(0)
    fx = FeatureExtractor()
    fx.init_normal()
    fx = fx.to(dev).eval()
    frames, ids, seg, _ = synthetic_raw_batch(32, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    n = 500_170_752                                       # out_layer.fc1.weight
    p, m, v = (torch.rand(n, device=dev) for _ in range(3))
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def stream_job(blocks):
        lib.probe_rmw_stream(p.data_ptr(), m.data_ptr(), v.data_ptr(), n, blocks, side.cuda_stream)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    reps = int(os.environ.get("PROBE_REPS", "5"))            # 5 x 12 GB behind each other: ~11 ms of HBM-bound work, started when the SECOND of two extractions starts
    for blocks in [int(b) for b in os.environ.get("PROBE_BLOCKS", "256,1024,4096").split(",")]:
        res = {}
        for mode in ("extract alone", "stream alone", "both"):
            ts = []
            for it in range(4):
                torch.cuda.synchronize()
                a0, a1, b0, b1, gate = ev(), ev(), ev(), ev(), ev()
                if mode != "stream alone":
                    fx.extract(frames, ids, seg, check_ids=False)          # extraction 1: fills the queue
                gate.record(main_s)
                if mode != "stream alone":
                    a0.record(main_s)
                    fx.extract(frames, ids, seg, check_ids=False)          # extraction 2: the timed one
                    a1.record(main_s)
                if mode != "extract alone":
                    side.wait_event(gate)
                    b0.record(side)
                    for _ in range(reps):
                        stream_job(blocks)
                    b1.record(side)
                torch.cuda.synchronize()
                ts.append((a0.elapsed_time(a1) if mode != "stream alone" else 0.0, b0.elapsed_time(b1) if mode != "extract alone" else 0.0))
            res[mode] = min(ts[1:], key=lambda t: t[0] + t[1])
        e0, s0, (e1, s1) = res["extract alone"][0], res["stream alone"][1], res["both"]
        print(f"blocks {blocks:5d}: extract alone {e0:7.2f} ms | {reps} streams alone {s0:6.2f} ms ({reps * 24.0 * n / s0 / 1e9:5.2f} TB/s) | together: "
              f"extract {e1:7.2f} ms (+{e1 - e0:5.2f}), streams {s1:6.2f} ms -> {e0 + s0 - max(e1, s1):5.2f} ms of {s0:5.2f} hidden", flush=True)


if __name__ == "__main__":
    main()
