#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two PMC summaries of tools/profile_round.sh (FETCH_SIZE pass + WRITE_SIZE pass):
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (counters are KiB; on gfx950 FETCH_SIZE reports half of the bytes of a
16-B-per-lane streaming read, MI355X_MICROARCH.md 'HBM'), keyed by bench.py's kernel signature where the dispatch
signature (kernel template + grid) identifies it, with the kernel-trace average duration of the same box next to it.
    python tools/make_traffic_json.py gpurun_out/prof_r02 r02 > profiles/pmc_traffic.json"""
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
f = json.load(open(os.path.join(d, f"{tag}_bench_pmc_fetch.json")))
w = json.load(open(os.path.join(d, f"{tag}_bench_pmc_write.json")))
# exclusive per-kernel durations: the --serial-streams trace when the round has one (the default schedule overlaps two streams)
_kt = os.path.join(d, f"{tag}_bench_kernel_trace_serial.json")
kt = json.load(open(_kt if os.path.exists(_kt) else os.path.join(d, f"{tag}_bench_kernel_trace.json")))
key = lambda r: (r["kernel"], r["workgroups"], r["workgroup_size"])  # noqa: E731
W, K = {key(r): r for r in w}, {key(r): r for r in kt}
by_label, by_sig = {}, {}
for r in f:
    k = key(r)
    if k not in W:
        continue
    fetch = r["counters_per_launch"].get("FETCH_SIZE", 0.0) * 1024
    write = W[k]["counters_per_launch"].get("WRITE_SIZE", 0.0) * 1024
    rec = {"read_bytes": int(2 * fetch), "write_bytes": int(write), "hbm_bytes": int(2 * fetch + write), "launches": r["launches"],
           "kernel_trace_avg_us": round(K[k]["avg_us"], 1) if k in K else None,
           "kernel_trace_min_us": round(K[k]["min_us"], 1) if k in K else None,
           "kernel_trace_max_us": round(K[k]["max_us"], 1) if k in K else None,
           "kernel_trace_calls": K[k]["calls"] if k in K else None}
    if r["bench_signature"]:
        by_label[r["bench_signature"]] = rec
    by_sig[f"{r['kernel']} [{r['workgroups']} x {r['workgroup_size']}]"] = rec
# MFMA-busy of the value loop's GEMM signatures from the encoder PMC pass of the same call (SQ_VALU_MFMA_BUSY_CYCLES over
# GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs; tools/mfma_busy_table.py prints the same numbers as a table)
_mf = os.path.join(d, f"{tag}_encoder_pmc_mfma.json")
if os.path.exists(_mf):
    for r in json.load(open(_mf)):
        if not r.get("bench_signature"):
            continue
        c = r["counters_sum"]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui <= 0:
            continue
        rec = by_label.setdefault(r["bench_signature"], {})
        rec["mfma_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0), 4)
        rec["effective_clock_ghz"] = round(gui / (r["total_ms"] * 1e-3) / 1e9, 3)
        rec["pmc_launches"] = r["launches"]
# HBM bytes of the value loop's GEMM signatures: FETCH_SIZE / WRITE_SIZE passes over the dual-encoder forward (same correction)
_ef, _ew = os.path.join(d, f"{tag}_encoder_pmc_fetch.json"), os.path.join(d, f"{tag}_encoder_pmc_write.json")
if os.path.exists(_ef) and os.path.exists(_ew):
    EW = {key(r): r for r in json.load(open(_ew))}
    for r in json.load(open(_ef)):
        if not r.get("bench_signature") or key(r) not in EW:
            continue
        fetch = r["counters_per_launch"].get("FETCH_SIZE", 0.0) * 1024
        write = EW[key(r)]["counters_per_launch"].get("WRITE_SIZE", 0.0) * 1024
        rec = by_label.setdefault(r["bench_signature"], {})
        rec.update(read_bytes=int(2 * fetch), write_bytes=int(write), hbm_bytes=int(2 * fetch + write), traffic_launches=r["launches"])
print(json.dumps({
    "_method": "tools/profile_round.sh " + tag + ": rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) on "
               "`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-online --no-profile` (head-only PPO steps), summarised per dispatch signature by "
               "tools/rocprof_summary.py; counters are KiB; read bytes = 2 x FETCH_SIZE (gfx950, 16-B/lane streaming reads), WRITE_SIZE as "
               "is.  kernel_trace_* come from the `--serial-streams` kernel-trace pass of the SAME gpurun call (same box, 13 PPO steps on one "
               "HIP stream: exclusive per-launch durations).  mfma_busy / effective_clock_ghz: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 "
               "x 1024 SIMDs) and (GRBM_GUI_ACTIVE / 8) / kernel time from the encoder PMC pass of the same call "
               "(`tools/encoder_bench.py --ppo-shapes --iters 1` under --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES); "
               "read_bytes / write_bytes / hbm_bytes of the encoder's GEMM signatures: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over the "
               "same encoder command.",
    "by_bench_label": by_label, "by_dispatch_signature": by_sig}, indent=1))
