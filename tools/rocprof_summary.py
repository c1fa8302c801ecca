#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output per DISPATCH SIGNATURE (kernel template + grid size + workgroup size), so that launches
of one kernel template that serve different bench signatures (e.g. the fused `gemm_TN_pp_M3072_N162816_K64_adamw` launch
and the other TN weight-gradient GEMMs) are reported separately, with count / avg / min / max duration.

    python tools/rocprof_summary.py kernel-trace <dir-or-csv> [--label-json labels.json] [--md out.md] [--json out.json]
    python tools/rocprof_summary.py pmc <dir-or-csv> [--md out.md] [--json out.json]

`kernel-trace`: reads *_kernel_trace.csv (rocprofv3 --kernel-trace --output-format csv).
`pmc`: reads *_counter_collection.csv (rocprofv3 --pmc ...), sums every counter per signature and also reports durations.
Kernel names are shortened (template arguments of torch kernels dropped) -- the raw names run to kilobytes."""
import argparse
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short_name(name: str) -> str:
    name = name.strip().strip('"')
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.replace("lr2gemm::", "").replace("g256::", "")
    m = re.match(r"^(at::native::[A-Za-z0-9_:]+)", name)
    if m:
        return m.group(1)[:80]
    i = name.find("(")            # drop the argument list
    if i > 0:
        name = name[:i]
    return name[:120]


# bench.py signatures whose launches can be told from the kernel template by their grid size (workgroups)
def gemm_label(short: str, grid_wg: int):
    """(label, note) for the dispatch signatures the bench's roofline object refers to."""
    if short.startswith("gemm_kernel<128, 128, 32, 64, 64, true, true, 3, true, true") and grid_wg == 24 * 1272:
        return "gemm_TN_pp_M3072_N162816_K64_adamw"       # 3072/128 x 162816/128 tiles, fused AdamW epilogue
    if short.startswith("gemm_kernel<64, 128, 64, 64, 32, false, false, 3, true, false") and grid_wg == 24 * 32:
        return "gemm_NT_pf_M64_N3072_K162816"             # out_layer.fc1 forward: 24 column tiles x 32 K splits
    if short.startswith("gemm_kernel<64, 128, 64, 64, 32, false, true, 3, true, false") and grid_wg == 1272:
        return "gemm_NN_pf_M64_N162816_K3072"             # out_layer.fc1 input gradient
    # the 256 x 256 NT kernel at M = 100 864 (ViT-B/16 over 512 frames): 394 tile rows x N / 256 tile columns
    if short.startswith("gemm256_nt_kernel") and grid_wg == 394 * 12:
        return "gemm_NT_pp_M100864_N3072_K768"            # FFN-1 (+ GELU): the value loop's dominant signature
    if short.startswith("gemm256_nt_kernel") and grid_wg == 394 * 9:
        return "gemm_NT_pp_M100864_N2304_K768"            # fused QKV projection
    if short.startswith("gemm256_nt_kernel") and grid_wg == 394 * 3:
        return "gemm_NT_pp_M100864_N768_K{768,3072}"      # output projection / FFN-2 (same grid)
    return None


def find_csv(path: str, suffix: str):
    if os.path.isfile(path):
        return [path]
    return sorted(glob.glob(os.path.join(path, "**", "*" + suffix), recursive=True))


def kernel_trace(paths):
    groups = defaultdict(list)
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
                    continue
                wg = int(row["Workgroup_Size_X"]) * int(row.get("Workgroup_Size_Y", 1)) * int(row.get("Workgroup_Size_Z", 1))
                grid = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y", 1)) * int(row.get("Grid_Size_Z", 1))
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                groups[(short_name(row["Kernel_Name"]), grid // max(wg, 1), wg)].append(dur)
    return groups


def pmc(paths):
    dur = defaultdict(dict)
    ctr = defaultdict(lambda: defaultdict(float))
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                wg = int(row["Workgroup_Size"])
                key = (short_name(row["Kernel_Name"]), int(row["Grid_Size"]) // max(wg, 1), wg)
                ctr[key][row["Counter_Name"]] += float(row["Counter_Value"])
                dur[key][row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return dur, ctr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["kernel-trace", "pmc"])
    ap.add_argument("path")
    ap.add_argument("--md")
    ap.add_argument("--json")
    ap.add_argument("--steps", type=float, default=0.0, help="steps inside the trace: adds a ms/step column")
    ap.add_argument("--title", default="")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    out_rows, lines = [], []
    if a.mode == "kernel-trace":
        groups = kernel_trace(find_csv(a.path, "kernel_trace.csv"))
        total = sum(sum(v) for v in groups.values())
        rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
        lines += [f"# {a.title}" if a.title else "# rocprofv3 kernel trace, per dispatch signature", "",
                  "| kernel (template) | workgroups x size | bench signature | calls | avg us | min us | max us | total ms"
                  + (" | ms / step" if a.steps else "") + " | % |", "|---|---|---|---|---|---|---|---|" + ("---|" if a.steps else "") + "---|"]
        for (name, grid, wg), d in rows[: a.top]:
            lab = gemm_label(name, grid) or ""
            rec = {"kernel": name, "workgroups": grid, "workgroup_size": wg, "bench_signature": lab or None, "calls": len(d),
                   "avg_us": sum(d) / len(d) / 1e3, "min_us": min(d) / 1e3, "max_us": max(d) / 1e3, "total_ms": sum(d) / 1e6}
            out_rows.append(rec)
            lines.append(f"| `{name}` | {grid} x {wg} | {lab} | {len(d)} | {rec['avg_us']:.1f} | {rec['min_us']:.1f} | {rec['max_us']:.1f} | "
                         f"{rec['total_ms']:.2f}" + (f" | {rec['total_ms'] / a.steps:.3f}" if a.steps else "") + f" | {100.0 * sum(d) / total:.1f} |")
        lines += ["", f"Total kernel time {total / 1e6:.2f} ms" + (f" = {total / 1e6 / a.steps:.2f} ms per step" if a.steps else "") + "."]
    else:
        dur, ctr = pmc(find_csv(a.path, "counter_collection.csv"))
        names = sorted({c for v in ctr.values() for c in v})
        rows = sorted(ctr.items(), key=lambda kv: -sum(dur[kv[0]].values()))
        lines += [f"# {a.title}" if a.title else "# rocprofv3 counters, per dispatch signature", "",
                  "| kernel (template) | workgroups x size | bench signature | launches | total ms | " + " | ".join(names) + " |",
                  "|---|---|---|---|---|" + "---|" * len(names)]
        for key, cv in rows[: a.top]:
            name, grid, wg = key
            lab = gemm_label(name, grid) or ""
            n = len(dur[key])
            rec = {"kernel": name, "workgroups": grid, "workgroup_size": wg, "bench_signature": lab or None, "launches": n,
                   "total_ms": sum(dur[key].values()) / 1e6, "counters_sum": dict(cv), "counters_per_launch": {k: v / n for k, v in cv.items()}}
            out_rows.append(rec)
            lines.append(f"| `{name}` | {grid} x {wg} | {lab} | {n} | {rec['total_ms']:.2f} | " + " | ".join(f"{cv.get(c, 0.0):.4g}" for c in names) + " |")
    text = "\n".join(lines) + "\n"
    if a.md:
        with open(a.md, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out_rows, f, indent=1)


if __name__ == "__main__":
    main()
