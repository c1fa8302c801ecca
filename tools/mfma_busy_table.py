#!/usr/bin/env python3
"""MFMA-busy and effective-clock table from the per-signature counter sums of tools/profile_round.sh's encoder PMC pass.
    python tools/mfma_busy_table.py profiles/r02_encoder_pmc_mfma.json > table.md
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) (GRBM_GUI_ACTIVE is summed over the 8 XCDs);
effective clock = (GRBM_GUI_ACTIVE / 8) / kernel time."""
import json
import sys

rows = json.load(open(sys.argv[1]))
print("| kernel | workgroups x size | launches | total ms | MFMA busy | effective clock |")
print("|---|---|---|---|---|---|")
tot_busy = tot_cyc = tot_ms = 0.0
for r in rows:
    c = r["counters_sum"]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    tot_busy += busy
    tot_cyc += gui
    tot_ms += r["total_ms"]
    if r["total_ms"] < 0.4:
        continue
    print(f"| `{r['kernel']}` | {r['workgroups']} x {r['workgroup_size']} | {r['launches']} | {r['total_ms']:.2f} | "
          f"{100.0 * busy / max(gui * 1024, 1):.1f} % | {gui / (r['total_ms'] * 1e-3) / 1e9:.2f} GHz |")
print()
print(f"Whole run (every kernel, {tot_ms:.1f} ms of kernel time): **{100.0 * tot_busy / (tot_cyc * 1024):.1f} % MFMA busy**.")
