#!/usr/bin/env python3
"""The measured table of DESIGN.md section 4.1 from the PMC summaries of one round.

    python3 tools/pmc_table.py r05            # reads profiles/r05_pmc_*.json

One row per summary: kernel, ms per launch (unprofiled), VALU instructions per launch, SIMD cycles per VALU
instruction at 2.4 GHz x 1024 SIMDs, parked share (SQ_WAIT_ANY / SQ_WAVE_CYCLES), VALU-active share,
LDS bank-conflict share, HBM traffic per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB counters).
Instructions per cell (pair) are printed where the summary knows the cells of a launch.
"""
import glob
import json
import os
import sys


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    rows = []
    for path in sorted(glob.glob(os.path.join(root, f"{tag}_pmc_*.json"))):
        d = json.load(open(path))
        c = {k: v["mean_per_launch"] for k, v in d["counters"].items()}
        ms = d.get("kernel_ms_unprofiled")
        valu = c.get("SQ_INSTS_VALU")
        cyc = ms * 1e-3 * 2.4e9 * 1024 / valu if ms and valu else None
        fr = d.get("fractions_of_wave_cycles", {})
        name = d["kernel"].replace("void ", "").replace("miopal::", "").replace("(anonymous namespace)::", "")
        name = name.split("(")[0]
        rows.append((os.path.basename(path)[len(tag) + 5:-5], name, ms, valu, d.get("valu_instr_per_lane_cell_pair"), cyc,
                     fr.get("SQ_WAIT_ANY"), fr.get("SQ_ACTIVE_INST_VALU"), d.get("lds_bank_conflict_fraction"),
                     d.get("hbm_traffic_bytes_per_launch"), c.get("SQ_WAVES")))
    f = lambda v, p: "-" if v is None else format(v, p)
    print("| summary | kernel | ms | VALU instr / launch | per cell pair | cycles / instr | parked | VALU active | waves | traffic GB |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r[0]} | `{r[1]}` | {f(r[2], '.3f')} | {f(r[3], '.3e')} | {f(r[4], '.2f')} | {f(r[5], '.2f')} | "
              f"{f(r[6], '.2f')} | {f(r[7], '.2f')} | {f(r[10], '.0f')} | {f(r[9] / 1e9 if r[9] else None, '.3f')} |")


if __name__ == "__main__":
    main()
