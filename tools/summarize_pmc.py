"""Condense rocprofv3 --pmc CSVs (separate passes) into one JSON under profiles/.

usage: summarize_pmc.py OUT.json KERNEL_SUBSTRING KERNEL_MS DIR [DIR ...]
"""
import collections
import csv
import glob
import json
import os
import sys

out_path, needle, kernel_ms = sys.argv[1], sys.argv[2], float(sys.argv[3])
counters = {}
name = None
for d in sys.argv[4:]:
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(path)):
            if needle in row["Kernel_Name"]:
                name = row["Kernel_Name"]
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
import hashlib
lib_path = os.environ.get("MIOPAL_LIBRARY") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "pyopal_amd", "libmiopal.so")
summary = {
    # bench.py only quotes these counters for the build they were measured on
    "library_sha256": hashlib.sha256(open(lib_path, "rb").read()).hexdigest(),
    "command": os.environ.get(
        "PMC_COMMAND",
        "rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py --steps 5 "
        "--warmup 1 --no-cpu-baseline (separate passes: SQ+GRBM, FETCH_SIZE, WRITE_SIZE)"),
    "kernel": name,
    "kernel_ms_unprofiled": kernel_ms,
    "counters": counters,
}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    f, w = counters["FETCH_SIZE"]["mean_per_launch"], counters["WRITE_SIZE"]["mean_per_launch"]
    summary["hbm_traffic_bytes_per_launch"] = (2 * f + w) * 1024
    summary["traffic_note"] = ("FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md "
                               "(gfx950 reports half the bytes of a wide coalesced streaming read)")
if "GRBM_GUI_ACTIVE" in counters:
    summary["effective_clock_ghz"] = counters["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8 / (kernel_ms * 1e-3) / 1e9
if "SQ_INSTS_VALU" in counters:
    summary["simd_cycles_per_valu_instr_at_2.4GHz"] = kernel_ms * 1e-3 * 2.4e9 * 1024 / counters["SQ_INSTS_VALU"]["mean_per_launch"]
cells = float(os.environ.get("PMC_CELLS") or 0)
if cells and "SQ_INSTS_VALU" in counters:
    # one wave instruction advances 64 lanes x 2 packed targets
    summary["dp_cells_per_search"] = cells
    summary["valu_instr_per_lane_cell_pair"] = counters["SQ_INSTS_VALU"]["mean_per_launch"] * 128 / cells
    summary["simd_cycles_per_lane_cell_pair_at_2.4GHz"] = kernel_ms * 1e-3 * 2.4e9 * 1024 * 128 / cells
    summary["cells_note"] = ("per launch of the named kernel when one launch covers the whole search "
                             "(side-stream and probe launches excluded by the kernel name)")
if "SQ_WAVE_CYCLES" in counters:
    wc = counters["SQ_WAVE_CYCLES"]["mean_per_launch"]
    summary["fractions_of_wave_cycles"] = {
        k: counters[k]["mean_per_launch"] / wc
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS") if k in counters}
if "SQ_LDS_IDX_ACTIVE" in counters and "SQ_LDS_BANK_CONFLICT" in counters:
    summary["lds_bank_conflict_fraction"] = (counters["SQ_LDS_BANK_CONFLICT"]["mean_per_launch"] /
                                             max(counters["SQ_LDS_IDX_ACTIVE"]["mean_per_launch"], 1.0))
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "counters"}, indent=1))
print({k: round(v["mean_per_launch"]) for k, v in counters.items()})
