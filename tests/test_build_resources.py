"""The compiler's resource remarks of the last in-tree build (pyopal_amd/csrc/*.rpt, written by
the Makefile): the lane-per-target kernels keep their whole DP column in registers, so a spill
(a change that pushes one instantiation over the budget of its occupancy) costs a factor, not
per cents - it must fail here, not show up as a slow bench."""
import glob
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pyopal_amd", "csrc")


def kernels():
    out = {}
    for path in glob.glob(os.path.join(CSRC, "*.rpt")):
        name = None
        for line in open(path, errors="replace"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                out[name] = {"file": os.path.basename(path)}
                continue
            m = re.search(r"remark:\s+(VGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|ScratchSize \[bytes/lane\]): (\d+)", line)
            if m and name:
                out[name][m.group(1)] = int(m.group(2))
    return out


def test_hot_kernels_do_not_spill():
    ks = kernels()
    if not ks:
        pytest.skip("no resource remarks: build with make -C pyopal_amd/csrc")
    hot = {n: k for n, k in ks.items() if "interseq" in n or "perpair" in n}
    assert len(hot) >= 40, sorted(hot)
    bad = {n: k for n, k in hot.items() if k.get("VGPRs Spill", 0) or k.get("ScratchSize [bytes/lane]", 0)}
    # No vector register may spill. Scalars (kernel arguments kept across the unit loop of the general
    # kernel) parked through a VGPR show up as a few bytes of scratch without a vector spill: allowed.
    for name in list(bad):
        if "interseq_kernel" in name and not bad[name].get("VGPRs Spill", 0) and bad[name]["ScratchSize [bytes/lane]"] <= 64:
            del bad[name]
    # The strips kernels (units of (batch, strip) taken in a loop around the sweep) park a few per-unit
    # values - boundary-row pointers, lane offsets - in scratch around the sweep at their tallest strips:
    # a handful of instructions per UNIT, allowed up to a bound; that none of them sits inside a column
    # loop is checked on the code itself by test_spills_of_the_strips_kernels_stay_out_of_the_column_loops.
    for name in list(bad):
        if "strips_kernel" in name and bad[name].get("VGPRs Spill", 0) <= 24 and bad[name]["ScratchSize [bytes/lane]"] <= 128:
            del bad[name]
    # The kernels with two pairs per lane (perpair_packed.hip, round 5) hold 64 rows x {H, E} of both pairs plus the
    # flags' accumulators: a few per-strip values - pointers, lengths - sit in scratch around the column loop, not
    # inside it (checked on the code below)
    for name in list(bad):
        if "perpair_packed" in name and bad[name].get("VGPRs Spill", 0) <= 24 and bad[name]["ScratchSize [bytes/lane]"] <= 128:
            del bad[name]
    # ... and the OV instantiation of the strips scan (the pair's own last row picked by a select tree: its two words of
    # rows and the tree's temporaries beside 64 rows x {H, E}) a few more, outside the column loops as well
    for name in list(bad):
        if "perpair_packed_scan_strips_kernelILb" in name and "ELb1EEE" in name and bad[name].get("VGPRs Spill", 0) <= 40 \
                and bad[name]["ScratchSize [bytes/lane]"] <= 192:
            del bad[name]
    assert not bad, bad


def test_spills_of_the_packed_pair_kernels_stay_out_of_the_column_loops():
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(CSRC), "..", "tools", "check_hot_loops.py")
    out = subprocess.run([sys.executable, tool, "perpair_packed"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("column blocks") >= 20, out.stdout


def test_spills_of_the_strips_kernels_stay_out_of_the_column_loops():
    # recompiles the two translation units with the tallest strips (hipcc cross-compiles: no GPU) and reads
    # the assembly: a basic block that holds a column of cells must not touch scratch memory
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(CSRC), "..", "tools", "check_hot_loops.py")
    out = subprocess.run([sys.executable, tool, "interseq_glbs16_b", "interseq_swbs16_b"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count("column blocks") >= 6, out.stdout
    # ... and the ISA the hand-over between strips relies on (common.h: outside the LLVM memory model): boundary rows
    # as 16-byte sc1 buffer accesses, progress counters stored behind a drained vmcnt
    assert out.stdout.count("hand-over:") >= 6 and "(0 without sc1)" in out.stdout, out.stdout


def test_pair_table_kernels_keep_three_wavefronts_per_simd():
    ks = kernels()
    if not ks:
        pytest.skip("no resource remarks: build with make -C pyopal_amd/csrc")
    pair = {n: k for n, k in ks.items() if "interseq_pair" in n}
    assert pair
    # kPairWaves = 12 wavefronts per workgroup, one workgroup per CU: 3 per SIMD, <= 168 VGPRs
    # (the NW / HW / OV form runs 8 wavefronts per workgroup: 2 per SIMD, up to 256 VGPRs)
    low = {n: k for n, k in pair.items()
           if k.get("Occupancy [waves/SIMD]", 0) < (2 if "pair_global" in n else 3)}
    assert not low, low
