"""Do the spills of a translation unit touch its column loops? Recompiles pyopal_amd/csrc/<tu>.hip with
--save-temps into a scratch directory (hipcc cross-compiles: no GPU needed) and lists, per kernel,
every basic block that holds at least 20 v_pk_maximum3_f16 (a column of cells) with its scratch_
instructions. Exit code 1 when any such block touches scratch.

For the strips kernels it also checks the ISA the hand-over of boundary rows relies on (common.h, stripPublish /
stripPoll: the scheme sits outside the LLVM memory model): every boundary-row access is a 16-byte buffer
instruction carrying sc1 (written through / read past the local L2), and every store of a progress counter
(global_store_dword ... sc1) has an `s_waitcnt vmcnt(0)` before it in its basic block with no vector-memory
instruction in between - a build in which the compiler turned either into something else fails here, not as
a wrong score one run in twenty.

usage: check_hot_loops.py TU [TU ...]     e.g. interseq_glbs16_b interseq_swbs16_b
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyopal_amd", "csrc")
bad = 0
for tu in sys.argv[1:]:
    work = tempfile.mkdtemp(prefix="hotloops_")
    try:
        # (the flags of pyopal_amd/csrc/Makefile: perpair_packed.o is scheduled for instruction-level parallelism)
        extra = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if tu == "perpair_packed" else []
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function"] + extra +
                       [f"-I{CSRC}", "--save-temps", "-c", os.path.join(CSRC, tu + ".hip"), "-o", os.path.join(work, "x.o")],
                       cwd=work, check=True, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(work) if f.endswith("gfx950.s")][0]
        kernel, blocks, cur = None, [], None
        handover = {}   # kernel -> [row accesses, of them without sc1, counter stores, of them not behind a drained vmcnt]
        for line in open(os.path.join(work, asm)):
            text = line.strip()
            m = re.match(r"^(_Z\w+):", text)
            if m:
                kernel = m.group(1)
            if re.match(r"^(\.LBB\d+_\d+|_Z\w+):", text):
                cur = {"kernel": kernel, "label": text.split(":")[0], "max3": 0, "scratch": 0, "drained": False}
                blocks.append(cur)
            elif cur is not None:
                if text.startswith("v_pk_maximum3_f16"):
                    cur["max3"] += 1
                elif text.startswith("scratch_"):
                    cur["scratch"] += 1
                if kernel and "strips_kernel" in kernel and "perpair" not in kernel:   # (the lane-per-pair strips hand nothing over)
                    h = handover.setdefault(kernel, [0, 0, 0, 0])
                    if text.startswith(("buffer_load_dwordx4", "buffer_store_dwordx4")):
                        h[0] += 1
                        h[1] += 0 if re.search(r"\bsc1\b", text) else 1
                        cur["drained"] = False
                    elif text.startswith("s_waitcnt") and "vmcnt(0)" in text:
                        cur["drained"] = True
                    elif text.startswith("global_store_dword ") and re.search(r"\bsc1\b", text):
                        h[2] += 1
                        h[3] += 0 if cur["drained"] else 1
                        cur["drained"] = False
                    elif text.startswith(("global_load", "global_store", "global_atomic", "buffer_", "flat_")):
                        cur["drained"] = False
        per_kernel = {}
        for b in blocks:
            k = per_kernel.setdefault(b["kernel"], {"columns": 0, "column_scratch": 0, "other_scratch": 0})
            if b["max3"] >= 20:
                k["columns"] += 1
                k["column_scratch"] += b["scratch"]
            else:
                k["other_scratch"] += b["scratch"]
        for name, k in sorted(per_kernel.items()):
            if name is None:
                continue
            short = re.sub(r"^_ZN6miopal\d+", "", name)
            print(f"{tu}: {short}: {k['columns']} column blocks, scratch instructions inside them {k['column_scratch']}, "
                  f"elsewhere (per group / per unit) {k['other_scratch']}")
            bad += k["column_scratch"]
            if name in handover:
                rows, rows_plain, counters, counters_early = handover[name]
                print(f"{tu}: {short}: hand-over: {rows} boundary-row accesses ({rows_plain} without sc1), "
                      f"{counters} progress-counter stores ({counters_early} not behind a drained vmcnt)")
                # (one store per kernel needs no drain: a unit that gave up poisons its counter, it publishes no rows;
                # the two sweeps that hand rows down - first and inner strips - each end in a drained store)
                bad += rows_plain + (0 if rows >= 4 and counters - counters_early >= 2 and counters_early <= 1 else 1)
    finally:
        shutil.rmtree(work, ignore_errors=True)
sys.exit(1 if bad else 0)
