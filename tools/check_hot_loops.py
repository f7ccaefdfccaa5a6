"""Do the spills of a translation unit touch its column loops? Recompiles pyopal_amd/csrc/<tu>.hip with
--save-temps into a scratch directory (hipcc cross-compiles: no GPU needed) and lists, per kernel,
every basic block that holds at least 20 v_pk_maximum3_f16 (a column of cells) with its scratch_
instructions. Exit code 1 when any such block touches scratch.

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
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                        f"-I{CSRC}", "--save-temps", "-c", os.path.join(CSRC, tu + ".hip"), "-o", os.path.join(work, "x.o")],
                       cwd=work, check=True, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(work) if f.endswith("gfx950.s")][0]
        kernel, blocks, cur = None, [], None
        for line in open(os.path.join(work, asm)):
            text = line.strip()
            m = re.match(r"^(_Z\w+):", text)
            if m:
                kernel = m.group(1)
            if re.match(r"^(\.LBB\d+_\d+|_Z\w+):", text):
                cur = {"kernel": kernel, "label": text.split(":")[0], "max3": 0, "scratch": 0}
                blocks.append(cur)
            elif cur is not None:
                if text.startswith("v_pk_maximum3_f16"):
                    cur["max3"] += 1
                elif text.startswith("scratch_"):
                    cur["scratch"] += 1
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
    finally:
        shutil.rmtree(work, ignore_errors=True)
sys.exit(1 if bad else 0)
