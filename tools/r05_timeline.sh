#!/bin/bash
# Round 5: the device timeline of the LAST `full` search of tools/quick_full_ab.py (packed routing): every dispatch with
# its start relative to the search's first kernel, its duration and the gap since the previous dispatch ended.
# usage (GPU box): tools/r05_timeline.sh N Q > gpurun_out/r05/timeline_QQ.txt
set -u
N=${1:-1000000}; Q=${2:-53}
export TMPDIR=/tmp
D=/tmp/prof_tl_$$; rm -rf $D
ONLY=${ONLY:-packed} rocprofv3 --kernel-trace -d $D -o t --output-format csv -- python3 tools/quick_full_ab.py $N $Q > /tmp/tl_$$.log 2>&1
python3 - $D <<'PY'
import csv, glob, sys
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), "copy"))
rows.sort()
# the last search: from the last dispatch of the end pass kernel
starts = [i for i, r in enumerate(rows) if "interseq_pair" in r[2]]
i0 = starts[-1]
t0 = rows[i0][0]
last_end = t0
for s, e, name, q in rows[i0:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - last_end) / 1e3:8.1f}  [{q}] {name}")
    last_end = max(last_end, e)
PY
rm -rf $D
