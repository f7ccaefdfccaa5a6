#!/bin/bash
# Round 5: `full` searches with the packed (two pairs per lane) later passes against the 32-bit ones, and whether both forms
# return the same arrays (tools/quick_full_ab.py), for the configurations DESIGN.md section 7 quotes.
# usage (GPU box): tools/r05_full_ab_table.sh > gpurun_out/r05_full_kernels_before_after.txt
cd "$(dirname "$0")/.."
for cfg in "1000000 53 3 1 sw" "1000000 53 11 1 sw" "1000000 150 3 1 sw" "1000000 300 3 1 sw" "1000000 300 11 1 sw" \
           "1000000 53 3 1 nw" "1000000 53 3 1 hw" "1000000 53 3 1 ov" "500000 300 3 1 hw"; do
  REPS=5 timeout -k 10 400 python3 tools/quick_full_ab.py $cfg 2>&1 | grep -v amdgpu || { echo "FAILED: $cfg"; exit 1; }
done
