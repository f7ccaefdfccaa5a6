#!/bin/bash
# Round 5: the rows of a window as a minor sort key of the direction jobs, with and without (MIOPAL_NO_SORT_BY_ROWS=1).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
out=gpurun_out/r05_sort_rows_ab.txt; : > $out
for cfg in "1000000 53 3 1" "1000000 53 11 1" "1000000 300 3 1" "1000000 300 11 1" "1000000 150 3 1"; do
  for sw in 0 1; do
    if [ $sw = 1 ]; then export MIOPAL_NO_SORT_BY_ROWS=1; else unset MIOPAL_NO_SORT_BY_ROWS; fi
    echo "NO_SORT_BY_ROWS=$sw" >> $out
    ONLY=packed REPS=7 timeout -k 10 200 python3 tools/quick_full_ab.py $cfg >> $out 2>&1 || exit 1
  done
done
