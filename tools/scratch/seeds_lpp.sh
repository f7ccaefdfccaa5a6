#!/bin/bash
cd "$(dirname "$0")/../.."; O=gpurun_out/r05_fuzz_lpp.txt; : > $O
for seed in 2001 2002 2003 2004; do
  echo "one lane per pair forced, 1500 larger + 6000 small cases, seed $seed" >> $O
  MIOPAL_NO_SMALL_SEARCH=1 MIOPAL_FORCE_LANE_PER_PAIR=1 MIOPAL_NO_HYBRID_TRACE=1 FUZZ_SEED=$seed FUZZ_N=1500 \
    timeout -k 10 600 python3 -m pytest tests/_fuzz_once.py -m gpu -x -q 2>&1 | tail -1 >> $O || exit 1
done
