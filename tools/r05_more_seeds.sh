#!/bin/bash
# Round 5: more seeds of the un-derandomised property run on the final build (tests/_fuzz_once.py): as routed by the host,
# and with one lane per pair forced - the two-pairs-per-lane passes of `full` on every small case.
cd "$(dirname "$0")/.."; O=gpurun_out/r05_fuzz_more.txt; : > $O
for seed in 611 722 833; do
  echo "seed $seed" >> $O
  FUZZ_SEED=$seed FUZZ_N=500 timeout -k 10 300 python3 -m pytest tests/_fuzz_once.py -m gpu -x -q 2>&1 | tail -1 >> $O || exit 1
done
for seed in 944 1055 1166; do
  echo "one lane per pair forced (MIOPAL_NO_SMALL_SEARCH, MIOPAL_FORCE_LANE_PER_PAIR, MIOPAL_NO_HYBRID_TRACE), seed $seed" >> $O
  MIOPAL_NO_SMALL_SEARCH=1 MIOPAL_FORCE_LANE_PER_PAIR=1 MIOPAL_NO_HYBRID_TRACE=1 FUZZ_SEED=$seed FUZZ_N=500 \
    timeout -k 10 400 python3 -m pytest tests/_fuzz_once.py -m gpu -x -q 2>&1 | tail -1 >> $O || exit 1
done
