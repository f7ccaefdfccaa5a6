#!/bin/bash
# A/B of library variants (tools/ab_build.sh) on the strips kernels, one box: each line is one process.
# usage: ab_strips.sh "variantA variantB ..." ("main" = the in-tree library)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in $1; do
  if [ "$lib" = main ]; then unset MIOPAL_LIBRARY; else export MIOPAL_LIBRARY=$root/variants/libmiopal_$lib.so; fi
  for w in "cfg4 nw score" "cfg4 sw score" "cfg4 hw end" "q300_1000000x300 nw score" "q300_1000000x300 sw score" "q150_1000000x300 ov score"; do
    echo -n "[$lib] "; python3 $root/tools/pmc_workload.py $w 5 2>&1 | grep TCUPS
  done
done
