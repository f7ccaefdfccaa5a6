#!/bin/bash
# A/B of a library variant on the one-strip kernels over database sizes: kernel ms (HIP events), one box.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for n in 100000 200000 393216 500000 786432 1000000 1500000 2000000; do
  a=$(QB_ROUNDS=3 python3 $root/tools/quick_bench.py $n 300 2>&1 | grep kernel | tail -1 | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  b=$(MIOPAL_LIBRARY=$root/variants/libmiopal_$1.so QB_ROUNDS=3 python3 $root/tools/quick_bench.py $n 300 2>&1 | grep kernel | tail -1 | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  echo "N=$n sw score Q=53: main $a ms | $1 $b ms"
done
for algo in nw hw; do
  a=$(QB_ROUNDS=3 python3 $root/tools/quick_bench.py 1000000 300 53 $algo 2>&1 | grep kernel | tail -1 | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  b=$(MIOPAL_LIBRARY=$root/variants/libmiopal_$1.so QB_ROUNDS=3 python3 $root/tools/quick_bench.py 1000000 300 53 $algo 2>&1 | grep kernel | tail -1 | sed 's/.*kernel \([0-9.]*\) ms.*/\1/')
  echo "N=1000000 $algo score Q=53: main $a ms | $1 $b ms"
done
