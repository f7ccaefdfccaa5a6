#!/bin/bash
# Round 5: kernel traces of a `full` search with and without the windows' rows as a minor sort key of the direction jobs.
# usage: tools/r05_sort_rows_trace.sh N Q [open ext]   (on the GPU box; writes gpurun_out/r05/sortrows_*.csv)
set -u
N=${1:-1000000}; Q=${2:-53}; GO=${3:-3}; GE=${4:-1}
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
export TMPDIR=/tmp
for sw in rows length; do
  if [ $sw = length ]; then export MIOPAL_NO_SORT_BY_ROWS=1; else unset MIOPAL_NO_SORT_BY_ROWS; fi
  D=/tmp/prof_${sw}_$$; rm -rf $D
  ONLY=packed rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/quick_full_ab.py $N $Q $GO $GE > $OUT/sortrows_${sw}_Q${Q}_${GO}.log 2>&1
  f=$(find $D -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f $OUT/sortrows_${sw}_Q${Q}_${GO}_kernel_stats.csv; echo "== $sw"; head -9 $f | cut -c1-170; fi
done
