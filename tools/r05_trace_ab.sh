#!/bin/bash
# Round 5: kernel traces of a `full` search with the packed later passes and with the 32-bit ones.
# usage: tools/r05_trace_ab.sh N Q [open ext]   (on the GPU box; writes gpurun_out/r05/trace_*.txt)
set -u
N=${1:-1000000}; Q=${2:-53}; GO=${3:-3}; GE=${4:-1}
OUT=$PWD/gpurun_out/r05; mkdir -p $OUT
export TMPDIR=/tmp
for which in packed old; do
  D=/tmp/prof_${which}_$$; rm -rf $D
  ONLY=$which rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/quick_full_ab.py $N $Q $GO $GE > $OUT/trace_${which}_Q${Q}.log 2>&1
  f=$(find $D -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f $OUT/trace_${which}_Q${Q}_kernel_stats.csv; head -14 $f | cut -c1-150; fi
done
