#!/bin/bash
# kernel traces of cfg3 `full` and of Q = 300 `full` (rocprofv3 --kernel-trace --stats, csv)
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in q53_1000000x300 q300_1000000x300; do
  rm -rf $O/r04_trace_$w
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r04_trace_$w -o t --output-format csv -- python3 $R/tools/pmc_workload.py $w sw full 3 > $O/r04_trace_$w.log 2>&1
  f=$(find $O/r04_trace_$w -name "*kernel_stats.csv" | head -1)
  cp $f $O/r04_${w}_full_kernel_stats.csv
  echo "== $w"; head -9 $f | cut -c1-160
  rm -rf $O/r04_trace_$w
done
