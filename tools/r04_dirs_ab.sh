#!/bin/bash
# A/B on one box: direction planes stored per column ([column][plane] lines, main) against four columns held in
# registers ([plane][column] lines, variants/libmiopal_held.so): kernel traces of cfg3 and Q = 300 `full`.
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in main held main held; do
  if [ $lib = main ]; then unset MIOPAL_LIBRARY; else export MIOPAL_LIBRARY=$R/variants/libmiopal_$lib.so; fi
  for w in q53_1000000x300 q300_1000000x300; do
    rm -rf $O/ab_trace
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/ab_trace -o t --output-format csv -- python3 $R/tools/pmc_workload.py $w sw full 3 > $O/ab_trace.log 2>&1
    f=$(find $O/ab_trace -name "*kernel_stats.csv" | head -1)
    echo "[$lib] $w: $(grep TCUPS $O/ab_trace.log | cut -d: -f2 | cut -d, -f1) | dir $(grep 'profile_kernel<4>' $f | cut -d, -f4 | cut -c1-8) ns | walk $(grep 'walk_planes' $f | cut -d, -f4 | cut -c1-8) ns"
    rm -rf $O/ab_trace
  done
done
