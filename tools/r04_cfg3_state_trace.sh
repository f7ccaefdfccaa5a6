#!/bin/bash
# kernel traces of the cfg3 `full` calls of tools/quick_cfg3_leg.py with and without what bench.py runs before them
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in none sd; do
  rm -rf $O/r04_state_$s
  PHASES=1 SUSTAIN=$s timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r04_state_$s -o t --output-format csv -- python3 $R/tools/quick_cfg3_leg.py > $O/r04_state_$s.log 2>&1
  f=$(find $O/r04_state_$s -name "*kernel_stats.csv" | head -1)
  cp $f $O/r04_cfg3_state_${s}_kernel_stats.csv
  echo "== $s"; head -14 $f | cut -c1-150
  t=$(find $O/r04_state_$s -name "*kernel_trace.csv" | head -1)
  (head -1 $t; tail -150 $t) > $O/r04_cfg3_state_${s}_kernel_trace_tail.csv
  rm -rf $O/r04_state_$s
done
