#!/bin/bash
# kernel trace of a `full` search of one algorithm: tools/scratch/trace_algo.sh N Q open ext algo
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
D=/tmp/prof_algo_$$; rm -rf $D
ONLY=packed REPS=5 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/quick_full_ab.py "$@" > /tmp/algo_$$.log 2>&1
grep median /tmp/algo_$$.log
f=$(find $D -name '*kernel_stats.csv' | head -1); head -9 $f | cut -c1-160; rm -rf $D
