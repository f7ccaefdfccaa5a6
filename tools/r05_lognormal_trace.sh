#!/bin/bash
# Round 5: kernel trace summary of tools/r05_lognormal_q53.py for one algorithm (GPU box).
export TMPDIR=/tmp
D=/tmp/prof_ln_$$; rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/r05_lognormal_q53.py ${1:-nw} 2>&1 | grep TCUPS
f=$(find $D -name '*kernel_stats.csv' | head -1); head -8 $f | cut -c1-170
rm -rf $D
