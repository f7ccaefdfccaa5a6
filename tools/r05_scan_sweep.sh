#!/bin/bash
# Round 5: kernel time of the packed start-cell scan of cfg3 under its two launch knobs (GPU box).
set -u
export TMPDIR=/tmp
for cfg in "3 24" "2 24" "4 24" "3 12" "3 40" "3 64"; do
  set -- $cfg
  D=/tmp/prof_sweep_$$; rm -rf $D
  MIOPAL_SCAN_BLOCKS_PER_CU=$1 MIOPAL_SCAN_REFILL_LANES=$2 ONLY=packed REPS=5 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/quick_full_ab.py 1000000 53 > /dev/null 2>&1
  f=$(find $D -name '*kernel_stats.csv' | head -1)
  echo "blocks/CU $1 refill $2: $(grep packed_scan_kernel $f | cut -d, -f2-4)"
  rm -rf $D
done
