#!/bin/bash
# kernel times of the packed passes for two builds of the library (GPU box): ab_lib.sh LIB_A LIB_B
export TMPDIR=/tmp
for lib in "$@"; do
  for q in 53 300; do
    D=/tmp/prof_ab_$$; rm -rf $D
    MIOPAL_LIBRARY=$PWD/$lib ONLY=packed REPS=3 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python3 tools/quick_full_ab.py 1000000 $q > /dev/null 2>&1
    f=$(find $D -name '*kernel_stats.csv' | head -1)
    echo "$lib Q=$q: $(grep 'packed_trace\|packed_scan' $f | cut -d, -f1,4 | sed 's/void miopal::(anonymous namespace):://' | tr '\n' ' ')"
    rm -rf $D
  done
done
