#!/bin/bash
# Round 4: the `full` pipeline after the tiled bit planes and the packed copy-out: tests of the passes, wall
# times of cfg3 / Q = 300 with and without the packed copy, kernel trace of cfg3.
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out && mkdir -p $O
python3 -m pytest tests/test_gpu_full_profile.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not cfg4 and not cfg5" > $O/r04_full_tests.log 2>&1 || { tail -30 $O/r04_full_tests.log; exit 1; }
tail -3 $O/r04_full_tests.log
for v in packed unpacked; do
  if [ $v = unpacked ]; then export MIOPAL_NO_PACKED_OPS=1; else unset MIOPAL_NO_PACKED_OPS; fi
  timeout -k 10 200 python3 tools/pmc_workload.py q53_1000000x300 sw full 10 2>&1 | tail -1 | sed "s/^/$v: /"
  MIOPAL_PHASE_TIMING=1 timeout -k 10 200 python3 tools/pmc_workload.py q53_1000000x300 sw full 3 > $O/r04_cfg3_$v.txt 2>&1
  MIOPAL_PHASE_TIMING=1 timeout -k 10 300 python3 tools/pmc_workload.py q300_1000000x300 sw full 2 > $O/r04_q300_$v.txt 2>&1; tail -1 $O/r04_q300_$v.txt | sed "s/^/$v: /"
done
unset MIOPAL_NO_PACKED_OPS
bash tools/r04_full_trace.sh
