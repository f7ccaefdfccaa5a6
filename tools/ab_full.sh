#!/bin/bash
# A/B of the `full` pipeline's lane-per-pair kernels: the query-profile form against the one before
# (MIOPAL_NO_PERPAIR_PROFILE=1), kernel trace of cfg3 and wall times of Q = 300 at gap 3/1 and 11/1.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export MIOPAL_NO_PERPAIR_PROFILE=1; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_full_$v -o cfg3 -- python3 $R/tools/pmc_workload.py q53_1000000x300 sw full 5 > $O/ab_full_cfg3_$v.txt 2>&1
  MIOPAL_PHASE_TIMING=1 timeout -k 10 200 python3 $R/tools/pmc_workload.py q300_1000000x300 sw full 2 > $O/ab_full_q300_$v.txt 2>&1
  PW_OPEN=11 MIOPAL_PHASE_TIMING=1 timeout -k 10 200 python3 $R/tools/pmc_workload.py q300_1000000x300 sw full 2 > $O/ab_full_q300g11_$v.txt 2>&1
done
