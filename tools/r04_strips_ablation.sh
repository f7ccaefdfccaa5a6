#!/bin/bash
# Round 4: what the wavefronts of the multi-strip kernels wait for. (1) the diagnostic build with s_memtime
# deltas around the wait sites (MIOPAL_STRIP_TIMING=1 prints them per search); (2) ablation builds - the same
# instruction stream without one part of the hand-over (wrong results, honest timing).
cd "$(dirname "$0")/.." && R=$PWD
for w in "cfg4 nw score" "cfg4 sw score" "q300_1000000x300 nw score" "q300_1000000x300 sw score"; do
  echo "== $w (timing build)"
  MIOPAL_LIBRARY=$R/variants/libmiopal_timing.so MIOPAL_STRIP_TIMING=1 timeout -k 10 300 python3 tools/pmc_workload.py $w 2 2>&1 | grep "strip timing\|TCUPS" | tail -2
done
bash tools/ab_strips.sh "main nopoll noloads nostores nopubwait nohandover main" 2>&1 | grep -v "hw end\|q150"
