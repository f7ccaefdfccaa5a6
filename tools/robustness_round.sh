#!/bin/bash
# Run ON THE GPU BOX: the robustness tools on the build in the tree, logs under gpurun_out/<tag>_*.txt
# (soak, many threads, leak check, and the un-derandomised property run: three seeds as routed by the host,
#  two with one lane per pair forced - the query-profile kernels of `full` on every small case).
set -e
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python3 tools/soak.py > $O/${tag}_soak.txt 2>&1
timeout -k 10 300 python3 tools/many_threads.py > $O/${tag}_many_threads.txt 2>&1
timeout -k 10 300 python3 tools/leakcheck.py > $O/${tag}_leakcheck.txt 2>&1
: > $O/${tag}_fuzz.txt
for seed in 101 202 303; do
    FUZZ_SEED=$seed FUZZ_N=500 timeout -k 10 300 python3 -m pytest tests/_fuzz_once.py -m gpu -x -q 2>&1 | tail -1 >> $O/${tag}_fuzz.txt
done
for seed in 404 505; do
    echo "one lane per pair forced (MIOPAL_NO_SMALL_SEARCH, MIOPAL_FORCE_LANE_PER_PAIR, MIOPAL_NO_HYBRID_TRACE), seed $seed" >> $O/${tag}_fuzz.txt
    MIOPAL_NO_SMALL_SEARCH=1 MIOPAL_FORCE_LANE_PER_PAIR=1 MIOPAL_NO_HYBRID_TRACE=1 FUZZ_SEED=$seed FUZZ_N=500 \
        timeout -k 10 400 python3 -m pytest tests/_fuzz_once.py -m gpu -x -q 2>&1 | tail -1 >> $O/${tag}_fuzz.txt
done
