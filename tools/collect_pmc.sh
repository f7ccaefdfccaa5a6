#!/bin/bash
# Run ON THE GPU BOX: kernel-trace summary + separate --pmc passes (SQ issue, SQ waits / LDS, FETCH_SIZE,
# WRITE_SIZE: they do not fit one pass, and PMC is never combined with other trace domains) of ONE
# workload of tools/pmc_workload.py, condensed per kernel into
#   gpurun_out/<tag>_pmc_<name>.json        (carries the sha256 of the libmiopal.so that was profiled)
#   gpurun_out/<tag>_<name>_kernel_stats.csv
# usage: collect_pmc.sh TAG NAME KERNEL_SUBSTRING[,KERNEL_SUBSTRING...] WORKLOAD ALGO MODE [REPS]
# Copy what should be judged into profiles/ afterwards.
set -e
tag=$1; name=$2; needles=$3; shift 3
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
p=$out/prof_${tag}_$name
rm -rf $p && mkdir -p $p
w="python3 $root/tools/pmc_workload.py $*"
rocprofv3 --kernel-trace --stats -d $p/trace --output-format csv -- $w > $p/trace.log 2>&1
cp $(ls $p/trace/*/*kernel_stats.csv | head -1) $out/${tag}_${name}_kernel_stats.csv
grep TCUPS $p/trace.log || true
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE -d $p/pmc_sq --output-format csv -- $w > $p/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $p/pmc_sq2 --output-format csv -- $w > $p/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $p/pmc_fetch --output-format csv -- $w > $p/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $p/pmc_write --output-format csv -- $w > $p/pmc_write.log 2>&1
for needle in ${needles//,/ }; do   # (a needle may carry template arguments: perpair_profile_kernel<4>)
    kms=$(python3 - "$out/${tag}_${name}_kernel_stats.csv" "$needle" <<'EOF'
import csv, sys
best = None
for row in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in row["Name"]:
        ns = float(row["AverageNs"])
        if best is None or ns > best: best = ns
print((best or 0) / 1e6)
EOF
)
    if [ "$kms" = "0.0" ]; then echo "kernel $needle not in the trace"; continue; fi
    suffix=$name; [ "$needles" != "$needle" ] && suffix=${name}_$(echo "$needle" | tr -d '>' | tr '<' '_')
    PMC_COMMAND="rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 tools/pmc_workload.py $* (separate passes: SQ issue + GRBM, SQ waits / LDS, FETCH_SIZE, WRITE_SIZE)" \
    PMC_CELLS=$(grep -o "cells=[0-9]*" $p/trace.log | head -1 | cut -d= -f2) \
    python3 $root/tools/summarize_pmc.py $out/${tag}_pmc_$suffix.json $needle $kms $p/pmc_sq $p/pmc_sq2 $p/pmc_fetch $p/pmc_write
done
