#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_headline_profile.sh r02'): kernel-trace summary of
# the default bench command plus three separate --pmc passes (SQ + GRBM, FETCH_SIZE, WRITE_SIZE: they do
# not fit one pass, and PMC is never combined with other trace domains), condensed into
#   gpurun_out/<tag>_bench_kernel_stats.csv, gpurun_out/<tag>_bench_line.json,
#   gpurun_out/<tag>_pmc_headline.json   (carries the sha256 of the libmiopal.so that was profiled)
# Copy them into profiles/ afterwards.
set -e
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--no-cpu-baseline --no-cfg5"
python3 $root/bench.py > $out/${tag}_bench_line.json 2> $out/${tag}_bench_line.err
rm -rf $out/prof_$tag && mkdir -p $out/prof_$tag
rocprofv3 --kernel-trace --stats -d $out/prof_$tag/trace --output-format csv -- python3 $root/bench.py $args > $out/prof_$tag/trace.log 2>&1
cp $(ls $out/prof_$tag/trace/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE -d $out/prof_$tag/pmc_sq --output-format csv -- python3 $root/bench.py $args --steps 5 --warmup 1 > $out/prof_$tag/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $out/prof_$tag/pmc_sq2 --output-format csv -- python3 $root/bench.py $args --steps 5 --warmup 1 > $out/prof_$tag/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/prof_$tag/pmc_fetch --output-format csv -- python3 $root/bench.py $args --steps 5 --warmup 1 > $out/prof_$tag/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/prof_$tag/pmc_write --output-format csv -- python3 $root/bench.py $args --steps 5 --warmup 1 > $out/prof_$tag/pmc_write.log 2>&1
kms=$(python3 -c "import json;print(json.load(open('$out/${tag}_bench_line.json'))['roofline']['kernel_ms'])")
PMC_COMMAND="rocprofv3 --kernel-trace --pmc <counters> --output-format csv -- python3 bench.py $args --steps 5 --warmup 1 (separate passes: SQ+GRBM, SQ waits, FETCH_SIZE, WRITE_SIZE)" \
python3 $root/tools/summarize_pmc.py $out/${tag}_pmc_headline.json interseq_pair_biased_kernel $kms $out/prof_$tag/pmc_sq $out/prof_$tag/pmc_sq2 $out/prof_$tag/pmc_fetch $out/prof_$tag/pmc_write
head -5 $out/${tag}_bench_kernel_stats.csv
