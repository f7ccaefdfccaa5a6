#!/bin/bash
# Run ON THE GPU BOX: every PMC summary the bench line quotes (bench.py, pmc_summary / leg_roofline /
# full_pipeline_roofline), on the build in the tree: gpurun_out/<tag>_pmc_*.json + kernel stats. Copy into profiles/.
# (round 5: the later passes of `full` are the kernels of perpair_packed.hip)
tag=${1:-r05}
part=${2:-all}
cd "$(dirname "$0")/.." && R=$PWD
mkdir -p gpurun_out
c() { bash tools/collect_pmc.sh $tag "$@" > gpurun_out/${tag}_collect_$1.log 2>&1; rm -rf gpurun_out/prof_${tag}_$1; tail -1 gpurun_out/${tag}_collect_$1.log | cut -c1-200; }
if [ $part = all ] || [ $part = a ]; then
bash tools/collect_headline_profile.sh $tag > gpurun_out/${tag}_collect_headline.log 2>&1
c cfg2_end      interseq_pair_biased_kernel                 q53_1000000x300 sw end 5
c q53_nw        interseq_pair_global_kernel                 q53_1000000x300 nw score 5
c q150_sw       interseq_pair_strips_kernel                 q150_1000000x300 sw score 5
c q300_sw       interseq_pair_strips_kernel                 q300_1000000x300 sw score 5
c q150_sw_end   interseq_pair_strips_kernel                 q150_1000000x300 sw end 5
c q300_sw_end   interseq_pair_strips_kernel                 q300_1000000x300 sw end 5
fi
if [ $part = all ] || [ $part = b ]; then
c cfg4_nw       interseq_pair_global_strips_kernel          cfg4tail nw score 3
c cfg4_hw       interseq_pair_global_strips_kernel          cfg4tail hw score 3
c cfg4_ov       interseq_pair_global_strips_kernel          cfg4tail ov score 3
c cfg4_sw       interseq_pair_strips_kernel                 cfg4tail sw score 3
fi
if [ $part = all ] || [ $part = c ]; then
c cfg3full      interseq_pair_biased_kernel,perpair_packed_scan_kernel,perpair_packed_trace_kernel,walk_planes_kernel,gather_ops_kernel,copy_out_packed_kernel  q53_1000000x300 sw full 3
c q300full      interseq_pair_strips_kernel,perpair_packed_scan_strips_kernel,perpair_packed_trace_kernel,walk_planes_kernel,copy_out_packed_kernel  q300_1000000x300 sw full 2
fi
rm -rf gpurun_out/prof_$tag
ls gpurun_out/${tag}_pmc_*.json | wc -l
