#!/bin/bash
# Run ON THE GPU BOX: every profile the bench line quotes, for the build that is in the tree now
# (the summaries carry its sha256): the headline kernel (default bench command), the BASELINE configs[3]
# kernels and the longer-query Smith-Waterman legs.   usage: collect_round_profiles.sh TAG
set -e
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash $root/tools/collect_headline_profile.sh $tag
bash $root/tools/collect_pmc.sh $tag cfg4_nw interseq_pair_global_strips_kernel cfg4tail nw score 3
bash $root/tools/collect_pmc.sh $tag cfg4_sw interseq_pair_strips_kernel cfg4tail sw score 3
bash $root/tools/collect_pmc.sh $tag q150_sw interseq_pair_strips_kernel q150_1000000x300 sw score 5
bash $root/tools/collect_pmc.sh $tag q300_sw interseq_pair_strips_kernel q300_1000000x300 sw score 5
bash $root/tools/collect_pmc.sh $tag q53_nw interseq_pair_global_kernel q53_1000000x300 nw score 5
# the passes behind the end pass of a `full` search: start-cell scan (<3> = region "all cells"), directions (<4>), walk
# (one-strip queries: the scan is perpair_scan_refill_kernel)
for w in cfg3full:q53_1000000x300:perpair_scan_refill_kernel "q300full:q300_1000000x300:perpair_profile_kernel<3>"; do
    name=${w%%:*}; rest=${w#*:}; scan=${rest#*:}; workload=${rest%%:*}
    bash $root/tools/collect_pmc.sh $tag $name "$scan,perpair_profile_kernel<4>,walk_kernel" $workload sw full 3
    mv "$root/gpurun_out/${tag}_pmc_${name}_${scan}.json" $root/gpurun_out/${tag}_pmc_${name}_scan.json
    mv "$root/gpurun_out/${tag}_pmc_${name}_perpair_profile_kernel<4>.json" $root/gpurun_out/${tag}_pmc_${name}_directions.json
    mv $root/gpurun_out/${tag}_pmc_${name}_walk_kernel.json $root/gpurun_out/${tag}_pmc_${name}_walk.json
done
ls -la $root/gpurun_out/${tag}_*
