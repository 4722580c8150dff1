#!/bin/bash
# Copies what scripts/collect_profiles.sh <tag> left in gpurun_out/ into profiles/ under the per-round names.
#   bash scripts/publish_profiles.sh <tag> <round> <version>     e.g.  r02v2 r02 v2
set -e
TAG=$1; R=$2; V=$3
G=gpurun_out; P=profiles
cp $G/${TAG}_bench.json $P/${R}_bench_${V}.json
cp $G/${TAG}_bench_ucf.json $P/${R}_bench_ucf_${V}.json
cp $G/${TAG}_bench_odernn.json $P/${R}_bench_odernn_${V}.json
cp $G/${TAG}_roofline_only.json $P/${R}_roofline_only_${V}.json
cp $G/${TAG}_roofline_only_kernel_stats.csv $P/${R}_roofline_only_kernel_stats_${V}.csv
cp $G/${TAG}_pmc_traffic.json $P/${R}_pmc_traffic_${V}.json
cp $G/${TAG}_iter_kernels.txt $P/${R}_iteration_kernels_${V}.txt
cp $G/${TAG}_iter_kernel_stats.csv $P/${R}_iteration_kernel_stats_${V}.csv
cp $G/${TAG}_iter_top.txt $P/${R}_iteration_top_dispatches_${V}.txt
H=$(mktemp)
echo "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --roofline-only (MI355X, ${R} build ${V}): per-kernel totals; the HIP-event timing printed by the same run is profiles/${R}_roofline_only_${V}.json" > $H; echo >> $H; echo >> $H
python3 scripts/stats_to_md.py $P/${R}_roofline_only_kernel_stats_${V}.csv $P/${R}_roofline_only_kernel_stats_${V}.md $H
echo "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --iteration-only 12 (MI355X, ${R} build ${V}): per-kernel totals over 3 warm-up + 12 timed training iterations; the per-iteration table of the LAST iteration is profiles/${R}_iteration_kernels_${V}.txt" > $H; echo >> $H; echo >> $H
python3 scripts/stats_to_md.py $P/${R}_iteration_kernel_stats_${V}.csv $P/${R}_iteration_kernel_stats_${V}.md $H
rm -f $H
ls $P | grep "${R}_.*_${V}"
