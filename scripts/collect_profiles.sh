#!/bin/bash
# Everything profiles/ quotes for one build, on the GPU box:  bash scripts/collect_profiles.sh <tag>   (writes gpurun_out/<tag>_*)
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python3 bench.py --config ucf --no-cpu-baseline > $OUT/${TAG}_bench_ucf.json 2>> $OUT/${TAG}_bench.err
python3 bench.py --config odernn --no-cpu-baseline > $OUT/${TAG}_bench_odernn.json 2>> $OUT/${TAG}_bench.err
rm -rf /tmp/prof_ro; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ro -o t -- python3 bench.py --roofline-only > $OUT/${TAG}_roofline_only.json 2> $OUT/${TAG}_roofline_only.err
cp $(find /tmp/prof_ro -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_roofline_only_kernel_stats.csv
rm -rf /tmp/prof_f; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o t -- python3 bench.py --roofline-only > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rm -rf /tmp/prof_w; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o t -- python3 bench.py --roofline-only > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 scripts/pmc_traffic_bench.py $(find /tmp/prof_f -name '*counter_collection.csv' | head -1) $(find /tmp/prof_w -name '*counter_collection.csv' | head -1) $OUT/${TAG}_pmc_traffic.json $TAG
bash scripts/prof_iteration.sh ${TAG} mnist > $OUT/${TAG}_iter_summary.txt
echo done
