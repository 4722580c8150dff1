#!/bin/bash
# rocprofv3 kernel trace of K training iterations -> per-kernel table of the LAST iteration (busy time, span, gaps).
#   bash scripts/prof_iteration.sh <tag> [config]      (run on the GPU box; writes gpurun_out/<tag>_iter_*.{txt,csv})
set -e
TAG=${1:-r02}
CFG=${2:-mnist}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- python3 bench.py --iteration-only 12 --config $CFG > $OUT/${TAG}_iter_run.log 2>&1
CSV=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
python3 scripts/summarise_iter.py $CSV > $OUT/${TAG}_iter_kernels.txt
STATS=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp $STATS $OUT/${TAG}_iter_kernel_stats.csv
python3 scripts/iter_top_dispatches.py $CSV 40 > $OUT/${TAG}_iter_top.txt; python3 scripts/iter_overlap.py $CSV | tee $OUT/${TAG}_iter_overlap.txt; head -12 $OUT/${TAG}_iter_kernels.txt
