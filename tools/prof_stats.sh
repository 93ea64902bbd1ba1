#!/bin/bash
# Runs on the GPU box: per-kernel average durations (rocprofv3 --kernel-trace --stats) of one bench.py command.
#   tools/prof_stats.sh <tag> [bench.py args]   ->  gpurun_out/stats_<tag>.csv
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
shift
OUT=$ROOT/gpurun_out/pstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_profile "$@" > $OUT/log.txt 2>&1
cp $(find $OUT -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/stats_$TAG.csv
python3 $ROOT/tools/trace_dist.py $(find $OUT -name '*kernel_trace.csv' | head -1) > $ROOT/gpurun_out/dist_$TAG.txt
rm -rf $OUT
