#!/bin/bash
# Runs on the GPU box (through gpurun): the rocprofv3 passes behind bench.py's roofline line.
#   1. --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   (separate passes; counters cannot share one)
#   4. --pmc SQ MFMA / wait / LDS counters + GRBM_GUI_ACTIVE (MFMA utilisation of the scorer kernels)
# Output under gpurun_out/prof_<tag>/ (tag = $1, default r02); tools/make_traffic.py turns it into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
shift || true
EXTRA="$@"   # further bench.py arguments, e.g. --config mslr30k --batch 4096
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline $EXTRA > $OUT/stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_profile --graph 0 $EXTRA > $OUT/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_profile --graph 0 $EXTRA > $OUT/write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -- python3 $ROOT/bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_profile --graph 0 $EXTRA > $OUT/sq.log 2>&1
ls -R $OUT | head -40
