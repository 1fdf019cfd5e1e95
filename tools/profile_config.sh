#!/bin/bash
# Per-kernel statistics of one bench configuration (rocprofv3 --kernel-trace --stats, tuner file filled first so the
# profiled run launches no tuning candidates): bash tools/profile_config.sh <tag> <config> [bench args]
set -e
TAG=$1; CFG=$2; shift 2
cd "$(dirname "$0")/.."
export TMPDIR=/tmp SMI_TUNE_FILE=/tmp/smi_tune_${CFG}.txt
OUT=gpurun_out; mkdir -p $OUT
python3 bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2> $OUT/${TAG}_fill.err
rm -rf $OUT/${TAG}_trace_$CFG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_$CFG -- python3 bench.py --config $CFG --steps 20 --warmup 2 --no-cpu-baseline "$@" > $OUT/${TAG}_bench_under_rocprof_$CFG.json 2> $OUT/${TAG}_trace_$CFG.err
find $OUT/${TAG}_trace_$CFG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_rocprofv3_kernel_stats_$CFG.csv
rm -rf $OUT/${TAG}_trace_$CFG
head -40 $OUT/${TAG}_rocprofv3_kernel_stats_$CFG.csv | cut -c1-200
