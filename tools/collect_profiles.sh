#!/bin/bash
# Round measurement set on the GPU box (run through gpurun): kernel trace + stats, then the two PMC passes for HBM-side
# traffic, each as its own rocprofv3 run with the program directly after `--` (MI355X_MICROARCH.md, HBM section).
# usage: bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_trace.err
find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmcf -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmcf.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmcw -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmcw.err
echo "write done"
python3 tools/pmc_traffic.py $OUT/${TAG}_pmcf $OUT/${TAG}_pmcw $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_summary.txt
rm -rf $OUT/${TAG}_pmcf $OUT/${TAG}_pmcw $OUT/${TAG}_trace
head -12 $OUT/${TAG}_pmc_summary.txt
