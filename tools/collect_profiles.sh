#!/bin/bash
# Round measurement set on the GPU box (run through gpurun): a first un-profiled bench run fills the tile tuner's file
# (SMI_TUNE_FILE), so the profiled runs launch NO tuning candidates; then kernel trace + stats, the per-step cut of the
# trace (tools/trace_steps.py), and the two PMC passes for HBM-side traffic, each as its own rocprofv3 run with the program
# directly after `--` (MI355X_MICROARCH.md, HBM section).
# usage: bash tools/collect_profiles.sh <tag> [<config>]     -> gpurun_out/<tag>_*
set -e
TAG=${1:-r03}
CFG=${2:-sdxl_1024_b2_r4}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
export SMI_TUNE_FILE=/tmp/smi_tune_${CFG}.txt
OUT=gpurun_out
mkdir -p $OUT
SMI_PROF_DUMP=1 python3 bench.py --config $CFG --steps 6 --warmup 2 > $OUT/${TAG}_bench_default_${CFG}.json 2> $OUT/${TAG}_bench_default_${CFG}.err
cp $SMI_TUNE_FILE $OUT/${TAG}_tune_${CFG}.txt
echo "bench done: $(wc -l < $SMI_TUNE_FILE) tuned keys"
rm -rf $OUT/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof_${CFG}.json 2> $OUT/${TAG}_trace.err
find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_rocprofv3_kernel_stats_${CFG}.csv
KT=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_steps.py $KT $OUT/${TAG}_bench_under_rocprof_${CFG}.json $OUT/${TAG}_trace_steps_${CFG}.json
gzip -c $KT > $OUT/${TAG}_kernel_trace_${CFG}.csv.gz
echo "trace done"
# Counter passes.  Round 3: one such pass ABORTED (HSA_STATUS_ERROR_INVALID_PACKET_FORMAT) inside
# bench.py's HIP-event profiled step -- 3 AQL packets per launch from us, each expanded by the profiler's queue interceptor,
# 433 packets outstanding.  The passes therefore (i) stop after the timed steps (--timed-only: no event-profiled step, which
# the counters do not need) and (ii) bound our queue depth (SMI_SYNC_EVERY=64: the engine waits for the stream every 64
# launches) instead of waiting after every launch as round 3 did.
SMI_SYNC_EVERY=64 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmcf -- python3 bench.py --config $CFG --steps 2 --warmup 1 --timed-only > /dev/null 2> $OUT/${TAG}_pmcf.err
echo "fetch done"
SMI_SYNC_EVERY=64 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmcw -- python3 bench.py --config $CFG --steps 2 --warmup 1 --timed-only > /dev/null 2> $OUT/${TAG}_pmcw.err
echo "write done"
python3 tools/pmc_traffic.py $OUT/${TAG}_pmcf $OUT/${TAG}_pmcw $OUT/${TAG}_pmc_traffic_${CFG}.json > $OUT/${TAG}_pmc_summary_${CFG}.txt
rm -rf $OUT/${TAG}_pmcf $OUT/${TAG}_pmcw $OUT/${TAG}_trace
head -12 $OUT/${TAG}_pmc_summary_${CFG}.txt
