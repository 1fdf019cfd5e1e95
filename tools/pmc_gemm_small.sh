#!/bin/bash
# MFMA-pipe / wait counters of the shapes that LOSE in the step: the 4096-row backward GEMMs and the
# level-1 K = 640 shapes, under the DEFAULT kernel selection (deep-prefetch 128 x 160 eight-wave tile by rule / the tuner's
# pick) -- two rocprofv3 --pmc passes, program directly after `--`.
# usage (through gpurun): bash tools/pmc_gemm_small.sh <tag>  -> gpurun_out/<tag>_pmc_gemm_small_counters.txt
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
export SMI_PMC_SHAPES="4096,1280,1280,1;4096,1280,3840,0;4096,1280,10240,0;4096,5120,1280,0;65536,640,640,1;65536,1920,640,0;65536,5120,640,0;16384,1280,1280,1"
export SMI_GEMM_TUNE=0
rm -rf /tmp/ps1 /tmp/ps2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/ps1 -- python3 tools/pmc_gemm.py > /dev/null 2> $OUT/${TAG}_pmcs1.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/ps2 -- python3 tools/pmc_gemm.py > /dev/null 2> $OUT/${TAG}_pmcs2.err
echo "# shapes (M,N,K,bias+res): $SMI_PMC_SHAPES -- three launches each, in this order; heuristic selection (SMI_GEMM_TUNE=0)" > $OUT/${TAG}_pmc_gemm_small_counters.txt
python3 tools/pmc_gemm_summary.py /tmp/ps1 /tmp/ps2 default >> $OUT/${TAG}_pmc_gemm_small_counters.txt
cat $OUT/${TAG}_pmc_gemm_small_counters.txt
