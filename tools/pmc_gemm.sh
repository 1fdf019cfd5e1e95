#!/bin/bash
# MFMA-pipe counters of the 8-phase (gemm3) and the persistent 256x320 (gemm4) kernels on the level-2 shapes of the step.
# usage (through gpurun): bash tools/pmc_gemm.sh <tag>  -> gpurun_out/<tag>_pmc_gemm_counters.txt
set -e
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
export SMI_PMC_SHAPES="8192,8192,8192,0;16384,10240,1280,0;16384,3840,1280,0;16384,1280,1280,1;16384,1280,5120,1"
: > $OUT/${TAG}_pmc_gemm_counters.txt
for G in 8ph 5ph; do
  export SMI_GEMM=$G
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pg1_$G -- python3 tools/pmc_gemm.py > /dev/null 2> $OUT/${TAG}_pmcg1_$G.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/pg2_$G -- python3 tools/pmc_gemm.py > /dev/null 2> $OUT/${TAG}_pmcg2_$G.err
  python3 tools/pmc_gemm_summary.py /tmp/pg1_$G /tmp/pg2_$G $G >> $OUT/${TAG}_pmc_gemm_counters.txt
done
cat $OUT/${TAG}_pmc_gemm_counters.txt
