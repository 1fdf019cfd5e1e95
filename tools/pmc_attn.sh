#!/bin/bash
# SQ / LDS / MFMA counters of the attention kernels (two rocprofv3 --pmc passes, program directly after `--`).
# usage (through gpurun): bash tools/pmc_attn.sh <tag>  -> gpurun_out/<tag>_pmc_attention_counters.txt
set -e
TAG=${1:-r03}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -rf /tmp/pa1 /tmp/pa2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pa1 -- python3 tools/pmc_attn.py > /dev/null 2> $OUT/${TAG}_pmca1.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --output-format csv -d /tmp/pa2 -- python3 tools/pmc_attn.py > /dev/null 2> $OUT/${TAG}_pmca2.err
python3 tools/pmc_attn_summary.py /tmp/pa1 /tmp/pa2 > $OUT/${TAG}_pmc_attention_counters.txt
cat $OUT/${TAG}_pmc_attention_counters.txt
