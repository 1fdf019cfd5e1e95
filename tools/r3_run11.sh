#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_clip_gpu.py -q -x -k "attention or clip_text" > $OUT/r3_t11.log 2>&1 || { tail -40 $OUT/r3_t11.log; exit 1; }
tail -2 $OUT/r3_t11.log
for v in 1 0 1 0; do
  echo "== MFMA16=$v"
  SMI_ATTN_MFMA16=$v python tools/bench_attn.py 2>&1 | grep -E "Nk4096|Nk1024" | grep "D64" | cut -c1-75
done
