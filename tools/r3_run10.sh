#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_clip_gpu.py -q -x -k "attention or clip_text" > $OUT/r3_t10.log 2>&1 || { tail -40 $OUT/r3_t10.log; exit 1; }
tail -2 $OUT/r3_t10.log
for cfg in "1 0" "0 0" "1 4" "1 8" "1 16"; do
  set -- $cfg
  echo "== PRESCALE=$1 STAGGER=$2"
  SMI_ATTN_PRESCALE=$1 SMI_ATTN_STAGGER=$2 python tools/bench_attn.py 2>&1 | grep -E "Nk4096|Nk1024" | grep "D64"
done
