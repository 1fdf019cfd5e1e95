#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_vae_gpu.py tests/test_clip_gpu.py tests/test_train_gpu.py -q -x -s -k "not headline_size" > $OUT/r3_t8.log 2>&1 || { tail -60 $OUT/r3_t8.log; exit 1; }
grep -E "rel err|eps |grads |c3lier|dora|moments|last |pooled|sign different|global rel|distance|bf16" $OUT/r3_t8.log | cut -c1-260
tail -2 $OUT/r3_t8.log
