#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
python -m pytest tests/test_train_gpu.py tests/test_dp_gpu.py tests/test_vae_gpu.py tests/test_clip_gpu.py -q -x -k "fused or elementwise or image or main or clip_text_encoder_matches_oracle" > $OUT/r3_t2.log 2>&1 || { tail -60 $OUT/r3_t2.log; exit 1; }
tail -3 $OUT/r3_t2.log
python bench.py --config tiny_image_sdxl --steps 3 --warmup 1 > $OUT/r3_bench_tiny_image.json 2> $OUT/r3_bench_tiny_image.err || { tail -30 $OUT/r3_bench_tiny_image.err; exit 1; }
cat $OUT/r3_bench_tiny_image.json | cut -c1-600
