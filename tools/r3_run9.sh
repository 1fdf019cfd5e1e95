#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_engine_gpu.py tests/test_train_gpu.py tests/test_dp_gpu.py tests/test_kernels_gpu.py -q -x -s -k "per_sample or image or lora_skinny" > $OUT/r3_t9.log 2>&1 || { tail -60 $OUT/r3_t9.log; exit 1; }
grep -E "distance|passed|failed" $OUT/r3_t9.log | tail -5
python3 bench.py --config image_sdxl_1024_b1_r4 --steps 6 --warmup 2 > $OUT/r3_b9_image.json 2> $OUT/r3_b9_image.err || { tail -30 $OUT/r3_b9_image.err; exit 1; }
python3 - <<'P'
import json
d=json.loads(open("gpurun_out/r3_b9_image.json").read().strip().splitlines()[-1])
print("image", round(d["ms_per_step"],2), "ms; vae", round(d["vae_encode_and_noise_ms"],1), "frac", round(d["roofline"]["frac"],3), {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
