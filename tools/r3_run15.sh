#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "lora or step_ops" > $OUT/r3_t15.log 2>&1 || { tail -40 $OUT/r3_t15.log; exit 1; }
tail -1 $OUT/r3_t15.log
for i in 1 2; do
  SMI_TUNE_FILE=/tmp/tune.txt python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r3_b15.json 2> $OUT/r3_b15.err || { tail -30 $OUT/r3_b15.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("gpurun_out/r3_b15.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],2), "ms", {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
done
