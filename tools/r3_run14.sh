#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
for r in 0 128 256 0 256; do
  SMI_WGRAD_ROWS=$r SMI_TUNE_FILE=/tmp/tune.txt python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r3_b14.json 2> $OUT/r3_b14.err || { tail -30 $OUT/r3_b14.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("gpurun_out/r3_b14.json").read().strip().splitlines()[-1])
print("rows=$r", round(d["ms_per_step"],2), "ms", {k:round(v["ms"],2) for k,v in d["kernel_classes"].items() if k in ("lora","norm")})
P
done
