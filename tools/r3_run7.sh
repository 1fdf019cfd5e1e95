#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
for v in 0 1 0 1; do
  SMI_GEMM_SPLITK_BIG=$v SMI_TUNE_FILE=/tmp/tune.txt python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/r3_b7_$v.json 2> $OUT/r3_b7_$v.err || { tail -30 $OUT/r3_b7_$v.err; exit 1; }
  python3 - <<P
import json
d=json.loads(open("gpurun_out/r3_b7_$v.json").read().strip().splitlines()[-1])
print("big=$v", round(d["ms_per_step"],2), "ms; preroll", round(d["preroll"]["ms"],1), {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
done
