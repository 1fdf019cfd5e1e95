#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
SMI_TUNE_FILE=/tmp/tune.txt python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/r3_b13.json 2> $OUT/r3_b13.err || { tail -30 $OUT/r3_b13.err; exit 1; }
python3 - <<P
import json
d=json.loads(open("gpurun_out/r3_b13.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],2), "ms; preroll", round(d["preroll"]["ms"],1), "frac", round(d["roofline"]["frac"],3), {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
