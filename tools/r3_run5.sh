#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "splitk or generations" > $OUT/r3_t5a.log 2>&1 || { tail -40 $OUT/r3_t5a.log; exit 1; }
tail -2 $OUT/r3_t5a.log
for c in sd14_512_b1_r4 sd15_512_b4_r4 sdxl_1024_b2_r4; do
  SMI_PROF_DUMP=1 python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $OUT/r3_b5_$c.json 2> $OUT/r3_b5_$c.err || { tail -30 $OUT/r3_b5_$c.err; exit 1; }
done
python3 - <<'P'
import json
for n in ("sd14_512_b1_r4","sd15_512_b4_r4","sdxl_1024_b2_r4"):
    d=json.loads(open(f"gpurun_out/r3_b5_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],2), "ms; preroll", round(d["preroll"]["ms"],1), {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
python -m pytest tests -m gpu -x -q > $OUT/r3_t5b.log 2>&1 || { tail -60 $OUT/r3_t5b.log; exit 1; }
tail -3 $OUT/r3_t5b.log
