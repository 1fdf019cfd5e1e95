#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or conv" > $OUT/r3_t4.log 2>&1 || { tail -40 $OUT/r3_t4.log; exit 1; }
tail -2 $OUT/r3_t4.log
for c in sd14_512_b1_r4 sd15_512_b4_r4 sdxl_1024_b2_r4; do
  SMI_TUNE_DUMP=1 SMI_PROF_DUMP=1 python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline > $OUT/r3_b4_$c.json 2> $OUT/r3_b4_$c.err || { tail -30 $OUT/r3_b4_$c.err; exit 1; }
done
python3 - <<'P'
import json
for n in ("sd14_512_b1_r4","sd15_512_b4_r4","sdxl_1024_b2_r4"):
    d=json.loads(open(f"gpurun_out/r3_b4_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],2), "ms; preroll", round(d["preroll"]["ms"],1), {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
