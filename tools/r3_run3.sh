#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
bash tools/collect_profiles.sh r03b sdxl_1024_b2_r4 > $OUT/r3_collect.log 2>&1 || { tail -30 $OUT/r3_collect.log; exit 1; }
tail -20 $OUT/r3_collect.log
python3 bench.py --config image_sdxl_1024_b1_r4 --steps 6 --warmup 2 > $OUT/r03b_bench_image_sdxl_1024_b1_r4.json 2> $OUT/r03b_bench_image.err || { tail -30 $OUT/r03b_bench_image.err; exit 1; }
cut -c1-400 $OUT/r03b_bench_image_sdxl_1024_b1_r4.json
python3 bench.py --config sd14_512_b1_r4 --steps 10 --warmup 2 > $OUT/r03b_bench_sd14_512_b1_r4.json 2> $OUT/r03b_bench_sd14.err || { tail -30 $OUT/r03b_bench_sd14.err; exit 1; }
python3 bench.py --config sd14_512_b1_r4_c3lier --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r03b_bench_sd14_512_b1_r4_c3lier.json 2> $OUT/r03b_bench_sd14c.err || { tail -30 $OUT/r03b_bench_sd14c.err; exit 1; }
python3 - <<'P'
import json
for n in ("sd14_512_b1_r4","sd14_512_b1_r4_c3lier","image_sdxl_1024_b1_r4"):
    d=json.loads(open(f"gpurun_out/r03b_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],2), "ms", round(d["value"],2), "steps/s frac", round(d["roofline"]["frac"],3), (d.get("cpu_baseline") or {}).get("value"))
P
