#!/bin/bash
# The measurement set of a round (run through gpurun; usage: bash tools/measure_round.sh <tag> [part]): profiles of the
# headline config + bench records of every BASELINE configuration (part 1), per-kernel counters + CLI timing (part 2).
# Two parts because one gpurun call is limited to 20 minutes.
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
TAG=${1:-r04}
PART=${2:-1}
if [ "$PART" = "1" ]; then
  bash tools/collect_profiles.sh $TAG sdxl_1024_b2_r4 > $OUT/${TAG}_collect.log 2>&1 || { tail -30 $OUT/${TAG}_collect.log; exit 1; }
  tail -4 $OUT/${TAG}_collect.log
  for c in sdxl_1024_b2_r8 sdxl_1024_b2_r4_dora sd15_512_b4_r4 sd14_512_b1_r4 sd14_512_b1_r4_c3lier image_sdxl_1024_b1_r4; do
    extra=""; [ "$c" != "sd14_512_b1_r4" ] && extra="--no-cpu-baseline"
    python3 bench.py --config $c --steps 8 --warmup 2 $extra > $OUT/${TAG}_bench_$c.json 2> $OUT/${TAG}_bench_$c.err || { tail -30 $OUT/${TAG}_bench_$c.err; exit 1; }
  done
  python3 - <<P
import json
for n in ("sdxl_1024_b2_r8","sdxl_1024_b2_r4_dora","sd15_512_b4_r4","sd14_512_b1_r4","sd14_512_b1_r4_c3lier","image_sdxl_1024_b1_r4"):
    d=json.loads(open(f"gpurun_out/${TAG}_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],2), "ms", round(d["value"],2), "steps/s frac", round(d["roofline"]["frac"],3), "preroll", (d.get("preroll") or {}).get("ms"))
P
else
  bash tools/pmc_attn.sh $TAG > /dev/null 2>&1 || true
  if [ -z "$SKIP_PMC" ]; then  # SKIP_PMC=1: no GEMM counter passes (GEMM kernels unchanged since the last set)
    bash tools/pmc_gemm_small.sh $TAG > /dev/null 2>&1 || true
    bash tools/pmc_gemm.sh $TAG > /dev/null 2>&1 || true
  fi
  python3 tools/cli_timing.py > $OUT/${TAG}_cli_timing.jsonl 2> $OUT/${TAG}_cli_timing.err || { tail -20 $OUT/${TAG}_cli_timing.err; exit 1; }
  cat $OUT/${TAG}_cli_timing.jsonl
fi
