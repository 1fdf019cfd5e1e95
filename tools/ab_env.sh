#!/bin/bash
# Same-box A/B of environment switches on the in-tree library: `bash tools/ab_env.sh <config> <rounds> "<env A>" "<env B>" ...`
# runs bench.py once per arm and round (arms interleaved) and prints step / pre-roll times.  An arm is a space-separated
# list of VAR=value pairs ("" = no switch).
cd "$(dirname "$0")/.."
CFG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    env $arm python3 bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $OUT/abenv_$i.json 2> $OUT/abenv_$i.err || { tail -20 $OUT/abenv_$i.err; exit 1; }
    python3 - "$arm" $OUT/abenv_$i.json <<'P'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
pre = d.get("preroll") or {}
print(f"[{sys.argv[1]:40s}] {d['config']['workload']}: {d['ms_per_step']:.2f} ms/step, pre-roll {pre.get('ms', float('nan')):.1f} ms (guidance 1: {pre.get('ms_at_guidance_1') or float('nan'):.1f})", flush=True)
P
  done
done
