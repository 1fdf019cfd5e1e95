#!/bin/bash
# Same-box A/B of two builds of the library: `bash tools/ab_lib.sh <base.so> [config] [rounds]` runs bench.py alternately
# with SMI_LIB=<base.so> and with the in-tree libsmi_hip.so and prints step / pre-roll times of every run.
cd "$(dirname "$0")/.."
BASE=$(realpath "$1"); CFG=${2:-sdxl_1024_b2_r4}; ROUNDS=${3:-2}
OUT=gpurun_out; mkdir -p $OUT
for r in $(seq 1 $ROUNDS); do
  for arm in base new; do
    if [ $arm = base ]; then export SMI_LIB=$BASE; else unset SMI_LIB; fi
    python3 bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $OUT/ab_$arm.json 2> $OUT/ab_$arm.err || { tail -20 $OUT/ab_$arm.err; exit 1; }
    python3 - $arm $OUT/ab_$arm.json <<'P'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
pre = d.get("preroll") or {}
print(f"{sys.argv[1]:5s} {d['config']['workload']}: {d['ms_per_step']:.2f} ms/step, pre-roll {pre.get('ms', float('nan')):.1f} ms (guidance 1: {pre.get('ms_at_guidance_1') or float('nan'):.1f}), frac {d['roofline']['frac']:.3f}", flush=True)
P
  done
done
