#!/bin/bash
# round-3 run 1: attention parity + same-box A/B of the attention kernels (old build vs new) and the bench under both
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p $OUT
BASE=$PWD/sliders_conceptmod_amd/build/libsmi_hip_base.so
python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" > $OUT/r3_t1.log 2>&1 || { tail -30 $OUT/r3_t1.log; exit 1; }
tail -2 $OUT/r3_t1.log
python tools/bench_attn.py > $OUT/r3_attn_new.log 2>&1
SMI_ATTN_DQ_REMAT=1 python tools/bench_attn.py > $OUT/r3_attn_remat.log 2>&1
SMI_LIB=$BASE python tools/bench_attn.py > $OUT/r3_attn_base.log 2>&1
paste -d'\n' $OUT/r3_attn_base.log $OUT/r3_attn_new.log $OUT/r3_attn_remat.log
SMI_TUNE_FILE=/tmp/tune.txt SMI_LIB=$BASE python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r3_bench_base.json 2> $OUT/r3_bench_base.err
SMI_TUNE_FILE=/tmp/tune.txt python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r3_bench_new.json 2> $OUT/r3_bench_new.err
cp /tmp/tune.txt $OUT/r3_tune.txt
python - <<'P'
import json
for t in ("base","new"):
    d=json.loads(open(f"gpurun_out/r3_bench_{t}.json").read().strip().splitlines()[-1])
    print(t, round(d["ms_per_step"],2), {k:round(v["ms"],2) for k,v in d["kernel_classes"].items()})
P
