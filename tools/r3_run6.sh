#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out
python -m pytest tests/test_kernels_gpu.py tests/test_clip_gpu.py -q -x -k "attention or clip_text" > $OUT/r3_t6.log 2>&1 || { tail -40 $OUT/r3_t6.log; exit 1; }
tail -2 $OUT/r3_t6.log
python tools/bench_attn.py > $OUT/r3_attn_db.log 2>&1
git_base=$PWD/sliders_conceptmod_amd/build/libsmi_hip_base.so
SMI_LIB=$git_base python tools/bench_attn.py > $OUT/r3_attn_base2.log 2>&1
paste -d'\n' $OUT/r3_attn_base2.log $OUT/r3_attn_db.log | grep -v amdgpu.ids
