#!/bin/bash
# A/B of the one-launch GroupNorm (SMI_GN_FUSED_MAX=0 turns it off): kernel tests, then SD-1.x benches interleaved.
python -m pytest tests/test_kernels_gpu.py -x -q -k "groupnorm" > gpurun_out/r4_gn_tests.log 2>&1; tail -2 gpurun_out/r4_gn_tests.log
for r in 1 2; do for v in 0 98304; do for c in sd14_512_b1_r4 sd15_512_b4_r4; do SMI_GN_FUSED_MAX=$v python3 bench.py --config $c --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c GN_FUSED_MAX=$v', round(d['ms_per_step'],2), round(d['preroll']['ms'],1), d['kernel_classes']['norm']['ms'])"; done; done; done 2>&1 | tee gpurun_out/r4_gn_ab.log
