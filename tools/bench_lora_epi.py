"""Cost of the fused LoRA epilogue: the same GEMM through smi_op_gemm with and without the rank-r delta (all rows
adapted -- in the engine only the last quarter of the forward rows is)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
from tools.bench_gemm import timeit, P  # noqa: E402

lib = _native.lib()

for M, N, K in [(4096, 1280, 1280), (4096, 3840, 1280), (16384, 3840, 1280), (16384, 1280, 1280), (4096, 1280, 3840)]:
    for r in (4, 8):
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        c = torch.empty(M, N, device="cuda", dtype=torch.float16)
        xa = torch.randn(M, r, device="cuda")
        up = torch.randn(N, r, device="cuda")
        t0 = timeit(lambda: lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, None, None, None, None, 0, 0.0, 0, None))
        t1 = timeit(lambda: lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, None, None, P(xa), P(up), r, 0.25, 0, None))
        print(f"M={M:6d} N={N:5d} K={K:5d} r={r}: plain {t0*1e6:7.1f} us, +LoRA epilogue {t1*1e6:7.1f} us "
              f"(+{(t1-t0)*1e6:6.1f} us, {100*(t1/t0-1):4.0f} %)", flush=True)
