"""SHA-256 of the outputs of one op per tuned-key class (dense + full LoRA epilogue, fused q|k|v LoRA shape, fused GEGLU,
3x3 conv with bias + residual, fp32 output) under whatever SMI_GEMM the process was started with: the tile autotuner
may pick any generation for a key, which is only sound while all of them give the same bits
(tests/test_kernels_gpu.py::test_gemm_generations_are_bit_identical_per_epilogue_class)."""
import ctypes as C
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native  # noqa: E402

lib = _native.lib()
P = _native.ptr


def dig(t):
    torch.cuda.synchronize()
    return hashlib.sha256(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:16]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc)
    out = {}
    ws = torch.empty(512 * 128 * 128, device="cuda", dtype=torch.float32)  # split-K scratch, as the engine lends it
    _native.check(lib.smi_op_gemm_scratch(P(ws), ws.numel() * 4), "scratch")
    for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
        tag = "f16" if code == 0 else "bf16"
        # shapes under the split-K rule: the override must not change their bits either
        for M, N, K, r in ((128, 1280, 11520, 4), (512, 1280, 5120, 0)):
            a, w = rn(M, K).to(dt), rn(N, K, sc=K ** -0.5).to(dt)
            bias, res = rn(N).to(dt), rn(M, N).to(dt)
            xa, up = rn(M, max(r, 1)), rn(N, max(r, 1))
            c = torch.empty(M, N, device="cuda", dtype=dt)
            _native.check(lib.smi_op_gemm(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa) if r else None,
                                          P(up) if r else None, r, 0.25, 0, None), "gemm")
            out[f"{tag}/splitk/{M}x{N}x{K}r{r}"] = dig(c)
        for M, N, K, r in ((4096, 1280, 1280, 4), (16384, 3840, 1280, 4), (2048, 640, 2560, 8)):
            a, w = rn(M, K).to(dt), rn(N, K, sc=K ** -0.5).to(dt)
            bias, res = rn(N).to(dt), rn(M, N).to(dt)
            xa, up = rn(M, r), rn(N, r)
            c = torch.empty(M, N, device="cuda", dtype=dt)
            _native.check(lib.smi_op_gemm(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.25, 0, None), "gemm")
            out[f"{tag}/lora_full_epilogue/{M}x{N}x{K}r{r}"] = dig(c)
            c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
            _native.check(lib.smi_op_gemm(code, P(a), P(w), P(c32), M, N, K, P(bias), None, None, None, 0, 0.0, 1, None), "gemm")
            out[f"{tag}/f32_out/{M}x{N}x{K}"] = dig(c32)
        # batched-pass form (delta on the last quarter of the rows; gemm4 interleaves frozen / adapted rows per tile)
        for M, N, K, r, seg in ((16384, 1280, 1280, 4, 0), (16384, 3840, 1280, 4, 1280), (4096, 1920, 640, 8, 640)):
            a, w = rn(M, K).to(dt), rn(N, K, sc=K ** -0.5).to(dt)
            bias, res = rn(N).to(dt), rn(M, N).to(dt)
            row0 = 3 * M // 4
            xa, up = rn(M - row0, r * (N // seg if seg else 1)), rn(N, r)
            c = torch.empty(M, N, device="cuda", dtype=dt)
            _native.check(lib.smi_op_gemm_rows(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.25,
                                               row0, seg, None), "gemm_rows")
            out[f"{tag}/lora_rows/{M}x{N}x{K}r{r}s{seg}"] = dig(c)
        for M, N, K in ((16384, 10240, 1280), (4096, 5120, 640)):
            a, w, bias = rn(M, K).to(dt), rn(N, K, sc=K ** -0.5).to(dt), rn(N).to(dt)
            o = torch.empty(M, N // 2, device="cuda", dtype=dt)
            pj = torch.empty(M, N, device="cuda", dtype=dt)
            _native.check(lib.smi_op_gemm_geglu(code, P(a), P(w), P(bias), P(o), P(pj), M, N, K, M // 2, None), "geglu")
            out[f"{tag}/geglu/{M}x{N}x{K}"] = dig(o) + dig(pj[M // 2:])
        for nb, H, ci, co in ((4, 32, 1280, 1280), (2, 64, 640, 320), (4, 64, 320, 640)):
            x = rn(nb, H, H, ci).to(dt)
            w = rn(co, 9 * ci, sc=(9 * ci) ** -0.5).to(dt)
            b = rn(co).to(dt)
            y = torch.empty(nb, H, H, co, device="cuda", dtype=dt)
            _native.check(lib.smi_op_conv3x3(code, P(x), P(w), P(b), P(y), nb, H, H, ci, co, 1, 0, 0, H, H, None), "conv")
            out[f"{tag}/conv/{nb}x{H}x{ci}->{co}"] = dig(y)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
