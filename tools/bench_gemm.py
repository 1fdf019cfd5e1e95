"""GEMM / conv micro-benchmark through the C ABI (smi_op_gemm / smi_op_conv3x3): TFLOP/s per shape."""
import ctypes as C
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native

lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gemm(M, N, K, epi=False, dt=torch.float16):
    a = torch.randn(M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    bias = torch.randn(N, device="cuda").to(dt) if epi else None
    res = torch.randn(M, N, device="cuda").to(dt) if epi else None
    t = timeit(lambda: lib.smi_op_gemm(0 if dt == torch.float16 else 1, P(a), P(w), P(c), M, N, K, P(bias), P(res),
                                       None, None, 0, 0.0, 0, None))
    print(f"gemm M={M:6d} N={N:6d} K={K:5d} epi={int(epi)}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)


def conv(nb, H, Cin, Cout, dt=torch.float16):
    x = torch.randn(nb, H, H, Cin, device="cuda").to(dt)
    w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt)
    b = torch.randn(Cout, device="cuda").to(dt)
    y = torch.empty(nb, H, H, Cout, device="cuda", dtype=dt)
    t = timeit(lambda: lib.smi_op_conv3x3(0, P(x), P(w), P(b), P(y), nb, H, H, Cin, Cout, 1, 0, 0, H, H, None))
    fl = 2.0 * nb * H * H * Cout * 9 * Cin
    print(f"conv nb={nb} {H}x{H} Cin={Cin:5d} Cout={Cout:5d}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    gemm(4096, 4096, 4096)
    gemm(8192, 8192, 8192)
    # batched 4-pass step, SD-XL 1024^2 B=2: 16 samples -> 65536 rows at 640 channels, 16384 rows at 1280
    for M, N, K in [(65536, 1920, 640), (65536, 640, 640), (65536, 5120, 640), (65536, 640, 2560),
                    (16384, 3840, 1280), (16384, 1280, 1280), (16384, 10240, 1280), (16384, 1280, 5120),
                    (4096, 1280, 1280), (4096, 1280, 5120), (1232, 2560, 2048)]:
        gemm(M, N, K)
        gemm(M, N, K, epi=True)
    for nb, H, Cin, Cout in [(16, 128, 320, 320), (16, 64, 640, 640), (16, 32, 1280, 1280), (16, 64, 1920, 640),
                             (16, 128, 960, 320), (16, 32, 2560, 1280)]:
        conv(nb, H, Cin, Cout)
