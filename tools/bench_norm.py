"""Per-shape timing of the GroupNorm(+SiLU) / LayerNorm kernels through the C ABI (smi_op_groupnorm / smi_op_layernorm):
algorithmic bytes (x read + y written; backward: x, dy read + dx written) over the measured time.
    python tools/bench_norm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native  # noqa: E402

lib = _native.lib()
P = _native.ptr


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    dt, code = torch.float16, 0
    print("GroupNorm+SiLU (32 groups): Nb HW C | fwd us (TB/s) | fwd+bwd us | bwd TB/s")
    for nb, hw, c in [(16, 16384, 320), (16, 16384, 640), (16, 16384, 960), (16, 4096, 640), (16, 4096, 1280),
                      (16, 4096, 1920), (16, 1024, 1280), (16, 1024, 2560), (4, 16384, 320), (4, 16384, 960),
                      (4, 4096, 640), (4, 4096, 1920), (4, 1024, 1280), (4, 1024, 2560),
                      # SD-1.x at UNet batch 2 / 8 (B = 1 pre-roll / step), the small maps of every pre-roll:
                      (2, 4096, 320), (2, 1024, 640), (2, 1024, 1280), (2, 1024, 1920), (2, 256, 1280), (2, 256, 2560),
                      (2, 64, 1280), (8, 4096, 320), (8, 1024, 640), (8, 1024, 1280), (8, 256, 1280), (8, 256, 2560),
                      (8, 64, 2560)]:
        x = torch.randn(nb, hw, c, device="cuda", dtype=dt)
        dy = torch.randn_like(x)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g = torch.randn(c, device="cuda", dtype=dt)
        b = torch.randn(c, device="cuda", dtype=dt)
        scr = torch.empty(2 * nb * c + nb * 64 + nb * 256 * 64 + 2 * nb * c + 1024, device="cuda")
        f = lambda: _native.check(lib.smi_op_groupnorm(code, P(x), P(g), P(b), P(y), None, None, P(scr), nb, hw, c, 32,
                                                       1e-5, 1, None), "gn")
        fb = lambda: _native.check(lib.smi_op_groupnorm(code, P(x), P(g), P(b), P(y), P(dy), P(dx), P(scr), nb, hw, c,
                                                        32, 1e-5, 1, None), "gn")
        tf, tfb = timeit(f), timeit(fb)
        n = nb * hw * c * 2
        print(f"{nb:3d} {hw:6d} {c:5d} | {tf:8.1f} ({2 * n / tf / 1e6:5.2f}) | {tfb:8.1f} | {3 * n / max(tfb - tf, 1e-3) / 1e6:5.2f}")
    print("LayerNorm: M C | fwd us (TB/s) | fwd+bwd us | bwd TB/s")
    for m, c in [(65536, 640), (16384, 1280), (16384, 640), (4096, 1280)]:
        x = torch.randn(m, c, device="cuda", dtype=dt)
        dy = torch.randn_like(x)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        g = torch.randn(c, device="cuda", dtype=dt)
        b = torch.randn(c, device="cuda", dtype=dt)
        mr = torch.empty(2 * m, device="cuda")
        f = lambda: _native.check(lib.smi_op_layernorm(code, P(x), P(g), P(b), P(y), None, None, P(mr), m, c, 1e-5, None), "ln")
        fb = lambda: _native.check(lib.smi_op_layernorm(code, P(x), P(g), P(b), P(y), P(dy), P(dx), P(mr), m, c, 1e-5, None), "ln")
        tf, tfb = timeit(f), timeit(fb)
        n = m * c * 2
        print(f"{m:6d} {c:5d} | {tf:8.1f} ({2 * n / tf / 1e6:5.2f}) | {tfb:8.1f} | {3 * n / max(tfb - tf, 1e-3) / 1e6:5.2f}")


if __name__ == "__main__":
    main()
