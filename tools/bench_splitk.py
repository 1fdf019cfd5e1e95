"""Split-K rule probe: for the small-grid GEMM shapes of the pre-roll / SD-1.x passes, time every (slice count, slice
kernel) pair against the library's own choice.  Needs SMI_SPLITK_DEBUG=1 in the environment (set below before the library
loads): launch_gemm then takes the slice count and the gemm2 variant of each call from SMI_SPLITK_S / SMI_SPLITK_V.

    python tools/bench_splitk.py [xl2|xl4|sd14|q|all]   (SPLITK_SLICES=1,2,3,4,6 SPLITK_VARIANTS=5,7 narrow the sweep)
"""
import ctypes as C
import os
import sys
os.environ["SMI_SPLITK_DEBUG"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native

lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())

# (M, N, K, epilogue, rank): epilogue 0 none, 1 bias + residual
SHAPES = {
    "xl2": [(2048, 1280, 1280, 0, 0), (2048, 1280, 1280, 1, 0), (2048, 1280, 1280, 1, 4), (2048, 3840, 1280, 0, 4),
            (2048, 1280, 5120, 1, 0), (8192, 640, 640, 1, 0), (8192, 1920, 640, 0, 4), (8192, 640, 2560, 1, 0)],
    "xl4": [(4096, 1280, 1280, 0, 0), (4096, 1280, 1280, 1, 0), (4096, 1280, 1280, 1, 4), (4096, 3840, 1280, 0, 4),
            (4096, 1280, 5120, 1, 0), (16384, 640, 640, 1, 0), (16384, 640, 2560, 1, 0)],
    "q": [(2048, 1280, 5120, 1, 0), (2048, 1280, 5120, 1, 4), (4096, 1280, 5120, 1, 0), (2048, 640, 2560, 1, 0),
          (512, 1280, 5120, 1, 0), (1024, 1280, 5120, 1, 0), (1024, 1280, 1280, 1, 0)],
    "sd14": [(8192, 320, 320, 1, 0), (8192, 960, 320, 0, 4), (8192, 320, 1280, 1, 0), (2048, 640, 640, 1, 0),
             (2048, 1920, 640, 0, 4), (2048, 640, 2560, 1, 0), (512, 1280, 1280, 1, 0), (512, 3840, 1280, 0, 4),
             (512, 1280, 5120, 1, 0), (128, 1280, 1280, 1, 0)],
}


SLICES = tuple(int(x) for x in os.environ.get("SPLITK_SLICES", "1,2,4,8").split(","))
VARIANTS = tuple(int(x) for x in os.environ.get("SPLITK_VARIANTS", "5,7,10,12,11").split(","))


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    scratch = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    lib.smi_op_gemm_scratch(P(scratch), scratch.numel())
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, shapes in SHAPES.items():
        if which not in ("all", name):
            continue
        for M, N, K, epi, r in shapes:
            a = torch.randn(M, K, device="cuda", generator=g).half()
            w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
            bias = torch.randn(N, device="cuda", generator=g).half() if epi else None
            res = torch.randn(M, N, device="cuda", generator=g).half() if epi else None
            xa = torch.randn(M, r, device="cuda", generator=g) if r else None
            up = torch.randn(N, r, device="cuda", generator=g) * 0.1 if r else None
            outs = {}

            def run(S, V):
                os.environ["SMI_SPLITK_S"], os.environ["SMI_SPLITK_V"] = str(S), str(V)
                c = torch.empty(M, N, device="cuda", dtype=torch.float16)
                fn = lambda: lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.5, 0, None)
                rc = fn()
                torch.cuda.synchronize()
                if rc != 0:
                    return None, None
                return timeit(fn), c

            t_def, c_def = run(0, 0)
            line = [f"{name} M={M:5d} N={N:4d} K={K:4d} epi={epi} r={r}: default {t_def:6.1f} us |"]
            best = (t_def, "default")
            for S in SLICES:
                if S > 1 and K // 64 // S < 2:
                    continue
                for V in VARIANTS:
                    if V in (10, 12, 11) and N % 160:
                        continue
                    if V == 11 and S > 1:
                        continue
                    t, c = run(S, V)
                    if t is None:
                        continue
                    err = (c.float() - c_def.float()).abs().max().item()
                    line.append(f" S{S}v{V} {t:5.1f}" + ("" if err < 0.05 else f"(!{err:.2g})"))
                    if t < best[0]:
                        best = (t, f"S{S}v{V}")
            line.append(f" | best {best[1]} {best[0]:.1f} us ({2 * M * N * K / best[0] / 1e6:.0f} TF/s)")
            print("".join(line), flush=True)


if __name__ == "__main__":
    main()
