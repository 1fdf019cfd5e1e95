"""Split-K slices + finish against the un-split kernel, every slice count the rule can pick (2..8, 16) on both slice kernels,
with and without an epilogue / a rank-r delta: fp32 outputs must agree to summation-order noise.  Runs with
SMI_SPLITK_DEBUG=1 (set here, before the library loads); tests/test_kernels_gpu.py runs it in a subprocess."""
import ctypes as C
import os
import sys
os.environ["SMI_SPLITK_DEBUG"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native

lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def main():
    scratch = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    lib.smi_op_gemm_scratch(P(scratch), scratch.numel())
    g = torch.Generator(device="cuda").manual_seed(0)
    worst = 0.0
    for M, N, K, epi, r in [(512, 1280, 1280, 1, 0), (200, 320, 2880, 1, 4), (1024, 640, 5120, 0, 8), (64, 1280, 1344, 1, 4),
                            (2048, 1280, 5120, 1, 4)]:
        a = torch.randn(M, K, device="cuda", generator=g).half()
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
        bias = torch.randn(N, device="cuda", generator=g).half() if epi else None
        res = torch.randn(M, N, device="cuda", generator=g).half() if epi else None
        xa = torch.randn(M, r, device="cuda", generator=g) if r else None
        up = torch.randn(N, r, device="cuda", generator=g) * 0.1 if r else None

        def run(S, V, f32):
            os.environ["SMI_SPLITK_S"], os.environ["SMI_SPLITK_V"] = str(S), str(V)
            c = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float32 if f32 else torch.float16)
            rc = lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.5, int(f32), None)
            torch.cuda.synchronize()
            assert rc == 0, (M, N, K, S, V)
            return c

        ref = a.float() @ w.float().t()
        if epi:
            ref = ref + bias.float() + res.float()
        if r:
            ref = ref + 0.5 * (xa @ up.t())
        base = run(1, 5, True)
        e0 = ((base - ref).abs().max() / ref.abs().max()).item()
        assert e0 < 2e-5, (M, N, K, "un-split vs torch", e0)
        base16 = run(1, 5, False)
        for S in (2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 16):
            if K // 64 // S < 1:
                continue
            for V in (5, 7):
                got = run(S, V, True)
                e = ((got - base).abs().max() / base.abs().max()).item()
                worst = max(worst, e)
                assert torch.isfinite(got).all() and e < 3e-6, (M, N, K, epi, r, S, V, e)
                got16 = run(S, V, False)
                # one 16-bit rounding of sums that differ by fp32 summation order: a rare last-bit flip
                assert (got16.float() - base16.float()).abs().max().item() <= 2.0 ** -10 * base16.float().abs().max().item() * 2
    print(f"split-K check ok: worst fp32 deviation from the un-split kernel {worst:.2e}")


if __name__ == "__main__":
    main()
