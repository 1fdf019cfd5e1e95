"""ff1 through the fused-GEGLU epilogue vs the same GEMM with a plain bias epilogue (what the gate costs)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K) in [(16384, 10240, 1280), (65536, 5120, 640)]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda").half()
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    out = torch.empty(M, N // 2, device="cuda", dtype=torch.float16)
    t0 = timeit(lambda: lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, P(b), None, None, None, 0, 0.0, 0, None))
    t1 = timeit(lambda: lib.smi_op_gemm_geglu(0, P(a), P(w), P(b), P(out), P(c), M, N, K, M, None))
    t2 = timeit(lambda: lib.smi_op_gemm_geglu(0, P(a), P(w), P(b), P(out), P(c), M, N, K, 3 * M // 4, None))
    print(f"{M}x{N}x{K}: gemm+bias {t0:.1f} us | fused geglu (no proj rows) {t1:.1f} us | with last quarter of proj kept {t2:.1f} us")
