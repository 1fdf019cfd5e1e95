"""One-shape-list GEMM timing under the SMI_GEMM override of the environment (A/B arms = separate processes)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
shapes = [tuple(int(x) for x in s.split(",")) for s in os.environ.get(
    "SMI_AB_SHAPES", "8192,7680,8192;16384,1280,5120;16384,1280,1280;16384,3840,1280;16384,10240,1280").split(";")]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    f = lambda: lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, None, None, None, None, 0, 0.0, 0, None)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        s.record()
        for _ in range(5):
            f()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 5)
    print(f"SMI_GEMM={os.environ.get('SMI_GEMM','auto'):5s} {M}x{N}x{K}: {best*1e3:8.1f} us  {2*M*N*K/best/1e9:7.1f} TF/s", flush=True)
