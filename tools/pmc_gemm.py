"""A few GEMM launches for rocprofv3 --pmc runs (SMI_GEMM selects the kernel generation: 8ph = gemm3, 5ph = gemm4).
SMI_PMC_SHAPES="M,N,K,epi;..." overrides the shape list (epi = 1 adds bias + residual)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
shapes = [(8192, 8192, 8192, 0), (16384, 10240, 1280, 0), (16384, 3840, 1280, 1)]
if os.environ.get("SMI_PMC_SHAPES"):
    shapes = [tuple(int(x) for x in s.split(",")) for s in os.environ["SMI_PMC_SHAPES"].split(";")]
for (M, N, K, epi) in shapes:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    bias = torch.randn(N, device="cuda").half() if epi else None
    res = torch.randn(M, N, device="cuda").half() if epi else None
    for _ in range(3):
        lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, P(bias), P(res), None, None, 0, 0.0, 0, None)
    torch.cuda.synchronize()
