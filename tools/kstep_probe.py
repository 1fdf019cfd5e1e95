import ctypes as C, sys, os, torch
sys.path.insert(0, "/root/repo")
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
def t(M,N,K):
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    c = torch.empty(M, N, device="cuda", dtype=torch.float16)
    f=lambda: lib.smi_op_gemm(0, P(a), P(w), P(c), M, N, K, None, None, None, None, 0, 0.0, 0, None)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/10*1e3
for (M,N) in [(2048,2048),(4096,2048),(4096,4096),(2048,1024)]:
    t1=t(M,N,2048); t2=t(M,N,8192)
    tiles=(M//128)*(N//128)
    print(f"{os.environ.get('SMI_GEMM','auto'):6s} M={M} N={N} tiles128={tiles}: K=2048 {t1:.1f} us, K=8192 {t2:.1f} us -> {(t2-t1)/96*1000:.0f} ns per 64-deep K-step, fixed {t1-(t2-t1)/96*32:.1f} us", flush=True)
