"""Does the ADDRESS FOOTPRINT of a pass cost time?  The engine's arenas are bump allocators: a headline step writes
~80 GB of distinct addresses.  This probe runs the same chain of launches (GEMM M x N x K with the output of one launch
the input of the next, then a LayerNorm over it) either ping-ponging between two fixed buffers or walking through a
`SMI_PROBE_GB`-sized region the way the arenas do, and prints the time per launch of both.

    python tools/probe_footprint.py            # on the GPU box"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
PTR = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
GB = float(os.environ.get("SMI_PROBE_GB", "64"))
M, N = 16384, 1280
K = N
slot = M * N * 2
region = torch.empty(int(GB * 2 ** 30), dtype=torch.uint8, device="cuda")
region.zero_()
nslot = region.numel() // slot
w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
g, b = torch.ones(N, device="cuda").half(), torch.zeros(N, device="cuda").half()
mr = torch.empty(M * 2, device="cuda")
x0 = torch.randn(M, K, device="cuda").half()
region[:slot].view(torch.float16).view(M, K).copy_(x0)


def chain(walk, n):
    i = 0
    for _ in range(n):
        j = (i + 1) % nslot if walk else 1 - (i % 2)
        k = (j + 1) % nslot if walk else 1 - j
        if walk and k == 0:
            k = 1
        _native.check(lib.smi_op_gemm(0, PTR(region, i * slot), C.c_void_p(w.data_ptr()), PTR(region, j * slot), M, N, K, None,
                                      None, None, None, 0, 0.0, 0, None), "gemm")
        _native.check(lib.smi_op_layernorm(0, PTR(region, j * slot), C.c_void_p(g.data_ptr()), C.c_void_p(b.data_ptr()),
                                           PTR(region, k * slot), None, None, C.c_void_p(mr.data_ptr()), M, N, 1e-5, None), "ln")
        i = k


for walk in (False, True, False, True):
    chain(walk, 50)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = min(1000, nslot // 2 - 2) if walk else 1000
    s.record()
    chain(walk, n)
    e.record()
    torch.cuda.synchronize()
    print(f"{'walking ' + str(int(GB)) + ' GB' if walk else 'two fixed buffers':>20s}: {s.elapsed_time(e) / n * 1e3:7.1f} us per (GEMM {M}x{N}x{K} + LayerNorm) pair, "
          f"{n} pairs", flush=True)
