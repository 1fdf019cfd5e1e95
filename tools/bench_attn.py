"""Attention micro-benchmark through the C ABI: forward (and backward) TF/s on the SD-XL / SD-1.x self-attention shapes."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3


SHAPES = [(16, 10, 4096, 4096, 64), (16, 20, 1024, 1024, 64), (16, 10, 4096, 77, 64),
          (16, 20, 1024, 77, 64), (4, 10, 4096, 4096, 64), (4, 20, 1024, 1024, 64),
          (8, 8, 4096, 4096, 40), (8, 8, 1024, 1024, 80)]
if "--xs" in sys.argv:  # cross-attention (77 text tokens) only: SD-XL step / pre-roll batches, SD-1.x levels
    SHAPES = [(16, 10, 4096, 77, 64), (16, 20, 1024, 77, 64), (4, 10, 4096, 77, 64), (4, 20, 1024, 77, 64),
              (8, 8, 4096, 77, 40), (8, 8, 1024, 77, 80), (8, 8, 256, 77, 160), (2, 8, 4096, 77, 40)]
for (B, H, Nq, Nk, D) in SHAPES:
    q = torch.randn(B, Nq, H, D, device="cuda").half()
    k, v = (torch.randn(B, Nk, H, D, device="cuda").half() for _ in range(2))
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Nq, device="cuda")
    t = timeit(lambda: lib.smi_op_attention_fwd(0, P(q), P(k), P(v), P(o), P(lse), B, H, Nq, Nk, D, D ** -0.5, None))
    do = torch.randn_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Nq, device="cuda")
    tb = timeit(lambda: lib.smi_op_attention_bwd(0, P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), P(dk), P(dv),
                                                 P(delta), B, H, Nq, Nk, D, D ** -0.5, None))
    tq = timeit(lambda: lib.smi_op_attention_bwd(0, P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), None, None,
                                                 P(delta), B, H, Nq, Nk, D, D ** -0.5, None))
    fl = 4.0 * B * H * Nq * Nk * D
    qo = 2.0 * B * Nq * H * D * 2  # bytes: Q read + O written (forward); dQ-only backward moves Q, dO, O in and dQ out
    print(f"attn B{B} H{H} Nq{Nq} Nk{Nk} D{D}: fwd {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF/s {qo/t/1e9:6.0f} GB/s(Q+O) | "
          f"bwd(dq+dkv) {tb*1e6:8.1f} us {2.5*fl/tb/1e12:6.1f} TF/s | dq only {tq*1e6:8.1f} us {2*qo/tq/1e9:6.0f} GB/s", flush=True)
