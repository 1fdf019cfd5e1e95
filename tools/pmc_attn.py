"""Attention forward + backward launches for rocprofv3 --pmc runs (SD-XL level-1 / level-2 self-attention, 16 and 4 samples)."""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
for (B, H, N, D) in [(16, 10, 4096, 64), (16, 20, 1024, 64), (4, 10, 4096, 64), (4, 20, 1024, 64)]:
    q, k, v = (torch.randn(B, N, H, D, device="cuda").half() for _ in range(3))
    o = torch.empty_like(q); lse = torch.empty(B, H, N, device="cuda")
    do = torch.randn_like(q); dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, N, device="cuda")
    for _ in range(3):
        lib.smi_op_attention_fwd(0, P(q), P(k), P(v), P(o), P(lse), B, H, N, N, D, D ** -0.5, None)
    if B == 4:
        for _ in range(3):
            lib.smi_op_attention_bwd(0, P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), P(dk), P(dv), P(delta), B, H, N, N, D, D ** -0.5, None)
    torch.cuda.synchronize()

# cross-attention on the 77 text tokens (round 4: the single-pass kernels attn_xs_fwd / attn_xs_bwd_dq): forward at the step's
# 16 samples, dQ-only backward (noxattn: no dK / dV) at its 4 adapted ones
for (B, H, N, D) in [(16, 10, 4096, 64), (16, 20, 1024, 64), (4, 10, 4096, 64), (4, 20, 1024, 64)]:
    q = torch.randn(B, N, H, D, device="cuda").half()
    k, v = (torch.randn(B, 77, H, D, device="cuda").half() for _ in range(2))
    o = torch.empty_like(q); lse = torch.empty(B, H, N, device="cuda")
    do = torch.randn_like(q); dq = torch.empty_like(q); delta = torch.empty(B, H, N, device="cuda")
    for _ in range(3):
        lib.smi_op_attention_fwd(0, P(q), P(k), P(v), P(o), P(lse), B, H, N, 77, D, D ** -0.5, None)
    if B == 4:
        for _ in range(3):
            lib.smi_op_attention_bwd(0, P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), None, None, P(delta), B, H, N, 77, D, D ** -0.5, None)
    torch.cuda.synchronize()
