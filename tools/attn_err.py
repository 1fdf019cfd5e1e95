"""Error of the attention kernels against fp32 torch on the same 16-bit inputs (rel-L2 of O, lse, dQ, dK, dV): run once per
kernel selection (SMI_ATTN_XS=0|1, SMI_ATTN_BWD_FUSED=0|1) to compare their rounding noise."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
rel = lambda a, b: float((a.float() - b).norm() / b.norm())
for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
    for (B, H, Nq, Nk, D) in [(2, 8, 1024, 77, 64), (2, 8, 256, 77, 40), (2, 8, 1024, 1024, 64)]:
        g = torch.Generator(device="cuda").manual_seed(0)
        q, k, v, do = (torch.randn(B, n, H, D, device="cuda", generator=g).to(dt) for n in (Nq, Nk, Nk, Nq))
        o = torch.empty_like(q); lse = torch.empty(B, H, Nq, device="cuda")
        lib.smi_op_attention_fwd(code, P(q), P(k), P(v), P(o), P(lse), B, H, Nq, Nk, D, D ** -0.5, None)
        qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
        s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * D ** -0.5
        oref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, -1), vf)
        gq, gk, gv = torch.autograd.grad(oref, (qf, kf, vf), do.float())
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        delta = torch.empty(B, H, Nq, device="cuda")
        lib.smi_op_attention_bwd(code, P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), P(dk), P(dv), P(delta), B, H, Nq, Nk, D, D ** -0.5, None)
        print(f"{str(dt)[6:]:9s} Nq{Nq} Nk{Nk} D{D}: O {rel(o, oref.detach()):.3e} lse {float((lse - torch.logsumexp(s, -1)).abs().max()):.2e} "
              f"dQ {rel(dq, gq):.3e} dK {rel(dk, gk):.3e} dV {rel(dv, gv):.3e}", flush=True)
