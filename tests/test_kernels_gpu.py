"""Per-kernel parity on a real MI355X: every HIP kernel, called through the C ABI (include/smi.h `smi_op_*`),
against a plain PyTorch fp32 reference of the same op computed from the SAME 16-bit-rounded inputs.

Tolerance convention (written per test): outputs are stored in fp16/bf16, so the comparison allows one storage
rounding of the result (2^-11 rel. for fp16, 2^-8 for bf16) on top of fp32 accumulation-order noise."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float16, torch.bfloat16]
EPS = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


@pytest.fixture(scope="module")
def lib():
    from sliders_conceptmod_amd import _native
    assert torch.cuda.is_available(), "these tests need the GPU box"
    return _native.lib()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def chk(lib, rc):
    assert rc == 0, lib.smi_last_error().decode()


def dcode(dt):
    return 0 if dt == torch.float16 else 1


def rnd(*shape, dt, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(dt)


def close(a, b, dt, mult=4.0, what=""):
    a, b = a.float(), b.float()
    tol = EPS[dt] * mult
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
    assert err <= tol * (ref + 1e-6) and rel <= tol, f"{what}: max err {err:.3e} (ref max {ref:.3e}), rel-norm {rel:.3e}, tol {tol:.2e}"


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 320), (4096, 640, 640), (77 * 2, 1280, 2048),
                                   (2, 1280, 320), (1000, 4, 2880), (300, 1284, 72)])
def test_gemm_plain_and_tails(lib, dt, M, N, K):
    a, w = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c), M, N, K, None, None, None, None, 0, 0.0, 0, None))
    close(c, a.float() @ w.float().t(), dt, what="gemm")


@pytest.mark.parametrize("dt", DT)
def test_gemm_asymmetric_identity_catches_transposed_writes(lib, dt):
    """A = I with an asymmetric W: a row/col swap in the C fragment map cannot hide (cdna guide section 3)."""
    M = N = K = 128
    a = torch.eye(M, device="cuda").to(dt)
    w = (torch.arange(N, device="cuda")[:, None] * 3 + torch.arange(K, device="cuda")[None, :] * 0.25).to(dt)
    c = torch.empty(M, N, device="cuda", dtype=dt)
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c), M, N, K, None, None, None, None, 0, 0.0, 0, None))
    assert torch.equal(c.float(), w.float().t())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("r", [4, 8, 3])
def test_gemm_full_epilogue(lib, dt, r):
    M, N, K = 520, 640, 320
    a, w = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    bias, res = rnd(N, dt=dt, seed=3), rnd(M, N, dt=dt, seed=4)
    xa = torch.randn(M, r, device="cuda")
    up = torch.randn(N, r, device="cuda") * 0.1
    s = 0.375
    c = torch.empty(M, N, device="cuda", dtype=dt)
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, s, 0, None))
    ref = a.float() @ w.float().t() + bias.float() + res.float() + s * (xa @ up.t())
    close(c, ref, dt, what="gemm+bias+res+lora")
    c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c32), M, N, K, P(bias), None, None, None, 0, 0.0, 1, None))
    torch.testing.assert_close(c32, a.float() @ w.float().t() + bias.float(), rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("dt", DT)
def test_splitk_rule_gemm_and_conv(lib, dt):
    """Split-K by rule (csrc/gemm.hip: few output tiles behind a long K loop -- SD-1.x at batch 1-2, the deepest UNet level,
    the time-embedding MLPs): fp32 slabs per K-slice + one finish kernel with the full epilogue.  Against fp32 torch on the
    same rounded inputs, with and without the scratch (the unsplit kernels), bit-reproducible run to run, every epilogue
    term, dense and conv (plain, stride 2, transposed gradient)."""
    ws = torch.empty(512 * 128 * 128, device="cuda", dtype=torch.float32)
    try:
        for (M, N, K, r) in ((128, 1280, 11520, 4), (512, 1280, 10240, 0), (16, 1280, 1280, 0), (2048, 640, 5120, 8),
                             (136, 324, 2048, 0)):
            a, w = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
            bias, res = rnd(N, dt=dt, seed=3), rnd(M, N, dt=dt, seed=4)
            xa = torch.randn(M, max(r, 1), device="cuda") if r else None
            up = torch.randn(N, max(r, 1), device="cuda") * 0.1 if r else None
            ref = a.float() @ w.float().t() + bias.float() + res.float() + (0.375 * (xa @ up.t()) if r else 0.0)
            outs = []
            for scratch in (True, False, True):
                chk(lib, lib.smi_op_gemm_scratch(P(ws) if scratch else None, ws.numel() * 4 if scratch else 0))
                c = torch.empty(M, N, device="cuda", dtype=dt)
                chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.375, 0,
                                         None))
                close(c, ref, dt, what=f"gemm {M}x{N}x{K} scratch={scratch}")
                outs.append(c)
            assert torch.equal(outs[0], outs[2]), "split-K result differs between two runs"
            chk(lib, lib.smi_op_gemm_scratch(P(ws), ws.numel() * 4))
            c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
            chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c32), M, N, K, P(bias), None, None, None, 0, 0.0, 1, None))
            torch.testing.assert_close(c32, a.float() @ w.float().t() + bias.float(), rtol=2e-5, atol=5e-4)
        for (nb, H, Cin, Cout) in ((2, 8, 1280, 1280), (1, 16, 1280, 640), (2, 8, 2560, 1280)):
            x = rnd(nb, H, H, Cin, dt=dt, seed=1)
            w = rnd(Cout, Cin, 3, 3, dt=dt, scale=(9 * Cin) ** -0.5, seed=2)
            b = rnd(Cout, dt=dt, seed=3)
            xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
            for stride in (1, 2):
                Ho = (H + 1) // 2 if stride == 2 else H
                ref = F.conv2d(xr, w.float(), b.float(), stride=stride, padding=1)
                gy = rnd(nb, Ho, Ho, Cout, dt=dt, seed=5)
                (gref,) = torch.autograd.grad(ref, xr, gy.float().permute(0, 3, 1, 2))
                for scratch in (True, False):
                    chk(lib, lib.smi_op_gemm_scratch(P(ws) if scratch else None, ws.numel() * 4 if scratch else 0))
                    y = torch.empty(nb, Ho, Ho, Cout, device="cuda", dtype=dt)
                    chk(lib, lib.smi_op_conv3x3(dcode(dt), P(x), P(pack_fwd(w)), P(b), P(y), nb, H, H, Cin, Cout, stride, 0,
                                                0, Ho, Ho, None))
                    close(y.permute(0, 3, 1, 2), ref, dt, what=f"conv s{stride} scratch={scratch}")
                    gx = torch.empty(nb, H, H, Cin, device="cuda", dtype=dt)
                    chk(lib, lib.smi_op_conv3x3(dcode(dt), P(gy), P(pack_grad(w, stride == 1)), None, P(gx), nb, Ho, Ho,
                                                Cout, Cin, stride, 0, 1 if stride == 2 else 0, H, H, None))
                    close(gx.permute(0, 3, 1, 2), gref, dt, mult=8, what=f"conv s{stride} dX scratch={scratch}")
    finally:
        chk(lib, lib.smi_op_gemm_scratch(None, 0))


# ---- shapes with >= 512 tiles of 256 x 256 run on the 8-phase kernel (gemm3.hip); the smaller ones above on gemm2.hip
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(8192, 4096, 64), (8192, 4096, 192), (8200, 4104, 320), (16384, 2048, 1280)])
def test_gemm_8phase_tile_epilogues_and_tails(lib, dt, M, N, K):
    a, w = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    bias, res = rnd(N, dt=dt, seed=3), rnd(M, N, dt=dt, seed=4)
    r, s = (8 if K == 192 else 4), 0.375  # rank 4 and rank 8 both take the vectorised LoRA epilogue
    xa = torch.randn(M, r, device="cuda")
    up = torch.randn(N, r, device="cuda") * 0.1
    c = torch.empty(M, N, device="cuda", dtype=dt)
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c), M, N, K, None, None, None, None, 0, 0.0, 0, None))
    base = a.float() @ w.float().t()
    close(c, base, dt, what="gemm")
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, s, 0, None))
    close(c, base + bias.float() + res.float() + s * (xa @ up.t()), dt, what="gemm+bias+res+lora")
    c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c32), M, N, K, P(bias), None, None, None, 0, 0.0, 1, None))
    torch.testing.assert_close(c32, base + bias.float(), rtol=2e-5, atol=2e-4)
    # the staggered-wave / counted-vmcnt schedule must be deterministic: identical bits run after run
    c2 = torch.empty_like(c)
    for _ in range(5):
        chk(lib, lib.smi_op_gemm(dcode(dt), P(a), P(w), P(c2), M, N, K, P(bias), P(res), P(xa), P(up), r, s, 0, None))
        assert torch.equal(c, c2)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(520, 1024, 320), (4096, 5120, 640), (16384, 10240, 320), (8200, 5120, 128)])
def test_gemm_fused_geglu_epilogue(lib, dt, M, N, K):
    """Both tile sizes (128-row v2 below 512 big tiles, 256x256 8-phase above) against proj -> hidden * gelu(gate)."""
    a, w, bias = rnd(M, K, dt=dt, seed=1), rnd(N, K, dt=dt, scale=K ** -0.5, seed=2), rnd(N, dt=dt, seed=3)
    row0 = (M // 3) & ~7
    out = torch.empty(M, N // 2, device="cuda", dtype=dt)
    proj = torch.full((M, N), 7.0, device="cuda", dtype=dt)
    chk(lib, lib.smi_op_gemm_geglu(dcode(dt), P(a), P(w), P(bias), P(out), P(proj), M, N, K, row0, None))
    pref = a.float() @ w.float().t() + bias.float()
    close(proj[row0:], pref[row0:], dt, what="geglu: kept projection rows")
    assert bool((proj[:row0] == 7.0).all()), "projection rows below proj_row0 must not be written"
    pq = pref.to(dt).float()  # the gate is applied to the 16-bit-rounded projection, as the unfused path does
    close(out, pq[:, :N // 2] * F.gelu(pq[:, N // 2:]), dt, what="geglu out")


def pack_fwd(w):  # [Cout, Cin, 3, 3] -> [Cout, 9*Cin] in (ky, kx, ci) order
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


def pack_grad(w, flip):  # -> [Cin, 9*Cout], taps flipped for stride-1 gradients
    wt = w.flip(2, 3) if flip else w
    return wt.permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nb,H,W,Cin,Cout", [(2, 16, 16, 64, 128), (1, 12, 20, 128, 64), (3, 8, 8, 320, 320),
                                             (9, 60, 62, 64, 1024)])  # last: >= 512 tiles of 256x256 -> gemm3.hip
def test_conv3x3_forward_and_input_gradient(lib, dt, nb, H, W, Cin, Cout):
    x = rnd(nb, H, W, Cin, dt=dt, seed=1)  # NHWC
    w = rnd(Cout, Cin, 3, 3, dt=dt, scale=(9 * Cin) ** -0.5, seed=2)
    b = rnd(Cout, dt=dt, seed=3)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    for stride, ups in [(1, 0), (2, 0), (1, 1)]:
        Ho = (H + 1) // 2 if stride == 2 else (2 * H if ups else H)
        Wo = (W + 1) // 2 if stride == 2 else (2 * W if ups else W)
        y = torch.empty(nb, Ho, Wo, Cout, device="cuda", dtype=dt)
        chk(lib, lib.smi_op_conv3x3(dcode(dt), P(x), P(pack_fwd(w)), P(b), P(y), nb, H, W, Cin, Cout, stride, ups, 0,
                                    Ho, Wo, None))
        xin = F.interpolate(xr, scale_factor=2.0, mode="nearest") if ups else xr
        ref = F.conv2d(xin, w.float(), b.float(), stride=stride, padding=1)
        close(y.permute(0, 3, 1, 2), ref, dt, what=f"conv s{stride} u{ups}")
        # input gradient through the same kernel: flipped/transposed filters; stride 2 uses the transposed gather
        gy = rnd(nb, Ho, Wo, Cout, dt=dt, seed=5)
        (gref,) = torch.autograd.grad(ref, xr, gy.float().permute(0, 3, 1, 2))
        if ups:
            gu = torch.empty(nb, Ho, Wo, Cin, device="cuda", dtype=dt)
            chk(lib, lib.smi_op_conv3x3(dcode(dt), P(gy), P(pack_grad(w, True)), None, P(gu), nb, Ho, Wo, Cout, Cin, 1,
                                        0, 0, Ho, Wo, None))
            gx = gu.float().view(nb, H, 2, W, 2, Cin).sum(dim=(2, 4))
            close(gx.permute(0, 3, 1, 2), gref, dt, mult=8, what="conv-upsample dX")
        else:
            gx = torch.empty(nb, H, W, Cin, device="cuda", dtype=dt)
            chk(lib, lib.smi_op_conv3x3(dcode(dt), P(gy), P(pack_grad(w, stride == 1)), None, P(gx), nb, Ho, Wo, Cout,
                                        Cin, stride, 0, 1 if stride == 2 else 0, H, W, None))
            close(gx.permute(0, 3, 1, 2), gref, dt, what=f"conv s{stride} dX")


def attn_ref(q, k, v, scale):
    s = torch.einsum("bqhd,bkhd->bhqk", q, k) * scale
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhqk,bkhd->bqhd", p, v), torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 4, 256, 256, 64), (1, 2, 64, 77, 40), (2, 3, 320, 77, 64),
                                          (1, 2, 128, 128, 160), (2, 8, 1024, 1024, 64), (1, 2, 96, 200, 80),
                                          (1, 2, 64, 64, 32), (1, 1, 48, 33, 16)])
def test_attention_forward_backward(lib, dt, B, H, Nq, Nk, D):
    q, k, v = rnd(B, Nq, H, D, dt=dt, seed=1), rnd(B, Nk, H, D, dt=dt, seed=2), rnd(B, Nk, H, D, dt=dt, seed=3)
    scale = D ** -0.5
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Nq, device="cuda")
    chk(lib, lib.smi_op_attention_fwd(dcode(dt), P(q), P(k), P(v), P(o), P(lse), B, H, Nq, Nk, D, scale, None))
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    oref, lref = attn_ref(qf, kf, vf, scale)
    close(o, oref, dt, what="attn O")
    torch.testing.assert_close(lse, lref, rtol=1e-4, atol=2e-3)
    do = rnd(B, Nq, H, D, dt=dt, seed=4)
    gq, gk, gv = torch.autograd.grad(oref, (qf, kf, vf), do.float())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Nq, device="cuda")
    chk(lib, lib.smi_op_attention_bwd(dcode(dt), P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), P(dk), P(dv), P(delta),
                                      B, H, Nq, Nk, D, scale, None))
    # P and dS are rounded to 16 bit before their MFMAs: allow a few storage roundings
    close(dq, gq, dt, mult=8, what="attn dQ")
    close(dk, gk, dt, mult=8, what="attn dK")
    close(dv, gv, dt, mult=8, what="attn dV")


@pytest.mark.parametrize("dt", DT)
def test_attention_rescale_branch_with_spiked_key(lib, dt):
    """Forces the online-softmax running max to jump in a LATE key tile (guide rule 26)."""
    B, H, N, D = 1, 1, 256, 64
    q, k, v = rnd(B, N, H, D, dt=dt, seed=1), rnd(B, N, H, D, dt=dt, seed=2), rnd(B, N, H, D, dt=dt, seed=3)
    k[0, 200, 0] = (q[0, 5, 0].float() * 6).to(dt)  # key 200 (4th tile) dominates query 5
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device="cuda")
    chk(lib, lib.smi_op_attention_fwd(dcode(dt), P(q), P(k), P(v), P(o), P(lse), B, H, N, N, D, D ** -0.5, None))
    oref, lref = attn_ref(q.float(), k.float(), v.float(), D ** -0.5)
    close(o, oref, dt, what="attn O (spiked)")
    torch.testing.assert_close(lse, lref, rtol=1e-4, atol=2e-3)  # what the backward recomputes P against


@pytest.mark.parametrize("Nq,Nk,D", [(192, 77, 64), (128, 77, 40), (64, 130, 64)])
def test_attention_backward_padded_keys_with_negative_lse(lib, Nq, Nk, D):
    """keys past Nk in the ragged last tile get p = exp(-lse); with every logit strongly negative (lse < -10) and
    a large (loss-scaled) dO their dS left the fp16 range and inf x 0 = NaN entered dQ.  fp16, dQ finite and == fp32 torch."""
    dt = torch.float16
    B, H = 2, 3
    # one unit direction u per (batch, head): q = 5 sqrt(D) u + noise, k = -6 u + unit noise  =>  scale * q . k ~ -30 +- 5
    # (the common component of k stays below its noise, so dQ = sum_k dS_k k_k is well conditioned: 5e-4 on an fp32 emulation)
    u = rnd(B, 1, H, D, dt=torch.float32, seed=7)
    u = u / u.norm(dim=-1, keepdim=True)
    q = (5.0 * D ** 0.5 * u + 0.3 * rnd(B, Nq, H, D, dt=torch.float32, seed=1)).to(dt)
    k = (-6.0 * u + rnd(B, Nk, H, D, dt=torch.float32, seed=2)).to(dt)
    v = rnd(B, Nk, H, D, dt=dt, seed=3)
    scale = D ** -0.5
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Nq, device="cuda")
    chk(lib, lib.smi_op_attention_fwd(dcode(dt), P(q), P(k), P(v), P(o), P(lse), B, H, Nq, Nk, D, scale, None))
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    oref, lref = attn_ref(qf, kf, vf, scale)
    assert lref.max().item() < -10.0, lref.max().item()
    close(o, oref, dt, what="attn O")
    do = rnd(B, Nq, H, D, dt=dt, seed=4) * 512.0  # as after the engine's loss scale
    gq, gk, gv = torch.autograd.grad(oref, (qf, kf, vf), do.float())
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Nq, device="cuda")
    chk(lib, lib.smi_op_attention_bwd(dcode(dt), P(q), P(k), P(v), P(o), P(lse), P(do), P(dq), P(dk), P(dv), P(delta),
                                      B, H, Nq, Nk, D, scale, None))
    assert torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all() and torch.isfinite(dv.float()).all()
    close(dq, gq, dt, mult=8, what="attn dQ (padded keys, lse < -10)")
    close(dk, gk, dt, mult=8, what="attn dK")
    close(dv, gv, dt, mult=8, what="attn dV")
    # and the dQ-only form the engine uses for cross-attention under `noxattn` (no dK / dV asked)
    dq2 = torch.empty_like(q)
    chk(lib, lib.smi_op_attention_bwd(dcode(dt), P(q), P(k), P(v), P(o), P(lse), P(do), P(dq2), None, None, P(delta),
                                      B, H, Nq, Nk, D, scale, None))
    assert torch.equal(dq2, dq)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("nb,HW,Cc,G,silu", [(2, 256, 320, 32, 1), (3, 64, 64, 16, 0), (2, 1000, 2560, 32, 1), (1, 160 * 160, 128, 32, 1),
                                                     (2, 16384 + 40, 64, 32, 0),
                                             (1, 4096, 640, 32, 1), (2, 100, 1920, 32, 0)])
def test_groupnorm_forward_backward(lib, dt, nb, HW, Cc, G, silu):
    x = rnd(nb, HW, Cc, dt=dt, seed=1) * 2 + 0.5
    gamma, beta = (1 + 0.1 * rnd(Cc, dt=dt, seed=2)).to(dt), rnd(Cc, dt=dt, scale=0.1, seed=3)
    dy = rnd(nb, HW, Cc, dt=dt, seed=4)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    # chunk partials: <= 64 chunks of >= 16 rows up to 128 x 128 maps; 256-row chunks + a 64-slot fold beyond (norm.hip)
    nchunk = 64 if HW <= 16384 else (HW + 255) // 256 + 64  # (upper bound of gn_num_chunks)
    scratch = torch.empty(2 * nb * Cc + nb * G * 2 + nb * nchunk * G * 2 + 2 * nb * Cc + 64, device="cuda")
    chk(lib, lib.smi_op_groupnorm(dcode(dt), P(x), P(gamma), P(beta), P(y), P(dy), P(dx), P(scratch), nb, HW, Cc, G,
                                  1e-5, silu, None))
    xr = x.float().permute(0, 2, 1).requires_grad_(True)  # [n, C, HW]
    ref = F.group_norm(xr, G, gamma.float(), beta.float(), 1e-5)
    if silu:
        ref = F.silu(ref)
    close(y.permute(0, 2, 1), ref, dt, what="groupnorm y")
    (gref,) = torch.autograd.grad(ref, xr, dy.float().permute(0, 2, 1))
    close(dx.permute(0, 2, 1), gref, dt, mult=6, what="groupnorm dx")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,Cc", [(1000, 320), (4096, 1280), (77, 640), (5, 64),
                                  # two rows per wave (M >= 16384) with an odd tail row; 4 and 2 vectors per lane (CLIP widths)
                                  (16384 + 3, 640), (16390, 1280), (300, 2048), (231, 768), (154, 1024)])
def test_layernorm_forward_backward(lib, dt, M, Cc):
    x = rnd(M, Cc, dt=dt, seed=1) * 1.5 - 0.3
    gamma, beta = (1 + 0.1 * rnd(Cc, dt=dt, seed=2)).to(dt), rnd(Cc, dt=dt, scale=0.1, seed=3)
    dy = rnd(M, Cc, dt=dt, seed=4)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    st = torch.empty(M, 2, device="cuda")
    chk(lib, lib.smi_op_layernorm(dcode(dt), P(x), P(gamma), P(beta), P(y), P(dy), P(dx), P(st), M, Cc, 1e-5, None))
    xr = x.float().requires_grad_(True)
    ref = F.layer_norm(xr, (Cc,), gamma.float(), beta.float(), 1e-5)
    close(y, ref, dt, what="layernorm y")
    (gref,) = torch.autograd.grad(ref, xr, dy.float())
    close(dx, gref, dt, mult=6, what="layernorm dx")


@pytest.mark.parametrize("dt", DT)
def test_geglu_forward_backward(lib, dt):
    M, C4 = 777, 1280
    pj, do = rnd(M, 2 * C4, dt=dt, seed=1), rnd(M, C4, dt=dt, seed=2)
    out, dpj = torch.empty(M, C4, device="cuda", dtype=dt), torch.empty_like(pj)
    chk(lib, lib.smi_op_geglu(dcode(dt), P(pj), P(out), P(do), P(dpj), M, C4, None))
    pr = pj.float().requires_grad_(True)
    h, g = pr.chunk(2, dim=-1)
    ref = h * F.gelu(g)
    close(out, ref, dt, what="geglu")
    (gref,) = torch.autograd.grad(ref, pr, do.float())
    close(dpj, gref, dt, what="geglu grad")


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,K,r", [(4096, 640, 4), (1000, 1280, 8), (154, 2048, 4), (70, 320, 3), (600, 64, 16)])
def test_lora_skinny_kernels(lib, dt, M, K, r):
    x = rnd(M, K, dt=dt, seed=1)
    a = torch.randn(r, K, device="cuda") * 0.05
    xa = torch.empty(M, r, device="cuda")
    chk(lib, lib.smi_op_lora_down(dcode(dt), P(x), P(a), P(xa), M, K, r, None))
    torch.testing.assert_close(xa, x.float() @ a.t(), rtol=1e-4, atol=1e-4)
    p = torch.randn(M, r, device="cuda")
    dw = torch.ones(r, K, device="cuda")  # the kernel accumulates (+=)
    scratch = torch.empty(((M + 63) // 64) * r * K + 256, device="cuda")  # partials + the one-job table (smi.h)
    chk(lib, lib.smi_op_lora_wgrad(dcode(dt), P(p), P(x), P(dw), M, K, r, 0.5, P(scratch), None))
    torch.testing.assert_close(dw, 1 + 0.5 * (p.t() @ x.float()), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,K,R", [(4096, 1280, 16), (1000, 640, 16), (16384, 640, 32), (4100, 3840, 16), (7, 128, 32)])
def test_lora_skinny_product(lib, dt, M, K, R):
    x, s = rnd(M, K, dt=dt, seed=1), rnd(R, K, dt=dt, scale=K ** -0.5, seed=2)
    out = torch.full((M, R), 9.0, device="cuda")
    chk(lib, lib.smi_op_lora_skinny(dcode(dt), P(x), P(s), P(out), M, R, K, None))
    torch.testing.assert_close(out, x.float() @ s.float().t(), rtol=2e-5, atol=2e-5)  # fp32 accumulate and output
    out2 = torch.empty_like(out)
    chk(lib, lib.smi_op_lora_skinny(dcode(dt), P(x), P(s), P(out2), M, R, K, None))
    assert torch.equal(out, out2)


def test_step_ops_match_torch(lib):
    from oracle import slider_ref as R
    g = torch.Generator(device="cuda").manual_seed(0)
    n = 2 * 4 * 32 * 32
    e2 = torch.randn(2 * n, device="cuda", generator=g)
    out = torch.empty(n, device="cuda")
    chk(lib, lib.smi_cfg_combine(P(e2), P(out), n, 3.0, None))
    torch.testing.assert_close(out, e2[:n] + 3.0 * (e2[n:] - e2[:n]), rtol=1e-6, atol=1e-6)
    t, po, ne, ng = (torch.randn(n, device="cuda", generator=g) for _ in range(4))
    loss, dt_ = torch.empty(1, device="cuda"), torch.empty(n, device="cuda")
    scratch = torch.empty(2048, device="cuda")
    for action, sign in (("enhance", 4.0), ("erase", -4.0)):
        chk(lib, lib.smi_slider_loss(P(t), P(po), P(ne), P(ng), sign, n, P(loss), P(dt_), P(scratch), None))
        tr = t.clone().requires_grad_(True)
        ref = R.slider_loss(action, 4.0, tr, po, ne, ng)
        ref.backward()
        torch.testing.assert_close(loss[0], ref.detach(), rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(dt_, tr.grad, rtol=1e-5, atol=1e-9)
    # clip + AdamW, 3 steps, against the oracle's restatement of torch's update
    p = torch.randn(5000, device="cuda", generator=g)
    pr = p.clone().cpu()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    mr, vr = torch.zeros(5000), torch.zeros(5000)
    for step in range(1, 4):
        gr = torch.randn(5000, device="cuda", generator=g) * 0.3
        chk(lib, lib.smi_clip_adamw(P(p), P(gr), P(m), P(v), 5000, 1e-2, 0.9, 0.999, 1e-8, 1e-2, step, 0.2, P(scratch),
                                    None))
        holder = torch.nn.Parameter(pr.clone())
        holder.grad = gr.cpu().clone()
        R.clip_grad_norm_([holder], 0.2)
        with torch.no_grad():
            R.adamw_step_(pr, holder.grad, mr, vr, step, 1e-2, weight_decay=1e-2)
        torch.testing.assert_close(p.cpu(), pr, rtol=2e-5, atol=2e-6)
    x = torch.randn(n, device="cuda", generator=g)
    e, nz = torch.randn(n, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g)
    xr = x.clone()
    chk(lib, lib.smi_sched_step(P(x), P(e), P(nz), 0.9, -0.2, 0.05, n, None))
    torch.testing.assert_close(x, 0.9 * xr - 0.2 * e + 0.05 * nz, rtol=1e-6, atol=1e-6)


def test_euler_a_step_draws_its_noise_on_the_device_when_asked():
    """EulerAncestralDiscreteScheduler.step with `noise_on_device` (or a device generator): the ancestral noise is the
    device generator's randn of the sample's shape (what diffusers' randn_tensor does without a CPU generator), the
    update is x + eps dt + noise sigma_up; the default stays the host draw the CPU oracle shares."""
    import sliders_conceptmod_amd.model_util as MU
    s = MU.create_noise_scheduler("euler_a")
    s.set_timesteps(12)
    t = s.timesteps[3]
    g = torch.Generator(device="cuda").manual_seed(5)
    x, e = torch.randn(2, 4, 16, 16, device="cuda", generator=g), torch.randn(2, 4, 16, 16, device="cuda", generator=g)
    sig, sig_to = float(s.sigmas[3]), float(s.sigmas[4])
    up = (sig_to ** 2 * (sig ** 2 - sig_to ** 2) / sig ** 2) ** 0.5
    down = (sig_to ** 2 - up ** 2) ** 0.5
    g1, g2 = torch.Generator(device="cuda").manual_seed(9), torch.Generator(device="cuda").manual_seed(9)
    got = s.step(e, t, x, generator=g1).prev_sample
    nz = torch.randn(x.shape, device="cuda", generator=g2)
    torch.testing.assert_close(got, x + e * (down - sig) + nz * up, rtol=1e-6, atol=1e-6)
    s.noise_on_device = True
    torch.cuda.manual_seed(11)
    got = s.step(e, t, x).prev_sample
    torch.cuda.manual_seed(11)
    nz = torch.randn(x.shape, device="cuda")
    torch.testing.assert_close(got, x + e * (down - sig) + nz * up, rtol=1e-6, atol=1e-6)
    s.noise_on_device = False
    torch.manual_seed(3)
    got = s.step(e, t, x).prev_sample
    torch.manual_seed(3)
    nz = torch.randn(x.shape).cuda()
    torch.testing.assert_close(got, x + e * (down - sig) + nz * up, rtol=1e-6, atol=1e-6)


def test_gemm_generations_are_bit_identical_per_epilogue_class():
    """The tile autotuner (csrc/gemm.hip) may serve a (shape, epilogue) key with any kernel generation / tile layout;
    that is only sound while they all produce the SAME BITS.  One op per key class -- full LoRA epilogue (bias + residual
    + rank-r delta, also in the batched-pass form with the delta on the last quarter of the rows), fp32 output, fused
    GEGLU, 3x3 conv -- hashed in separate processes under every SMI_GEMM override.  "5ph" is the 256 x 320
    persistent kernel (gemm4.hip): its delta comes off the fp32 MFMA and must equal the VALU fmaf chain of the others."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("128", "256", "160", "160w", "64", "64w", "64x160", "8ph", "5ph", ""):
        env = dict(os.environ)
        env.pop("SMI_GEMM", None)
        if mode:
            env["SMI_GEMM"] = mode
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "gemm_digest.py")], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        res[mode or "auto"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    base = res["128"]
    assert len(base) >= 20
    for mode, d in res.items():
        diff = [k for k in base if d[k] != base[k]]
        assert not diff, f"SMI_GEMM={mode} differs from the 128x128 kernel on {diff}"


def test_gemm4_persistent_256x320_kernel_screen():
    """gemm4.hip (256 x 320 persistent tile, LDS-DMA pipeline running across output tiles, fp32-MFMA LoRA delta,
    interleaved frozen / adapted rows, in-register GEGLU, implicit-GEMM conv) forced with SMI_GEMM=5ph in a child process:
    every case is compared with fp32 torch, bit-for-bit with the heuristic selection of the older generations, and
    bit-for-bit with its own repeated runs (race screen of the counted-vmcnt schedule) -- tools/check_gemm4.py, here in
    its quick form (every mode and epilogue class, fewer large shapes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_gemm4.py")], capture_output=True, text=True,
                       timeout=900, cwd=root, env={**os.environ, "SMI_CHECK_QUICK": "1"})  # (the full list: run the tool)
    tail = "\n".join(l for l in r.stdout.splitlines() if not l.startswith("["))[-3000:]
    assert r.returncode == 0 and "ALL OK" in r.stdout, tail + r.stderr[-1500:]


def test_splitk_every_slice_count_matches_the_unsplit_kernel():
    """The split-K rule picks any slice count from 2 to 16 (round 4: chosen to fill two workgroups per CU, not a power of
    two) on the 64 x 128 or 128 x 128 slice kernel; tools/check_splitk.py forces each (count, kernel) pair in a child
    process (SMI_SPLITK_DEBUG) and compares the finished result -- bias, residual, rank-4 / rank-8 delta included --
    with the un-split kernel: fp32 outputs to 3e-6 of the largest element, 16-bit outputs to a last-bit flip."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_splitk.py")], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and "split-K check ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_groupnorm_one_launch_forms_are_bit_identical():
    """GroupNorm statistics + apply in one launch (round 4: workgroups of a sample meet at a counter between the two bodies,
    csrc/norm.hip gn_coop_kernel) against the two-launch form (SMI_GN_COOP=0), forward and backward, fp16 / bf16, with
    and without SiLU, pre-roll and step batches: same bits for y, dx and the saved statistics, equal repeats, and no
    workgroup gave up its bounded wait -- tools/gn_digest.py in two child processes."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for coop in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "gn_digest.py")], capture_output=True, text=True,
                           timeout=600, cwd=root, env={**os.environ, "SMI_GN_COOP": coop})
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
    assert outs[0]["timeouts"] == 0 and outs[1]["timeouts"] == 0
    assert outs[0] == outs[1], {k: (outs[0][k], outs[1][k]) for k in outs[0] if outs[0][k] != outs[1][k]}
