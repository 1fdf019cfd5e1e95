"""End-to-end parity of the HIP engine (through the product UNet2DConditionModel / LoRANetwork and the C ABI) against
the CPU oracle (oracle/unet_ref.py + oracle/slider_ref.py) on the same seeded inputs.

Tolerances are relative L2 norms, written at each assert.  The engine stores activations in fp16/bf16 (fp32
accumulate), so its distance to the pure-fp32 oracle is storage-rounding noise accumulated over ~100 layers:
measured ~1e-3 (fp16) and ~8e-3 (bf16) on these nets.  That noise level is calibrated independently of the engine:
the oracle itself, run with `storage_dtype` (rounding at exactly the points where the engine writes HBM), lands at
the same distance from fp32 -- the engine must not be noisier than 1.5x that.  (Rounding decisions of two runs
decorrelate after a few layers, so engine-vs-rounded-oracle is ~sqrt(2) x that distance, not ~0.)"""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import slider_ref as R
from oracle import unet_ref as OU

CFGS = {"tiny_sd1x": OU.tiny_sd1x_config, "tiny_sdxl": OU.tiny_sdxl_config}
LOOSE = {torch.float16: 1.8e-3, torch.bfloat16: 1.4e-2}   # vs pure fp32 oracle: 1.5 x measured (1.17e-3 / 9.2e-3)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def build_pair(model, dtype, method="noxattn", rank=4, lora_seed=2, ocfg=None, c3lier=False):
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.unet as PU
    ocfg = ocfg or CFGS[model]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0).requires_grad_(False).eval()
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
    pu.load_state_dict(ou.state_dict())
    pu = pu.to("cuda", dtype).requires_grad_(False).eval()
    torch.manual_seed(1)
    if c3lier:
        onet = R.LoRANetworkRef(ou, rank, 1.0, 1.0, method, target_replace=R.C3LIER_TARGET_REPLACE)
        torch.manual_seed(1)
        pnet = L.LoRANetwork(pu, rank=rank, multiplier=1.0, alpha=1.0, train_method=method,
                             target_replace=L.UNET_TARGET_REPLACE_MODULE_TRANSFORMER + L.UNET_TARGET_REPLACE_MODULE_CONV)
    else:
        onet = R.LoRANetworkRef(ou, rank, 1.0, 1.0, method)
        torch.manual_seed(1)
        pnet = L.LoRANetwork(pu, rank=rank, multiplier=1.0, alpha=1.0, train_method=method)
    g = torch.Generator().manual_seed(lora_seed)
    with torch.no_grad():
        for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
            assert lo.lora_name == lp.lora_name
            w = torch.randn(lo.lora_up.weight.shape, generator=g) * 0.05
            lo.lora_up.weight.copy_(w)
            lp.lora_up.weight.copy_(w)
            torch.testing.assert_close(lp.lora_down.weight, lo.lora_down.weight, rtol=0, atol=0)
    pnet.to("cuda")
    return ocfg, ou, onet, pu, pnet


def inputs(ocfg, n, hw, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 4, hw, hw, generator=g)
    ctx = torch.randn(n, 77, ocfg.cross_attention_dim, generator=g)
    add = None
    if ocfg.addition_embed_type == "text_time":
        pdim = ocfg.projection_class_embeddings_input_dim - 6 * ocfg.addition_time_embed_dim
        add = {"text_embeds": torch.randn(n, pdim, generator=g),
               "time_ids": torch.tensor([[hw * 8.0, hw * 8.0, 0, 0, hw * 8.0, hw * 8.0]] * n)}
    return x, ctx, add


def cuda_add(add):
    return None if add is None else {k: v.cuda() for k, v in add.items()}


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_forward_parity_lora_off_and_on(model, dtype):
    ocfg, ou, onet, pu, pnet = build_pair(model, dtype)
    x, ctx, add = inputs(ocfg, 2, 16)
    for t in (1.0, 499.0, 981.0):
        for on in (False, True):
            if on:
                onet.__enter__(), pnet.__enter__()
            else:
                onet.__exit__(None, None, None), pnet.__exit__(None, None, None)
            with torch.no_grad():
                got = pu(x.cuda(), t, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
                ou.storage_dtype = None
                ref32 = ou(x, t, ctx, add).sample
                ou.storage_dtype = dtype
                refq = ou(x, t, ctx, add).sample
                ou.storage_dtype = None
            assert got.dtype == torch.float32 and got.shape == ref32.shape
            e_eng, e_q, e_x = rel(got, ref32), rel(refq, ref32), rel(got, refq)
            print(f"{model} {dtype} t={t} lora={on}: engine-fp32 {e_eng:.2e}  rounded_oracle-fp32 {e_q:.2e}  "
                  f"engine-rounded_oracle {e_x:.2e}")
            assert e_eng < LOOSE[dtype], f"{model} t={t} lora={on}: vs fp32 oracle {e_eng:.2e}"
            assert e_eng < 1.5 * e_q + 1e-4, f"engine noisier than its storage rounding explains: {e_eng:.2e} vs {e_q:.2e}"
            assert e_x < 2.0 * e_q + 1e-4
    # LoRA must actually change the output (guards against a silently ignored adaptor)
    assert rel(got, pu(x.cuda(), 981.0, encoder_hidden_states=ctx.cuda(),
                       added_cond_kwargs=cuda_add(add)).sample) == 0.0
    pnet.__exit__(None, None, None)
    off = pu(x.cuda(), 981.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
    assert rel(got, off) > 1e-3


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("method", ["noxattn", "full", "xattn"])
def test_lora_gradients_match_oracle(model, dtype, method):
    ocfg, ou, onet, pu, pnet = build_pair(model, dtype, method=method)
    x, ctx, add = inputs(ocfg, 2, 16)
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(2, 4, 16, 16, generator=g) * 1e-4  # the size of d(MSE)/d(eps) at real shapes: exercises the loss scale
    with onet:
        ref = ou(x, 499.0, ctx, add).sample
    (ref * gy).sum().backward()
    with pnet:
        got = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
    assert got.requires_grad
    (got * gy.cuda()).sum().backward()
    assert pnet.flat.grad is not None
    sd_g = {l.lora_name: (l.lora_down.grad, l.lora_up.grad) for l in pnet.unet_loras}
    # Per-module bar.  A module's gradient error is noise of 16-bit activation gradients and 16-bit LoRA GEMM operands whose
    # size follows the ACTIVATION gradients, not the module's own gradient norm: a module whose true gradient is tiny shows
    # a large ratio to its own norm from one build to the next (0.15 <-> 0.22 on one bf16 to_v.down, round 4, with every
    # global figure unchanged).  So a module is measured against max(its own norm, the median module norm of
    # its kind); the global bar below is the one that tracks parity.  Bars = 1.5 x measured (round 4: fp16 per-module
    # <= 9.6e-3, global <= 3.1e-3; bf16 <= 2.2e-1 / 2.5e-2 -- the bf16 extreme is ONE deepest-level to_v.down of the tiny
    # SD-XL net, a sum over 32 rows; every other module is <= 6.6e-2, and the attention kernels themselves measure the same
    # error against fp32 torch in both builds, tools/attn_err.py).  The same module moves the fp16 extreme from build to build
    # as well: 0.96e-2 / 1.14e-2 / 1.53e-2 across round 4's split-K slice-count rules (each slice count checked against the
    # un-split kernel to 1.2e-6, tools/check_splitk.py), while bf16 went 2.3e-1 -> 1.2e-1 on the same change: fp16 bar =
    # 1.5 x the largest of them.
    tol = 2.3e-2 if dtype == torch.float16 else 3.3e-1
    worst = worst_own = 0.0
    tot_num = tot_den = 0.0
    ref_norm = {"down": [], "up": []}
    for lo in onet.unet_loras:
        ref_norm["down"].append(lo.lora_down.weight.grad.norm().item())
        ref_norm["up"].append(lo.lora_up.weight.grad.norm().item())
    floor = {k: sorted(v)[len(v) // 2] for k, v in ref_norm.items()}
    for lo in onet.unet_loras:
        gd, gu = sd_g[lo.lora_name]
        for a, b, what in ((gd, lo.lora_down.weight.grad, "down"), (gu, lo.lora_up.weight.grad, "up")):
            assert b is not None and b.abs().max() > 0, f"oracle grad missing for {lo.lora_name}.{what}"
            err = (a.detach().cpu() - b).norm().item()
            r = err / max(b.norm().item(), floor[what])
            worst = max(worst, r)
            worst_own = max(worst_own, err / b.norm().item())
            tot_num += err ** 2
            tot_den += b.norm().item() ** 2
            assert r < tol, f"{lo.lora_name}.{what}: rel err {r:.3e}"
    glob = (tot_num / tot_den) ** 0.5
    assert glob < (4.7e-3 if dtype == torch.float16 else 3.7e-2), f"global LoRA-grad rel err {glob:.3e}"
    print(f"{model} {dtype} {method}: worst per-module grad err {worst:.2e} (vs own norm {worst_own:.2e}), global {glob:.2e}")


@pytest.mark.parametrize("dtype", [torch.float16])
def test_interleaved_frozen_pass_does_not_clobber_tape(dtype):
    """forward(grad) -> forward(no grad) -> backward: the frozen pass must not disturb saved activations."""
    ocfg, ou, onet, pu, pnet = build_pair("tiny_sd1x", dtype)
    x, ctx, add = inputs(ocfg, 2, 16)
    gy = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9)).cuda() * 1e-3
    with pnet:
        a = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda()).sample
    (a * gy).sum().backward()
    g1 = pnet.flat.grad.clone()
    pnet.flat.grad = None
    with pnet:
        b = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda()).sample
    with torch.no_grad():
        pnet.__exit__(None, None, None)
        pu(x.cuda() * 0.5, 20.0, encoder_hidden_states=ctx.cuda() * 2).sample
    (b * gy).sum().backward()
    torch.testing.assert_close(pnet.flat.grad, g1, rtol=0, atol=0)  # deterministic kernels: bitwise equal


def test_smaller_batch_than_engine_capacity():
    ocfg, ou, onet, pu, pnet = build_pair("tiny_sdxl", torch.float16)
    x, ctx, add = inputs(ocfg, 4, 16)
    pnet.__exit__(None, None, None)
    with torch.no_grad():
        full = pu(x.cuda(), 300.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
        sub = pu(x[:2].cuda(), 300.0, encoder_hidden_states=ctx[:2].cuda(),
                 added_cond_kwargs={k: v[:2].cuda() for k, v in add.items()}).sample
    torch.testing.assert_close(sub, full[:2], rtol=0, atol=0)


@pytest.mark.parametrize("model", list(CFGS))
def test_batched_guidance_pass_equals_separate_passes(model):
    """smi_unet_forward_batched: [frozen | frozen | frozen | adapted] in one pass == four passes, bit for bit,
    forward and LoRA gradients (per-sample arithmetic does not depend on the batch composition)."""
    ocfg, ou, onet, pu, pnet = build_pair(model, torch.float16)
    n = 2
    xs, ctxs, adds = [], [], []
    for i in range(4):
        x, ctx, add = inputs(ocfg, n, 16, seed=3)
        _, ctx, add2 = inputs(ocfg, n, 16, seed=10 + i)
        xs.append(x), ctxs.append(ctx), adds.append(add2)
    flat, n_down, _ = pnet.engine_params()
    down, up = flat[:n_down].detach(), flat[n_down:].detach()
    dt = torch.float16

    def te(a, k):
        return None if a is None else (a[k].cuda().to(dt) if k == "text_embeds" else a[k].cuda().float())

    gy = torch.randn(n, 4, 16, 16, generator=torch.Generator().manual_seed(9)).cuda() * 1e-3
    # four separate passes (engine capacity n, all adapted)
    eng = pu._ensure_engine(n, 16, 16, 77)
    sep = []
    for i in range(4):
        lora = i == 3
        sep.append(eng.forward(xs[i].cuda(), 499.0, ctxs[i].cuda().to(dt), te(adds[i], "text_embeds"),
                               te(adds[i], "time_ids"), down if lora else None, up if lora else None,
                               1.0 if lora else 0.0, lora))
    g_sep = torch.zeros_like(flat)
    eng.backward(gy, g_sep[:n_down], g_sep[n_down:])
    # one batched pass
    eng4 = pu._ensure_engine(4 * n, 16, 16, 77, n_adapted=n)
    cat = lambda ts: None if ts[0] is None else torch.cat(ts).contiguous()
    out = eng4.forward(torch.cat(xs).cuda(), 499.0, torch.cat(ctxs).cuda().to(dt),
                       cat([te(a, "text_embeds") for a in adds]), cat([te(a, "time_ids") for a in adds]), down, up, 1.0,
                       True, n_adapted=n)
    g_bat = torch.zeros_like(flat)
    eng4.backward(gy, g_bat[:n_down], g_bat[n_down:])
    for i in range(4):
        torch.testing.assert_close(out[i * n:(i + 1) * n], sep[i], rtol=0, atol=0)
    assert float(g_sep.abs().max()) > 0
    torch.testing.assert_close(g_bat, g_sep, rtol=0, atol=0)


@pytest.mark.parametrize("model", list(CFGS))
def test_rank8_and_ragged_nonsquare_latents(model):
    """BASELINE configs[3] trains rank 8; and latents need not be square or a multiple of the tile sizes: 24 x 40 latents
    (ragged attention key tiles: 960 / 240 / 60 tokens; conv rows that are not multiples of 128) with a rank-8 network,
    forward and LoRA gradients against the oracle."""
    dtype = torch.float16
    ocfg, ou, onet, pu, pnet = build_pair(model, dtype, method="noxattn", rank=8)
    g = torch.Generator().manual_seed(3)
    n, H, W = 2, 24, 40
    x = torch.randn(n, 4, H, W, generator=g)
    ctx = torch.randn(n, 77, ocfg.cross_attention_dim, generator=g)
    add = None
    if ocfg.addition_embed_type == "text_time":
        pdim = ocfg.projection_class_embeddings_input_dim - 6 * ocfg.addition_time_embed_dim
        add = {"text_embeds": torch.randn(n, pdim, generator=g),
               "time_ids": torch.tensor([[H * 8.0, W * 8.0, 0, 0, H * 8.0, W * 8.0]] * n)}
    gy = torch.randn(n, 4, H, W, generator=g) * 1e-4
    with onet:
        ref = ou(x, 499.0, ctx, add).sample
    (ref * gy).sum().backward()
    with pnet:
        got = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
    (got * gy.cuda()).sum().backward()
    assert rel(got, ref) < LOOSE[dtype], f"forward {rel(got, ref):.2e}"
    num = den = 0.0
    for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
        assert lp.lora_down.weight.shape == (8, lo.lora_down.weight.shape[1])
        for a, b in ((lp.lora_down.grad, lo.lora_down.weight.grad), (lp.lora_up.grad, lo.lora_up.weight.grad)):
            num += (a.detach().cpu() - b).norm().item() ** 2
            den += b.norm().item() ** 2
    glob = (num / den) ** 0.5
    print(f"rank-8 / ragged: global LoRA-grad rel err {glob:.2e}")
    assert glob < 4.7e-3, f"rank-8 / ragged global LoRA-grad rel err {glob:.3e}"


def test_alternating_resolutions_replan_without_repacking():
    """Engine lifetime (reference: dynamic_resolution draws a new bucket EVERY step, T/train_util.py:1085-1097): switching
    latent sizes must only re-plan the activation arenas -- no weight-packing kernel may run again (smi_engine_stats)
    -- and must give bit-identical results to a fresh engine at that size."""
    ocfg, ou, onet, pu, pnet = build_pair("tiny_sdxl", torch.float16)
    sizes = [(16, 16), (24, 16), (16, 16), (8, 24), (24, 16)]
    got = []
    with torch.no_grad(), pnet:
        for i, (h, w) in enumerate(sizes):
            g = torch.Generator().manual_seed(30 + h * 100 + w)
            x = torch.randn(2, 4, h, w, generator=g).cuda()
            ctx = torch.randn(2, 77, ocfg.cross_attention_dim, generator=g).cuda()
            pdim = ocfg.projection_class_embeddings_input_dim - 6 * ocfg.addition_time_embed_dim
            add = {"text_embeds": torch.randn(2, pdim, generator=g).cuda(),
                   "time_ids": torch.tensor([[h * 8.0, w * 8.0, 0, 0, h * 8.0, w * 8.0]] * 2).cuda()}
            got.append(pu(x, 499.0, encoder_hidden_states=ctx, added_cond_kwargs=add).sample.clone())
            st = pu._engine.stats()
            if i == 0:
                packs = st["pack_launches"]
                assert packs > 0
            assert st["pack_launches"] == packs, f"weights re-packed on a shape change ({h}x{w})"
            assert st["replans"] == i, st
        eng = pu._engine
    assert torch.equal(got[0], got[2]) and torch.equal(got[1], got[4])
    # fresh engine at the second size: same bits
    ocfg2, _, _, pu2, pnet2 = build_pair("tiny_sdxl", torch.float16)
    h, w = sizes[1]
    g = torch.Generator().manual_seed(30 + h * 100 + w)
    x = torch.randn(2, 4, h, w, generator=g).cuda()
    ctx = torch.randn(2, 77, ocfg.cross_attention_dim, generator=g).cuda()
    pdim = ocfg.projection_class_embeddings_input_dim - 6 * ocfg.addition_time_embed_dim
    add = {"text_embeds": torch.randn(2, pdim, generator=g).cuda(),
           "time_ids": torch.tensor([[h * 8.0, w * 8.0, 0, 0, h * 8.0, w * 8.0]] * 2).cuda()}
    with torch.no_grad(), pnet2:
        fresh = pu2(x, 499.0, encoder_hidden_states=ctx, added_cond_kwargs=add).sample
    assert pu2._engine.stats()["replans"] == 0
    assert torch.equal(fresh, got[1])
    assert eng is pu._engine


def test_backward_through_a_superseded_forward_fails_loudly():
    """The engine keeps ONE tape.  A second grad-enabled forward (or a change of shape) drops the first one's saved
    activations: backward through the older output must raise, not silently replay the newer tape."""
    from sliders_conceptmod_amd._native import SmiError
    ocfg, ou, onet, pu, pnet = build_pair("tiny_sd1x", torch.float16)
    x, ctx, add = inputs(ocfg, 2, 16)
    with pnet:
        a = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda()).sample
        b = pu(x.cuda(), 301.0, encoder_hidden_states=ctx.cuda()).sample
    with pytest.raises(SmiError, match="saved activations are gone"):
        a.sum().backward()
    b.sum().backward()  # the live tape still works
    assert pnet.flat.grad is not None and float(pnet.flat.grad.abs().max()) > 0
    with pnet:
        c = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda()).sample
    with torch.no_grad():
        x2 = torch.randn(2, 4, 24, 24).cuda()
        pu(x2, 499.0, encoder_hidden_states=ctx.cuda())  # another shape: replan drops the tape
    with pytest.raises(SmiError, match="saved activations are gone"):
        c.sum().backward()


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("method", ["noxattn", "full"])
def test_c3lier_forward_and_gradients_match_oracle(model, method):
    """`network.type: c3lier` (the shipped SD-1.x config, T/data/config.yaml:7): adaptors on the resnet convs (3x3 down
    conv -> 1x1 up, T/lora.py:100-114), time_emb_proj, conv_shortcut and the sampler convs (stride 2 / nearest-2x) next
    to the attention ones; forward and every LoRA gradient against the oracle's autograd."""
    dtype = torch.float16
    ocfg, ou, onet, pu, pnet = build_pair(model, dtype, method=method, c3lier=True)
    kinds = {"conv1": 0, "conv2": 0, "time_emb_proj": 0, "conv_shortcut": 0, "samplers_0_conv": 0, "attn1": 0}
    if method == "full":
        kinds["attn2"] = 0
    for l in pnet.unet_loras:
        for k in kinds:
            kinds[k] += k in l.lora_name
    assert all(v > 0 for v in kinds.values()), kinds
    x, ctx, add = inputs(ocfg, 2, 16)
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(2, 4, 16, 16, generator=g) * 1e-4
    with torch.no_grad():
        onet.__exit__(None, None, None)
        off = ou(x, 499.0, ctx, add).sample
    with onet:
        ref = ou(x, 499.0, ctx, add).sample
    (ref * gy).sum().backward()
    with pnet:
        got = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
    (got * gy.cuda()).sum().backward()
    e = rel(got, ref)
    assert rel(ref, off) > 1e-2  # the conv adaptors move the output a lot more than the attention-only ones
    assert e < 1.8e-3, e  # measured 8.1e-4 .. 1.19e-3 (1.5 x)
    by_kind = {}
    tot_num = tot_den = 0.0
    for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
        assert lo.lora_name == lp.lora_name
        kind = next(k for k in kinds if k in lo.lora_name)
        for a, b in ((lp.lora_down.grad, lo.lora_down.weight.grad), (lp.lora_up.grad, lo.lora_up.weight.grad)):
            assert a.shape == b.shape and b.abs().max() > 0, lo.lora_name
            n_, d_ = (a.cpu() - b).norm().item() ** 2, b.norm().item() ** 2
            acc = by_kind.setdefault(kind, [0.0, 0.0])
            acc[0] += n_
            acc[1] += d_
            tot_num += n_
            tot_den += d_
    errs = {k: (v[0] / v[1]) ** 0.5 for k, v in by_kind.items()}
    glob = (tot_num / tot_den) ** 0.5
    print(f"{model} {method} c3lier: eps {e:.2e}, LoRA-grad global {glob:.2e}, by kind " +
          ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    # measured (fp16): global 2.28e-3 .. 2.41e-3, worst kind 3.23e-3 (attn1 on tiny SD-XL); bars = 1.5 x
    assert glob < 3.6e-3, glob
    assert all(v < 4.9e-3 for v in errs.values()), errs


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("method", ["noxattn", "full"])
def test_dora_forward_and_gradients_match_oracle(model, method):
    """`--peft_type dora` (T/dora.py:124-162, T/train_lora_xl.py:87-90): dW = (W + up down) g / ||W + up down||_col - W as
    a second GEMM on the adapted rows; gradients of lora_down / lora_up / dora_scale through the dense G = dY^T X against
    the oracle's autograd (oracle.slider_ref.DoRAModuleRef, pinned to the reference's own module by
    tests/test_oracle_golden.py::test_dora_linear_forward_backward)."""
    import sliders_conceptmod_amd.dora as D
    import sliders_conceptmod_amd.unet as PU
    dtype = torch.float16
    ocfg = CFGS[model]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0).requires_grad_(False).eval()
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
    pu.load_state_dict(ou.state_dict())
    torch.manual_seed(1)
    onet = R.DoRANetworkRef(ou, 4, 1.0, 1.0, method)
    torch.manual_seed(1)
    pnet = D.DoRANetwork(pu, rank=4, multiplier=1.0, target_replace=["Attention"], train_method=method)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
            assert lo.lora_name == lp.lora_name
            torch.testing.assert_close(lp.lora_down.weight, lo.lora_down.weight, rtol=0, atol=0)
            torch.testing.assert_close(lp.dora_scale, lo.dora_scale, rtol=1e-6, atol=1e-7)
            w = torch.randn(lo.lora_up.weight.shape, generator=g) * 0.05
            sc = 1.0 + 0.1 * torch.randn(lo.dora_scale.shape, generator=g)
            lo.lora_up.weight.copy_(w)
            lp.lora_up.weight.copy_(w)
            lo.dora_scale.mul_(sc)
            lp.dora_scale.mul_(sc)
    pu = pu.to("cuda", dtype).requires_grad_(False).eval()
    pnet.to("cuda")
    x, ctx, add = inputs(ocfg, 2, 16)
    gy = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9)) * 1e-4
    with torch.no_grad():
        onet.__exit__(None, None, None)
        off = ou(x, 499.0, ctx, add).sample
    with onet:
        ref = ou(x, 499.0, ctx, add).sample
    (ref * gy).sum().backward()
    with pnet:
        got = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
    (got * gy.cuda()).sum().backward()
    e = rel(got, ref)
    assert rel(ref, off) > 1e-3 and e < 1.8e-3, (rel(ref, off), e)  # measured 8.3e-4 .. 1.18e-3
    errs = {}
    for what in ("down", "up", "scale"):
        num = den = 0.0
        for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
            a, b = {"down": (lp.lora_down.grad, lo.lora_down.weight.grad), "up": (lp.lora_up.grad, lo.lora_up.weight.grad),
                    "scale": (lp.dora_scale_grad, lo.dora_scale.grad)}[what]
            assert b is not None and float(b.abs().max()) > 0
            num += float((a.cpu() - b).norm() ** 2)
            den += float(b.norm() ** 2)
        errs[what] = (num / den) ** 0.5
    print(f"{model} {method} dora: eps {e:.2e}, grads " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert all(v < 5.3e-3 for v in errs.values()), errs  # measured 2.2e-3 .. 3.5e-3 (1.5 x)
    pnet.__exit__(None, None, None)
    with torch.no_grad():
        assert rel(pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample, off) < LOOSE[dtype]


@pytest.mark.parametrize("model", list(CFGS))
def test_per_sample_multipliers_equal_separate_passes(model):
    """smi_unet_forward_multi: the image-slider step's two sides (adaptor at +s on one CFG pair, -s on the other) as ONE UNet
    pass with one multiplier per adapted sample must give the eps of two separate passes bit for bit (per-sample arithmetic
    does not depend on batch composition; the multiplier only rescales the rows of xa = x down^T), and -- after one
    backward over all four samples -- the LoRA gradient of the two separate backward passes accumulated (fp32 order).
    A network with conv adaptors must refuse unequal multipliers."""
    from sliders_conceptmod_amd import _native
    dtype = torch.float16
    ocfg, ou, onet, pu, pnet = build_pair(model, dtype)
    x, ctx, add = inputs(ocfg, 4, 16)
    cadd = cuda_add(add)
    te, ti = (cadd["text_embeds"].half().contiguous(), cadd["time_ids"].float().contiguous()) if cadd else (None, None)
    xc, cc = x.cuda().float().contiguous(), ctx.cuda().half().contiguous()
    flat, n_down, _ = pnet.engine_params()
    down, up = flat[:n_down], flat[n_down:]
    eng = pu._ensure_engine(4, 16, 16, 77)
    gy = (torch.randn(4, 4, 16, 16, generator=torch.Generator().manual_seed(9)) * 1e-4).cuda()
    s = 1.5
    # one pass, per-sample multipliers
    eps = eng.forward(xc, 499.0, cc, te, ti, down, up, [s, s, -s, -s], True)
    g1 = torch.zeros_like(flat)
    eng.backward(gy.contiguous(), g1[:n_down], g1[n_down:])
    # two passes
    g2 = torch.zeros_like(flat)
    parts = []
    for sl, m in ((slice(0, 2), s), (slice(2, 4), -s)):
        e = eng.forward(xc[sl].contiguous(), 499.0, cc[sl].contiguous(), None if te is None else te[sl].contiguous(),
                        None if ti is None else ti[sl].contiguous(), down, up, m, True)
        eng.backward(gy[sl].contiguous(), g2[:n_down], g2[n_down:])
        parts.append(e)
    assert torch.equal(eps, torch.cat(parts)), "per-sample multipliers changed a sample's forward arithmetic"
    relg = float((g1 - g2).norm() / g2.norm())
    assert relg < 1e-5, relg
    assert rel(eps[:2], eps[2:]) > 1e-4  # (+s and -s really differ)
    # conv adaptors: refused
    _, _, _, pu2, pnet2 = build_pair(model, dtype, c3lier=True)
    flat2, nd2, _ = pnet2.engine_params()
    eng2 = pu2._ensure_engine(4, 16, 16, 77)
    with pytest.raises(_native.SmiError, match="per-sample multipliers"):
        eng2.forward(xc, 499.0, cc, te, ti, flat2[:nd2], flat2[nd2:], [s, s, -s, -s], False)


@pytest.mark.parametrize("model", list(CFGS))
def test_grouped_time_emb_proj_equals_one_gemm_per_resnet(model):
    """engine.hip build_temb_group: every resnet's time_emb_proj stacked into ONE GEMM per pass, conv1 reading its column
    block as a row vector with a leading dimension -- against `SMI_TEMB_GROUP=0` (one GEMM per resnet), in two processes
    (the switch is read once).  Same products, different split-K partition of the K loop: rounding-level agreement, and
    the adapted pass with gradients must agree as well (no gradient flows into the views)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import json, sys, torch
sys.path.insert(0, {root!r})
from tests.test_engine_gpu import build_pair, inputs, cuda_add
ocfg, ou, onet, pu, pnet = build_pair({model!r}, torch.float16)
x, ctx, add = inputs(ocfg, 2, 16)
pnet.__enter__()
eps = pu(x.cuda(), 499.0, encoder_hidden_states=ctx.cuda(), added_cond_kwargs=cuda_add(add)).sample
eps.square().mean().backward()
g = torch.cat([p.grad.flatten() for p in pnet.parameters() if p.grad is not None])
print(json.dumps({{"eps": eps.flatten().tolist(), "g": g[:: max(1, g.numel() // 4096)].tolist()}}))
"""
    res = {}
    for arm in ("1", "0"):
        env = dict(os.environ, SMI_TEMB_GROUP=arm)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        res[arm] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    a, b = torch.tensor(res["1"]["eps"]), torch.tensor(res["0"]["eps"])
    ga, gb = torch.tensor(res["1"]["g"]), torch.tensor(res["0"]["g"])
    e, eg = rel(a, b), rel(ga, gb)
    print(f"{model}: grouped vs per-resnet time_emb_proj: eps rel diff {e:.2e}, LoRA-gradient rel diff {eg:.2e}")
    assert e < 1e-3 and eg < 5e-3, (e, eg)
    assert float(gb.abs().max()) > 0
