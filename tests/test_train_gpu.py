"""The product trainers (sliders_conceptmod_amd.train_lora / train_lora_xl `train()`, i.e. the CLI code path) on a
real MI355X against the trajectory the REFERENCE's own harness produced on the CPU oracle
(tests/golden/make_goldens.py section 5): same seeds, same RNG draw order, same step order, same optimiser.
fp16 engine vs fp32 reference run: tolerances are the storage-noise level measured in test_engine_gpu.py."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref as OU

CFGS = {"tiny_sd1x": OU.tiny_sd1x_config, "tiny_sdxl": OU.tiny_sdxl_config}
SIGN_FLIP_BAR = 0.01  # measured 0.08 % (SD-1.x: 2 of 2560) / 0.29 % (SD-XL: 6 of 2048); share of lora_up elements whose sign may differ from the reference's Adam trajectory (see the test)


class GoldenEncoder:
    """Injects the fixture's prompt embeddings: prompt text == the role name."""

    def __init__(self, cfg, keys):
        g = torch.Generator().manual_seed(4)
        self.emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in keys}
        self.pooled = None
        if cfg.addition_embed_type == "text_time":
            pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
            self.pooled = {k: torch.randn(1, pdim, generator=g) for k in keys}


def make(model, tmp_path, xl, optimizer_args=""):
    import sliders_conceptmod_amd.config_util as CU
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.prompt_util as PRU
    import sliders_conceptmod_amd.unet as PU
    ocfg = CFGS[model]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0)
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
    pu.load_state_dict(ou.state_dict())
    keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if xl else [])
    ge = GoldenEncoder(ocfg, keys)

    class Enc(MU.SyntheticTextEncoder):
        def __init__(self):
            pass

        def encode(self, prompt):
            return (ge.emb[prompt], ge.pooled[prompt]) if xl else ge.emb[prompt]

    cfg = CU.RootConfig(
        prompts_file="unused", pretrained_model=CU.PretrainedModelConfig(name_or_path="injected"),
        network=CU.NetworkConfig(type="lierla", rank=4, alpha=1.0, training_method="noxattn"),
        train=CU.TrainConfig(precision="float16", noise_scheduler="euler_a" if xl else "ddim", iterations=6, lr=2e-3,
                             optimizer="AdamW", optimizer_args=optimizer_args, lr_scheduler="constant",
                             max_denoising_steps=8, cfg=1.0),
        save=CU.SaveConfig(name="t", path=str(tmp_path), per_steps=1000), logging=CU.LoggingConfig(),
        other=CU.OtherConfig())
    kw = dict(target="target", positive="positive", neutral="neutral", unconditional="unconditional", action="enhance",
              guidance_scale=4.0, resolution=128, batch_size=1)
    if xl:
        kw["negative"] = "negative"
    prompts = [PRU.PromptSettings(**kw)]
    sched = MU.create_noise_scheduler("euler_a" if xl else "ddim")
    return cfg, prompts, (None, Enc(), pu, sched)


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_cli_train_reproduces_reference_trajectory(goldens, tmp_path, model):
    t, man = goldens
    meta = man[f"traj/{model}"]
    xl = model.endswith("xl")
    cfg, prompts, models = make(model, tmp_path, xl)
    losses = []
    torch.manual_seed(1)  # LoRA init, then the training loop continues the same stream (as in the fixture run)
    if xl:
        from sliders_conceptmod_amd.train_lora_xl import train
        net = train(cfg, prompts, torch.device("cuda:0"), on_step_complete=None, rank=4, save_file=True, models=models)
    else:
        from sliders_conceptmod_amd.train_lora import train
        net = train(cfg, prompts, torch.device("cuda:0"), models=models,
                    on_step_complete=lambda i, l: losses.append(l))
    out = tmp_path / ("t_last.safetensors" if xl else "t_last.pt")
    assert out.exists()
    if xl:
        losses = net.training_losses
    assert len(losses) == 6
    for a, b in zip(losses, meta["losses"]):  # measured: within 0.6 % (fp16 engine vs the fp32 reference run)
        assert abs(a - b) <= 0.015 * abs(b), (losses, meta["losses"])
    sd = net.state_dict()
    assert set(sd.keys()) == set(meta["norms"].keys())
    # weights after 6 AdamW steps: Adam's sign-like first steps amplify 16-bit gradient noise on tiny gradients, so
    # compare per-tensor norms (2 %) and the direction of the stored modules (cosine > 0.97)
    for k, nrm in meta["norms"].items():
        if k.endswith("alpha"):
            continue
        assert float(sd[k].float().norm()) == pytest.approx(nrm, rel=2e-2, abs=2e-5), k
    flips = total = 0
    worst_cos = 1.0
    for k in [k for k in t if k.startswith(f"traj/{model}/sd/") and "lora_up" in k]:
        a, b = sd[k.split("/sd/")[1]].float().cpu().flatten(), t[k].float().flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
        worst_cos = min(worst_cos, cos)
        assert cos > 0.97, f"{k}: cosine {cos:.4f}"
        # lora_up starts at zero, so after a few sign-like Adam steps an element's SIGN is the sign of its first
        # gradients: the share of elements whose sign differs from the reference run is the share of gradient elements
        # that sit inside the 16-bit noise (the bar as a number)
        flips += int((torch.sign(a) != torch.sign(b)).sum())
        total += a.numel()
    frac = flips / max(total, 1)
    print(f"{model}: lora_up elements with a sign different from the reference trajectory: {flips} / {total} = "
          f"{frac:.3%}; worst per-tensor cosine {worst_cos:.4f}")
    assert frac < SIGN_FLIP_BAR, frac


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_cli_train_final_lora_tensors_elementwise(goldens, tmp_path, model, fused):
    """The saved LoRA tensors themselves, element by element, against the reference harness' "smooth" trajectory
    (tests/golden/make_goldens.py section 7: the same six steps with Adam eps = 1e-3 >> |gradient|, so the update is
    proportional to the gradient instead of its sign and 16-bit gradient noise is not amplified).  SD-1.x goes through
    the config's `optimizer_args`, SD-XL through train(optimizer_kwargs=...) (its AdamW is hard-coded, as in the
    reference); `fused` runs the same steps through SliderStep (native loss / clip / AdamW kernels)."""
    t, man = goldens
    meta = man[f"traj_smooth/{model}"]
    xl = model.endswith("xl")
    eps = meta["optimizer_kwargs"]["eps"]
    cfg, prompts, models = make(model, tmp_path, xl, optimizer_args="" if xl else f"eps={eps}")
    losses = []
    torch.manual_seed(1)
    if xl:
        from sliders_conceptmod_amd.train_lora_xl import train
        net = train(cfg, prompts, torch.device("cuda:0"), rank=4, save_file=True, models=models, fused_step=fused,
                    optimizer_kwargs={"eps": eps})
        losses = net.training_losses
    else:
        from sliders_conceptmod_amd.train_lora import train
        net = train(cfg, prompts, torch.device("cuda:0"), models=models,
                    on_step_complete=lambda i, l: losses.append(l), fused_step=fused)
    for a, b in zip(losses, meta["losses"]):
        assert abs(a - b) <= 0.015 * abs(b), (losses, meta["losses"])
    # reload what was written (the saved file is the product) and compare every stored tensor
    out = tmp_path / ("t_last.safetensors" if xl else "t_last.pt")
    if xl:
        from safetensors.torch import load_file
        saved = load_file(str(out))
    else:
        saved = torch.load(out, weights_only=True)
    keys = [k for k in t if k.startswith(f"traj_smooth/{model}/sd/")]
    assert len(keys) == 2 * len(net.unet_loras)
    live_sd = net.state_dict()
    # element-wise scale: the largest element of the same kind (all lora_up / all lora_down tensors of the network)
    gmax = {kind: max(float(t[k].abs().max()) for k in keys if kind in k) for kind in ("lora_up", "lora_down")}
    worst_el = worst_t = 0.0
    num = den = 0.0
    for k in keys:
        name = k.split("/sd/")[1]
        ref = t[k].float()
        live = live_sd[name].float().cpu()
        kind = "lora_up" if "lora_up" in name else "lora_down"
        # fp32 master weights vs the fp32 reference run
        worst_el = max(worst_el, float((live - ref).abs().max()) / gmax[kind])
        worst_t = max(worst_t, float((live - ref).norm() / ref.norm()))
        num += float((live - ref).norm() ** 2)
        den += float(ref.norm() ** 2)
        # the file holds them in the train dtype (fp16), as the reference writes them: one storage rounding on top
        tol16 = 2.0 ** -10 * float(ref.abs().max()) + 6e-8  # half an fp16 ulp at the top + the subnormal step
        assert float((saved[name].float() - live).abs().max()) <= tol16, name
    glob = (num / den) ** 0.5
    print(f"{model} fused={fused}: global rel L2 {glob:.2e}, worst tensor rel L2 {worst_t:.2e}, worst element error / "
          f"largest element of its kind {worst_el:.2e}")
    # measured (fp16 engine vs fp32 reference harness run): global 2.1e-4; worst single tensor 1.3e-2 (a module whose
    # gradient sits at the 16-bit noise level); worst single element 1.3e-3 (SD-1.x) / 3.0e-3 (SD-XL) of the largest
    # element of its kind.  Bars = 1.5-2x measured; north_star's 1e-3 relative is met globally, not by every element.
    assert glob <= 4e-4, glob
    assert worst_el <= 5e-3, worst_el
    assert worst_t <= 2.5e-2, worst_t


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_fused_step_matches_autograd_step(goldens, tmp_path, model):
    """--fused_step (SliderStep: native pre-roll, one batched UNet pass, native loss / clip / AdamW, no autograd) == the
    drop-in autograd loop, same seeds -- both text trainers (train_lora.py: DDIM, no clip; train_lora_xl.py: Euler-a,
    clip 0.2, cosine LR)."""
    xl = model.endswith("xl")
    res, losses = [], []
    for fused in (False, True):
        cfg, prompts, models = make(model, tmp_path, xl)
        torch.manual_seed(1)
        ls = []
        if xl:
            from sliders_conceptmod_amd.train_lora_xl import train
            net = train(cfg, prompts, torch.device("cuda:0"), rank=4, save_file=True, models=models, fused_step=fused)
            ls = net.training_losses
        else:
            from sliders_conceptmod_amd.train_lora import train
            net = train(cfg, prompts, torch.device("cuda:0"), models=models, fused_step=fused,
                        on_step_complete=lambda i, l: ls.append(l))
        res.append(net.flat.detach().clone())
        losses.append(list(ls))
    a, b = res
    rel = float((a - b).norm() / b.norm())
    print(f"{model}: fused vs autograd parameter distance {rel:.2e}; losses {losses}")
    assert rel < 1.1e-3, f"fused vs autograd parameter distance {rel:.2e}"  # measured 6.9e-4 (SD-1.x), 5.5e-6 (SD-XL)
    # the first loss is bit-equal; later ones see Adam's sign-like first updates amplify fp32-order differences (measured
    # 2e-3 on SD-1.x at lr 2e-3)
    assert losses[0][0] == pytest.approx(losses[1][0], rel=1e-5)
    for x, y in zip(*losses):
        assert abs(x - y) <= 1e-2 * abs(y), losses


def _latent_folders(root, n=3, hw=16):
    g = torch.Generator().manual_seed(5)
    for folder, bias in (("big", 0.3), ("small", -0.3)):
        (root / folder).mkdir(parents=True)
        for i in range(n):
            torch.save(torch.randn(4, hw, hw, generator=g) + bias, root / folder / f"p{i}.pt")


def _image_cfg(tmp_path, xl, iterations=4, eps=None):
    import sliders_conceptmod_amd.config_util as CU
    import sliders_conceptmod_amd.prompt_util as PRU
    cfg = CU.RootConfig(
        prompts_file="unused",
        pretrained_model=CU.PretrainedModelConfig(name_or_path="synthetic://tiny_sdxl" if xl else "synthetic://tiny_sd1x"),
        network=CU.NetworkConfig(type="lierla", rank=4, alpha=1.0, training_method="noxattn"),
        train=CU.TrainConfig(precision="float16", noise_scheduler="ddim", iterations=iterations, lr=1e-3,
                             optimizer="AdamW", lr_scheduler="constant", max_denoising_steps=10),
        save=CU.SaveConfig(name="img", path=str(tmp_path / "out"), per_steps=1000), logging=CU.LoggingConfig(),
        other=CU.OtherConfig())
    prompts = [PRU.PromptSettings(target="t", positive="big", neutral="small", unconditional="", resolution=128),
               PRU.PromptSettings(target="t", positive="large", neutral="tiny", unconditional="", resolution=128)]
    return cfg, prompts


@pytest.mark.parametrize("xl", [True, False])
def test_image_slider_fused_step_matches_autograd_loop(tmp_path, xl):
    """train_lora_scale(_xl).train(fused_step=True) -- step.ImageSliderStep: two adapted forward + backward pairs at
    slider +s / -s accumulating into one flat gradient, native AdamW, no autograd graph -- against the autograd loop
    (image_slider_step + torch.optim.AdamW) on the same folders, seeds and prompt file (two entries: the per-step
    prompt-pair draw of I/train_lora-scale-xl.py:183 is exercised)."""
    import random
    from sliders_conceptmod_amd.train_lora_scale_xl import train
    _latent_folders(tmp_path / "data")
    res, losses = [], []
    for fused in (False, True):
        cfg, prompts = _image_cfg(tmp_path, xl)
        torch.manual_seed(0)
        random.seed(0)
        net = train(cfg, prompts, torch.device("cuda:0"), str(tmp_path / "data"), ["big", "small"], [1, -1], xl=xl,
                    fused_step=fused)
        res.append(net.flat.detach().clone())
        losses.append(net.training_losses)
    rel = float((res[1] - res[0]).norm() / res[0].norm())
    print(f"xl={xl}: fused vs autograd parameter distance {rel:.2e}")
    assert rel < 3e-4, rel  # measured 1.07e-4 (SD-XL), 9.4e-5 (SD-1.x)
    for (ah, al), (bh, bl) in zip(*losses):
        assert abs(ah - bh) <= 1e-2 * abs(ah) and abs(al - bl) <= 1e-2 * abs(al), losses


def test_image_slider_two_sided_step_matches_oracle(goldens):
    """Row a-10: slider at +s on the 'high' latent and -s on the 'low' latent, two backward()s accumulate; against the
    oracle's autograd on the same inputs (LoRA multiplier = +/- s, MSE to the true noise)."""
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.unet as PU
    from sliders_conceptmod_amd.prompt_util import PromptEmbedsXL
    from sliders_conceptmod_amd.train_lora_scale_xl import image_slider_step
    from oracle import sched_ref as S, slider_ref as R
    ocfg = CFGS["tiny_sdxl"]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0).requires_grad_(False).eval()
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
    pu.load_state_dict(ou.state_dict())
    pu = pu.to("cuda", torch.float16).requires_grad_(False).eval()
    torch.manual_seed(1)
    onet = R.LoRANetworkRef(ou, 4, 1.0, 1.0, "noxattn")
    torch.manual_seed(1)
    pnet = L.LoRANetwork(pu, rank=4, alpha=1.0, train_method="noxattn")
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
            w = torch.randn(lo.lora_up.weight.shape, generator=g) * 0.05
            lo.lora_up.weight.copy_(w)
            lp.lora_up.weight.copy_(w)
    pnet.to("cuda")
    g = torch.Generator().manual_seed(5)
    emb = {k: (torch.randn(1, 77, 64, generator=g), torch.randn(1, 64, generator=g)) for k in ("pos", "neu")}
    lat = {k: torch.randn(1, 4, 16, 16, generator=g) for k in ("low", "high")}
    noise = torch.randn(1, 4, 16, 16, generator=g)
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]])
    osch, psch = S.create_noise_scheduler_ref("ddim"), MU.create_noise_scheduler("ddim")
    osch.set_timesteps(1000), psch.set_timesteps(1000)
    t = psch.timesteps[600]
    scale = 2.0
    # oracle: same arithmetic with autograd
    for sgn, key, ek in ((+1.0, "high", "pos"), (-1.0, "low", "neu")):
        onet.set_lora_slider(sgn * scale)
        x = osch.add_noise(lat[key], noise, torch.tensor([int(t)]))
        te, pe = emb[ek]
        with onet:
            pred = R.predict_noise_xl(ou, osch, osch.timesteps[600], x, torch.cat([te, te]), torch.cat([pe, pe]),
                                      torch.cat([tid, tid]), guidance_scale=1.0)
        torch.nn.functional.mse_loss(pred, noise).backward()
    nl = psch.add_noise(lat["low"], noise, t).cuda()
    nh = psch.add_noise(lat["high"], noise, t).cuda()
    pos = PromptEmbedsXL(emb["pos"][0].cuda().half(), emb["pos"][1].cuda().half())
    neu = PromptEmbedsXL(emb["neu"][0].cuda().half(), emb["neu"][1].cuda().half())
    image_slider_step(pu, pnet, psch, nl, nh, noise.cuda(), noise.cuda(), t, pos, neu, tid.cuda(), scale)
    num = den = 0.0
    for lo, lp in zip(onet.unet_loras, pnet.unet_loras):
        for a, b in ((lp.lora_down.grad, lo.lora_down.weight.grad), (lp.lora_up.grad, lo.lora_up.weight.grad)):
            num += float((a.cpu() - b).norm() ** 2)
            den += float(b.norm() ** 2)
    rel = (num / den) ** 0.5
    assert rel < 1e-2, f"two-sided image-slider LoRA gradient vs oracle: {rel:.2e}"


@pytest.mark.parametrize("with_negative", [True, False])
def test_dedup_uncond_step_is_identical_to_the_full_step(with_negative):
    """SliderStep(dedup_uncond=True) runs each distinct frozen sample once (8B -> 6B / 5B samples); loss, gradient and
    updated parameters must equal the full 4-pass step's (same kernels on the same per-sample inputs)."""
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.unet as PU
    from sliders_conceptmod_amd.step import SliderStep
    ocfg = CFGS["tiny_sdxl"]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0)
    res = []
    for dedup in (False, True):
        pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
        pu.load_state_dict(ou.state_dict())
        pu = pu.to("cuda", torch.float16).requires_grad_(False).eval()
        torch.manual_seed(1)
        net = L.LoRANetwork(pu, rank=4, alpha=1.0, train_method="noxattn").to("cuda")
        with torch.no_grad():
            net.flat_up.copy_(torch.randn(net.flat_up.shape, generator=torch.Generator().manual_seed(2)) * 2e-2)
        sched = MU.create_noise_scheduler("euler_a")
        sched.set_timesteps(1000)
        t = sched.timesteps[400]
        g = torch.Generator().manual_seed(4)
        keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if with_negative else [])
        emb = {k: torch.randn(1, 77, 64, generator=g) for k in keys}
        pooled = {k: torch.randn(1, 64, generator=g) for k in keys}
        tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]])
        step = SliderStep(pu, net, sched, lr=1e-3, weight_decay=1e-6, max_grad_norm=0.2, dedup_uncond=dedup)
        cond = step.make_conditioning(emb, 2, pooled, tid)
        lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(3)).cuda()
        losses = [float(step.train_step(lat, t, cond, "enhance", 2.0).item()) for _ in range(2)]
        res.append((losses, step.grad.clone(), net.flat.detach().clone()))
    (l0, g0, p0), (l1, g1, p1) = res
    assert l0 == pytest.approx(l1, rel=1e-6), (l0, l1)
    assert float((g0 - g1).abs().max()) <= 1e-6 * float(g0.abs().max())
    assert float((p0 - p1).abs().max()) <= 1e-7


def test_native_preroll_matches_the_drop_in_diffusion_xl():
    """SliderStep.preroll == train_util.diffusion_xl (the reference's pre-roll loop, T/train_util.py:677-708) on the
    same engine: same UNet calls, CFG mix and Euler-a steps (host RNG seeded identically)."""
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.unet as PU
    from sliders_conceptmod_amd import train_util as TU
    from sliders_conceptmod_amd.step import SliderStep
    ocfg = CFGS["tiny_sdxl"]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0)
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
    pu.load_state_dict(ou.state_dict())
    pu = pu.to("cuda", torch.float16).requires_grad_(False).eval()
    torch.manual_seed(1)
    net = L.LoRANetwork(pu, rank=4, alpha=1.0, train_method="noxattn").to("cuda")
    with torch.no_grad():
        net.flat_up.copy_(torch.randn(net.flat_up.shape, generator=torch.Generator().manual_seed(2)) * 2e-2)
    g = torch.Generator().manual_seed(4)
    keys = ["target", "positive", "neutral", "unconditional"]
    emb = {k: torch.randn(1, 77, 64, generator=g) for k in keys}
    pooled = {k: torch.randn(1, 64, generator=g) for k in keys}
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]])
    lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(3)).cuda()
    out = []
    for native in (True, False):
        sched = MU.create_noise_scheduler("euler_a")
        sched.set_timesteps(12)
        lat0 = lat * sched.init_noise_sigma
        torch.manual_seed(11)
        if native:
            step = SliderStep(pu, net, sched)
            cond = step.make_conditioning(emb, 2, pooled, tid)
            out.append(step.preroll(lat0, cond, 4, 3.0))
        else:
            te = TU.concat_embeddings(emb["unconditional"], emb["target"], 2).cuda().half()
            pe = TU.concat_embeddings(pooled["unconditional"], pooled["target"], 2).cuda().half()
            ti = TU.concat_embeddings(tid, tid, 2).cuda()
            with net:
                out.append(TU.diffusion_xl(pu, sched, lat0, te, pe, ti, guidance_scale=3.0, total_timesteps=4))
    a, b = out
    assert torch.isfinite(a).all()
    assert float((a - b.float()).abs().max()) <= 1e-5 * float(b.abs().max())


def test_two_rank_bench_flow_on_one_gpu():
    """The N > 1 path of bench.py (one process per rank, barrier + max-over-ranks timing, LoRA-gradient all-reduce before
    the clip, rank 0 prints the JSON line) rehearsed with two ranks sharing this box's single GPU over gloo
    (SMI_BENCH_ONE_DEVICE=1; RCCL refuses two ranks on one device).  The measured path is one rank per GPU over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SMI_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--config", "tiny_sdxl", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]  # whole-job aggregate: world / step time
    assert d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"]


def test_cli_main_runs_the_shipped_c3lier_config_end_to_end(tmp_path, monkeypatch):
    """`python -m sliders_conceptmod_amd.train_lora --config_file <shipped data/config.yaml> --alpha 1 --rank 4 --name age
    --attributes "white, black"` through main(): the shipped YAMLs (c3lier, the age-slider prompts with attributes), the
    product config_util / prompt_util, model factory, trainer and writer -- only the model is swapped for the tiny
    synthetic SD-1.x architecture and the run shortened.  The written file must strict-load into the ORACLE's
    restatement of the reference's LoRANetwork with the c3lier target list (the format eval scripts consume)."""
    import yaml
    import sliders_conceptmod_amd.config_util as CU
    from sliders_conceptmod_amd import train_lora
    from oracle import slider_ref as R
    raw = yaml.safe_load(open(CU.resolve_data_path("data/config.yaml")))
    raw["pretrained_model"]["name_or_path"] = "synthetic://tiny_sd1x"
    raw["train"].update(iterations=4, max_denoising_steps=6, precision="float16")
    raw["save"].update(path=str(tmp_path / "models"), per_steps=2)
    praw = yaml.safe_load(open(CU.resolve_data_path("data/prompts.yaml")))
    for e in praw:
        e["resolution"] = 128
    (tmp_path / "prompts.yaml").write_text(yaml.safe_dump(praw))
    raw["prompts_file"] = str(tmp_path / "prompts.yaml")
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(raw))
    torch.manual_seed(5)
    args = train_lora.build_parser().parse_args(["--config_file", str(tmp_path / "config.yaml"), "--alpha", "1.0",
                                                 "--rank", "4", "--name", "age", "--attributes", "white, black"])
    train_lora.main(args)
    out_dir = tmp_path / "models" / "age_alpha1.0_rank4_noxattn"
    assert (out_dir / "age_alpha1.0_rank4_noxattn_last.pt").exists()
    assert (out_dir / "age_alpha1.0_rank4_noxattn_2steps.pt").exists()  # per_steps = 2, T/train_lora.py:317-327
    sd = torch.load(out_dir / "age_alpha1.0_rank4_noxattn_last.pt", weights_only=True)
    ou = OU.UNet2DConditionModel(OU.tiny_sd1x_config())
    onet = R.LoRANetworkRef(ou, 4, 1.0, 1.0, "noxattn", target_replace=R.C3LIER_TARGET_REPLACE)
    onet.load_state_dict(sd, strict=True)
    assert len(onet.unet_loras) == 150
    conv_up = [v for k, v in sd.items() if "conv1.lora_up" in k]
    assert conv_up and all(v.ndim == 4 and v.dtype == torch.float16 for v in conv_up)
    assert any(float(v.float().abs().max()) > 0 for v in conv_up), "conv adaptors did not train"


@pytest.mark.parametrize("sched_name,model,gscale", [("euler_a", "tiny_sdxl", 3.0), ("ddim", "tiny_sd1x", 3.0),
                                                     ("euler_a", "tiny_sdxl", 1.0)])
def test_native_preroll_matches_the_oracle(sched_name, model, gscale):
    """Row f-1: the on-device pre-roll (SliderStep.preroll: engine forward, smi_cfg_combine, smi_sched_step per denoising
    step, adaptor ON, CFG scale 3) against the ORACLE's diffusion(_xl) (T/train_util.py:306-327, 677-708) on the CPU --
    an independent implementation, not the same engine.  Euler-ancestral noise is drawn on the host from the same seed
    on both sides.  gscale 1.0 is the SD-XL trainer's default `train.cfg`: the oracle runs the reference's doubled batch
    (u + 1 (t - u)), the product the conditional half alone (SliderStep.preroll_skip_dead_half) and, for comparison, the
    doubled batch too."""
    import sliders_conceptmod_amd.model_util as MU
    from sliders_conceptmod_amd.step import SliderStep
    from oracle import sched_ref as S, slider_ref as R
    from test_engine_gpu import build_pair
    xl = model.endswith("xl")
    ocfg, ou, onet, pu, pnet = build_pair(model, torch.float16)
    g = torch.Generator().manual_seed(4)
    keys = ["target", "positive", "neutral", "unconditional"]
    emb = {k: torch.randn(1, 77, ocfg.cross_attention_dim, generator=g) for k in keys}
    pooled = tid = None
    if xl:
        pdim = ocfg.projection_class_embeddings_input_dim - 6 * ocfg.addition_time_embed_dim
        pooled = {k: torch.randn(1, pdim, generator=g) for k in keys}
        tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]])
    lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(3))
    osch, psch = S.create_noise_scheduler_ref(sched_name), MU.create_noise_scheduler(sched_name)
    osch.set_timesteps(12), psch.set_timesteps(12)
    lat0 = lat * float(psch.init_noise_sigma)
    te = R.concat_embeddings(emb["unconditional"], emb["target"], 2)
    torch.manual_seed(11)
    with torch.no_grad(), onet:
        if xl:
            ref = R.diffusion_xl(ou, osch, lat0, te, R.concat_embeddings(pooled["unconditional"], pooled["target"], 2),
                                 R.concat_embeddings(tid, tid, 2), guidance_scale=gscale, total_timesteps=5)
        else:
            ref = R.diffusion(ou, osch, lat0, te, total_timesteps=5, guidance_scale=gscale)
    step = SliderStep(pu, pnet, psch)
    cond = step.make_conditioning(emb, 2, pooled, tid)
    torch.manual_seed(11)
    got = step.preroll(lat0.cuda(), cond, 5, gscale)
    e = float((got.cpu() - ref).norm() / ref.norm())
    print(f"{model} {sched_name} g={gscale}: 5-step pre-roll vs oracle rel err {e:.2e}")
    assert e < 3e-3, e
    if gscale == 1.0:  # the doubled batch at scale 1 lands on the same latents (to the engine's own rounding noise)
        full = SliderStep(pu, pnet, psch, preroll_skip_dead_half=False)
        torch.manual_seed(11)
        got2 = full.preroll(lat0.cuda(), cond, 5, gscale)
        e2 = float((got2.cpu() - ref).norm() / ref.norm())
        d = float((got2 - got).norm() / got.norm())
        print(f"   doubled batch: vs oracle {e2:.2e}; conditional-half-only vs doubled {d:.2e}")
        assert e2 < 3e-3 and d < 3e-3, (e2, d)


def test_xl_trainer_peft_type_dora(tmp_path):
    """train_lora_xl.train(peft_type="dora") (T/train_lora_xl.py:87-90): DoRANetwork on the engine through the CLI code
    path; the saved file strict-loads into the oracle's restatement of the reference's DoRANetwork."""
    from safetensors.torch import load_file
    from sliders_conceptmod_amd.train_lora_xl import train
    from oracle import slider_ref as R
    cfg, prompts, models = make("tiny_sdxl", tmp_path, True)
    cfg.train.iterations = 3
    torch.manual_seed(1)
    net = train(cfg, prompts, torch.device("cuda:0"), rank=4, save_file=True, models=models, peft_type="dora")
    assert len(net.training_losses) == 3 and all(l == l and l > 0 for l in net.training_losses)
    sd = load_file(str(tmp_path / "t_last.safetensors"))
    onet = R.DoRANetworkRef(OU.UNet2DConditionModel(OU.tiny_sdxl_config()), 4, 1.0, 1.0, "noxattn")
    onet.load_state_dict(sd, strict=True)
    assert any(k.endswith(".dora_scale") for k in sd)
    assert float(net.flat_up.detach().abs().max()) > 0  # lora_up left zero: the adaptor trained
    with pytest.raises(ValueError):
        train(cfg, prompts, torch.device("cuda:0"), rank=4, save_file=False, models=make("tiny_sdxl", tmp_path, True)[2],
              peft_type="ia3")


def test_inference_side_slider_sweep_matches_oracle():
    """Row f-4, second half: the eval scripts' sampling loop (E/generate_images_sd1.py:170-190) -- adaptor gated off while
    t > start_noise, then on at the slider scale, CFG 7.5, DDIM -- on the engine vs the same loop on the oracle."""
    import sliders_conceptmod_amd.model_util as MU
    from sliders_conceptmod_amd import train_util as TU
    from oracle import sched_ref as S, slider_ref as R
    from test_engine_gpu import build_pair
    ocfg, ou, onet, pu, pnet = build_pair("tiny_sd1x", torch.float16)
    g = torch.Generator().manual_seed(4)
    te = torch.cat([torch.randn(1, 77, ocfg.cross_attention_dim, generator=g) for _ in range(2)])
    lat = torch.randn(1, 4, 16, 16, generator=g)
    osch, psch = S.create_noise_scheduler_ref("ddim"), MU.create_noise_scheduler("ddim")
    outs = {}
    for scale in (0.0, 2.0, -2.0):
        osch.set_timesteps(10)
        x = lat.clone()
        with torch.no_grad():
            for t in osch.timesteps:
                onet.set_lora_slider(0 if t > 600 else scale)
                with onet:
                    eps = R.predict_noise(ou, osch, t, x, te, guidance_scale=7.5)
                x = osch.step(eps, t, x).prev_sample
        got = TU.slider_sweep_latents(pu, pnet, psch, lat.cuda(), te.cuda().half(), scale, 600, 7.5, 10)
        e = float((got.cpu() - x).norm() / x.norm())
        outs[scale] = got
        print(f"slider sweep scale {scale:+.0f}: final latents vs oracle rel err {e:.2e}")
        assert e < 5e-3, e
    assert float((outs[2.0] - outs[0.0]).abs().max()) > 1e-3 and float((outs[2.0] - outs[-2.0]).abs().max()) > 1e-3
