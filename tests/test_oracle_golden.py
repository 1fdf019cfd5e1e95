"""The oracle (oracle/*.py, CPU restatement) against the fixtures produced by the reference's own harness code
(tests/golden/make_goldens.py).  CPU only."""
import hashlib
import json

import pytest
import torch

from oracle import sched_ref as S
from oracle import slider_ref as R
from oracle import unet_ref as U

CFGS = {"sd1x": U.sd1x_config, "sdxl": U.sdxl_config, "tiny_sd1x": U.tiny_sd1x_config,
        "tiny_sdxl": U.tiny_sdxl_config}


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("method", ["noxattn", "innoxattn", "selfattn", "xattn", "full", "xattn-strict"])
def test_selection_matches_reference(goldens, model, method):
    _, man = goldens
    with torch.device("meta"):
        unet = U.UNet2DConditionModel(CFGS[model]())
    sel = R.select_lora_targets(unet, method)
    names = [s[0] for s in sel]
    g = man["selection"][f"{model}/{method}"]
    assert len(names) == g["n_modules"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["names_sha"]
    shapes = {}
    for n, _p, child in sel:
        shapes[f"{n}.alpha"] = []
        shapes[f"{n}.lora_down.weight"] = [4, child.in_features]
        shapes[f"{n}.lora_up.weight"] = [child.out_features, 4]
    assert hashlib.sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest() == g["shapes_sha"]
    assert sum(4 * (c.in_features + c.out_features) for _, _, c in sel) == g["n_params"]


def test_bad_method_raises():
    with torch.device("meta"):
        unet = U.UNet2DConditionModel(U.tiny_sd1x_config())
    with pytest.raises(NotImplementedError):
        R.select_lora_targets(unet, "bogus")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lora_linear_forward_backward(goldens, seed):
    t, man = goldens
    g = lambda k: t[f"lora_linear/{seed}/{k}"]
    meta = man[f"lora_linear/{seed}"]
    lin = torch.nn.Linear(24, 40)
    lin.weight.data.copy_(g("w"))
    lin.bias.data.copy_(g("b"))
    mod = R.LoRAModuleRef("lin", lin, 1.0, 4, meta["alpha"])
    assert mod.scale == meta["scale"]
    mod.lora_down.weight.data.copy_(g("down"))
    mod.lora_up.weight.data.copy_(g("up"))
    mod.apply_to()
    mod.multiplier = meta["multiplier"]
    x = g("x").clone().requires_grad_(True)
    y = lin(x)
    y.backward(g("gy"))
    torch.testing.assert_close(y, g("y"), rtol=0, atol=0)
    torch.testing.assert_close(x.grad, g("gx"), rtol=0, atol=0)
    torch.testing.assert_close(mod.lora_down.weight.grad, g("gdown"), rtol=0, atol=0)
    torch.testing.assert_close(mod.lora_up.weight.grad, g("gup"), rtol=0, atol=0)


def test_lora_down_init_is_kaiming_a1(goldens):
    t, _ = goldens
    torch.manual_seed(100)
    lin = torch.nn.Linear(24, 40)
    mod = R.LoRAModuleRef("lin", lin, 1.0, 4, 1.0)
    torch.testing.assert_close(mod.lora_down.weight.detach(), t["lora_linear/0/down_init"], rtol=0, atol=0)
    assert float(mod.lora_up.weight.detach().abs().max()) == 0.0


def test_lora_conv_forward_backward(goldens):
    t, man = goldens
    g = lambda k: t[f"lora_conv/{k}"]
    conv = torch.nn.Conv2d(8, 12, 3, 1, 1)
    conv.weight.data.copy_(g("w"))
    conv.bias.data.copy_(g("b"))
    mod = R.LoRAModuleRef("conv", conv, 1.0, 4, man["lora_conv"]["alpha"])
    mod.lora_down.weight.data.copy_(g("down"))
    mod.lora_up.weight.data.copy_(g("up"))
    mod.apply_to()
    x = g("x").clone().requires_grad_(True)
    y = conv(x)
    y.backward(g("gy"))
    torch.testing.assert_close(y, g("y"), rtol=0, atol=0)
    torch.testing.assert_close(x.grad, g("gx"), rtol=0, atol=0)
    torch.testing.assert_close(mod.lora_up.weight.grad, g("gup"), rtol=0, atol=0)


@pytest.mark.parametrize("action", ["enhance", "erase"])
def test_loss(goldens, action):
    t, man = goldens
    tgt = t["loss/target"].clone().requires_grad_(True)
    loss = R.slider_loss(action, man["loss"]["guidance_scale"], tgt, t["loss/positive"], t["loss/neutral"],
                         t["loss/negative"])
    loss.backward()
    torch.testing.assert_close(loss.reshape(1), t[f"loss/{action}/value"], rtol=0, atol=0)
    torch.testing.assert_close(tgt.grad, t[f"loss/{action}/gtarget"], rtol=0, atol=0)


def _synth_embeds(cfg, keys, seed):
    g = torch.Generator().manual_seed(seed)
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in keys}
    pooled = None
    if cfg.addition_embed_type == "text_time":
        pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        pooled = {k: torch.randn(1, pdim, generator=g) for k in keys}
    return emb, pooled


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_predict_noise_matches_reference(goldens, model):
    t, man = goldens
    meta = man[f"predict/{model}"]
    cfg = CFGS[model]()
    xl = cfg.addition_embed_type == "text_time"
    unet = U.init_synthetic_(U.UNet2DConditionModel(cfg), seed=0).requires_grad_(False).eval()
    torch.manual_seed(1)
    net = R.LoRANetworkRef(unet, 4, 1.0, 1.0, "noxattn" if not xl else "full")
    g = torch.Generator().manual_seed(2)
    for l in net.unet_loras:
        l.lora_up.weight.data.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.05)
    sched = S.create_noise_scheduler_ref(meta["scheduler"])
    sched.set_timesteps(1000)
    emb, pooled = _synth_embeds(cfg, ["unconditional", "target"], 4)
    B = meta["B"]
    latents = t[f"predict/{model}/latents"]
    ts = sched.timesteps[500]
    assert float(ts) == meta["timestep"]
    te = R.concat_embeddings(emb["unconditional"], emb["target"], B)
    for lora_on in [False, True]:
        net.__enter__() if lora_on else net.__exit__(None, None, None)
        for gs in [1.0, 3.0]:
            with torch.no_grad():
                if not xl:
                    out = R.predict_noise(unet, sched, ts, latents, te, guidance_scale=gs)
                else:
                    tid = R.get_add_time_ids(128, 128)
                    out = R.predict_noise_xl(unet, sched, ts, latents, te,
                                             R.concat_embeddings(pooled["unconditional"], pooled["target"], B),
                                             R.concat_embeddings(tid, tid, B), guidance_scale=gs)
            torch.testing.assert_close(out, t[f"predict/{model}/lora{int(lora_on)}_g{gs}"], rtol=0, atol=0)


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_training_trajectory_matches_reference(goldens, model):
    """The oracle's own step driver + clip + AdamW reproduce the reference-driven 6-step run."""
    t, man = goldens
    meta = man[f"traj/{model}"]
    cfg = CFGS[model]()
    xl = cfg.addition_embed_type == "text_time"
    unet = U.init_synthetic_(U.UNet2DConditionModel(cfg), seed=0).requires_grad_(False).eval()
    torch.manual_seed(1)
    net = R.LoRANetworkRef(unet, 4, 1.0, 1.0, "noxattn")
    keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if xl else [])
    emb, pooled = _synth_embeds(cfg, keys, 4)
    sched = S.create_noise_scheduler_ref("euler_a" if xl else "ddim")
    params = [p for l in net.unet_loras for p in l.parameters()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    lr0, wd = (1e-4, 1e-6) if xl else (2e-3, 1e-2)
    max_steps = meta["max_denoising_steps"]
    losses = []  # the loop continues the RNG stream left by the LoRA init (seed 1), like a real run
    for it in range(meta["iterations"]):
        import math
        lr = (1e-6 + (lr0 - 1e-6) * (1 + math.cos(math.pi * it / 50)) / 2) if xl else lr0
        sched.set_timesteps(max_steps)
        torch.randint(0, 1, (1,))
        timesteps_to = torch.randint(1, max_steps, (1,)).item()
        latents = R.get_initial_latents(sched, 1, 128, 128, 1)
        xlkw = {"pooled": pooled, "time_ids": R.get_add_time_ids(128, 128)} if xl else None
        te = R.concat_embeddings(emb["unconditional"], emb["target"], 1)
        with net:
            if not xl:
                denoised = R.diffusion(unet, sched, latents, te, total_timesteps=timesteps_to, guidance_scale=3)
            else:
                denoised = R.diffusion_xl(unet, sched, latents, te,
                                          R.concat_embeddings(pooled["unconditional"], pooled["target"], 1),
                                          R.concat_embeddings(xlkw["time_ids"], xlkw["time_ids"], 1),
                                          guidance_scale=1.0, total_timesteps=timesteps_to)
        sched.set_timesteps(1000)
        cur_t = sched.timesteps[int(timesteps_to * 1000 / max_steps)]
        for p in params:
            p.grad = None
        loss, preds = R.slider_step(unet, net, sched, denoised, cur_t, emb, 1, "enhance", 4.0, 1.0, xlkw)
        if it == 0:
            assert float(cur_t) == float(t[f"traj/{model}/step0_timestep"])
            torch.testing.assert_close(denoised, t[f"traj/{model}/step0_denoised"], rtol=0, atol=0)
            torch.testing.assert_close(preds["target"].detach(), t[f"traj/{model}/step0_target"], rtol=0, atol=0)
        losses.append(loss.item())
        loss.backward()
        if xl:
            R.clip_grad_norm_(params, 0.2)
        with torch.no_grad():
            for p, mm, vv in zip(params, m, v):
                R.adamw_step_(p, p.grad, mm, vv, it + 1, lr, weight_decay=wd)
    assert losses == pytest.approx(meta["losses"], rel=2e-4)
    sd = net.state_dict()
    for k, nrm in meta["norms"].items():
        assert float(sd[k].float().norm()) == pytest.approx(nrm, rel=2e-3, abs=1e-6), k
    for k in [k for k in t if k.startswith(f"traj/{model}/sd/")]:
        torch.testing.assert_close(sd[k.split("/sd/")[1]], t[k], rtol=2e-3, atol=2e-5)


# ---- c3lier (conv / time-embedding adaptors; T/lora.py:19-26,100-114,243-249; T/train_lora.py:44-46) -----------------
@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("method", ["noxattn", "full"])
def test_c3lier_selection_matches_reference(goldens, model, method):
    _, man = goldens
    with torch.device("meta"):
        unet = U.UNet2DConditionModel(CFGS[model]())
        net = R.LoRANetworkRef(unet, 4, 1.0, 1.0, method, target_replace=R.C3LIER_TARGET_REPLACE)
    names = [l.lora_name for l in net.unet_loras]
    g = man["selection_c3lier"][f"{model}/{method}"]
    assert len(names) == g["n_modules"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["names_sha"]
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert hashlib.sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest() == g["shapes_sha"]
    assert sum(p.numel() for p in net.parameters()) == g["n_params"]


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_c3lier_init_consumes_the_rng_like_the_reference(goldens, model):
    """Modules after a duplicate visit only get the reference's init if the dropped duplicates are built too."""
    t, man = goldens
    unet = U.init_synthetic_(U.UNet2DConditionModel(CFGS[model]()), seed=0)
    torch.manual_seed(man[f"c3lier_init/{model}"]["seed"])
    net = R.LoRANetworkRef(unet, 4, 1.0, 1.0, "noxattn", target_replace=R.C3LIER_TARGET_REPLACE)
    assert int(torch.randint(0, 2 ** 31 - 1, (1,))) == man[f"c3lier_init/{model}"]["after_init_randint"]
    by_name = {l.lora_name: l for l in net.unet_loras}
    keys = [k for k in t if k.startswith(f"c3lier_init/{model}/")]
    assert len(keys) >= 5
    for k in keys:
        name = k.split("/")[2].rsplit(".lora_down.weight", 1)[0]
        torch.testing.assert_close(by_name[name].lora_down.weight.detach(), t[k], rtol=0, atol=0)


# ---- DoRA (T/dora.py:53-162) ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 1])
def test_dora_linear_forward_backward(goldens, seed):
    t, man = goldens
    g = lambda k: t[f"dora_linear/{seed}/{k}"]
    meta = man[f"dora_linear/{seed}"]
    lin = torch.nn.Linear(24, 40)
    lin.weight.data.copy_(g("w"))
    lin.bias.data.copy_(g("b"))
    mod = R.DoRAModuleRef("lin", lin, 1.0, 4, meta["alpha"])
    assert mod.scale == meta["scale"]
    torch.testing.assert_close(mod.dora_scale.detach(), g("dora_scale_init"), rtol=1e-6, atol=1e-7)
    assert list(mod.state_dict().keys()) == meta["state_dict_keys"]
    mod.lora_down.weight.data.copy_(g("down"))
    mod.lora_up.weight.data.copy_(g("up"))
    mod.dora_scale.data.copy_(g("dora_scale"))
    mod.apply_to()
    mod.multiplier = meta["multiplier"]
    x = g("x").clone().requires_grad_(True)
    y = lin(x)
    y.backward(g("gy"))
    torch.testing.assert_close(y.detach(), g("y"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(x.grad, g("gx"), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mod.lora_down.weight.grad, g("gdown"), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(mod.lora_up.weight.grad, g("gup"), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(mod.dora_scale.grad, g("gscale"), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
@pytest.mark.parametrize("method", ["noxattn", "full"])
def test_dora_selection_matches_reference(goldens, model, method):
    _, man = goldens
    unet = U.UNet2DConditionModel(CFGS[model]())
    net = R.DoRANetworkRef(unet, 4, 1.0, 1.0, method)
    g = man["selection_dora"][f"{model}/{method}"]
    names = [l.lora_name for l in net.unet_loras]
    assert len(names) == g["n_modules"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["names_sha"]
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert hashlib.sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest() == g["shapes_sha"]
    assert list(net.state_dict().keys())[:4] == g["keys_head"]
    assert sum(p.numel() for p in net.parameters()) == g["n_params"]
