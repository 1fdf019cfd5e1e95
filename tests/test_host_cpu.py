"""Host-side logic of the product package (no GPU): adaptor API, state-dict layout, C-ABI exports, workspace plan."""
import ctypes
import dataclasses
import hashlib
import json
import os
import re

import pytest
import torch

import sliders_conceptmod_amd.lora as L
import sliders_conceptmod_amd.unet as PU
from sliders_conceptmod_amd import _native, build as smi_build
from oracle import unet_ref as OU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFGS = {"sd1x": OU.sd1x_config, "sdxl": OU.sdxl_config, "tiny_sd1x": OU.tiny_sd1x_config,
        "tiny_sdxl": OU.tiny_sdxl_config}


def pcfg(ocfg):
    return PU.UNetConfig(**dataclasses.asdict(ocfg))


@pytest.fixture(scope="module")
def libpath():
    return smi_build.build()


@pytest.mark.parametrize("model", list(CFGS))
def test_product_skeleton_matches_oracle_state_dict(model):
    with torch.device("meta"):
        a = PU.UNet2DConditionModel(pcfg(CFGS[model]()))
        b = OU.UNet2DConditionModel(CFGS[model]())
    sa = {k: tuple(v.shape) for k, v in a.state_dict().items()}
    sb = {k: tuple(v.shape) for k, v in b.state_dict().items()}
    assert list(sa.items()) == list(sb.items())


def test_public_configs_match():
    assert dataclasses.asdict(PU.sd1x_config()) == dataclasses.asdict(OU.sd1x_config())
    assert dataclasses.asdict(PU.sdxl_config()) == dataclasses.asdict(OU.sdxl_config())


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("method", ["noxattn", "innoxattn", "selfattn", "xattn", "full", "xattn-strict"])
def test_lora_network_names_and_shapes_match_reference(goldens, model, method):
    _, man = goldens
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(pcfg(CFGS[model]()))
    net = L.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method=method)
    g = man["selection"][f"{model}/{method}"]
    names = [l.lora_name for l in net.unet_loras]
    assert len(names) == g["n_modules"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["names_sha"]
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert hashlib.sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest() == g["shapes_sha"]
    assert net._n_down + net._n_up == g["n_params"]
    assert sum(p.numel() for p in net.parameters()) == g["n_params"]


def test_bad_train_method_raises_like_reference(goldens):
    _, man = goldens
    assert man["bad_method_error"] == "NotImplementedError"
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(pcfg(OU.tiny_sd1x_config()))
    with pytest.raises(NotImplementedError):
        L.LoRANetwork(unet, train_method="bogus")


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_lora_init_is_seed_compatible_with_reference(goldens, model):
    t, _ = goldens
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(pcfg(CFGS[model]()))
    torch.manual_seed(1)
    net = L.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    sd = net.state_dict()
    keys = [k for k in t if k.startswith(f"traj/{model}/init/")]
    assert len(keys) == 2
    for k in keys:
        torch.testing.assert_close(sd[k.split("/init/")[1]], t[k], rtol=0, atol=0)
    assert float(net.flat_up.detach().abs().max()) == 0.0 and net.flat.is_leaf


def test_context_manager_and_slider_semantics():
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(pcfg(OU.tiny_sd1x_config()))
    net = L.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
    assert net.engine_params()[2] == 1.0  # constructor multiplier (lora.py:126)
    with net:
        assert all(l.multiplier == 1.0 for l in net.unet_loras)
    assert all(l.multiplier == 0 for l in net.unet_loras)  # lora.py:299-301
    net.set_lora_slider(-2.5)
    with net:
        assert net.engine_params()[2] == -2.5
    assert net.prepare_optimizer_params()[0]["params"][0] is net.flat
    assert net.unet_loras[0].scale == 0.25


def test_save_and_strict_reload(tmp_path):
    unet = PU.UNet2DConditionModel(pcfg(OU.tiny_sd1x_config()))
    net = L.LoRANetwork(unet, rank=4, alpha=1.0, train_method="noxattn")
    with torch.no_grad():
        net.flat_up.normal_(0, 0.1)
    assert float(net.flat.detach()[net._n_down:].abs().max()) > 0  # flat_up is a live view of the single leaf
    for ext in (".pt", ".safetensors"):
        f = tmp_path / f"w{ext}"
        net.save_weights(f, dtype=torch.bfloat16)
        if ext == ".pt":
            sd = torch.load(f, weights_only=True)
        else:
            from safetensors.torch import load_file
            sd = load_file(str(f))
        assert all(v.dtype == torch.bfloat16 for k, v in sd.items())
        net2 = L.LoRANetwork(unet, rank=4, alpha=1.0, train_method="noxattn")
        net2.load_state_dict(sd)
        torch.testing.assert_close(net2.flat_up.detach(), net.flat_up.detach().bfloat16().float())
        bad = dict(sd)
        bad.pop(next(iter(bad)))
        with pytest.raises(RuntimeError):
            net2.load_state_dict(bad)
    net.to(dtype=torch.bfloat16)  # reference does .to(device, dtype=weight_dtype); master weights stay fp32
    assert net.flat.dtype == torch.float32


def test_unet_without_gpu_fails_loudly():
    unet = PU.UNet2DConditionModel(pcfg(OU.tiny_sd1x_config())).half()
    with pytest.raises(_native.SmiError):
        unet(torch.zeros(2, 4, 8, 8), 10, torch.zeros(2, 77, 64))


def test_library_exports_every_declared_symbol(libpath):
    hdr = open(os.path.join(ROOT, "include", "smi.h")).read()
    declared = set(re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", hdr)) - {"smi_engine"}
    lib = ctypes.CDLL(libpath)
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"{sym} declared in include/smi.h but not exported"
    assert declared == set(_native.EXPORTED_SYMBOLS)


@pytest.mark.parametrize("model,batch,hw", [("tiny_sd1x", 2, 16), ("tiny_sdxl", 2, 16), ("sd1x", 8, 64),
                                             ("sdxl", 4, 128)])
def test_workspace_plan_dry_run(libpath, model, batch, hw):
    """smi_workspace_bytes runs the engine's forward+backward graph in dry mode (no GPU needed)."""
    cfg = pcfg(CFGS[model]())
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(cfg)
    net = L.LoRANetwork(unet, rank=4, alpha=1.0, train_method="noxattn")
    cc = _native.make_config(cfg, torch.float16)
    with_lora = _native.workspace_bytes(cc, net.engine_sites(), batch, hw, hw, 77)
    without = _native.workspace_bytes(cc, [], batch, hw, hw, 77)
    assert with_lora > without > 0
    nparam = sum(p.numel() for p in unet.parameters())
    assert without > nparam * 2  # at least the transposed copies
    if model == "sdxl":
        assert with_lora < 200e9, f"SD-XL 1024^2 batch 4 plan = {with_lora/1e9:.1f} GB must fit in 288 GB HBM"
    print(f"{model}: workspace {with_lora/1e9:.2f} GB (no-LoRA {without/1e9:.2f} GB)")


def test_engine_rejects_unsupported_lora_target(libpath):
    cfg = pcfg(OU.tiny_sd1x_config())
    cc = _native.make_config(cfg, torch.float16)
    with pytest.raises(_native.SmiError, match="not a layer this engine adapts"):
        _native.workspace_bytes(cc, [{"target": "time_embedding.linear_1", "off_down": 0, "off_up": 0,
                                      "rank": 4, "scale": 0.25}], 2, 16, 16, 77)


# ---- c3lier (reference: T/lora.py:19-26,100-114,243-249; T/train_lora.py:44-46; T/data/config.yaml:7) -------------
C3 = L.UNET_TARGET_REPLACE_MODULE_TRANSFORMER + L.UNET_TARGET_REPLACE_MODULE_CONV


@pytest.mark.parametrize("model", list(CFGS))
@pytest.mark.parametrize("method", ["noxattn", "full"])
def test_c3lier_names_and_shapes_match_reference(goldens, model, method):
    _, man = goldens
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(pcfg(CFGS[model]()))
    net = L.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method=method, target_replace=C3)
    g = man["selection_c3lier"][f"{model}/{method}"]
    names = [l.lora_name for l in net.unet_loras]
    assert len(names) == g["n_modules"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["names_sha"]
    shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
    assert hashlib.sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest() == g["shapes_sha"]
    assert sum(p.numel() for p in net.parameters()) == g["n_params"]


@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
def test_c3lier_init_is_seed_compatible_with_reference(goldens, model):
    t, man = goldens
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(pcfg(CFGS[model]()))
    torch.manual_seed(man[f"c3lier_init/{model}"]["seed"])
    net = L.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn", target_replace=C3)
    assert int(torch.randint(0, 2 ** 31 - 1, (1,))) == man[f"c3lier_init/{model}"]["after_init_randint"]
    sd = net.state_dict()
    keys = [k for k in t if k.startswith(f"c3lier_init/{model}/")]
    assert len(keys) >= 5
    for k in keys:
        torch.testing.assert_close(sd[k.split("/")[2]], t[k], rtol=0, atol=0)


def test_conv_lora_rank_is_clamped_and_scale_uses_it():
    conv = torch.nn.Conv2d(2, 8, 3, 1, 1)
    r, dshape, ushape = L.lora_shapes(conv, 4)
    assert (r, dshape, ushape) == (2, (2, 2, 3, 3), (8, 2, 1, 1))  # min(rank, in, out): T/lora.py:104


def test_shipped_default_configs_equal_the_references(goldens):
    """sliders_conceptmod_amd/data/*.yaml (the CLI defaults `data/config.yaml`, `data/config-xl.yaml`) parse to the same
    values as the reference's shipped files (recorded by make_goldens.py through the REFERENCE's config_util /
    prompt_util), paths aside; found from any working directory."""
    import sliders_conceptmod_amd.config_util as CU
    import sliders_conceptmod_amd.prompt_util as PRU
    _, man = goldens
    for name in ("config.yaml", "config-xl.yaml"):
        got = json.loads(CU.load_config_from_yaml(f"data/{name}").model_dump_json())
        want = man["config_defaults"]["shipped"][name]
        for d in (got, want):
            d.pop("prompts_file")
            d["pretrained_model"].pop("name_or_path")
        assert got == want, name
    got = [json.loads(p.model_dump_json()) for p in PRU.load_prompts_from_yaml("data/prompts.yaml")]
    assert got == man["prompts_shipped"]
    assert CU.load_config_from_yaml("data/config.yaml").network.type == "c3lier"  # T/data/config.yaml:7
    xl = PRU.load_prompts_from_yaml("data/prompts-xl.yaml")
    assert xl[0].negative == "person, very young" and xl[0].resolution == 1024


def test_c3lier_workspace_plan_covers_every_site(libpath):
    """Dry run of the engine with the c3lier site list of the real SD-1.x architecture: every target resolves."""
    cfg = pcfg(OU.sd1x_config())
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(cfg)
    net = L.LoRANetwork(unet, rank=4, alpha=1.0, train_method="noxattn", target_replace=C3)
    cc = _native.make_config(cfg, torch.float16)
    assert len(net.unet_loras) == 150 and net.flat.numel() == 2906880
    assert _native.workspace_bytes(cc, net.engine_sites(), 8, 64, 64, 77) > 0


# ---- VAE front end (image sliders, I/train_util.py:200-235) ------------------------------------------------------------
def test_vae_container_matches_oracle_keys_and_fails_loudly_off_gpu():
    from oracle import vae_ref as OV
    import sliders_conceptmod_amd.vae as PV
    ov = OV.AutoencoderKLRef(OV.sd_vae_config())
    pv = PV.AutoencoderKL(PV.sd_vae_config())
    assert list(ov.state_dict().keys()) == list(pv.state_dict().keys())
    assert {k: tuple(v.shape) for k, v in ov.state_dict().items()} == {k: tuple(v.shape) for k, v in pv.state_dict().items()}
    assert sum(p.numel() for p in ov.encoder.parameters()) == 34163592 and pv.config.scaling_factor == 0.18215
    full = dict(ov.state_dict())
    full["decoder.conv_in.weight"] = torch.zeros(1)  # a full diffusers VAE state dict: decoder entries are ignored
    full["post_quant_conv.weight"] = torch.zeros(1)
    pv.load_state_dict(full)
    with pytest.raises(_native.SmiError):
        pv.half().encode(torch.zeros(1, 3, 64, 64))


def test_vae_loads_deprecated_attention_names():
    """SD-1.x era VAE checkpoints name the mid-block attention query / key / value / proj_attn (some as
    1 x 1 conv kernels); diffusers renames them on load, so must this container -- strictly, values intact."""
    import sliders_conceptmod_amd.vae as PV
    src = PV.init_synthetic_(PV.AutoencoderKL(PV.sd_vae_config()), seed=3)
    new_sd = {k: v.clone() for k, v in src.state_dict().items()}
    old_sd = {}
    ren = ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn."))
    n_renamed = 0
    for k, v in new_sd.items():
        for a, b in ren:
            if a in k:
                k = k.replace(a, b)
                n_renamed += 1
                if k.endswith("weight"):
                    v = v.reshape(v.shape[0], v.shape[1], 1, 1)  # the conv-kernel form of the oldest checkpoints
                break
        old_sd[k] = v
    assert n_renamed == 8
    dst = PV.AutoencoderKL(PV.sd_vae_config())
    dst.load_state_dict(old_sd)  # strict
    for k, v in dst.state_dict().items():
        assert torch.equal(v, new_sd[k]), k


def test_engine_cache_key_tells_dora_from_lora():
    """a DoRANetwork and a LoRANetwork with the same rank / alpha / targets have identical down / up offsets;
    the engine key must still differ (the packed engine holds DoRA sites and delta-weight buffers or it does not)."""
    import dataclasses
    import sliders_conceptmod_amd.dora as D
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.unet as PU
    from oracle import unet_ref as OU
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(OU.tiny_sdxl_config())))
    keys = []
    for cls in (L.LoRANetwork, D.DoRANetwork):
        net = cls(pu, rank=4, multiplier=1.0, alpha=1.0, train_method="noxattn")
        sites = net.engine_sites()
        keys.append(tuple((s["target"], s["off_down"], s["off_up"], s["rank"], s["scale"], s.get("off_dora", -1))
                          for s in sites))
    assert [k[:5] for k in keys[0]] == [k[:5] for k in keys[1]]  # the collision the advisor described ...
    assert keys[0] != keys[1]                                     # ... which the sixth field resolves
    import inspect
    assert "off_dora" in inspect.getsource(PU.UNet2DConditionModel._ensure_engine)


def test_vae_image_processor_preprocess():
    import numpy as np
    from PIL import Image
    from oracle import vae_ref as OV
    from sliders_conceptmod_amd.vae import VaeImageProcessor
    arr = np.random.default_rng(0).integers(0, 255, (64, 72, 3)).astype(np.uint8)
    p = VaeImageProcessor(8)
    t = p.preprocess(Image.fromarray(arr))
    assert t.shape == (1, 3, 64, 72) and t.dtype == torch.float32
    torch.testing.assert_close(t, OV.preprocess(torch.from_numpy(arr)), rtol=0, atol=0)
    torch.testing.assert_close(p.preprocess(arr), t, rtol=0, atol=0)
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    assert p.preprocess(Image.fromarray(arr[:61, :70])).shape == (1, 3, 56, 64)  # floored to multiples of 8


def test_vae_workspace_plan_dry_run(libpath):
    import ctypes as C
    c = _native.VaeConfigC()
    c.dtype, c.in_channels, c.latent_channels, c.n_levels = 0, 3, 4, 4
    for i, v in enumerate((128, 256, 512, 512)):
        c.block_out_channels[i] = v
    c.layers_per_block, c.norm_num_groups = 2, 32
    out = C.c_size_t(0)
    _native.check(_native.lib().smi_vae_workspace_bytes(C.byref(c), 2, 1024, 1024, C.byref(out)), "plan")
    assert 1e9 < out.value < 40e9  # 12.9 GB: two 1024^2 images, fp32 16384^2 scores of the mid-block attention included
    assert _native.lib().smi_vae_workspace_bytes(C.byref(c), 1, 100, 64, C.byref(out)) != 0  # not a multiple of 8


# ---- DoRA (T/dora.py) -----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model", ["tiny_sd1x", "tiny_sdxl"])
@pytest.mark.parametrize("method", ["noxattn", "full"])
def test_dora_network_names_shapes_and_key_order_match_reference(goldens, model, method):
    import sliders_conceptmod_amd.dora as D
    _, man = goldens
    unet = PU.UNet2DConditionModel(pcfg(CFGS[model]()))
    torch.manual_seed(0)
    for p in unet.parameters():
        p.data.normal_(0, 0.05)
    net = D.DoRANetwork(unet, rank=4, multiplier=1.0, delimiter="_", target_replace=["Attention"], prefix="lora_unet",
                        train_method=method)
    g = man["selection_dora"][f"{model}/{method}"]
    names = [l.lora_name for l in net.unet_loras]
    assert len(names) == g["n_modules"]
    assert hashlib.sha256("\n".join(names).encode()).hexdigest() == g["names_sha"]
    sd = net.state_dict()
    shapes = {k: list(v.shape) for k, v in sd.items()}
    assert hashlib.sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest() == g["shapes_sha"]
    assert list(sd.keys())[:4] == g["keys_head"]
    assert sum(p.numel() for p in net.parameters()) == g["n_params"]
    l0 = net.unet_loras[0]  # dora_scale starts at the frozen weight's column norms (T/dora.py:95-105); scale = 1 / rank
    w = dict(unet.named_modules())[l0.target_path].weight
    torch.testing.assert_close(l0.dora_scale.detach(), w.detach().float().norm(dim=0, keepdim=True))
    assert l0.scale == 0.25 and float(net.flat_up.detach().abs().max()) == 0.0
    with net:
        assert net.engine_params()[2] == 1.0
    assert all(l.multiplier == 0 for l in net.unet_loras)
    net2 = D.DoRANetwork(unet, rank=4, target_replace=["Attention"], train_method=method)
    net2.load_state_dict(sd)
    torch.testing.assert_close(net2.flat.detach(), net.flat.detach())


def test_posterior_rows_sample_like_separate_posteriors():
    """vae.DiagonalGaussianDistribution.rows (train_util.get_noisy_image_pair): the posterior of samples [lo, hi) of a batched
    encode equals the posterior built from those samples' moments alone, and sampling the rows one after the other from a
    generator seeded per row reproduces the one-call-per-image draws of the reference loop (I/train_lora-scale-xl.py:220-247)."""
    import torch
    from sliders_conceptmod_amd.vae import DiagonalGaussianDistribution as D
    g = torch.Generator().manual_seed(0)
    moments = torch.randn(3, 8, 4, 4, generator=g)
    moments[:, 4:] = moments[:, 4:] * 20  # exercises the clamp to [-30, 20]
    whole = D(moments)
    for i in range(3):
        part, alone = whole.rows(i, i + 1), D(moments[i:i + 1])
        assert torch.equal(part.mean, alone.mean) and torch.equal(part.logvar, alone.logvar) and torch.equal(part.std, alone.std)
        a = part.sample(torch.Generator().manual_seed(7))
        b = alone.sample(torch.Generator().manual_seed(7))
        assert torch.equal(a, b) and a.shape == (1, 4, 4, 4)
    assert torch.equal(whole.rows(0, 3).mode(), whole.mode())
