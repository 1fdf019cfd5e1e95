"""AutoencoderKL encoder on the HIP engine (row a-10: `get_noisy_image`, trainscripts/imagesliders/train_util.py:200-235)
against the CPU oracle (oracle/vae_ref.py -- parity-unpinned restatement of the un-vendored diffusers encoder) on the same
seeded weights and images.  fp16 engine vs fp32 oracle: tolerances are storage-rounding level, as for the UNet."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sched_ref as S
from oracle import vae_ref as OV


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def pair(ocfg, dtype):
    import sliders_conceptmod_amd.vae as PV
    ov = OV.init_synthetic_(OV.AutoencoderKLRef(ocfg), seed=7).eval()
    pv = PV.AutoencoderKL(PV.VAEConfig(**dataclasses.asdict(ocfg)))
    pv.load_state_dict(ov.state_dict())
    return ov, pv.to("cuda", dtype).requires_grad_(False).eval()


# bars = 1.5 x measured (tiny VAE: fp16 1.53e-3, bf16 1.22e-2)
@pytest.mark.parametrize("dtype,bar", [(torch.float16, 2.3e-3), (torch.bfloat16, 1.85e-2)])
@pytest.mark.parametrize("size", [(64, 64), (96, 64)])
def test_tiny_vae_moments_match_oracle(dtype, bar, size):
    ov, pv = pair(OV.tiny_vae_config(), dtype)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, *size, generator=g) * 2 - 1
    with torch.no_grad():
        ref = ov.moments(x)
    dist = pv.encode(x.cuda()).latent_dist
    got = torch.cat([dist.mean, dist.logvar], dim=1)
    assert got.shape == ref.shape == (2, 8, size[0] // 8, size[1] // 8)
    e = rel(got, torch.cat([ref[:, :4], ref[:, 4:].clamp(-30, 20)], dim=1))
    print(f"tiny VAE {dtype} {size}: moments rel err {e:.2e}")
    assert e < bar, e
    # batch composition must not matter (per-sample GroupNorm / attention): sample 1 alone == sample 1 in the batch
    alone = pv.encode(x[1:].cuda()).latent_dist.mean
    assert torch.equal(alone, dist.mean[1:])


def test_real_sd_vae_encoder_256px_matches_oracle():
    """The real SD / SD-XL VAE encoder architecture (34.2 M parameters, 512-wide single-head attention over 32 x 32
    pixels) at 256 x 256: posterior mean and logvar against the oracle."""
    torch.set_num_threads(16)
    ov, pv = pair(OV.sdxl_vae_config(), torch.float16)
    assert sum(p.numel() for p in ov.encoder.parameters()) == 34163592  # the public SD VAE encoder
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        ref = ov.moments(x)
    dist = pv.encode(x.cuda()).latent_dist
    e_mean, e_lv = rel(dist.mean, ref[:, :4]), rel(dist.logvar, ref[:, 4:].clamp(-30, 20))
    print(f"SD VAE encoder 256px fp16: mean rel err {e_mean:.2e}, logvar rel err {e_lv:.2e}")
    assert e_mean < 2.4e-3 and e_lv < 2.4e-3, (e_mean, e_lv)  # measured 1.58e-3 / 1.47e-3 (1.5 x)


def test_get_noisy_image_follows_the_reference_order():
    """I/train_util.py:200-235 on the engine vs the oracle's restatement: same preprocessing, posterior, scaling, noise
    drawn from the CALLER's generator, add_noise at scheduler.timesteps[total_timesteps].  The posterior sample itself uses
    the device RNG (generator None), so the latents are compared through the posterior they must come from."""
    import sliders_conceptmod_amd.model_util as MU
    from sliders_conceptmod_amd import train_util as TU
    ov, pv = pair(OV.tiny_vae_config(), torch.float16)
    img = (torch.rand(64, 64, 3, generator=torch.Generator().manual_seed(9)) * 255).to(torch.uint8)
    osch, psch = S.create_noise_scheduler_ref("ddim"), MU.create_noise_scheduler("ddim")
    osch.set_timesteps(50), psch.set_timesteps(50)
    noised, noise = TU.get_noisy_image(img.numpy(), pv, torch.Generator().manual_seed(11), None, psch,
                                       total_timesteps=20)
    o_noised, o_noise, o_lat, o_dist = OV.get_noisy_image(img, ov, torch.Generator().manual_seed(11), osch, 20)
    assert noised.shape == (1, 4, 8, 8) and noised.is_cuda
    torch.testing.assert_close(noise.cpu(), o_noise, rtol=0, atol=0)  # the caller's generator, host draw
    t = int(psch.timesteps[20])
    a = float(psch.alphas_cumprod[t])
    lat = (noised.cpu() - (1 - a) ** 0.5 * noise.cpu()) / a ** 0.5 / pv.config.scaling_factor  # = the posterior sample
    z = (lat - o_dist.mean) / o_dist.std
    assert float(z.abs().max()) < 6.0 and 0.5 < float(z.std()) < 1.5, (float(z.abs().max()), float(z.std()))
    assert pv.config.scaling_factor == ov.cfg.scaling_factor


@pytest.mark.parametrize("cfg,size", [("tiny", 64), ("sdxl", 256)])
def test_get_noisy_image_pair_equals_one_call_per_image(cfg, size):
    """train_util.get_noisy_image_pair (one batched VAE encode for the two images of an image-slider step) against the
    reference's one call per image with `torch.manual_seed(seed)` in front of each: same posterior samples (the device RNG
    is re-seeded per image either way), same noise, same noised latents.  A sample's arithmetic does not depend on the batch
    it is encoded in, so the results are compared at rounding level, and bit for bit where no launch changes its K split."""
    import sliders_conceptmod_amd.model_util as MU
    from sliders_conceptmod_amd import train_util as TU
    _, pv = pair(OV.tiny_vae_config() if cfg == "tiny" else OV.sdxl_vae_config(), torch.float16)
    g = torch.Generator().manual_seed(3)
    imgs = [(torch.rand(size, size, 3, generator=g) * 255).to(torch.uint8).numpy() for _ in range(2)]
    sch = MU.create_noise_scheduler("ddim")
    sch.set_timesteps(50)
    one = [TU.get_noisy_image(im, pv, torch.manual_seed(21), None, sch, total_timesteps=17) for im in imgs]
    two = TU.get_noisy_image_pair(imgs, pv, lambda: torch.manual_seed(21), None, sch, total_timesteps=17)
    for (a, na), (b, nb) in zip(one, two):
        torch.testing.assert_close(na, nb, rtol=0, atol=0)
        e = rel(b, a)
        print(f"{cfg} {size}px pair vs single encode: rel diff {e:.2e}, bit-equal {bool((a == b).all())}")
        assert e < 2e-3, e
    assert rel(one[0][0], one[1][0]) > 1e-2  # the two images do differ


@pytest.mark.parametrize("xl", [True, False])
def test_image_slider_trainer_encodes_image_folders(tmp_path, xl):
    """train_lora_scale_xl.train() on folders of PNG pairs (the reference's data layout): resize, VAE-encode on the engine,
    two-sided step, save.  Scales given in DESCENDING order with matching folders (pairing by original index)."""
    from PIL import Image
    import numpy as np
    import sliders_conceptmod_amd.config_util as CU
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.prompt_util as PRU
    from sliders_conceptmod_amd.train_lora_scale_xl import train
    rng = np.random.default_rng(0)
    for folder, bias in (("big", 40), ("small", -40)):
        (tmp_path / "data" / folder).mkdir(parents=True)
        for name in ("a.png", "b.png"):
            arr = np.clip(rng.integers(0, 255, (80, 72, 3)) + bias, 0, 255).astype(np.uint8)
            Image.fromarray(arr).save(tmp_path / "data" / folder / name)
    cfg = CU.RootConfig(
        prompts_file="unused",
        pretrained_model=CU.PretrainedModelConfig(name_or_path="synthetic://tiny_sdxl" if xl else "synthetic://tiny_sd1x"),
        network=CU.NetworkConfig(type="c3lier", rank=4, alpha=1.0, training_method="noxattn"),
        train=CU.TrainConfig(precision="float16", noise_scheduler="ddim", iterations=3, lr=1e-4, optimizer="AdamW",
                             lr_scheduler="constant", max_denoising_steps=10),
        save=CU.SaveConfig(name="img", path=str(tmp_path / "out"), per_steps=1000), logging=CU.LoggingConfig(),
        other=CU.OtherConfig())
    prompts = [PRU.PromptSettings(target="t", positive="big", neutral="small", unconditional="")]
    torch.manual_seed(0)
    net = train(cfg, prompts, torch.device("cuda:0"), str(tmp_path / "data"), ["big", "small"], [1, -1],
                image_size=(64, 64), xl=xl)
    out = tmp_path / "out" / "img_last.pt"
    assert out.exists()
    sd = torch.load(out, weights_only=True)
    assert any("conv1" in k for k in sd) and any("attn1" in k for k in sd)  # c3lier on the image trainer too
    assert float(net.flat_up.detach().abs().max()) > 0
    with pytest.raises(Exception, match="number of folders"):
        train(cfg, prompts, torch.device("cuda:0"), str(tmp_path / "data"), ["big"], [1, -1], image_size=(64, 64), xl=xl)
    with pytest.raises(ValueError, match="no counterpart"):
        train(cfg, prompts, torch.device("cuda:0"), str(tmp_path / "data"), ["big", "small"], [1, -2],
              image_size=(64, 64), xl=xl)
