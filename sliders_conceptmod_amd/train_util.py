"""Step helpers with the reference's signatures (conceptmod/textsliders/train_util.py:27-47, 91-105, 267-327,
449-489, 677-708, 976-1097).  They are model-agnostic host code: `unet` is the engine-backed
sliders_conceptmod_amd.unet.UNet2DConditionModel (or anything with the same call surface)."""
from typing import Optional

import torch

from . import _native

UNET_IN_CHANNELS = 4
VAE_SCALE_FACTOR = 8
UNET_ATTENTION_TIME_EMBED_DIM = 256
TEXT_ENCODER_2_PROJECTION_DIM = 1280
UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM = 2816


def get_random_noise(batch_size: int, height: int, width: int, generator: torch.Generator = None) -> torch.Tensor:
    return torch.randn((batch_size, UNET_IN_CHANNELS, height // VAE_SCALE_FACTOR, width // VAE_SCALE_FACTOR),
                       generator=generator, device="cpu")


def get_initial_latents(scheduler, n_imgs: int, height: int, width: int, n_prompts: int, generator=None):
    noise = get_random_noise(n_imgs, height, width, generator=generator).repeat(n_prompts, 1, 1, 1)
    return noise * scheduler.init_noise_sigma.to(noise.device)


def concat_embeddings(unconditional, conditional, n_imgs: int):
    return torch.cat([unconditional, conditional]).repeat_interleave(n_imgs, dim=0)


def _cfg(noise_pred: torch.Tensor, guidance_scale: float) -> torch.Tensor:
    if noise_pred.is_cuda and noise_pred.dtype == torch.float32 and not noise_pred.requires_grad:
        n_half = noise_pred.numel() // 2
        out = torch.empty((noise_pred.shape[0] // 2,) + tuple(noise_pred.shape[1:]), dtype=torch.float32,
                          device=noise_pred.device)
        _native.check(_native.lib().smi_cfg_combine(_native.ptr(noise_pred.contiguous()), _native.ptr(out), n_half,
                                                    float(guidance_scale), _native.stream_ptr()), "smi_cfg_combine")
        return out
    u, t = noise_pred.chunk(2)  # differentiable path (the adapted pass under autograd)
    return u + guidance_scale * (t - u)


def predict_noise(unet, scheduler, timestep, latents, text_embeddings, guidance_scale=7.5):
    latent_model_input = torch.cat([latents] * 2)
    latent_model_input = scheduler.scale_model_input(latent_model_input, timestep)
    noise_pred = unet(latent_model_input, timestep, encoder_hidden_states=text_embeddings).sample
    return _cfg(noise_pred, guidance_scale)


@torch.no_grad()
def diffusion(unet, scheduler, latents, text_embeddings, total_timesteps: int = 1000, start_timesteps=0, **kwargs):
    for timestep in scheduler.timesteps[start_timesteps:total_timesteps]:
        noise_pred = predict_noise(unet, scheduler, timestep, latents, text_embeddings, **kwargs)
        latents = scheduler.step(noise_pred, timestep, latents).prev_sample
    return latents


def predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings, add_time_ids,
                     guidance_scale=7.5, guidance_rescale=0.7):
    latent_model_input = torch.cat([latents] * 2)
    latent_model_input = scheduler.scale_model_input(latent_model_input, timestep)
    added_cond_kwargs = {"text_embeds": add_text_embeddings, "time_ids": add_time_ids}
    noise_pred = unet(latent_model_input, timestep, encoder_hidden_states=text_embeddings,
                      added_cond_kwargs=added_cond_kwargs).sample
    # the reference computes rescale_noise_cfg here and discards it (train_util.py:485-489); the plain mix is returned
    return _cfg(noise_pred, guidance_scale)


@torch.no_grad()
def diffusion_xl(unet, scheduler, latents, text_embeddings, add_text_embeddings, add_time_ids,
                 guidance_scale: float = 1.0, total_timesteps: int = 1000, start_timesteps=0):
    for timestep in scheduler.timesteps[start_timesteps:total_timesteps]:
        noise_pred = predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings,
                                      add_time_ids, guidance_scale=guidance_scale, guidance_rescale=0.7)
        latents = scheduler.step(noise_pred, timestep, latents).prev_sample
    return latents


@torch.no_grad()
def slider_sweep_latents(unet, network, scheduler, latents, text_embeddings, scale: float, start_noise: int,
                         guidance_scale: float = 7.5, num_inference_steps: int = 50, added_cond=None):
    """The inference-side slider loop of the eval scripts (eval-scripts/generate_images_sd1.py:170-190,
    generate_images_xl.py:327-343): denoise from `latents` (already x init_noise_sigma) with classifier-free guidance, the
    adaptor OFF (set_lora_slider(0)) while t > start_noise and at `scale` afterwards, every UNet call inside `with network`.
    `text_embeddings` = cat([uncond, cond]) as there; `added_cond` = (add_text_embeddings, add_time_ids) for SD-XL.
    Returns the final latents (decoding them is the VAE decoder's job, outside this package)."""
    scheduler.set_timesteps(num_inference_steps)
    for t in scheduler.timesteps:
        network.set_lora_slider(scale=0 if t > start_noise else scale)
        with network:
            if added_cond is None:
                noise_pred = predict_noise(unet, scheduler, t, latents, text_embeddings, guidance_scale=guidance_scale)
            else:
                noise_pred = predict_noise_xl(unet, scheduler, t, latents, text_embeddings, added_cond[0], added_cond[1],
                                              guidance_scale=guidance_scale)
        latents = scheduler.step(noise_pred, t, latents).prev_sample
    network.set_lora_slider(scale=1)
    return latents


@torch.no_grad()
def get_noisy_image(img, vae, generator, unet, scheduler, total_timesteps: int = 1000, start_timesteps=0, **kwargs):
    """Image-slider front end (trainscripts/imagesliders/train_util.py:200-235), same signature and order of operations:
    preprocess -> vae.encode(image).latent_dist.sample(None) -> x vae.config.scaling_factor -> noise = randn(shape,
    generator) -> scheduler.add_noise(latents, noise, scheduler.timesteps[total_timesteps : total_timesteps + 1]).
    Returns (noised latents, noise).  `unet`, `start_timesteps` are unused there too."""
    from .vae import VaeImageProcessor
    vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
    image = VaeImageProcessor(vae_scale_factor=vae_scale_factor).preprocess(img).to(vae.device)
    init_latents = vae.encode(image).latent_dist.sample(None)
    init_latents = vae.config.scaling_factor * init_latents
    shape = init_latents.shape
    # randn_tensor(shape, generator=generator, device=device): a CPU generator draws on the host and the result is moved
    gdev = "cpu" if generator is None or generator.device.type == "cpu" else init_latents.device
    noise = torch.randn(shape, generator=generator, device=gdev, dtype=init_latents.dtype).to(init_latents.device)
    timestep = scheduler.timesteps[total_timesteps:total_timesteps + 1]
    init_latents = scheduler.add_noise(init_latents, noise, timestep)
    return init_latents, noise


def get_noisy_image_pair(imgs, vae, make_generator, unet, scheduler, total_timesteps: int = 1000, start_timesteps=0,
                         **kwargs):
    """`[get_noisy_image(img, vae, make_generator(), ...) for img in imgs]` for images of ONE size with a single batched
    VAE encode (the two 1024 x 1024 images of an image-slider step: 2 x 9.1 ms -> one pass of twice the rows).  The
    encoder consumes no random numbers, so calling `make_generator()` -- the reference re-seeds with
    `torch.manual_seed(seed)` per image, I/train_lora-scale-xl.py:220-247 -- right before each image's posterior sample
    keeps the reference's draw ORDER.  The moments are the per-image ones up to fp32 summation order only: a sample's
    kernels are the same whatever its batch, except that the split-K rule (csrc/gemm.hip) picks its slice count from the
    launch's tile count, i.e. from the batch -- the pair-vs-single test therefore holds to a tolerance (2e-3), not bitwise."""
    from .vae import VaeImageProcessor
    vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
    proc = VaeImageProcessor(vae_scale_factor=vae_scale_factor)
    image = torch.cat([proc.preprocess(img) for img in imgs]).to(vae.device)
    dist = vae.encode(image).latent_dist
    timestep = scheduler.timesteps[total_timesteps:total_timesteps + 1]
    out = []
    for i in range(len(imgs)):
        generator = make_generator()
        init_latents = vae.config.scaling_factor * dist.rows(i, i + 1).sample(None)
        gdev = "cpu" if generator is None or generator.device.type == "cpu" else init_latents.device
        noise = torch.randn(init_latents.shape, generator=generator, device=gdev,
                            dtype=init_latents.dtype).to(init_latents.device)
        out.append((scheduler.add_noise(init_latents, noise, timestep), noise))
    return out


def get_add_time_ids(height: int, width: int, dynamic_crops: bool = False, dtype: torch.dtype = torch.float32):
    if dynamic_crops:
        random_scale = torch.rand(1).item() * 2 + 1
        original_size = (int(height * random_scale), int(width * random_scale))
        crops_coords_top_left = (torch.randint(0, original_size[0] - height, (1,)).item(),
                                 torch.randint(0, original_size[1] - width, (1,)).item())
        target_size = (height, width)
    else:
        original_size = (height, width)
        crops_coords_top_left = (0, 0)
        target_size = (height, width)
    add_time_ids = list(original_size + crops_coords_top_left + target_size)
    passed_add_embed_dim = UNET_ATTENTION_TIME_EMBED_DIM * len(add_time_ids) + TEXT_ENCODER_2_PROJECTION_DIM
    if passed_add_embed_dim != UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM:
        raise ValueError(f"Model expects an added time embedding vector of length "
                         f"{UNET_PROJECTION_CLASS_EMBEDDING_INPUT_DIM}, but a vector of {passed_add_embed_dim} was "
                         f"created.")
    return torch.tensor([add_time_ids], dtype=dtype)


def get_optimizer(name: str):
    name = name.lower()
    if name == "adam":
        return torch.optim.Adam
    if name == "adamw":
        return torch.optim.AdamW
    if name.startswith("dadapt") or name.endswith("8bit") or name in ("lion", "prodigy"):
        # the reference imports dadaptation / bitsandbytes / lion_pytorch / prodigyopt here; none is in this image
        raise ValueError(f"optimizer '{name}' needs a package that is not installed; use adam or adamw")
    raise ValueError("Optimizer must be adam, adamw, lion or Prodigy")


def get_lr_scheduler(name: Optional[str], optimizer, max_iterations: Optional[int], lr_min: Optional[float], **kwargs):
    if name == "cosine":
        return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=max_iterations, eta_min=lr_min, **kwargs)
    elif name == "cosine_with_restarts":
        return torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(optimizer, T_0=max_iterations // 10, T_mult=2,
                                                                    eta_min=lr_min, **kwargs)
    elif name == "step":
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=max_iterations // 100, gamma=0.999, **kwargs)
    elif name == "constant":
        return torch.optim.lr_scheduler.ConstantLR(optimizer, factor=1, **kwargs)
    elif name == "linear":
        # the reference passes LinearLR(factor=0.5, ...), which is not a torch keyword and raises TypeError
        # (tests/golden: error/lr_scheduler_linear); the evident intent is start_factor
        return torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=0.5, total_iters=max_iterations // 100,
                                                 **kwargs)
    raise ValueError("Scheduler must be cosine, cosine_with_restarts, step, linear or constant")


def get_random_resolution_in_bucket(bucket_resolution: int = 512):
    max_resolution = bucket_resolution
    min_resolution = bucket_resolution // 2
    step = 64
    min_step = min_resolution // step
    max_step = max_resolution // step
    height = torch.randint(min_step, max_step, (1,)).item() * step
    width = torch.randint(min_step, max_step, (1,)).item() * step
    return height, width
