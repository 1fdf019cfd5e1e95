"""Image-slider trainer (reference: trainscripts/imagesliders/train_lora-scale-xl.py:42-548, SD-1.x twin
train_lora-scale.py) -- the two-sided step on the HIP engine.

Per step (I/train_lora-scale-xl.py:178-401): pick |scale| s, a paired (low, high) latent, one seed for both noises;
noisy latents = scheduler.add_noise(latent, noise, t) (I/train_util.py:200-235); then
    set_lora_slider(+s): eps = predict_noise_xl(high_noised, positive prompt) -> MSE(eps, high_noise) -> backward()
    set_lora_slider(-s): eps = predict_noise_xl(low_noised,  neutral  prompt) -> MSE(eps, low_noise)  -> backward()
gradients accumulate, one optimizer.step().  (The reference also runs two frozen passes whose results the loss never
uses, :263-308; they are skipped here -- they cannot change the result.)

Data: `--folder_main` holds one sub-folder per scale (`--folders`, e.g. "bigsize,smallsize"), each with the same file
names.  Image files (.png / .jpg / .jpeg / .webp) are resized and VAE-encoded every step exactly as the reference does
(`train_util.get_noisy_image`, I/train_lora-scale-xl.py:216-247) on the HIP VAE encoder; `<name>.pt` / `.safetensors`
tensors [4, h, w] are taken as PRE-ENCODED latents (already multiplied by the VAE scaling factor)."""
import argparse
import ast
import os
import random
from pathlib import Path

import torch
from tqdm import tqdm

from . import config_util, model_util, parallel, prompt_util, train_util
from .lora import DEFAULT_TARGET_REPLACE, UNET_TARGET_REPLACE_MODULE_CONV, LoRANetwork
from .train_lora_xl import encode_xl


def image_slider_step(unet, network, scheduler, lat_low, lat_high, noise_low, noise_high, timestep, pos, neu,
                      add_time_ids, scale: float, guidance_scale: float = 1.0, uncond=None):
    """The two adapted passes + two backward()s of one image-slider step; returns (loss_high, loss_low).
    The reference pairs every prompt with the UNCONDITIONAL one in the CFG batch at guidance_scale = 1
    (I/train_lora-scale-xl.py:321-337, I/train_lora-scale.py:283-318): pass `uncond` (a PromptEmbedsXL for SD-XL, a
    [1, 77, D] tensor for SD-1.x) for exactly that.  With `uncond=None` an SD-XL prompt is paired with itself, which at
    guidance 1 is the same prediction (u + 1 (t - u) == t) -- kept for callers that hold no unconditional embedding."""
    bs = lat_high.shape[0]
    xl = hasattr(pos, "pooled_embeds")

    def cond(e):
        u = e if uncond is None else uncond
        return dict(text_embeddings=train_util.concat_embeddings(u.text_embeds, e.text_embeds, bs),
                    add_text_embeddings=train_util.concat_embeddings(u.pooled_embeds, e.pooled_embeds, bs),
                    add_time_ids=train_util.concat_embeddings(add_time_ids, add_time_ids, bs))

    losses = []
    for sgn, lat, noise, emb in ((+1.0, lat_high, noise_high, pos), (-1.0, lat_low, noise_low, neu)):
        network.set_lora_slider(scale=sgn * scale)
        with network:
            if xl:
                pred = train_util.predict_noise_xl(unet, scheduler, timestep, lat, **cond(emb),
                                                   guidance_scale=guidance_scale)
            else:
                pred = train_util.predict_noise(unet, scheduler, timestep, lat,
                                                train_util.concat_embeddings(uncond, emb, bs),
                                                guidance_scale=guidance_scale)
        loss = torch.nn.functional.mse_loss(pred.float(), noise.float())  # fp32 at the loss, I/..-xl.py:282,338
        loss.backward()  # gradients of the two sides accumulate (I/train_lora-scale-xl.py:345,377)
        losses.append(loss.detach())
    network.set_lora_slider(scale=1)
    return losses[0], losses[1]


def _load_latent(path):
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        return next(iter(load_file(str(path)).values()))
    return torch.load(path, weights_only=True)


_IMAGE_EXT = (".png", ".jpg", ".jpeg", ".webp")  # I/train_lora-scale-xl.py:217


def train(config, prompts, device, folder_main, folders, scales, models=None, rank=4, vae=None, image_size=None,
          xl=True, fused_step=None, on_step_complete=None):
    """`vae` (an AutoencoderKL, default: loaded next to the UNet) is only needed when the folders hold image files;
    `image_size` is what the pairs are resized to (the reference hard-codes (512, 512) for SD-XL,
    I/train_lora-scale-xl.py:220, and (256, 256) for SD-1.x, I/train_lora-scale.py:219).  `xl=False` is the SD-1.x twin
    (train_lora_scale.py).  `fused_step` runs the two-sided step through step.ImageSliderStep (no autograd graph, native
    AdamW) -- same arithmetic, tested against the autograd loop: None (default) = whenever the configured optimiser is
    Adam / AdamW without amsgrad, True = fused or ValueError, False = the reference-style autograd loop."""
    weight_dtype = config_util.parse_precision(config.train.precision)
    tokenizers, text_encoders, unet, noise_scheduler = models or model_util.load_models(
        config.pretrained_model.name_or_path, scheduler_name=config.train.noise_scheduler, xl=xl)
    if isinstance(text_encoders, (list, tuple)):
        for te in text_encoders:
            te.to(device, dtype=weight_dtype)
            te.eval()
    rank_, world = parallel.world_info()
    parallel.sync_control_rng(device=device)  # one control RNG (torch + random) for all ranks, before the adaptor init
    unet.to(device, dtype=weight_dtype)
    unet.requires_grad_(False)
    unet.eval()
    modules = list(DEFAULT_TARGET_REPLACE)  # I/train_lora-scale-xl.py:57-59: c3lier adds the conv classes
    if config.network.type == "c3lier":
        modules += UNET_TARGET_REPLACE_MODULE_CONV
    network = LoRANetwork(unet, rank=rank, multiplier=1.0, alpha=config.network.alpha,
                          train_method=config.network.training_method, target_replace=modules).to(device,
                                                                                                   dtype=weight_dtype)
    parallel.broadcast_(network.flat.data)
    optimizer_kwargs = {}  # I/train_lora-scale-xl.py:110-117: "k=v k=v" -> keyword arguments of the optimiser
    if config.train.optimizer_args is not None and len(config.train.optimizer_args) > 0:
        for arg in config.train.optimizer_args.split(" "):
            key, value = arg.split("=")
            optimizer_kwargs[key] = ast.literal_eval(value)
    optimizer = train_util.get_optimizer(config.train.optimizer)(network.prepare_optimizer_params(),
                                                                 lr=config.train.lr, **optimizer_kwargs)
    lr_scheduler = train_util.get_lr_scheduler(config.train.lr_scheduler, optimizer,
                                               max_iterations=config.train.iterations, lr_min=config.train.lr / 100)
    # every prompt entry is encoded (target / positive / neutral / unconditional, I/train_lora-scale-xl.py:136-166) and one
    # is drawn per step from the torch RNG BEFORE timesteps_to (:183-191): multi-entry prompt files and the draw order
    # are the reference's
    cache, pairs = {}, []
    with torch.no_grad():
        if xl:
            enc = lambda p_: encode_xl(text_encoders, tokenizers, p_, device, weight_dtype)  # noqa: E731
        else:
            from .train_lora import encode
            enc = lambda p_: encode(text_encoders, tokenizers, p_, device, weight_dtype)  # noqa: E731
        for settings in prompts:
            for p_ in (settings.target, settings.positive, settings.neutral, settings.unconditional):
                if p_ not in cache:
                    cache[p_] = enc(p_)
            pairs.append((settings, cache[settings.positive], cache[settings.neutral], cache[settings.unconditional]))
    # folder <-> scale pairing by ORIGINAL index, as the reference does (folders[scales == s][0],
    # I/train_lora-scale-xl.py:213-214); a length mismatch raises as there (:452-453)
    if len(scales) != len(folders):
        raise Exception("the number of folders need to match the number of scales")
    scales = [float(s) for s in scales]
    folder_of = {}
    for s, f in zip(scales, folders):
        folder_of.setdefault(s, f)
    for s in scales:
        if -s not in folder_of:
            raise ValueError(f"scale {s:g} has no counterpart {-s:g} in --scales: the two-sided step needs both")
    names = sorted(os.listdir(os.path.join(folder_main, folder_of[scales[0]])))
    images = [n for n in names if n.lower().endswith(_IMAGE_EXT)]
    if images:
        names = images
        if vae is None:
            vae = model_util.load_vae(config.pretrained_model.name_or_path, xl=xl)
        vae.to(device, dtype=weight_dtype)
        vae.requires_grad_(False)
        vae.eval()
    size = image_size or ((512, 512) if xl else (256, 256))
    save_path = Path(config.save.path)
    save_dtype = config_util.parse_precision(config.train.precision)
    stepper = None
    oname = config.train.optimizer.lower()
    wd = optimizer_kwargs.get("weight_decay", 1e-2 if oname == "adamw" else 0.0)
    unsupported = set(optimizer_kwargs) - {"weight_decay", "betas", "eps", "amsgrad"}
    fusable = (oname in ("adam", "adamw") and not (oname == "adam" and wd != 0.0) and not optimizer_kwargs.get("amsgrad")
               and not unsupported)
    if fused_step and not fusable:
        raise ValueError("--fused_step implements Adam / AdamW (decoupled weight decay; weight_decay, betas, eps) only")
    if fused_step or (fused_step is None and fusable):
        from .step import ImageSliderStep
        stepper = ImageSliderStep(unet, network, noise_scheduler, lr=config.train.lr, weight_decay=wd,
                                  eps=optimizer_kwargs.get("eps", 1e-8), betas=optimizer_kwargs.get("betas", (0.9, 0.999)))
    cond_cache = {}
    network.training_losses = []
    for i in tqdm(range(config.train.iterations), disable=rank_ != 0):
        noise_scheduler.set_timesteps(config.train.max_denoising_steps, device=device)
        optimizer.zero_grad()
        settings, pos, neu, unc = pairs[torch.randint(0, len(pairs), (1,)).item()]
        # I/train_lora-scale-xl.py:191-193 draws from [1, max); the SD-1.x script from [1, max - 1) (I/train_lora-scale.py:186)
        timesteps_to = torch.randint(1, config.train.max_denoising_steps - (0 if xl else 1), (1,)).item()
        height, width = settings.resolution, settings.resolution
        if settings.dynamic_resolution:
            height, width = train_util.get_random_resolution_in_bucket(settings.resolution)
        scale_to_look = abs(random.choice(scales))
        f_low, f_high = folder_of[-scale_to_look], folder_of[scale_to_look]
        name = names[(random.randint(0, len(names) - 1) + rank_) % len(names)]  # ranks take different pairs
        seed = random.randint(0, 2 * 15)
        if images:  # the reference's path: resize, VAE-encode and noise both images with the same seed (:220-247)
            from PIL import Image
            imgs = [Image.open(os.path.join(folder_main, f, name)).resize(size) for f in (f_low, f_high)]
            # one batched VAE encode for both; the per-image `torch.manual_seed(seed)` of the reference stays per image
            (nl, noise_low), (nh, noise_high) = train_util.get_noisy_image_pair(
                imgs, vae, lambda: torch.manual_seed(seed), unet, noise_scheduler, start_timesteps=0,
                total_timesteps=timesteps_to)
        else:
            lat_low = _load_latent(os.path.join(folder_main, f_low, name)).unsqueeze(0).float()
            lat_high = _load_latent(os.path.join(folder_main, f_high, name)).unsqueeze(0).float()
            t = noise_scheduler.timesteps[timesteps_to]
            g = torch.Generator().manual_seed(seed)
            noise_low = torch.randn(lat_low.shape, generator=g)
            g = torch.Generator().manual_seed(seed)
            noise_high = torch.randn(lat_high.shape, generator=g)
            nl = noise_scheduler.add_noise(lat_low, noise_low, t).to(device)
            nh = noise_scheduler.add_noise(lat_high, noise_high, t).to(device)
        if settings.batch_size != nh.shape[0]:  # the reference repeats the embeddings batch_size times over ONE image pair
            raise ValueError(f"prompt batch_size {settings.batch_size} != {nh.shape[0]} image pair(s) per step")
        noise_scheduler.set_timesteps(1000)
        # the added time ids follow the PROMPT's resolution (and dynamic_crops), not the image size (:249-254)
        tid = train_util.get_add_time_ids(height, width, dynamic_crops=settings.dynamic_crops,
                                          dtype=torch.float32).to(device) if xl else None
        cur_t = noise_scheduler.timesteps[int(timesteps_to * 1000 / config.train.max_denoising_steps)]
        if stepper is not None:
            ckey = (id(settings), height, width)
            cc = None if settings.dynamic_crops else cond_cache.get(ckey)
            if cc is None:
                if xl:
                    cc = tuple(stepper.make_conditioning(e.text_embeds, 1, e.pooled_embeds, tid, uncond=unc.text_embeds,
                                                         uncond_pooled=unc.pooled_embeds) for e in (pos, neu))
                else:
                    cc = tuple(stepper.make_conditioning(e, 1, uncond=unc) for e in (pos, neu))
                cond_cache[ckey] = cc
            with torch.no_grad():
                losses = stepper.train_step(nl, nh, noise_low.to(device), noise_high.to(device), cur_t, cc[0], cc[1],
                                            scale_to_look, lr=lr_scheduler.get_last_lr()[0])
            lh, ll = (float(v) for v in losses.tolist())
        else:
            lh, ll = image_slider_step(unet, network, noise_scheduler, nl, nh, noise_low.to(device),
                                       noise_high.to(device), cur_t, pos, neu, tid, scale_to_look, uncond=unc)
            lh, ll = float(lh), float(ll)
            parallel.allreduce_mean_(network.flat.grad)
        optimizer.step()  # fused: no gradients, a no-op that keeps torch's "optimizer before scheduler" order
        lr_scheduler.step()
        network.training_losses.append((lh, ll))
        if on_step_complete is not None:
            on_step_complete(i, lh, ll)
        if rank_ == 0 and i % config.save.per_steps == 0 and i != 0 and i != config.train.iterations - 1:
            save_path.mkdir(parents=True, exist_ok=True)  # I/train_lora-scale-xl.py:402-412
            network.save_weights(save_path / f"{config.save.name}_{i}steps.pt", dtype=save_dtype)
    if rank_ == 0:
        save_path.mkdir(parents=True, exist_ok=True)
        network.save_weights(save_path / f"{config.save.name}_last.pt", dtype=save_dtype)
    return network


def launch_device(args):
    """One process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment): RCCL process
    group, device = LOCAL_RANK -- exactly as train_lora_xl.main does.  `SMI_DIST_BACKEND=gloo` is for rehearsals on a
    box with fewer GPUs than ranks (tests)."""
    if str(args.device) == "cpu":
        raise ValueError("--device cpu: the product path has no CPU fallback (the CPU oracle lives under oracle/)")
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        backend = os.environ.get("SMI_DIST_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else int(args.device)
        torch.cuda.set_device(local)
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group(backend)
        return torch.device("cuda", local)
    return torch.device(f"cuda:{args.device}")


def main(args, xl: bool = True, models=None, vae=None):
    config = config_util.load_config_from_yaml(args.config_file)
    if args.name is not None:
        config.save.name = args.name
    attributes = [a.strip() for a in args.attributes.split(",")] if args.attributes is not None else []
    config.network.alpha = args.alpha
    config.network.rank = args.rank
    config.save.name += f"_alpha{args.alpha}_rank{config.network.rank}_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    device = launch_device(args)
    folders = [f.strip() for f in args.folders.split(",")]
    scales = [float(s.strip()) for s in args.scales.split(",")]
    return train(config, prompts, device, args.folder_main, folders, scales, rank=args.rank, xl=xl, models=models,
                 vae=vae, fused_step=getattr(args, "fused_step", None))


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=False, default="data/config-xl.yaml")
    p.add_argument("--alpha", type=float, required=True)
    p.add_argument("--rank", type=int, default=4)
    p.add_argument("--device", default=0)
    p.add_argument("--name", type=str, default=None)
    p.add_argument("--attributes", type=str, default=None)
    p.add_argument("--folder_main", type=str, required=True, help="directory holding one sub-folder per scale")
    p.add_argument("--stylecheck", type=str, default=None)
    p.add_argument("--folders", type=str, default="verylow, low, high, veryhigh")
    p.add_argument("--scales", type=str, default="-2, -1, 1, 2")
    from .train_lora import add_fused_step_flags
    add_fused_step_flags(p)  # (--no_dedup_uncond is accepted and has no effect here: the image step has no frozen passes)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
