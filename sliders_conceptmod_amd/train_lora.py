"""Text-slider trainer for SD-1.x -- same CLI, YAML schemas, step order, RNG draw order and output naming as the
reference script (conceptmod/textsliders/train_lora.py:32-419), on the HIP engine.

    python -m sliders_conceptmod_amd.train_lora --config_file cfg.yaml --alpha 1 --rank 4 --device 0 --name ageslider \
           --attributes "male, female"

Differences, all recorded in DESIGN.md: the reference's stale 6-argument PromptEmbedsPair / `unconditional_latents`
call works (prompt_util.py here accepts both forms); `--device cpu` is rejected (no CPU path); under torch.distributed.run the batch is
sharded over ranks and the LoRA gradient is all-reduced."""
import argparse
import ast
import sys
from pathlib import Path

import torch
from tqdm import tqdm

from . import config_util, model_util, parallel, prompt_util, train_util
from .config_util import RootConfig
from .lora import DEFAULT_TARGET_REPLACE, UNET_TARGET_REPLACE_MODULE_CONV, LoRANetwork
from .prompt_util import PromptEmbedsCache, PromptEmbedsPair, PromptSettings


def encode(text_encoder, tokenizer, prompt, device, dtype):
    if isinstance(text_encoder, model_util.SyntheticTextEncoder):
        return text_encoder.encode(prompt).to(device, dtype)
    tokens = tokenizer[0](prompt, padding="max_length", max_length=tokenizer[0].model_max_length, truncation=True,
                          return_tensors="pt").input_ids
    return text_encoder[0](tokens.to(text_encoder[0].device))[0].to(device, dtype)


def train(config: RootConfig, prompts: list, device, models=None, on_step_complete=None, save_file=True,
          fused_step=None, dedup_uncond: bool = True):
    """`fused_step` (not in the reference signature): None (default) = the fused path whenever the configured optimiser is
    Adam / AdamW, else the reference-style autograd loop; True = fused or ValueError; False = the reference-style loop.
    `dedup_uncond`: inside the fused step, run each distinct frozen sample once (bit-identical results, step.py)."""
    metadata = {"prompts": ",".join([p.model_dump_json() for p in prompts]), "config": config.model_dump_json()}
    save_path = Path(config.save.path)
    # train_lora.py:44-46: `modules = DEFAULT_TARGET_REPLACE; modules += UNET_TARGET_REPLACE_MODULE_CONV` mutates the list
    # that LoRANetwork's default argument is bound to, which is how c3lier reaches the network there; here it is passed
    modules = list(DEFAULT_TARGET_REPLACE)
    if config.network.type == "c3lier":
        modules += UNET_TARGET_REPLACE_MODULE_CONV
    weight_dtype = config_util.parse_precision(config.train.precision)
    save_weight_dtype = config_util.parse_precision(config.train.precision)  # sic: train.precision (train_lora.py:55)
    if weight_dtype == torch.float32:
        raise ValueError("the HIP engine computes in fp16/bf16 storage with fp32 accumulation; set train.precision")
    tokenizer, text_encoder, unet, noise_scheduler = models or model_util.load_models(
        config.pretrained_model.name_or_path, scheduler_name=config.train.noise_scheduler,
        v2=config.pretrained_model.v2, v_pred=config.pretrained_model.v_pred)
    # data parallelism: one control RNG for all ranks, seeded BEFORE the adaptor init draws from it (parallel.py)
    rank, world = parallel.world_info()
    parallel.sync_control_rng(device=device)
    noise_scheduler.dp_shard = (rank, world)
    if isinstance(text_encoder, (list, tuple)):  # train_lora.py:64-65: text_encoder.to(device, dtype=weight_dtype).eval()
        for te in text_encoder:
            te.to(device, dtype=weight_dtype)
            te.eval()
    unet.to(device, dtype=weight_dtype)
    unet.enable_xformers_memory_efficient_attention()
    unet.requires_grad_(False)
    unet.eval()
    network = LoRANetwork(unet, rank=config.network.rank, multiplier=1.0, alpha=config.network.alpha,
                          train_method=config.network.training_method, target_replace=modules).to(device,
                                                                                                   dtype=weight_dtype)
    parallel.broadcast_(network.flat.data)
    optimizer_module = train_util.get_optimizer(config.train.optimizer)
    optimizer_kwargs = {}
    if config.train.optimizer_args is not None and len(config.train.optimizer_args) > 0:
        for arg in config.train.optimizer_args.split(" "):
            key, value = arg.split("=")
            optimizer_kwargs[key] = ast.literal_eval(value)
    optimizer = optimizer_module(network.prepare_optimizer_params(), lr=config.train.lr, **optimizer_kwargs)
    lr_scheduler = train_util.get_lr_scheduler(config.train.lr_scheduler, optimizer,
                                               max_iterations=config.train.iterations, lr_min=config.train.lr / 100)
    criteria = torch.nn.MSELoss()

    cache = PromptEmbedsCache()
    prompt_pairs = []
    with torch.no_grad():
        for settings in prompts:
            for prompt in [settings.target, settings.positive, settings.neutral, settings.unconditional]:
                if cache[prompt] is None:
                    cache[prompt] = encode(text_encoder, tokenizer, prompt, device, weight_dtype)
            prompt_pairs.append(PromptEmbedsPair(criteria, cache[settings.target], cache[settings.positive],
                                                 cache[settings.unconditional], cache[settings.neutral],
                                                 settings=settings))
    del tokenizer, text_encoder

    # The fused step (default): the pre-roll and the 4-pass step run through SliderStep (one batched UNet pass, native loss /
    # AdamW, no autograd graph) -- the path bench.py measures.  Same arithmetic and RNG draw order as the loop below
    # (tested); `--no_fused_step` keeps the reference-style loop.
    stepper = None
    name = config.train.optimizer.lower()
    wd = optimizer_kwargs.get("weight_decay", 1e-2 if name == "adamw" else 0.0)
    fusable = name in ("adam", "adamw") and not (name == "adam" and wd != 0.0) and not optimizer_kwargs.get("amsgrad")
    if fused_step and not fusable:
        raise ValueError("--fused_step implements Adam / AdamW (decoupled weight decay) only")
    if fused_step or (fused_step is None and fusable):
        from .step import SliderStep
        stepper = SliderStep(unet, network, noise_scheduler, lr=config.train.lr, weight_decay=wd,
                             eps=optimizer_kwargs.get("eps", 1e-8), betas=optimizer_kwargs.get("betas", (0.9, 0.999)),
                             max_grad_norm=0.0, cfg_scale=1.0, dedup_uncond=dedup_uncond)
    cond_cache = {}

    pbar = tqdm(range(config.train.iterations), disable=rank != 0)
    loss = None
    for i in pbar:
        if stepper is not None:
            with torch.no_grad():
                noise_scheduler.set_timesteps(config.train.max_denoising_steps, device=device)
                prompt_pair = prompt_pairs[torch.randint(0, len(prompt_pairs), (1,)).item()]
                timesteps_to = torch.randint(1, config.train.max_denoising_steps, (1,)).item()
                height, width = prompt_pair.resolution, prompt_pair.resolution
                if prompt_pair.dynamic_resolution:
                    height, width = train_util.get_random_resolution_in_bucket(prompt_pair.resolution)
                bs = prompt_pair.batch_size
                latents = train_util.get_initial_latents(noise_scheduler, bs, height, width, 1)
                if world > 1:
                    latents = latents[parallel.shard_slice(bs, rank, world)]
                    bs = bs // world
                latents = latents.to(device, dtype=torch.float32)
                c = cond_cache.get((id(prompt_pair), bs))
                if c is None:
                    emb = {k: getattr(prompt_pair, k) for k in ("target", "positive", "neutral", "unconditional")}
                    c = cond_cache[(id(prompt_pair), bs)] = stepper.make_conditioning(emb, bs)
                denoised_latents = stepper.preroll(latents, c, timesteps_to, 3)
                noise_scheduler.set_timesteps(1000)
                current_timestep = noise_scheduler.timesteps[
                    int(timesteps_to * 1000 / config.train.max_denoising_steps)]
                loss = stepper.train_step(denoised_latents, current_timestep, c, prompt_pair.action,
                                          prompt_pair.guidance_scale, lr=lr_scheduler.get_last_lr()[0])
            lv = float(loss.item())  # the one host sync per step, as the reference's loss.item() (train_lora.py:292)
            pbar.set_description(f"Loss*1k: {lv * 1000:.4f}")
            optimizer.step()  # no gradients: a no-op that keeps torch's "optimizer before scheduler" order
            lr_scheduler.step()
            if on_step_complete is not None:
                on_step_complete(i, lv)
            if save_file and rank == 0 and i % config.save.per_steps == 0 and i != 0 and \
                    i != config.train.iterations - 1:
                save_path.mkdir(parents=True, exist_ok=True)
                network.save_weights(save_path / f"{config.save.name}_{i}steps.pt", dtype=save_weight_dtype)
            continue
        with torch.no_grad():
            noise_scheduler.set_timesteps(config.train.max_denoising_steps, device=device)
            optimizer.zero_grad()
            prompt_pair = prompt_pairs[torch.randint(0, len(prompt_pairs), (1,)).item()]
            timesteps_to = torch.randint(1, config.train.max_denoising_steps, (1,)).item()
            height, width = prompt_pair.resolution, prompt_pair.resolution
            if prompt_pair.dynamic_resolution:
                height, width = train_util.get_random_resolution_in_bucket(prompt_pair.resolution)
            # all ranks draw the GLOBAL latent batch from the same control RNG and take their slice
            bs = prompt_pair.batch_size
            latents = train_util.get_initial_latents(noise_scheduler, bs, height, width, 1)
            if world > 1:
                latents = latents[parallel.shard_slice(bs, rank, world)]
                bs = bs // world
            latents = latents.to(device, dtype=torch.float32)
            with network:
                denoised_latents = train_util.diffusion(
                    unet, noise_scheduler, latents,
                    train_util.concat_embeddings(prompt_pair.unconditional, prompt_pair.target, bs),
                    start_timesteps=0, total_timesteps=timesteps_to, guidance_scale=3)
            noise_scheduler.set_timesteps(1000)
            current_timestep = noise_scheduler.timesteps[int(timesteps_to * 1000 / config.train.max_denoising_steps)]
            positive_latents = train_util.predict_noise(
                unet, noise_scheduler, current_timestep, denoised_latents,
                train_util.concat_embeddings(prompt_pair.unconditional, prompt_pair.positive, bs), guidance_scale=1)
            neutral_latents = train_util.predict_noise(
                unet, noise_scheduler, current_timestep, denoised_latents,
                train_util.concat_embeddings(prompt_pair.unconditional, prompt_pair.neutral, bs), guidance_scale=1)
            unconditional_latents = train_util.predict_noise(
                unet, noise_scheduler, current_timestep, denoised_latents,
                train_util.concat_embeddings(prompt_pair.unconditional, prompt_pair.unconditional, bs),
                guidance_scale=1)
        with network:
            target_latents = train_util.predict_noise(
                unet, noise_scheduler, current_timestep, denoised_latents,
                train_util.concat_embeddings(prompt_pair.unconditional, prompt_pair.target, bs), guidance_scale=1)
        loss = prompt_pair.loss(target_latents=target_latents, positive_latents=positive_latents,
                                neutral_latents=neutral_latents, unconditional_latents=unconditional_latents)
        pbar.set_description(f"Loss*1k: {loss.item() * 1000:.4f}")
        loss.backward()
        if world > 1:  # mean over ranks == the gradient of the global-batch MSE
            parallel.allreduce_mean_(network.flat.grad)
        optimizer.step()
        lr_scheduler.step()
        if on_step_complete is not None:
            on_step_complete(i, float(loss.item()))
        if save_file and rank == 0 and i % config.save.per_steps == 0 and i != 0 and i != config.train.iterations - 1:
            save_path.mkdir(parents=True, exist_ok=True)
            network.save_weights(save_path / f"{config.save.name}_{i}steps.pt", dtype=save_weight_dtype)
    if save_file and rank == 0:
        save_path.mkdir(parents=True, exist_ok=True)
        network.save_weights(save_path / f"{config.save.name}_last.pt", dtype=save_weight_dtype)
    return network


def main(args):
    config = config_util.load_config_from_yaml(args.config_file)
    if args.name is not None:
        config.save.name = args.name
    attributes = []
    if args.attributes is not None:
        attributes = [a.strip() for a in args.attributes.split(",")]
    config.network.alpha = args.alpha
    config.network.rank = args.rank
    config.save.name += f"_alpha{args.alpha}"
    config.save.name += f"_rank{config.network.rank}"
    config.save.name += f"_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    if str(args.device) == "cpu":
        raise ValueError("--device cpu: the product path has no CPU fallback (the CPU oracle lives under oracle/)")
    import os
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    else:
        device = torch.device(f"cuda:{args.device}")
    train(config, prompts, device, fused_step=args.fused_step, dedup_uncond=not args.no_dedup_uncond)


def add_fused_step_flags(parser):
    """Shared by the four trainers: the fused step is the default, `--no_fused_step` keeps the reference-style loop."""
    g = parser.add_mutually_exclusive_group()
    g.add_argument("--fused_step", dest="fused_step", action="store_true", default=None,
                   help="insist on the fused step (pre-roll + 4-pass step through step.SliderStep / ImageSliderStep: one "
                        "batched UNet pass, native loss / clip / AdamW, no autograd graph -- the path bench.py measures); "
                        "it is the default whenever the optimiser is Adam / AdamW")
    g.add_argument("--no_fused_step", dest="fused_step", action="store_false",
                   help="the reference-style loop: one UNet call per guidance pass, torch autograd and optimiser")
    parser.add_argument("--no_dedup_uncond", action="store_true",
                        help="fused step: run the unconditional half of every frozen pass again, as the reference does, "
                             "instead of once (results are bit-identical either way)")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_file", required=False, default="data/config.yaml")
    parser.add_argument("--alpha", type=float, required=True, help="LoRA weight.")
    parser.add_argument("--rank", type=int, required=False, default=4, help="Rank of LoRA.")
    parser.add_argument("--device", required=False, default=0, help="Device to train on.")
    parser.add_argument("--name", type=str, required=False, default=None, help="Name of the slider.")
    parser.add_argument("--attributes", type=str, required=False, default=None,
                        help="attritbutes to disentangle (comma seperated string)")
    add_fused_step_flags(parser)
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
