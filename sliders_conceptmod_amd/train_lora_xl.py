"""Text-slider trainer for SD-XL -- the fork's maintained entry point (conceptmod/textsliders/train_lora_xl.py:41-516):
five prompts (adds `negative`), hard-coded AdamW(lr 1e-4, wd 1e-6) + CosineAnnealingLR(T_max 50, eta_min 1e-6),
clip_grad_norm_ 0.2, `.safetensors` output, `on_step_complete` callback, in-memory return when save_file=False.

    python -m sliders_conceptmod_amd.train_lora_xl --config_file cfg.yaml --alpha 1 --rank 4 --device 0 --name x

By default the step runs through sliders_conceptmod_amd.step.SliderStep (one batched UNet pass, no autograd graph, native
loss / clip / AdamW kernels, each distinct frozen sample once) -- the code path bench.py measures; `--no_fused_step` keeps the
reference-style loop (one UNet call per guidance pass, torch autograd + optimiser), same arithmetic (tested)."""
import argparse
import os
import sys
from pathlib import Path

import torch
from tqdm import tqdm

from . import config_util, model_util, parallel, prompt_util, train_util
from .config_util import RootConfig
from .lora import LoRANetwork
from .prompt_util import PromptEmbedsCache, PromptEmbedsPair, PromptEmbedsXL


def encode_xl(text_encoder, tokenizer, prompt, device, dtype) -> PromptEmbedsXL:
    if isinstance(text_encoder, model_util.SyntheticTextEncoder):
        te, pooled = text_encoder.encode(prompt)
        return PromptEmbedsXL(te.to(device, dtype), pooled.to(device, dtype))
    embeds, pooled = [], None
    for tok, enc in zip(tokenizer, text_encoder):  # penultimate layer of each encoder, pooled of the last
        ids = tok(prompt, padding="max_length", max_length=tok.model_max_length, truncation=True,
                  return_tensors="pt").input_ids
        out = enc(ids.to(enc.device), output_hidden_states=True)
        pooled = out[0]
        embeds.append(out.hidden_states[-2])
    return PromptEmbedsXL(torch.cat(embeds, dim=-1).to(device, dtype), pooled.to(device, dtype))


def train(config: RootConfig, prompts: list, device, on_step_complete=None, peft_type="lora", rank=4, save_file=True,
          models=None, fused_step=None, optimizer_kwargs=None, dedup_uncond=True):
    """`fused_step` (not in the reference signature): None / True = the fused path (the script's optimiser is a hard-coded
    AdamW, train_lora_xl.py:104), False = the reference-style autograd loop; `dedup_uncond`: inside the fused step, each
    distinct frozen sample runs once (bit-identical results, step.py).  `optimizer_kwargs` (not in the reference signature) overrides keyword arguments of the hard-coded AdamW
    (train_lora_xl.py:104), e.g. {"eps": 1e-3} for the element-wise trajectory parity test; the per-step losses of the
    run are left in `network.training_losses`."""
    if peft_type not in ("lora", "dora"):
        raise ValueError(f"peft_type must be lora or dora, got {peft_type}")
    save_path = Path(config.save.path)
    weight_dtype = config_util.parse_precision(config.train.precision)
    save_weight_dtype = config_util.parse_precision(config.train.precision)
    guidance_scale = config.train.cfg
    tokenizers, text_encoders, unet, noise_scheduler = models or model_util.load_models(
        config.pretrained_model.name_or_path, scheduler_name=config.train.noise_scheduler, xl=True)
    # data parallelism: one control RNG for all ranks, seeded BEFORE the adaptor init draws from it (parallel.py)
    rank_, world = parallel.world_info()
    parallel.sync_control_rng(device=device)
    noise_scheduler.dp_shard = (rank_, world)
    if isinstance(text_encoders, (list, tuple)):  # train_lora_xl.py:72-75
        for te in text_encoders:
            te.to(device, dtype=weight_dtype)
            te.requires_grad_(False)
            te.eval()
    unet.to(device, dtype=weight_dtype)
    unet.requires_grad_(False)
    unet.eval()
    if peft_type == "dora":  # train_lora_xl.py:87-90
        from .dora import DoRANetwork as peft_class
    else:
        peft_class = LoRANetwork
    network = peft_class(unet, rank=rank, multiplier=1.0, delimiter="_", target_replace=["Attention"],
                         prefix="lora_unet", train_method=config.network.training_method).to(device,
                                                                                             dtype=weight_dtype)
    parallel.broadcast_(network.flat.data)
    okw = dict(lr=1e-4, weight_decay=1e-6)
    okw.update(optimizer_kwargs or {})
    optimizer = torch.optim.AdamW(network.parameters(), **okw)
    lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=50, eta_min=1e-6)
    criteria = torch.nn.MSELoss()

    cache = PromptEmbedsCache()
    prompt_pairs, raw = [], []
    with torch.no_grad():
        for settings in prompts:
            names = [settings.target, settings.positive, settings.negative, settings.neutral, settings.unconditional]
            for prompt in names:
                if prompt is not None and cache[prompt] is None:
                    cache[prompt] = encode_xl(text_encoders, tokenizers, prompt, device, weight_dtype)
            neg = cache[settings.negative] if settings.negative is not None else None
            prompt_pairs.append(PromptEmbedsPair(criteria, cache[settings.target], cache[settings.positive],
                                                 cache[settings.unconditional], cache[settings.neutral], neg,
                                                 settings))
    del tokenizers, text_encoders

    stepper = None
    if okw.get("amsgrad") and fused_step:
        raise ValueError("--fused_step implements AdamW without amsgrad")
    if (fused_step or fused_step is None) and not okw.get("amsgrad"):
        from .step import SliderStep
        stepper = SliderStep(unet, network, noise_scheduler, lr=okw["lr"], weight_decay=okw["weight_decay"],
                             eps=okw.get("eps", 1e-8), betas=okw.get("betas", (0.9, 0.999)), max_grad_norm=0.2,
                             cfg_scale=guidance_scale, dedup_uncond=dedup_uncond)
    cond_cache = {}  # fused path: conditioning tensors per (prompt pair, batch, size); rebuilt only with dynamic crops
    network.training_losses = []
    pbar = tqdm(range(config.train.iterations), disable=rank_ != 0)
    for i in pbar:
        with torch.no_grad():
            noise_scheduler.set_timesteps(config.train.max_denoising_steps, device=device)
            optimizer.zero_grad()
            prompt_pair = prompt_pairs[torch.randint(0, len(prompt_pairs), (1,)).item()]
            timesteps_to = torch.randint(1, config.train.max_denoising_steps, (1,)).item()
            height, width = prompt_pair.resolution, prompt_pair.resolution
            if prompt_pair.dynamic_resolution:
                height, width = train_util.get_random_resolution_in_bucket(prompt_pair.resolution)
            bs = prompt_pair.batch_size
            latents = train_util.get_initial_latents(noise_scheduler, bs, height, width, 1)
            if world > 1:
                latents = latents[parallel.shard_slice(bs, rank_, world)]
                bs = bs // world
            latents = latents.to(device, dtype=torch.float32)
            add_time_ids = train_util.get_add_time_ids(height, width, dynamic_crops=prompt_pair.dynamic_crops,
                                                       dtype=torch.float32).to(device)

            def cond(e):
                return dict(
                    text_embeddings=train_util.concat_embeddings(prompt_pair.unconditional.text_embeds,
                                                                 e.text_embeds, bs),
                    add_text_embeddings=train_util.concat_embeddings(prompt_pair.unconditional.pooled_embeds,
                                                                     e.pooled_embeds, bs),
                    add_time_ids=train_util.concat_embeddings(add_time_ids, add_time_ids, bs))

            c = None
            if stepper is not None:  # fused: conditioning tensors per (prompt pair, batch, size), pre-roll without autograd
                ckey = (id(prompt_pair), bs, height, width)
                c = None if prompt_pair.dynamic_crops else cond_cache.get(ckey)
                if c is None:
                    emb = {k: getattr(prompt_pair, k).text_embeds for k in
                           ("target", "positive", "neutral", "unconditional", "negative")}
                    pooled = {k: getattr(prompt_pair, k).pooled_embeds for k in emb}
                    c = cond_cache[ckey] = stepper.make_conditioning(emb, bs, pooled, add_time_ids)
                denoised_latents = stepper.preroll(latents, c, timesteps_to, guidance_scale)
            else:
                with network:
                    denoised_latents = train_util.diffusion_xl(unet, noise_scheduler, latents,
                                                               **cond(prompt_pair.target), start_timesteps=0,
                                                               total_timesteps=timesteps_to,
                                                               guidance_scale=guidance_scale)
            noise_scheduler.set_timesteps(1000)
            current_timestep = noise_scheduler.timesteps[int(timesteps_to * 1000 / config.train.max_denoising_steps)]
            if stepper is None:
                positive_latents = train_util.predict_noise_xl(unet, noise_scheduler, current_timestep,
                                                               denoised_latents, **cond(prompt_pair.positive),
                                                               guidance_scale=guidance_scale)
                neutral_latents = train_util.predict_noise_xl(unet, noise_scheduler, current_timestep,
                                                              denoised_latents, **cond(prompt_pair.neutral),
                                                              guidance_scale=guidance_scale)
                negative_latents = train_util.predict_noise_xl(unet, noise_scheduler, current_timestep,
                                                               denoised_latents, **cond(prompt_pair.negative),
                                                               guidance_scale=guidance_scale)
        if stepper is not None:
            lr_now = lr_scheduler.get_last_lr()[0]
            loss = stepper.train_step(denoised_latents, current_timestep, c, prompt_pair.action,
                                      prompt_pair.guidance_scale, lr=lr_now)
            optimizer.step()  # no gradients: a no-op that keeps torch's "optimizer before scheduler" order
        else:
            with network:
                target_latents = train_util.predict_noise_xl(unet, noise_scheduler, current_timestep,
                                                             denoised_latents, **cond(prompt_pair.target),
                                                             guidance_scale=guidance_scale)
            loss = prompt_pair.loss(target_latents=target_latents, positive_latents=positive_latents,
                                    neutral_latents=neutral_latents, negative_latents=negative_latents)
            loss.backward()
            if world > 1:
                torch.distributed.all_reduce(network.flat.grad)
                network.flat.grad.div_(world)
            torch.nn.utils.clip_grad_norm_(network.parameters(), max_norm=0.2)
            optimizer.step()
        lr_scheduler.step()
        lv = float(loss.item())  # the one host sync per step, as the reference's loss.item() (train_lora_xl.py:346)
        network.training_losses.append(lv)
        pbar.set_description(f"Loss*1k: {lv * 1000:.4f}")
        if save_file and rank_ == 0 and i % config.save.per_steps == 0 and i != 0 and i != config.train.iterations - 1:
            save_path.mkdir(parents=True, exist_ok=True)
            network.save_weights(save_path / f"{config.save.name}_{i}steps.safetensors", dtype=save_weight_dtype)
        if on_step_complete is not None:
            on_step_complete(i)
    if save_file:
        if rank_ == 0:
            save_path.mkdir(parents=True, exist_ok=True)
            network.save_weights(save_path / f"{config.save.name}_last.safetensors", dtype=save_weight_dtype)
        return network
    return network.get_state_dict(save_weight_dtype)


def train_lora(target, positive, negative, unconditional, alpha=1.0, rank=4, device=0, name=None, attributes=None,
               batch_size=1, config_file="data/config-xl.yaml", resolution=512, steps=None, on_step_complete=None,
               peft_type="lora", save_file=True, models=None, neutral=None, guidance_scale=4, action="enhance"):
    """Programmatic entry (train_lora_xl.py:421-458): builds the prompt settings in memory and trains."""
    config = config_util.load_config_from_yaml(config_file)
    if steps is not None:
        config.train.iterations = steps
    if name is not None:
        config.save.name = name
    config.network.alpha = alpha
    config.network.rank = rank
    config.save.name += f"_alpha{alpha}_rank{rank}_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    settings = prompt_util.PromptSettings(target=target, positive=positive, negative=negative,
                                          unconditional=unconditional, neutral=neutral if neutral is not None else target,
                                          action=action, guidance_scale=guidance_scale, resolution=resolution,
                                          batch_size=batch_size)
    dev = torch.device(f"cuda:{device}") if not isinstance(device, torch.device) else device
    return train(config, [settings], dev, on_step_complete, peft_type, rank, save_file, models)


def main(args):
    config = config_util.load_config_from_yaml(args.config_file)
    if args.name is not None:
        config.save.name = args.name
    attributes = [a.strip() for a in args.attributes.split(",")] if args.attributes is not None else []
    config.network.alpha = args.alpha
    config.network.rank = args.rank
    config.save.name += f"_alpha{args.alpha}"
    config.save.name += f"_rank{config.network.rank}"
    config.save.name += f"_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    if str(args.device) == "cpu":
        raise ValueError("--device cpu: the product path has no CPU fallback (the CPU oracle lives under oracle/)")
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")
        device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    else:
        device = torch.device(f"cuda:{args.device}")
    train(config, prompts, device, None, args.peft_type, args.rank, True, None, args.fused_step,
          dedup_uncond=not args.no_dedup_uncond)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_file", required=False, default="data/config-xl.yaml")
    parser.add_argument("--alpha", type=float, required=True, help="LoRA weight.")
    parser.add_argument("--rank", type=int, required=False, default=4, help="Rank of LoRA.")
    parser.add_argument("--device", required=False, default=0, help="Device to train on.")
    parser.add_argument("--name", type=str, required=False, default=None, help="Name of the slider.")
    parser.add_argument("--attributes", type=str, required=False, default=None)
    parser.add_argument("--peft_type", type=str, required=False, default="lora")
    from .train_lora import add_fused_step_flags
    add_fused_step_flags(parser)
    return parser


if __name__ == "__main__":
    main(build_parser().parse_args())
