"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's slider train step.

Each function cites the reference lines it follows (paths relative to /root/reference):
  T = conceptmod/textsliders, I = trainscripts/imagesliders.
PARITY STATUS: pinned.  tests/golden/make_goldens.py imports the reference's own T/lora.py, T/train_util.py and
T/prompt_util.py (with a type-only `diffusers` stub), runs them on oracle/unet_ref.py and writes the fixtures that
tests/test_oracle_golden.py checks this file against.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn


# ----------------------------------------------------------------------------------------------------------
# LoRA adaptor (T/lora.py:76-138, 141-301)
# ----------------------------------------------------------------------------------------------------------
class LoRAModuleRef(nn.Module):
    """y = org(x) + up(down(x)) * multiplier * (alpha / rank)            (T/lora.py:134-138, :118-119)"""

    def __init__(self, lora_name: str, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=None):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        if isinstance(org_module, nn.Linear):
            self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
            self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        else:  # Conv2d: k x k down, 1 x 1 up (T/lora.py:100-114)
            self.lora_dim = min(lora_dim, org_module.in_channels, org_module.out_channels)
            self.lora_down = nn.Conv2d(org_module.in_channels, self.lora_dim, org_module.kernel_size,
                                       org_module.stride, org_module.padding, bias=False)
            self.lora_up = nn.Conv2d(self.lora_dim, org_module.out_channels, (1, 1), (1, 1), bias=False)
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = alpha / self.lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=1)  # T/lora.py:123
        nn.init.zeros_(self.lora_up.weight)  # T/lora.py:124
        self.multiplier = multiplier
        self.org_module = [org_module]  # list: not registered as a child

    def apply_to(self):
        org = self.org_module[0]
        self.org_forward = org.forward
        org.forward = self.forward
        del self.org_module

    def forward(self, x):
        return self.org_forward(x) + self.lora_up(self.lora_down(x)) * self.multiplier * self.scale


C3LIER_TARGET_REPLACE = ("Attention", "ResnetBlock2D", "Downsample2D", "Upsample2D", "DownBlock2D", "UpBlock2D")
"""T/lora.py:14-26 after T/train_lora.py:44-46 (`network.type: c3lier`)."""


def select_lora_targets(unet: nn.Module, train_method: str, target_replace=("Attention",), prefix="lora_unet",
                        with_duplicates=False):
    """Name/class-name walk of T/lora.py:194-251.  Returns [(lora_name, module_path, child_module)]; with
    `with_duplicates` also the second visits of a name (the reference builds a LoRAModule for those too and then
    drops it, T/lora.py:243-249), as 4-tuples ending in True."""
    out, names = [], []
    for name, module in unet.named_modules():
        if train_method in ("noxattn", "noxattn-hspace", "noxattn-hspace-last"):
            if "attn2" in name or "time_embed" in name:
                continue
        elif train_method == "innoxattn":
            if "attn2" in name:
                continue
        elif train_method == "selfattn":
            if "attn1" not in name:
                continue
        elif train_method in ("xattn", "xattn-strict"):
            if "attn2" not in name:
                continue
        elif train_method == "full":
            pass
        else:
            raise NotImplementedError(f"train_method: {train_method} is not implemented.")
        if module.__class__.__name__ in target_replace:
            for child_name, child in module.named_modules():
                if "add_" in child_name:
                    continue
                if child.__class__.__name__ in ("Linear", "Conv2d", "LoRACompatibleLinear", "LoRACompatibleConv"):
                    if train_method == "xattn-strict" and "out" in child_name:
                        continue
                    if train_method == "noxattn-hspace" and "mid_block" not in name:
                        continue
                    if train_method == "noxattn-hspace-last" and (
                            "mid_block" not in name or ".1" not in name or "conv2" not in child_name):
                        continue
                    lora_name = (prefix + "." + name + "." + child_name).replace(".", "_")
                    if lora_name not in names:
                        names.append(lora_name)
                        out.append((lora_name, name + "." + child_name, child) + ((False,) if with_duplicates else ()))
                    elif with_duplicates:
                        out.append((lora_name, name + "." + child_name, child, True))
    return out


class LoRANetworkRef(nn.Module):
    def __init__(self, unet, rank=4, multiplier=1.0, alpha=1.0, train_method="full", target_replace=("Attention",)):
        super().__init__()
        self.lora_scale = 1
        self.multiplier = multiplier
        self.lora_dim = rank
        self.alpha = alpha
        self.unet_loras: List[LoRAModuleRef] = []
        for lora_name, _path, child, dup in select_lora_targets(unet, train_method, target_replace,
                                                                with_duplicates=True):
            m = LoRAModuleRef(lora_name, child, multiplier, rank, alpha)  # built (RNG consumed) even when dropped
            if not dup:
                self.unet_loras.append(m)
        assert len({l.lora_name for l in self.unet_loras}) == len(self.unet_loras)
        for lora in self.unet_loras:
            lora.apply_to()
            self.add_module(lora.lora_name, lora)

    def prepare_optimizer_params(self):
        params = []
        for lora in self.unet_loras:
            params.extend(lora.parameters())
        return [{"params": params}]

    def set_lora_slider(self, scale):
        self.lora_scale = scale

    def __enter__(self):  # T/lora.py:295-297
        for lora in self.unet_loras:
            lora.multiplier = 1.0 * self.lora_scale

    def __exit__(self, *a):  # T/lora.py:299-301
        for lora in self.unet_loras:
            lora.multiplier = 0


# ----------------------------------------------------------------------------------------------------------
# DoRA (`--peft_type dora`: T/dora.py:53-162, T/train_lora_xl.py:87-90)
# ----------------------------------------------------------------------------------------------------------
class DoRAModuleRef(nn.Module):
    """y = org(x) + F.linear(x, dW) * scale * multiplier with dW = (W + up down) * (dora_scale / ||W + up down||_col) - W,
    the column norm (over the output dimension, one per input column) DETACHED (T/dora.py:124-162); dora_scale starts at
    the frozen weight's column norms (:95-105).  Linear targets (what train_lora_xl.py adapts)."""

    def __init__(self, lora_name: str, org_module: nn.Linear, multiplier=1.0, lora_dim=4, alpha=1):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        self.lora_down = nn.Linear(org_module.in_features, lora_dim, bias=False)
        self.lora_up = nn.Linear(lora_dim, org_module.out_features, bias=False)
        w = org_module.weight
        self.dora_scale = nn.Parameter(torch.norm(w.detach().float(), dim=0, keepdim=True).to(w.dtype))  # [1, in]
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.scale = alpha / self.lora_dim
        self.register_buffer("alpha", torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=1)
        nn.init.zeros_(self.lora_up.weight)
        self.multiplier = multiplier
        self.org_module = [org_module]

    def apply_to(self):
        org = self.org_module[0]
        self.org_forward = org.forward
        org.forward = self.forward
        self.org_weight = org.weight.detach()
        del self.org_module

    def forward(self, x):
        w = self.org_weight
        v = w + self.lora_up.weight @ self.lora_down.weight
        norm = v.norm(dim=0, keepdim=True).detach()
        dw = v * (self.dora_scale / norm) - w
        return self.org_forward(x) + torch.nn.functional.linear(x, dw.to(x.dtype)) * self.scale * self.multiplier


class DoRANetworkRef(nn.Module):
    """T/dora.py:165-330.  NOTE the reference passes `self.alpha` into DoRAModule's `prefix` slot (:247-249 vs :57-65),
    so every module keeps the default alpha = 1 whatever the network was given: reproduced."""

    def __init__(self, unet, rank=4, multiplier=1.0, alpha=1.0, train_method="full", target_replace=("Attention",)):
        super().__init__()
        self.multiplier = multiplier
        self.lora_dim = rank
        self.alpha = alpha
        self.unet_loras = []
        for lora_name, _path, child, dup in select_lora_targets(unet, train_method, target_replace,
                                                                with_duplicates=True):
            m = DoRAModuleRef(lora_name, child, multiplier, rank, 1)
            if not dup:
                self.unet_loras.append(m)
        for lora in self.unet_loras:
            lora.apply_to()
            self.add_module(lora.lora_name, lora)

    def prepare_optimizer_params(self):
        params = []
        for lora in self.unet_loras:
            params.extend(lora.parameters())
        return [{"params": params}]

    def __enter__(self):  # T/dora.py:322-324: 1.0, no lora_scale
        for lora in self.unet_loras:
            lora.multiplier = 1.0

    def __exit__(self, *a):
        for lora in self.unet_loras:
            lora.multiplier = 0


# ----------------------------------------------------------------------------------------------------------
# step helpers (T/train_util.py)
# ----------------------------------------------------------------------------------------------------------
def concat_embeddings(unconditional, conditional, n_imgs):  # T/train_util.py:267-272
    return torch.cat([unconditional, conditional]).repeat_interleave(n_imgs, dim=0)


def get_initial_latents(scheduler, n_imgs, height, width, n_prompts, generator=None):  # T/train_util.py:27-39,91-105
    noise = torch.randn((n_imgs, 4, height // 8, width // 8), generator=generator, device="cpu").repeat(
        n_prompts, 1, 1, 1)
    return noise * scheduler.init_noise_sigma.to(noise.device)


def get_add_time_ids(height, width, dtype=torch.float32):  # T/train_util.py:976-1011 (dynamic_crops=False branch)
    return torch.tensor([[height, width, 0, 0, height, width]], dtype=dtype)


def predict_noise(unet, scheduler, timestep, latents, text_embeddings, guidance_scale=7.5):
    """T/train_util.py:276-302"""
    x = torch.cat([latents] * 2)
    x = scheduler.scale_model_input(x, timestep)
    eps = unet(x, timestep, encoder_hidden_states=text_embeddings).sample
    u, t = eps.chunk(2)
    return u + guidance_scale * (t - u)


def predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings, add_time_ids,
                     guidance_scale=7.5, guidance_rescale=0.7):
    """T/train_util.py:449-489 (the rescale result is discarded there, :485-489; the plain CFG mix is returned)"""
    x = torch.cat([latents] * 2)
    x = scheduler.scale_model_input(x, timestep)
    eps = unet(x, timestep, encoder_hidden_states=text_embeddings,
               added_cond_kwargs={"text_embeds": add_text_embeddings, "time_ids": add_time_ids}).sample
    u, t = eps.chunk(2)
    return u + guidance_scale * (t - u)


@torch.no_grad()
def diffusion(unet, scheduler, latents, text_embeddings, total_timesteps=1000, start_timesteps=0, **kw):
    """T/train_util.py:306-327"""
    for timestep in scheduler.timesteps[start_timesteps:total_timesteps]:
        eps = predict_noise(unet, scheduler, timestep, latents, text_embeddings, **kw)
        latents = scheduler.step(eps, timestep, latents).prev_sample
    return latents


@torch.no_grad()
def diffusion_xl(unet, scheduler, latents, text_embeddings, add_text_embeddings, add_time_ids, guidance_scale=1.0,
                 total_timesteps=1000, start_timesteps=0):
    """T/train_util.py:677-708"""
    for timestep in scheduler.timesteps[start_timesteps:total_timesteps]:
        eps = predict_noise_xl(unet, scheduler, timestep, latents, text_embeddings, add_text_embeddings,
                               add_time_ids, guidance_scale=guidance_scale)
        latents = scheduler.step(eps, timestep, latents).prev_sample
    return latents


# ----------------------------------------------------------------------------------------------------------
# loss (T/prompt_util.py:134-174; four-prompt form I/prompt_util.py:108-148)
# ----------------------------------------------------------------------------------------------------------
def slider_loss(action: str, guidance_scale: float, target, positive, neutral, negative):
    """MSE(target, neutral -/+ eta * (positive - negative)).  In the four-prompt (upstream) formulation
    `negative` is the unconditional prediction."""
    if action == "erase":
        goal = neutral - guidance_scale * (positive - negative)
    elif action == "enhance":
        goal = neutral + guidance_scale * (positive - negative)
    else:
        raise ValueError("action must be erase or enhance")
    return torch.nn.functional.mse_loss(target, goal)


# ----------------------------------------------------------------------------------------------------------
# the 4-pass train step (T/train_lora.py:155-300, T/train_lora_xl.py:165-351)
# ----------------------------------------------------------------------------------------------------------
def slider_step(unet, network, scheduler, denoised_latents, current_timestep, emb: Dict[str, torch.Tensor],
                batch_size: int, action="enhance", eta=4.0, cfg_scale=1.0, xl: Optional[dict] = None):
    """3 frozen passes (positive, neutral, negative|unconditional) + 1 adapted pass + loss; returns
    (loss, dict of the four predictions).  `emb` maps 'target','positive','neutral','unconditional'[,'negative']
    to text embeddings [1,77,D]; for XL `xl` = {'pooled': {same keys -> [1,P]}, 'time_ids': [1,6]}.
    Frozen passes run with the adaptor off (multiplier 0, T/lora.py:299-301), the target pass inside
    `with network` (T/train_lora.py:261-273)."""
    sub = "negative" if "negative" in emb else "unconditional"

    def run(key):
        te = concat_embeddings(emb["unconditional"], emb[key], batch_size)
        if xl is None:
            return predict_noise(unet, scheduler, current_timestep, denoised_latents, te, guidance_scale=cfg_scale)
        pooled = concat_embeddings(xl["pooled"]["unconditional"], xl["pooled"][key], batch_size)
        tid = concat_embeddings(xl["time_ids"], xl["time_ids"], batch_size)
        return predict_noise_xl(unet, scheduler, current_timestep, denoised_latents, te, pooled, tid,
                                guidance_scale=cfg_scale)

    network.__exit__(None, None, None)
    with torch.no_grad():
        positive = run("positive")
        neutral = run("neutral")
        negative = run(sub)
    with network:
        target = run("target")
    loss = slider_loss(action, eta, target, positive, neutral, negative)
    return loss, {"target": target, "positive": positive, "neutral": neutral, "negative": negative}


def clip_grad_norm_(params, max_norm: float):
    """torch.nn.utils.clip_grad_norm_ (T/train_lora_xl.py:349): g *= min(1, max_norm / (||g||_2 + 1e-6))."""
    grads = [p.grad for p in params if p.grad is not None]
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step_(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW single-tensor update (the optimiser the reference uses: T/train_util.py:1040,
    T/train_lora_xl.py:104)."""
    p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
