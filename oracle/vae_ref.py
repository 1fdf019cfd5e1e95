"""ORACLE (test infrastructure, not shipped): CPU restatement of the image-slider front end --
`AutoencoderKL.encode(...).latent_dist.sample()` and the reference's `get_noisy_image`
(trainscripts/imagesliders/train_util.py:200-235).  Plain fp32 PyTorch; used only by tests/ as the checker.

PARITY UNPINNED: the encoder arithmetic lives in `diffusers` (requirements.txt:3, not installed, not vendored, no
fixtures or weights in the reference).  This file follows the published `AutoencoderKL` architecture of SD-1.x / SD-XL:
    Encoder: conv_in 3->128 | 4 x DownEncoderBlock2D (128, 256, 512, 512; 2 ResnetBlock2D each, no time embedding,
             GroupNorm(32, eps 1e-6), SiLU; Downsample2D = pad (0,1,0,1) + 3x3 stride-2 conv, not on the last block)
             | UNetMidBlock2D (ResnetBlock2D, single-head Attention over the pixels with its own GroupNorm and residual,
             ResnetBlock2D) | GroupNorm + SiLU + conv_out 512->8
    quant_conv 1x1 8->8; DiagonalGaussianDistribution: mean, logvar = chunk(2); logvar clamped to [-30, 20];
    sample = mean + exp(0.5 logvar) * eps;  latents = sample * scaling_factor (0.18215 SD-1.x, 0.13025 SD-XL).
Independent checks: the state-dict keys follow diffusers' names; the parameter count of the default config equals the
public SD VAE encoder + quant_conv (34,163,592 + 72).  What IS the reference's own here -- get_noisy_image's order of
operations (encode, scale, draw noise with the caller's generator, add_noise at timesteps[total_timesteps]) -- is
restated in `get_noisy_image` below line by line.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VAEConfig:
    in_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


def sd_vae_config() -> VAEConfig:
    return VAEConfig()


def sdxl_vae_config() -> VAEConfig:
    return VAEConfig(scaling_factor=0.13025)


def tiny_vae_config() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 128, 128, 128), norm_num_groups=16)


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class Attention(nn.Module):
    """Single-head self-attention over the pixels with its own GroupNorm and residual (the VAE's AttnBlock)."""

    def __init__(self, ch, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        B, C, H, W = x.shape
        h = self.group_norm(x).reshape(B, C, H * W).transpose(1, 2)
        q, k, v = self.to_q(h), self.to_k(h), self.to_v(h)
        p = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(B, C, H, W) + x


class UNetMidBlock2D(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(ch, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, groups), ResnetBlock2D(ch, ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, 1, 1)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, out in enumerate(boc):
            self.down_blocks.append(DownEncoderBlock2D(ch, out, cfg.layers_per_block, cfg.norm_num_groups,
                                                       i != len(boc) - 1))
            ch = out
        self.mid_block = UNetMidBlock2D(ch, cfg.norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch, 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(self.conv_act(self.conv_norm_out(x)))


class DiagonalGaussianDistribution:
    def __init__(self, moments):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKLRef(nn.Module):
    """Encoder half + quant_conv (what the image sliders use); state-dict keys as in diffusers."""

    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def moments(self, x):
        return self.quant_conv(self.encoder(x))

    def encode(self, x):
        return _EncoderOutput(DiagonalGaussianDistribution(self.moments(x)))


@torch.no_grad()
def init_synthetic_(vae: nn.Module, seed: int = 0):
    """Seeded fan-in-scaled weights snapped to bf16-representable values (an fp16 / bf16 / fp32 copy holds the same)."""
    import math
    g = torch.Generator().manual_seed(seed)
    for name, p in vae.named_parameters():
        if p.ndim >= 2:
            w = torch.randn(p.shape, generator=g) * (0.9 / math.sqrt(p[0].numel()))
        elif name.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            w = 0.02 * torch.randn(p.shape, generator=g)
        p.copy_(w.to(torch.bfloat16).to(p.dtype))
    return vae


def preprocess(img: torch.Tensor) -> torch.Tensor:
    """VaeImageProcessor.preprocess for an already-sized image given as uint8 [H, W, 3] (or a batch [n, H, W, 3]):
    to float in [0, 1], NCHW, then 2 x - 1."""
    if img.ndim == 3:
        img = img.unsqueeze(0)
    return img.permute(0, 3, 1, 2).float().div(255.0) * 2.0 - 1.0


@torch.no_grad()
def get_noisy_image(img_u8, vae, generator, scheduler, total_timesteps: int = 1000):
    """I/train_util.py:200-235: encode -> latent_dist.sample(None) -> x scaling_factor -> noise = randn(shape, generator)
    -> scheduler.add_noise(latents, noise, scheduler.timesteps[total_timesteps : total_timesteps + 1]).
    Returns (noised latents, noise, the clean scaled latents, the posterior) -- the last two for the parity tests."""
    image = preprocess(img_u8)
    dist = vae.encode(image).latent_dist
    init_latents = vae.cfg.scaling_factor * dist.sample(None)
    noise = torch.randn(init_latents.shape, generator=generator)
    timestep = scheduler.timesteps[total_timesteps:total_timesteps + 1]
    return scheduler.add_noise(init_latents, noise, timestep), noise, init_latents, dist
