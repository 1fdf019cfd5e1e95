"""`AutoencoderKL` drop-in for the image sliders -- the encoder half the reference uses through
`vae.encode(image).latent_dist.sample(None)`, `vae.config.scaling_factor`, `vae.config.block_out_channels`, `vae.device`
(trainscripts/imagesliders/train_util.py:213-222) -- as a parameter container with diffusers' module names
(`encoder.down_blocks.0.resnets.0.conv1.weight`, `encoder.mid_block.attentions.0.to_q.weight`, `quant_conv.weight`, ...:
a diffusers VAE state dict loads by key, decoder entries are ignored) whose arithmetic runs in the HIP engine
(csrc/engine.hip `forward_vae`).  No PyTorch forward: without the HIP library, or on a CPU device, encode() raises.
Also `VaeImageProcessor.preprocess` as the reference uses it (PIL image or array -> [-1, 1] NCHW)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _native


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


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _Down(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, down):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Down(cout)]) if down else None


class _Attn(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])


class _Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.attentions = nn.ModuleList([_Attn(ch, groups)])
        self.resnets = nn.ModuleList([_Resnet(ch, ch, groups), _Resnet(ch, ch, groups)])


class _Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, 1, 1)
        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, out in enumerate(boc):
            self.down_blocks.append(_DownBlock(ch, out, cfg.layers_per_block, cfg.norm_num_groups, i != len(boc) - 1))
            ch = out
        self.mid_block = _Mid(ch, cfg.norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch, 2 * cfg.latent_channels, 3, padding=1)


class DiagonalGaussianDistribution:
    """Posterior of AutoencoderKL.encode: moments = (mean | logvar), logvar clamped to [-30, 20]."""

    def __init__(self, moments: torch.Tensor):
        self.mean, logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        # diffusers draws on the parameters' device (randn_tensor); a CPU generator draws on the host
        dev = self.mean.device if generator is None or generator.device.type != "cpu" else "cpu"
        eps = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=self.mean.dtype)
        return self.mean + self.std * eps.to(self.mean.device)

    def mode(self) -> torch.Tensor:
        return self.mean

    def rows(self, lo: int, hi: int) -> "DiagonalGaussianDistribution":
        """The posterior of samples [lo, hi) of a batched encode (train_util.get_noisy_image_pair)."""
        d = object.__new__(DiagonalGaussianDistribution)
        d.mean, d.logvar, d.std = self.mean[lo:hi], self.logvar[lo:hi], self.std[lo:hi]
        return d


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


_DEPRECATED_ATTN_NAMES = ((".query.", ".to_q."), (".key.", ".to_k."), (".value.", ".to_v."), (".proj_attn.", ".to_out.0."))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        self.config = cfg
        self.encoder = _Encoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self._engines = {}

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    @property
    def device(self):
        return self.quant_conv.weight.device

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts a full diffusers AutoencoderKL state dict: the decoder / post_quant_conv entries are not used."""
        sd = {}
        for k, v in state_dict.items():
            if not (k.startswith("encoder.") or k.startswith("quant_conv.")):
                continue
            # SD-1.x era checkpoints name the mid-block attention query / key / value / proj_attn (diffusers renames them
            # when it loads, `_convert_deprecated_attention_blocks`); some store them as 1 x 1 conv kernels [C, C, 1, 1]
            for old, new in _DEPRECATED_ATTN_NAMES:
                if old in k:
                    k = k.replace(old, new)
                    if v.ndim == 4 and v.shape[-2:] == (1, 1):
                        v = v.reshape(v.shape[0], v.shape[1])
                    break
            sd[k] = v
        return super().load_state_dict(sd, strict=strict)

    def _engine(self, n, h, w):
        if self.device.type != "cuda":
            raise _native.SmiError("the VAE encoder runs only on an MI355X through the HIP engine; move it to a cuda "
                                   "device with vae.to(device, dtype) (there is no CPU fallback)")
        if self.dtype not in _native.DTYPE_CODE:
            raise _native.SmiError(f"engine dtypes are float16/bfloat16, got {self.dtype}")
        key = (self.dtype, str(self.device), h, w)
        e = self._engines.get(key)
        if e is None or e.batch < n:
            if e is not None:
                e.close()
            state = {k: v.detach() for k, v in self.state_dict().items()}
            e = self._engines[key] = _native.VaeEngine(self.config, self.dtype, state, n, h, w, self.device)
        return e

    @torch.no_grad()
    def encode(self, x: torch.Tensor):
        """x: [n, 3, h, w] in [-1, 1] -> object with `.latent_dist` (`.sample(generator)`, `.mode()`, `.mean`, `.std`)."""
        n, _c, h, w = x.shape
        x = x.to(self.device, torch.float32).contiguous()
        return _EncoderOutput(DiagonalGaussianDistribution(self._engine(n, h, w).moments(x)))


class VaeImageProcessor:
    """`preprocess` of diffusers' VaeImageProcessor as get_noisy_image uses it: a PIL image, an HWC uint8 array or a CHW /
    NCHW float tensor in [0, 1] -> float32 NCHW in [-1, 1], height and width floored to a multiple of vae_scale_factor."""

    def __init__(self, vae_scale_factor: int = 8):
        self.vae_scale_factor = vae_scale_factor

    def preprocess(self, image) -> torch.Tensor:
        f = self.vae_scale_factor
        if isinstance(image, torch.Tensor):
            t = image if image.ndim == 4 else image.unsqueeze(0)
            t = t.float()
            h, w = t.shape[-2] - t.shape[-2] % f, t.shape[-1] - t.shape[-1] % f
            if (h, w) != tuple(t.shape[-2:]):
                t = torch.nn.functional.interpolate(t, size=(h, w), mode="bilinear", antialias=True)
            return 2.0 * t - 1.0
        if hasattr(image, "resize") and hasattr(image, "size") and not isinstance(image, np.ndarray):  # PIL
            w, h = image.size
            w, h = w - w % f, h - h % f
            if (w, h) != image.size:
                from PIL import Image
                image = image.resize((w, h), resample=Image.LANCZOS)
            image = np.array(image.convert("RGB"))
        arr = np.asarray(image)
        if arr.ndim == 3:
            arr = arr[None]
        t = torch.from_numpy(arr.astype(np.float32) / 255.0).permute(0, 3, 1, 2)
        return 2.0 * t - 1.0


@torch.no_grad()
def init_synthetic_(vae: nn.Module, seed: int = 0):
    """Same seeded synthetic weights as oracle/vae_ref.py:init_synthetic_ (kept textually independent of oracle/)."""
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
