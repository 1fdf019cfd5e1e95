"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the model arithmetic on the hot path.

The reference (ntc-ai/sliders-conceptmod) holds no model arithmetic of its own: every FLOP of
`unet(sample, t, encoder_hidden_states=..., added_cond_kwargs=...).sample`
(call sites conceptmod/textsliders/train_util.py:290-294 and :471-476) runs inside the un-vendored
third-party package `diffusers` (requirements.txt:3 pins diffusers==0.20.2; absent from this image).
This file restates the published `UNet2DConditionModel` architecture for the two configurations the
reference trains (SD-1.x: model_util.py:27 "CompVis/stable-diffusion-v1-4"; SD-XL base) in plain fp32
PyTorch.

PARITY STATUS: the UNet arithmetic is **parity unpinned** -- the reference ships no golden vectors, no
tests and no weights for it, and diffusers is not importable here.  What IS pinned against the reference's
own code (see tests/golden/make_goldens.py): module/class names are laid out so that the reference's
name-based `LoRANetwork.create_modules` (conceptmod/textsliders/lora.py:194-251) selects exactly the same
children here as it would on a diffusers UNet, and `LoRAModule.forward` (lora.py:134-138),
`predict_noise(_xl)` (train_util.py:276-302, 449-489) and `PromptEmbedsPair.loss`
(prompt_util.py:134-174) from the reference are run *on top of this UNet* to generate the goldens.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

`storage_dtype`: the HIP engine stores activations in fp16/bf16 and accumulates in fp32.  Setting
`unet.storage_dtype = torch.float16` makes this restatement round at exactly the points where the engine
writes a tensor to HBM (documented in DESIGN.md "rounding points"), so the HIP-vs-oracle comparison can be
much tighter than the fp16 round-off noise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    num_attention_heads: Tuple[int, ...] = (8, 8, 8, 8)  # diffusers' (mis-named) attention_head_dim
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    use_linear_projection: bool = False
    addition_embed_type: Optional[str] = None  # "text_time" for SD-XL
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    mid_block_transformer_layers: Optional[int] = None  # default: last entry of transformer_layers_per_block

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


def sd1x_config() -> UNetConfig:
    """SD-1.4 / SD-1.5 UNet (public config of CompVis/stable-diffusion-v1-4, runwayml/stable-diffusion-v1-5)."""
    return UNetConfig()


def sdxl_config() -> UNetConfig:
    """SD-XL base 1.0 UNet (public config of stabilityai/stable-diffusion-xl-base-1.0)."""
    return UNetConfig(
        block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        layers_per_block=2,
        transformer_layers_per_block=(1, 2, 10),
        num_attention_heads=(5, 10, 20),
        cross_attention_dim=2048,
        use_linear_projection=True,
        addition_embed_type="text_time",
    )


def tiny_sd1x_config() -> UNetConfig:
    """Same topology as SD-1.x (conv proj_in/out, 4 levels, 1 tf layer) at 1/5 width; head_dim 16/32/64/64.
    Channel counts stay multiples of 64 (the engine's MFMA K tile) like every real SD channel count."""
    return UNetConfig(
        block_out_channels=(64, 128, 256, 256),
        num_attention_heads=(4, 4, 4, 4),
        cross_attention_dim=64,
        norm_num_groups=16,
    )


def tiny_sdxl_config() -> UNetConfig:
    """Same topology as SD-XL (linear proj, text_time conditioning, no attention at level 0, 1/2/3 tf layers),
    head_dim 32, at 1/5 width."""
    return UNetConfig(
        block_out_channels=(64, 128, 256),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 3),
        num_attention_heads=(2, 4, 8),
        cross_attention_dim=64,
        norm_num_groups=16,
        use_linear_projection=True,
        addition_embed_type="text_time",
        addition_time_embed_dim=32,
        projection_class_embeddings_input_dim=6 * 32 + 64,
    )


# --------------------------------------------------------------------------------------------------------
# storage rounding hook
# --------------------------------------------------------------------------------------------------------
class _Q:
    """Process-wide rounding mode; set through UNet2DConditionModel.storage_dtype."""
    dtype: Optional[torch.dtype] = None


def q(x: torch.Tensor) -> torch.Tensor:
    if _Q.dtype is None:
        return x
    # straight-through: round the value, keep the autograd graph (the engine's backward rounds separately)
    return x + (x.detach().to(_Q.dtype).to(x.dtype) - x.detach())


# --------------------------------------------------------------------------------------------------------
# embeddings
# --------------------------------------------------------------------------------------------------------
def get_timestep_embedding(timesteps: torch.Tensor, dim: int, flip_sin_to_cos: bool = True,
                           downscale_freq_shift: float = 0.0, max_period: int = 10000) -> torch.Tensor:
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels: int):
        super().__init__()
        self.num_channels = num_channels

    def forward(self, t):
        return get_timestep_embedding(t, self.num_channels, flip_sin_to_cos=True, downscale_freq_shift=0.0)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        x = q(self.linear_1(x))
        x = q(self.act(x))
        return q(self.linear_2(x))


# --------------------------------------------------------------------------------------------------------
# resnet / sampling blocks
# --------------------------------------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, groups: int, eps: float = 1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0) if in_channels != out_channels else None

    def forward(self, x, temb_act):
        # temb_act = q(silu(emb)) is computed once per forward by the caller (the engine does the same)
        h = q(self.nonlinearity(self.norm1(x)))
        t = q(self.time_emb_proj(temb_act))[:, :, None, None]
        h = q(self.conv1(h) + t)
        h = q(self.nonlinearity(self.norm2(h)))
        h = self.conv2(self.dropout(h))
        sc = q(self.conv_shortcut(x)) if self.conv_shortcut is not None else x
        return q(sc + h)


class Downsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return q(self.conv(x))


class Upsample2D(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return q(self.conv(x))


# --------------------------------------------------------------------------------------------------------
# transformer
# --------------------------------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        ctx_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(ctx_dim, inner, bias=False)
        self.to_v = nn.Linear(ctx_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, context=None):
        context = x if context is None else context
        B, N, _ = x.shape
        H = self.heads
        qh = q(self.to_q(x)).view(B, N, H, -1).transpose(1, 2)
        kh = q(self.to_k(context)).view(B, context.shape[1], H, -1).transpose(1, 2)
        vh = q(self.to_v(context)).view(B, context.shape[1], H, -1).transpose(1, 2)
        s = torch.matmul(qh, kh.transpose(-1, -2)) * self.scale
        p = torch.softmax(s, dim=-1)
        o = q(torch.matmul(p, vh).transpose(1, 2).reshape(B, N, -1))
        # NOTE: no rounding here -- the engine adds bias, LoRA delta and the residual in the GEMM epilogue
        # and rounds once; the caller (BasicTransformerBlock) applies q() after the residual add.
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = q(self.proj(x)).chunk(2, dim=-1)
        return q(h * F.gelu(gate))


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, heads: int, dim_head: int, cross_attention_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = q(self.attn1(q(self.norm1(x))) + x)
        x = q(self.attn2(q(self.norm2(x)), context) + x)
        x = q(self.ff(q(self.norm3(x))) + x)
        return x


class Transformer2DModel(nn.Module):
    def __init__(self, heads: int, dim_head: int, in_channels: int, num_layers: int, cross_attention_dim: int,
                 groups: int, use_linear_projection: bool):
        super().__init__()
        inner = heads * dim_head
        self.use_linear_projection = use_linear_projection
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        if use_linear_projection:
            self.proj_in = nn.Linear(in_channels, inner)
        else:
            self.proj_in = nn.Conv2d(in_channels, inner, 1, 1, 0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        if use_linear_projection:
            self.proj_out = nn.Linear(inner, in_channels)
        else:
            self.proj_out = nn.Conv2d(inner, in_channels, 1, 1, 0)

    def forward(self, x, context):
        B, C, H, W = x.shape
        res = x
        h = q(self.norm(x))
        if self.use_linear_projection:
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
            h = q(self.proj_in(h))
        else:
            h = q(self.proj_in(h))
            h = h.permute(0, 2, 3, 1).reshape(B, H * W, -1)
        for blk in self.transformer_blocks:
            h = blk(h, context)
        if self.use_linear_projection:
            h = self.proj_out(h)
            h = h.reshape(B, H, W, -1).permute(0, 3, 1, 2)
        else:
            h = h.reshape(B, H, W, -1).permute(0, 3, 1, 2)
            h = self.proj_out(h)
        return q(h + res)


# --------------------------------------------------------------------------------------------------------
# down / mid / up blocks
# --------------------------------------------------------------------------------------------------------
class DownBlock2D(nn.Module):
    def __init__(self, cfg: UNetConfig, in_ch: int, out_ch: int, add_downsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, cfg.time_embed_dim, cfg.norm_num_groups)
            for i in range(cfg.layers_per_block)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None

    def forward(self, h, temb, context):
        outs = []
        for r in self.resnets:
            h = r(h, temb)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, cfg: UNetConfig, in_ch: int, out_ch: int, add_downsample: bool, heads: int, tf_layers: int):
        super().__init__()
        self.attentions = nn.ModuleList([
            Transformer2DModel(heads, out_ch // heads, out_ch, tf_layers, cfg.cross_attention_dim,
                               cfg.norm_num_groups, cfg.use_linear_projection)
            for _ in range(cfg.layers_per_block)])
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_ch if i == 0 else out_ch, out_ch, cfg.time_embed_dim, cfg.norm_num_groups)
            for i in range(cfg.layers_per_block)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None

    def forward(self, h, temb, context):
        outs = []
        for r, a in zip(self.resnets, self.attentions):
            h = r(h, temb)
            h = a(h, context)
            outs.append(h)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, cfg: UNetConfig, ch: int, heads: int, tf_layers: int):
        super().__init__()
        self.attentions = nn.ModuleList([
            Transformer2DModel(heads, ch // heads, ch, tf_layers, cfg.cross_attention_dim, cfg.norm_num_groups,
                               cfg.use_linear_projection)])
        self.resnets = nn.ModuleList([
            ResnetBlock2D(ch, ch, cfg.time_embed_dim, cfg.norm_num_groups),
            ResnetBlock2D(ch, ch, cfg.time_embed_dim, cfg.norm_num_groups)])

    def forward(self, h, temb, context):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, context)
        h = self.resnets[1](h, temb)
        return h


class UpBlock2D(nn.Module):
    def __init__(self, cfg: UNetConfig, in_ch: int, prev_ch: int, out_ch: int, add_upsample: bool):
        super().__init__()
        n = cfg.layers_per_block + 1
        res = []
        for i in range(n):
            skip = in_ch if i == n - 1 else out_ch
            rin = prev_ch if i == 0 else out_ch
            res.append(ResnetBlock2D(rin + skip, out_ch, cfg.time_embed_dim, cfg.norm_num_groups))
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, h, skips, temb, context):
        for r in self.resnets:
            h = torch.cat([h, skips.pop()], dim=1)
            h = r(h, temb)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, cfg: UNetConfig, in_ch: int, prev_ch: int, out_ch: int, add_upsample: bool, heads: int,
                 tf_layers: int):
        super().__init__()
        n = cfg.layers_per_block + 1
        res, att = [], []
        for i in range(n):
            skip = in_ch if i == n - 1 else out_ch
            rin = prev_ch if i == 0 else out_ch
            res.append(ResnetBlock2D(rin + skip, out_ch, cfg.time_embed_dim, cfg.norm_num_groups))
            att.append(Transformer2DModel(heads, out_ch // heads, out_ch, tf_layers, cfg.cross_attention_dim,
                                          cfg.norm_num_groups, cfg.use_linear_projection))
        self.attentions = nn.ModuleList(att)
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None

    def forward(self, h, skips, temb, context):
        for r, a in zip(self.resnets, self.attentions):
            h = torch.cat([h, skips.pop()], dim=1)
            h = r(h, temb)
            h = a(h, context)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


# --------------------------------------------------------------------------------------------------------
# the UNet
# --------------------------------------------------------------------------------------------------------
class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionModel(nn.Module):
    """Attribute registration order follows diffusers (down_blocks, up_blocks, mid_block) so that
    `named_modules()` -- which the reference walks at lora.py:206 -- yields the same order."""

    def __init__(self, cfg: UNetConfig):
        super().__init__()
        self.cfg = cfg
        boc = cfg.block_out_channels
        ted = cfg.time_embed_dim
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        if cfg.addition_embed_type == "text_time":
            self.add_time_proj = Timesteps(cfg.addition_time_embed_dim)
            self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, ted)
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])

        out_ch = boc[0]
        for i, typ in enumerate(cfg.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            last = i == len(boc) - 1
            if typ == "CrossAttnDownBlock2D":
                blk = CrossAttnDownBlock2D(cfg, in_ch, out_ch, not last, cfg.num_attention_heads[i],
                                           cfg.transformer_layers_per_block[i])
            else:
                blk = DownBlock2D(cfg, in_ch, out_ch, not last)
            self.down_blocks.append(blk)

        mid_layers = cfg.mid_block_transformer_layers or cfg.transformer_layers_per_block[-1]
        self.mid_block = UNetMidBlock2DCrossAttn(cfg, boc[-1], cfg.num_attention_heads[-1], mid_layers)

        rev = list(reversed(boc))
        rev_heads = list(reversed(cfg.num_attention_heads))
        rev_layers = list(reversed(cfg.transformer_layers_per_block))
        out_ch = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev_ch, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            last = i == len(boc) - 1
            if typ == "CrossAttnUpBlock2D":
                blk = CrossAttnUpBlock2D(cfg, in_ch, prev_ch, out_ch, not last, rev_heads[i], rev_layers[i])
            else:
                blk = UpBlock2D(cfg, in_ch, prev_ch, out_ch, not last)
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self.storage_dtype: Optional[torch.dtype] = None

    # the reference calls these on the diffusers object (train_lora.py:67-70); harmless no-ops here
    def enable_xformers_memory_efficient_attention(self):
        return None

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None):
        _Q.dtype = self.storage_dtype
        try:
            return UNetOutput(self._forward(sample, timestep, encoder_hidden_states, added_cond_kwargs))
        finally:
            _Q.dtype = None

    def _forward(self, sample, timestep, context, added):
        N = sample.shape[0]
        sample = q(sample.float())
        context = q(context.float())
        t = torch.as_tensor(timestep, dtype=torch.float32, device=sample.device).reshape(-1)
        t = t.expand(N)
        emb = self.time_embedding(q(self.time_proj(t)))
        if self.cfg.addition_embed_type == "text_time":
            text_embeds = q(added["text_embeds"].float())
            time_ids = added["time_ids"].float()
            te = q(self.add_time_proj(time_ids.flatten())).reshape(N, -1)
            emb = q(emb + self.add_embedding(torch.cat([text_embeds, te], dim=-1)))
        temb = q(F.silu(emb))

        h = q(self.conv_in(sample))
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, temb, context)
            skips.extend(outs)
        h = self.mid_block(h, temb, context)
        for blk in self.up_blocks:
            h = blk(h, skips, temb, context)
        h = q(self.conv_act(self.conv_norm_out(h)))
        return self.conv_out(h)  # fp32 out (the engine's conv_out epilogue writes fp32)


# --------------------------------------------------------------------------------------------------------
# synthetic weights (SURVEY.md section 8(d) "Synthetic inputs")
# --------------------------------------------------------------------------------------------------------
@torch.no_grad()
def init_synthetic_(unet: UNet2DConditionModel, seed: int = 0) -> UNet2DConditionModel:
    """Seeded fan-in-scaled normal weights, small biases, norm gamma near 1: keeps activations O(1) through the
    whole UNet so fp16 storage neither overflows nor collapses.  Weight values are snapped to bf16-representable
    numbers so that an fp16/bf16/fp32 copy of the model holds *identical* weights (removes weight-rounding noise
    from parity comparisons)."""
    g = torch.Generator().manual_seed(seed)
    for name, p in unet.named_parameters():
        if p.ndim >= 2:
            fan_in = p[0].numel()
            w = torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(fan_in))
        elif name.endswith("weight"):  # norm gamma
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            w = 0.02 * torch.randn(p.shape, generator=g)
        p.copy_(w.to(torch.bfloat16).to(torch.float32))
    return unet
