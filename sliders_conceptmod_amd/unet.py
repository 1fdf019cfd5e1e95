"""`UNet2DConditionModel` drop-in for the slider trainer: same constructor-free surface the reference uses on the
diffusers object -- `unet(sample, timestep, encoder_hidden_states=..., added_cond_kwargs=...).sample`,
`.to(device, dtype)`, `.eval()`, `.requires_grad_(False)`, `.named_modules()`,
`.enable_xformers_memory_efficient_attention()` (reference: conceptmod/textsliders/train_lora.py:67-70,
train_util.py:290-294, 471-476) -- but the module tree is only a *parameter container*: class names and attribute
names mirror diffusers so that name-based LoRA injection (lora.py:194-251) selects the same children and
diffusers-format state dicts load by key, while all arithmetic runs in the HIP engine (csrc/engine.hip).
There is deliberately no PyTorch implementation of the forward here: without the HIP library the call raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _native


@dataclass
class UNetConfig:
    """Public config values of the diffusers UNet2DConditionModel."""
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",)
    up_block_types: Tuple[str, ...] = ("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3
    layers_per_block: int = 2
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    num_attention_heads: Tuple[int, ...] = (8, 8, 8, 8)
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    use_linear_projection: bool = False
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    mid_block_transformer_layers: Optional[int] = None

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


def sd1x_config() -> UNetConfig:
    return UNetConfig()


def sdxl_config() -> UNetConfig:
    return UNetConfig(
        block_out_channels=(320, 640, 1280),
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        transformer_layers_per_block=(1, 2, 10), num_attention_heads=(5, 10, 20), cross_attention_dim=2048,
        use_linear_projection=True, addition_embed_type="text_time")


# ---- parameter containers (names and class names follow diffusers) ------------------------------------------------
class Timesteps(nn.Module):
    def __init__(self, num_channels):
        super().__init__()
        self.num_channels = num_channels


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=1e-5)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=1e-5)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None


class Downsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(ctx, inner, bias=False)
        self.to_v = nn.Linear(ctx, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, num_layers, cross_attention_dim, groups, linear):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner) if linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner, in_channels) if linear else nn.Conv2d(inner, in_channels, 1)


def _resnets(cfg, ins, out):
    return nn.ModuleList([ResnetBlock2D(i, out, cfg.time_embed_dim, cfg.norm_num_groups) for i in ins])


def _attns(cfg, n, ch, heads, layers):
    return nn.ModuleList([Transformer2DModel(heads, ch // heads, ch, layers, cfg.cross_attention_dim,
                                             cfg.norm_num_groups, cfg.use_linear_projection) for _ in range(n)])


class DownBlock2D(nn.Module):
    def __init__(self, cfg, in_ch, out_ch, add_downsample):
        super().__init__()
        self.resnets = _resnets(cfg, [in_ch] + [out_ch] * (cfg.layers_per_block - 1), out_ch)
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, cfg, in_ch, out_ch, add_downsample, heads, layers):
        super().__init__()
        self.attentions = _attns(cfg, cfg.layers_per_block, out_ch, heads, layers)
        self.resnets = _resnets(cfg, [in_ch] + [out_ch] * (cfg.layers_per_block - 1), out_ch)
        self.downsamplers = nn.ModuleList([Downsample2D(out_ch)]) if add_downsample else None


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, cfg, ch, heads, layers):
        super().__init__()
        self.attentions = _attns(cfg, 1, ch, heads, layers)
        self.resnets = _resnets(cfg, [ch, ch], ch)


def _up_ins(cfg, in_ch, prev_ch, out_ch):
    n = cfg.layers_per_block + 1
    return [(prev_ch if i == 0 else out_ch) + (in_ch if i == n - 1 else out_ch) for i in range(n)]


class UpBlock2D(nn.Module):
    def __init__(self, cfg, in_ch, prev_ch, out_ch, add_upsample):
        super().__init__()
        self.resnets = _resnets(cfg, _up_ins(cfg, in_ch, prev_ch, out_ch), out_ch)
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, cfg, in_ch, prev_ch, out_ch, add_upsample, heads, layers):
        super().__init__()
        self.attentions = _attns(cfg, cfg.layers_per_block + 1, out_ch, heads, layers)
        self.resnets = _resnets(cfg, _up_ins(cfg, in_ch, prev_ch, out_ch), out_ch)
        self.upsamplers = nn.ModuleList([Upsample2D(out_ch)]) if add_upsample else None


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class _UNetFn(torch.autograd.Function):
    """Autograd bookkeeping only: forward and backward are single calls into the HIP engine."""

    @staticmethod
    def forward(ctx, save, engine, sample, timestep, ehs, text_embeds, time_ids, flat, n_down, multiplier):
        # `save` is decided by the caller: grad mode is always off inside Function.forward
        down = up = None
        if flat is not None:
            down, up = flat[:n_down], flat[n_down:]
        eps = engine.forward(sample, timestep, ehs, text_embeds, time_ids, down, up, multiplier, save)
        ctx.engine = engine if save else None
        ctx.gen = engine.tape_generation() if save else 0
        ctx.n_down = n_down
        ctx.keep = (sample, ehs, text_embeds, time_ids, flat)  # borrowed by the engine until backward
        return eps

    @staticmethod
    def backward(ctx, d_eps):
        if ctx.engine is None:
            return (None,) * 10
        flat = ctx.keep[4]
        # the engine holds ONE tape: a later saved forward, a change of shape (replan) or an earlier backward drops it
        if ctx.engine.tape_generation() != ctx.gen:
            raise _native.SmiError(
                "backward through a UNet output whose saved activations are gone: the engine keeps the tape of the LAST "
                "grad-enabled forward only (a later grad-enabled forward, a call at another resolution / batch, or a "
                "previous backward released it). Call backward before the next grad-enabled UNet call.")
        g = torch.zeros_like(flat)
        ctx.engine.backward(d_eps.contiguous().float(), g[:ctx.n_down], g[ctx.n_down:])
        return None, None, None, None, None, None, None, g, None, None


class UNet2DConditionModel(nn.Module):
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
                self.down_blocks.append(CrossAttnDownBlock2D(cfg, in_ch, out_ch, not last, cfg.num_attention_heads[i],
                                                             cfg.transformer_layers_per_block[i]))
            else:
                self.down_blocks.append(DownBlock2D(cfg, in_ch, out_ch, not last))
        mid_layers = cfg.mid_block_transformer_layers or cfg.transformer_layers_per_block[-1]
        self.mid_block = UNetMidBlock2DCrossAttn(cfg, boc[-1], cfg.num_attention_heads[-1], mid_layers)
        rev, rh, rl = list(reversed(boc)), list(reversed(cfg.num_attention_heads)), list(
            reversed(cfg.transformer_layers_per_block))
        out_ch = rev[0]
        for i, typ in enumerate(cfg.up_block_types):
            prev_ch, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            last = i == len(boc) - 1
            if typ == "CrossAttnUpBlock2D":
                self.up_blocks.append(CrossAttnUpBlock2D(cfg, in_ch, prev_ch, out_ch, not last, rh[i], rl[i]))
            else:
                self.up_blocks.append(UpBlock2D(cfg, in_ch, prev_ch, out_ch, not last))
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, boc[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self._engine: Optional[_native.Engine] = None
        self._engine_key = None
        self.__dict__["_lora_network"] = None  # set by LoRANetwork (sliders_conceptmod_amd.lora); not a child module

    # ---- reference call surface -------------------------------------------------------------------------------
    def enable_xformers_memory_efficient_attention(self):
        return None  # attention is always the fused HIP kernel

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _ensure_engine(self, n, h, w, ctx_len, n_adapted=None):
        """Engine for UNet batch n of which n_adapted (default n) samples may be LoRA-adapted / differentiated."""
        n_adapted = n if n_adapted is None else n_adapted
        net = self._lora_network
        sites = net.engine_sites() if net is not None else []
        # packed weights depend on dtype / device / adapted sites only; shapes are plans of the same engine
        key = (self.dtype, str(self.device), tuple((s["target"], s["off_down"], s["off_up"], s["rank"], s["scale"],
                                                    s.get("off_dora", -1)) for s in sites))  # LoRA vs DoRA differ here
        if self._engine is not None and self._engine_key == key:
            e = self._engine
            if (h, w, ctx_len) == (e.h, e.w, e.ctx_len) and n <= e.batch and n_adapted <= e.batch_adapted:
                return e
            if (h, w, ctx_len) == (e.h, e.w, e.ctx_len):  # same latents, larger batch: grow, never shrink
                n, n_adapted = max(n, e.batch), max(n_adapted, e.batch_adapted)
            e.plan(n, min(n, n_adapted), h, w, ctx_len)
            return e
        if self.device.type != "cuda":
            raise _native.SmiError("the UNet runs only on an MI355X through the HIP engine; move it to a cuda "
                                   "device with unet.to(device, dtype) (there is no CPU fallback)")
        if self.dtype not in _native.DTYPE_CODE:
            raise _native.SmiError(f"engine dtypes are float16/bfloat16, got {self.dtype}")
        if self._engine is not None:  # other dtype / device / adaptor set: the packed weights change
            self._engine.close()
            self._engine = None
        state = {k: v.detach() for k, v in self.state_dict().items()}
        self._engine = _native.Engine(self.cfg, self.dtype, state, sites, n, h, w, ctx_len, self.device,
                                      batch_adapted=min(n, n_adapted))
        self._engine_key = key
        return self._engine

    def forward(self, sample, timestep, encoder_hidden_states, added_cond_kwargs=None, **_):
        n, _c, h, w = sample.shape
        ehs = encoder_hidden_states.to(self.dtype).contiguous()
        eng = self._ensure_engine(n, h, w, ehs.shape[1])
        text_embeds = time_ids = None
        if self.cfg.addition_embed_type == "text_time":
            text_embeds = added_cond_kwargs["text_embeds"].to(self.dtype).contiguous()
            time_ids = added_cond_kwargs["time_ids"].float().contiguous()
        t = float(timestep)
        net = self._lora_network
        flat, n_down, mult = None, 0, 0.0
        if net is not None:
            flat, n_down, mult = net.engine_params()
        save = bool(torch.is_grad_enabled() and mult != 0 and flat is not None and flat.requires_grad)
        eps = _UNetFn.apply(save, eng, sample.float().contiguous(), t, ehs, text_embeds, time_ids, flat, n_down,
                            float(mult))
        return UNetOutput(eps)


@torch.no_grad()
def init_synthetic_(unet: nn.Module, seed: int = 0):
    """Same seeded synthetic weights as oracle/unet_ref.py:init_synthetic_ (kept textually independent of oracle/)."""
    import math
    g = torch.Generator().manual_seed(seed)
    for name, p in unet.named_parameters():
        if p.ndim >= 2:
            w = torch.randn(p.shape, generator=g) * (0.7 / math.sqrt(p[0].numel()))
        elif name.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        else:
            w = 0.02 * torch.randn(p.shape, generator=g)
        p.copy_(w.to(torch.bfloat16).to(p.dtype))
    return unet
