"""CLIP text encoders for the prompt front end -- drop-ins for `transformers.CLIPTextModel` (SD-1.x; SD-XL encoder 1)
and `CLIPTextModelWithProjection` (SD-XL encoder 2) as the reference calls them:
    text_encoder(tokens)[0]                                             (conceptmod/textsliders/train_util.py:119-120)
    out = text_encoder(tokens, output_hidden_states=True); out[0]; out.hidden_states[-2]          (train_util.py:139-144)
They are parameter containers with the transformers parameter names (a transformers state dict loads by key); the
arithmetic runs in the HIP engine (csrc/engine.hip `forward_clip`: fused causal attention, LayerNorm, GEMM kernels).  No
PyTorch forward: without the HIP library, or on a CPU device, a call raises.  Tokenisation stays host text processing
(`transformers.CLIPTokenizer`)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _native


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    projection_dim: Optional[int] = None  # set for CLIPTextModelWithProjection
    eos_token_id: int = 49407


def clip_l_config() -> CLIPTextConfig:            # openai/clip-vit-large-patch14 text tower (SD-1.x, SD-XL encoder 1)
    return CLIPTextConfig()


def open_clip_bigg_config() -> CLIPTextConfig:    # laion CLIP-ViT-bigG-14 text tower (SD-XL encoder 2)
    return CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                          hidden_act="gelu", projection_dim=1280)


class _Attn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = (nn.Linear(d, d) for _ in range(4))


class _MLP(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d, inter), nn.Linear(inter, d)


class _Layer(nn.Module):
    def __init__(self, d, inter):
        super().__init__()
        self.self_attn = _Attn(d)
        self.layer_norm1 = nn.LayerNorm(d)
        self.mlp = _MLP(d, inter)
        self.layer_norm2 = nn.LayerNorm(d)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg.hidden_size, cfg.intermediate_size) for _ in range(cfg.num_hidden_layers)])


class _TextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size)


class _HiddenStates:
    """What the reference reads of `hidden_states`: [-1] (last layer, before the final norm is NOT kept by the engine)
    is not needed; [-2] is the penultimate layer's output (train_util.py:142)."""

    def __init__(self, penultimate):
        self._pen = penultimate

    def __getitem__(self, i):
        if i != -2:
            raise IndexError("the engine returns hidden_states[-2] only (what text_encode_xl reads)")
        return self._pen


class CLIPTextOutput:
    def __init__(self, first, last_hidden_state, pooler_output, text_embeds, hidden_states):
        self._first = first
        self.last_hidden_state, self.pooler_output, self.text_embeds = last_hidden_state, pooler_output, text_embeds
        self.hidden_states = hidden_states

    def __getitem__(self, i):
        if i == 0:
            return self._first
        raise IndexError(i)


class CLIPTextModel(nn.Module):
    with_projection = False

    def __init__(self, cfg: CLIPTextConfig):
        super().__init__()
        self.config = cfg
        self.text_model = _TextTransformer(cfg)
        if self.with_projection:
            self.text_projection = nn.Linear(cfg.hidden_size, cfg.projection_dim, bias=False)
        self._engine = None

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}  # a buffer in older checkpoints
        return super().load_state_dict(sd, strict=strict)

    def _eng(self, n):
        if self.device.type != "cuda":
            raise _native.SmiError("the CLIP text encoder runs only on an MI355X through the HIP engine; move it to a "
                                   "cuda device with text_encoder.to(device, dtype) (there is no CPU fallback)")
        if self.dtype not in _native.DTYPE_CODE:
            raise _native.SmiError(f"engine dtypes are float16/bfloat16, got {self.dtype}")
        if self._engine is None or self._engine.batch < n or self._engine.dtype != self.dtype:
            if self._engine is not None:
                self._engine.close()
            cfg = self.config
            if not self.with_projection:
                cfg = CLIPTextConfig(**{**cfg.__dict__, "projection_dim": None})
            state = {k: v.detach() for k, v in self.state_dict().items()}
            self._engine = _native.ClipEngine(cfg, self.dtype, state, max(n, 1), self.device)
        return self._engine

    @torch.no_grad()
    def forward(self, input_ids, output_hidden_states: bool = False, **_):
        ids = input_ids.to(torch.int64)
        eos = self.config.eos_token_id
        # transformers: the pooled token is the EOS token -- the highest id for the original vocabulary (eos_token_id 2 in
        # old configs), else the first position holding eos_token_id
        eos_pos = ids.argmax(dim=-1) if eos == 2 else (ids == eos).int().argmax(dim=-1)
        last, pen, pooled = self._eng(ids.shape[0]).encode(ids, eos_pos)
        hs = _HiddenStates(pen) if output_hidden_states else None
        if self.with_projection:
            return CLIPTextOutput(pooled, last, None, pooled, hs)
        return CLIPTextOutput(last, last, pooled, None, hs)


class CLIPTextModelWithProjection(CLIPTextModel):
    with_projection = True
