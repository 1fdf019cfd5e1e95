"""prompts.yaml schema and the guidance loss of the reference (conceptmod/textsliders/prompt_util.py:63-201;
four-prompt upstream form trainscripts/imagesliders/prompt_util.py:86-148).

One loss serves both formulations: `negative` is optional and falls back to `unconditional`, which reproduces
the fork's five-prompt call (`negative_latents=`, train_lora_xl.py:334-339) and the upstream four-prompt call
(`unconditional_latents=`, train_lora.py:284-289) -- the stale 6-argument call in the fork's own train_lora.py
 works again through this.

    loss = loss_fn(target, neutral + sign * guidance_scale * (positive - negative)),   sign = -1 erase, +1 enhance"""
from typing import Dict, Literal, Optional, Sequence, Union

import torch
import yaml
from pydantic import BaseModel, model_validator

ACTION_TYPES = Literal["erase", "enhance"]
_ACTION_SIGN = {"erase": -1.0, "enhance": +1.0}
# the prompt fields an `--attributes` prefix is applied to
_PROMPT_KEYS = ("target", "positive", "neutral", "negative", "unconditional")


class PromptEmbedsXL:
    """SD-XL conditioning of one prompt: (token embeddings of both encoders concatenated, pooled embedding)."""
    __slots__ = ("text_embeds", "pooled_embeds")

    def __init__(self, text_embeds, pooled_embeds) -> None:
        self.text_embeds, self.pooled_embeds = text_embeds, pooled_embeds


PROMPT_EMBEDDING = Union[torch.Tensor, PromptEmbedsXL]


class PromptEmbedsCache:
    """prompt string -> embedding; a miss reads as None (the trainers test `cache[p] is None` before encoding)."""

    def __init__(self):
        self.prompts: Dict[str, PROMPT_EMBEDDING] = {}

    def __getitem__(self, prompt: str):
        return self.prompts.get(prompt)

    def __setitem__(self, prompt: str, embedding) -> None:
        self.prompts[prompt] = embedding


class PromptSettings(BaseModel):
    """One entry of prompts.yaml.  Only `target` is mandatory: `positive` defaults to it, `neutral` to `unconditional`
    (itself "" by default), `negative` stays None (= use `unconditional` in the loss)."""
    target: str
    positive: Optional[str] = None
    negative: Optional[str] = None
    unconditional: str = ""
    neutral: Optional[str] = None
    action: ACTION_TYPES = "erase"
    guidance_scale: float = 1.0
    resolution: int = 512
    dynamic_resolution: bool = False
    batch_size: int = 1
    dynamic_crops: bool = False

    @model_validator(mode="before")
    @classmethod
    def _defaults_from_other_prompts(cls, raw):
        if "target" not in raw:
            raise ValueError("target must be specified")
        raw.setdefault("positive", raw["target"])
        raw.setdefault("unconditional", "")
        raw.setdefault("neutral", raw["unconditional"])
        return raw


class PromptEmbedsPair:
    """The embeddings of one prompt entry plus its loss.  Accepts the fork's 7-argument construction
    (..., neutral, negative, settings) and upstream's 6-argument one (..., neutral, settings)."""
    _COPIED = ("guidance_scale", "resolution", "dynamic_resolution", "batch_size", "dynamic_crops", "action")

    def __init__(self, loss_fn, target, positive, unconditional, neutral, negative=None, settings=None) -> None:
        if settings is None and isinstance(negative, PromptSettings):
            negative, settings = None, negative
        self.loss_fn = loss_fn
        self.target, self.positive, self.unconditional, self.neutral = target, positive, unconditional, neutral
        self.negative = unconditional if negative is None else negative
        for name in self._COPIED:
            setattr(self, name, getattr(settings, name))

    def embeddings(self):
        return [self.target, self.positive, self.negative, self.unconditional, self.neutral]

    def _guided(self, sign, target_latents, positive_latents, neutral_latents, negative_latents):
        return self.loss_fn(target_latents,
                            neutral_latents + (sign * self.guidance_scale) * (positive_latents - negative_latents))

    def _erase(self, **latents):
        return self._guided(_ACTION_SIGN["erase"], **latents)

    def _enhance(self, **latents):
        return self._guided(_ACTION_SIGN["enhance"], **latents)

    def loss(self, **latents):
        """Keywords: target_latents, positive_latents, neutral_latents and negative_latents (or upstream's
        unconditional_latents)."""
        if "unconditional_latents" in latents:
            latents["negative_latents"] = latents.pop("unconditional_latents")
        if self.action not in _ACTION_SIGN:
            raise ValueError("action must be erase or enhance")
        return self._guided(_ACTION_SIGN[self.action], **latents)


def load_prompts_from_yaml(path, attributes: Sequence[str] = ()):
    """Reads prompts.yaml; with attributes, every entry is repeated once per attribute with the attribute prefixed to
    each prompt it has (the reference prefixes `negative` unconditionally and KeyErrors on its own shipped
    prompts.yaml, prompt_util.py:191 -- a missing key is simply skipped here)."""
    from .config_util import resolve_data_path
    with open(resolve_data_path(path), "r") as fh:
        entries = yaml.safe_load(fh)
    if not entries:
        raise ValueError("prompts file is empty")
    if attributes:
        entries = [{k: (f"{att} {v}" if k in _PROMPT_KEYS else v) for k, v in entry.items()}
                   for entry in entries for att in attributes]
    return [PromptSettings(**dict(entry)) for entry in entries]
