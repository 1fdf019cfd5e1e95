"""prompts.yaml schema and the guidance loss of the reference (conceptmod/textsliders/prompt_util.py:63-201;
four-prompt upstream form trainscripts/imagesliders/prompt_util.py:86-148).

One loss serves both formulations: `negative` is optional and falls back to `unconditional`, which reproduces
the fork's five-prompt call (`negative_latents=`, train_lora_xl.py:334-339) and the upstream four-prompt call
(`unconditional_latents=`, train_lora.py:284-289) -- the stale 6-argument call in the fork's own train_lora.py
 works again through this."""
import copy
from typing import Literal, Optional, Union

import torch
import yaml
from pydantic import BaseModel, model_validator

ACTION_TYPES = Literal["erase", "enhance"]


class PromptEmbedsXL:
    def __init__(self, *args) -> None:
        self.text_embeds = args[0]
        self.pooled_embeds = args[1]


PROMPT_EMBEDDING = Union[torch.Tensor, PromptEmbedsXL]


class PromptEmbedsCache:
    def __init__(self):
        self.prompts = {}

    def __setitem__(self, name: str, value) -> None:
        self.prompts[name] = value

    def __getitem__(self, name: str):
        return self.prompts.get(name)


class PromptSettings(BaseModel):
    target: str
    positive: Optional[str] = None  # if None, target will be used
    negative: Optional[str] = None
    unconditional: str = ""
    neutral: Optional[str] = None  # if None, unconditional will be used
    action: ACTION_TYPES = "erase"
    guidance_scale: float = 1.0
    resolution: int = 512
    dynamic_resolution: bool = False
    batch_size: int = 1
    dynamic_crops: bool = False

    @model_validator(mode="before")
    @classmethod
    def fill_prompts(cls, values):
        keys = values.keys()
        if "target" not in keys:
            raise ValueError("target must be specified")
        if "positive" not in keys:
            values["positive"] = values["target"]
        if "unconditional" not in keys:
            values["unconditional"] = ""
        if "neutral" not in keys:
            values["neutral"] = values["unconditional"]
        return values


class PromptEmbedsPair:
    def __init__(self, loss_fn, target, positive, unconditional, neutral, negative=None, settings=None) -> None:
        if settings is None and isinstance(negative, PromptSettings):  # upstream 6-positional-argument form
            settings, negative = negative, None
        self.loss_fn = loss_fn
        self.target = target
        self.positive = positive
        self.negative = negative if negative is not None else unconditional
        self.unconditional = unconditional
        self.neutral = neutral
        self.guidance_scale = settings.guidance_scale
        self.resolution = settings.resolution
        self.dynamic_resolution = settings.dynamic_resolution
        self.batch_size = settings.batch_size
        self.dynamic_crops = settings.dynamic_crops
        self.action = settings.action

    def embeddings(self):
        return [self.target, self.positive, self.negative, self.unconditional, self.neutral]

    def _erase(self, target_latents, positive_latents, neutral_latents, negative_latents):
        return self.loss_fn(target_latents,
                            neutral_latents - self.guidance_scale * (positive_latents - negative_latents))

    def _enhance(self, target_latents, positive_latents, neutral_latents, negative_latents):
        return self.loss_fn(target_latents,
                            neutral_latents + self.guidance_scale * (positive_latents - negative_latents))

    def loss(self, **kwargs):
        if "unconditional_latents" in kwargs:  # upstream four-prompt keyword
            kwargs["negative_latents"] = kwargs.pop("unconditional_latents")
        if self.action == "erase":
            return self._erase(**kwargs)
        elif self.action == "enhance":
            return self._enhance(**kwargs)
        raise ValueError("action must be erase or enhance")


def load_prompts_from_yaml(path, attributes=[]):
    with open(path, "r") as f:
        prompts = yaml.safe_load(f)
    if prompts is None or len(prompts) == 0:
        raise ValueError("prompts file is empty")
    if len(attributes) != 0:
        newprompts = []
        for i in range(len(prompts)):
            for att in attributes:
                copy_ = copy.deepcopy(prompts[i])
                for key in ("target", "positive", "neutral", "negative", "unconditional"):
                    if key in copy_:  # the reference indexes 'negative' unconditionally and KeyErrors on its own
                        copy_[key] = att + " " + copy_[key]  # shipped prompts.yaml (prompt_util.py:191)
                newprompts.append(copy_)
    else:
        newprompts = copy.deepcopy(prompts)
    return [PromptSettings(**prompt) for prompt in newprompts]
