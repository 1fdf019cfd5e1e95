"""The reference's YAML configuration surface (conceptmod/textsliders/config_util.py:14-107; the image-slider copy,
trainscripts/imagesliders/config_util.py:14-104, differs only in the `train.lr` default, 1e-4).

The schema is data: one table of sections -> fields -> (type, default), turned into pydantic models of the reference's
names, so a config written for the reference validates here field for field (and a typo'd key or a wrong type fails
with the same pydantic ValidationError)."""
from typing import Literal, Optional

import torch
import yaml
from pydantic import BaseModel, create_model

from .lora import TRAINING_METHODS

_DTYPES = {torch.float32: ("fp32", "float32"), torch.float16: ("fp16", "float16"), torch.bfloat16: ("bf16", "bfloat16")}
PRECISION_TYPES = Literal["fp32", "fp16", "bf16", "float32", "float16", "bfloat16"]
NETWORK_TYPES = Literal["lierla", "c3lier"]
_SCHEDULERS = Literal["ddim", "ddpm", "lms", "euler_a", "builtin"]

_REQUIRED = ...
# section -> {field: (annotation, default)}; the YAML key of a section is its lower-cased name without "Config"
_SECTIONS = {
    "PretrainedModelConfig": dict(name_or_path=(str, _REQUIRED), v2=(bool, False), v_pred=(bool, False),
                                  clip_skip=(Optional[int], None)),
    "NetworkConfig": dict(type=(NETWORK_TYPES, "lierla"), rank=(int, 4), alpha=(float, 1.0),
                          training_method=(TRAINING_METHODS, "full")),
    "TrainConfig": dict(precision=(PRECISION_TYPES, "bfloat16"), noise_scheduler=(_SCHEDULERS, "ddim"),
                        iterations=(int, 500), lr=(float, 1e-3), eta_min=(float, 1e-4),
                        lambda_similarity=(float, 0.1), optimizer=(str, "adamw"), optimizer_args=(str, ""),
                        lr_scheduler=(str, "constant"), cfg=(float, 1.0), max_denoising_steps=(int, 50)),
    "SaveConfig": dict(name=(str, "untitled"), path=(str, "./output"), per_steps=(int, 200),
                       precision=(PRECISION_TYPES, "float32")),
    "LoggingConfig": dict(use_wandb=(bool, False), verbose=(bool, False)),
    "OtherConfig": dict(use_xformers=(bool, False)),
}


def _section(name: str):
    model = create_model(name, __base__=BaseModel, **_SECTIONS[name])
    model.__module__ = __name__
    return model


PretrainedModelConfig = _section("PretrainedModelConfig")
NetworkConfig = _section("NetworkConfig")
TrainConfig = _section("TrainConfig")
SaveConfig = _section("SaveConfig")
LoggingConfig = _section("LoggingConfig")
OtherConfig = _section("OtherConfig")

# sections a config file may leave out (or set to null); they are filled with their defaults after validation
_OPTIONAL = {"train": TrainConfig, "save": SaveConfig, "logging": LoggingConfig, "other": OtherConfig}

RootConfig = create_model(
    "RootConfig", __base__=BaseModel, prompts_file=(str, _REQUIRED),
    pretrained_model=(PretrainedModelConfig, _REQUIRED), network=(NetworkConfig, _REQUIRED),
    **{key: (Optional[cls], None) for key, cls in _OPTIONAL.items()})
RootConfig.__module__ = __name__


def parse_precision(precision: str) -> torch.dtype:
    for dtype, names in _DTYPES.items():
        if precision in names:
            return dtype
    raise ValueError(f"Invalid precision type: {precision}")


def resolve_data_path(path: str) -> str:
    """`data/...` paths (the CLI defaults and the shipped configs' prompts_file) are looked up in the working directory
    first, as the reference does, then next to this package (sliders_conceptmod_amd/data/)."""
    import os
    if os.path.exists(path):
        return path
    alt = os.path.join(os.path.dirname(os.path.abspath(__file__)), path)
    return alt if os.path.exists(alt) else path


def load_config_from_yaml(config_path: str):
    config_path = resolve_data_path(config_path)
    with open(config_path, "r") as fh:
        raw = yaml.safe_load(fh)  # plain data only: nothing in the file is executed
    root = RootConfig(**raw)
    for key, cls in _OPTIONAL.items():
        if getattr(root, key) is None:
            setattr(root, key, cls())
    return root
