"""LoRA adaptor with the reference's API (conceptmod/textsliders/lora.py:76-301; image-slider variant
trainscripts/imagesliders/lora.py:49-256) on top of the HIP engine.

Same surface: `LoRANetwork(unet, rank, multiplier, delimiter, alpha, target_replace, prefix, train_method)`,
`.unet_loras` (each with `.lora_name/.lora_down/.lora_up/.multiplier/.scale/.alpha`), `.prepare_optimizer_params()`,
`.save_weights(file, dtype, metadata)`, `.get_state_dict(dtype)`, `.set_lora_slider(scale)`, context-manager
semantics (inside `with network:` multiplier = 1.0 * lora_scale, after exit 0: lora.py:295-301), `state_dict()` /
`load_state_dict()` with keys `<lora_name>.alpha`, `<lora_name>.lora_down.weight [r, in]`,
`<lora_name>.lora_up.weight [out, r]` (strict-loadable by the reference's eval scripts,
eval-scripts/generate_images_sd1.py:98-106).

Different underneath: no forward monkey-patching.  All parameters live in ONE flat fp32 leaf tensor
`network.flat = [all lora_down matrices back to back | all lora_up matrices]`; the engine reads them by offset, fuses
the rank-r delta into the GEMM epilogues, and returns one flat gradient (so the optimiser step, the global-norm clip
and the data-parallel all-reduce each touch a single buffer).  `module.lora_down.weight` is a view into it.
Master weights stay fp32 whatever dtype `.to()` asks for (the reference keeps them in the train dtype); files are
still written in the requested dtype.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from typing import List, Literal, Optional

import torch
import torch.nn as nn

UNET_TARGET_REPLACE_MODULE_TRANSFORMER = ["Attention"]
UNET_TARGET_REPLACE_MODULE_CONV = ["ResnetBlock2D", "Downsample2D", "Upsample2D", "DownBlock2D", "UpBlock2D"]
LORA_PREFIX_UNET = "lora_unet"
DEFAULT_TARGET_REPLACE = UNET_TARGET_REPLACE_MODULE_TRANSFORMER

TRAINING_METHODS = Literal["noxattn", "innoxattn", "selfattn", "xattn", "full", "xattn-strict", "noxattn-hspace",
                           "noxattn-hspace-last"]


class _WeightView:
    """Stand-in for the nn.Linear / nn.Conv2d the reference keeps in `lora_down` / `lora_up`: `.weight` is a live view
    ([out, in] for Linear, [out, in, kh, kw] for Conv2d -- the shapes the reference's state_dict holds)."""

    def __init__(self, owner: "LoRANetwork", which: str, offset: int, shape):
        self._owner, self._which, self._offset, self._shape = owner, which, offset, tuple(shape)
        self.in_features, self.out_features = shape[1], shape[0]
        self._numel = math.prod(self._shape)

    @property
    def weight(self) -> torch.Tensor:
        flat = self._owner.flat_down if self._which == "down" else self._owner.flat_up
        return flat[self._offset:self._offset + self._numel].view(self._shape)

    @property
    def grad(self) -> Optional[torch.Tensor]:
        g = self._owner.flat.grad
        if g is None:
            return None
        base = self._offset + (0 if self._which == "down" else self._owner._n_down)
        return g[base:base + self._numel].view(self._shape)


def lora_shapes(org_module: nn.Module, lora_dim: int):
    """(rank actually used, lora_down.weight shape, lora_up.weight shape) -- lora.py:94-114: Linear [r, in] / [out, r];
    Conv2d: k x k down with the layer's stride / padding, 1 x 1 up, rank clamped to min(rank, in, out)."""
    if isinstance(org_module, nn.Linear):
        return lora_dim, (lora_dim, org_module.in_features), (org_module.out_features, lora_dim)
    if isinstance(org_module, nn.Conv2d):
        r = min(lora_dim, org_module.in_channels, org_module.out_channels)
        kh, kw = org_module.kernel_size
        return r, (r, org_module.in_channels, kh, kw), (org_module.out_channels, r, 1, 1)
    raise NotImplementedError(f"LoRA on {org_module.__class__.__name__}")


class LoRAModule:
    """One adapted Linear or Conv2d: y = org(x) + up(down(x)) * multiplier * (alpha / rank)   (lora.py:76-138)."""

    def __init__(self, owner, lora_name, target_path, org_module, multiplier, lora_dim, alpha, off_down, off_up):
        self.lora_name = lora_name
        self.target_path = target_path
        self.lora_dim, down_shape, up_shape = lora_shapes(org_module, lora_dim)
        alpha = lora_dim if alpha is None or alpha == 0 else alpha  # sic: the UNCLAMPED rank (lora.py:118)
        self.scale = alpha / self.lora_dim
        self.alpha = torch.tensor(alpha)
        self.multiplier = multiplier
        self.is_conv = isinstance(org_module, nn.Conv2d)
        self.in_dim, self.out_dim = down_shape[1], up_shape[0]
        self.off_down, self.off_up = off_down, off_up
        self.lora_down = _WeightView(owner, "down", off_down, down_shape)
        self.lora_up = _WeightView(owner, "up", off_up, up_shape)

    def parameters(self):
        return [self.lora_down.weight, self.lora_up.weight]


def select_targets(root_module: nn.Module, train_method: str, target_replace_modules, prefix: str, delimiter: str,
                   with_duplicates: bool = False):
    """The name / class-name walk of lora.py:194-251, returning (lora_name, dotted path, child module).

    With the conv classes in `target_replace_modules` (c3lier) a resnet conv is reached twice -- through its
    DownBlock2D / UpBlock2D and again through its ResnetBlock2D; the reference BUILDS a LoRAModule both times (drawing
    its init from the global RNG) and keeps the first (lora.py:243-249).  `with_duplicates` returns those second visits
    too, as (lora_name, path, child, True), so the init can consume the RNG identically."""
    out, names = [], []
    for name, module in root_module.named_modules():
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
        if module.__class__.__name__ in target_replace_modules:
            for child_name, child_module in module.named_modules():
                if "add_" in child_name:
                    continue
                if child_module.__class__.__name__ in ["Linear", "Conv2d", "LoRACompatibleLinear",
                                                       "LoRACompatibleConv"]:
                    if train_method == "xattn-strict" and "out" in child_name:
                        continue
                    if train_method == "noxattn-hspace" and "mid_block" not in name:
                        continue
                    if train_method == "noxattn-hspace-last" and (
                            "mid_block" not in name or ".1" not in name or "conv2" not in child_name):
                        continue
                    lora_name = (prefix + "." + name + "." + child_name).replace(".", delimiter)
                    if lora_name not in names:
                        names.append(lora_name)
                        out.append((lora_name, name + "." + child_name, child_module) +
                                   ((False,) if with_duplicates else ()))
                    elif with_duplicates:
                        out.append((lora_name, name + "." + child_name, child_module, True))
    return out


class LoRANetwork(nn.Module):
    def __init__(self, unet, rank: int = 4, multiplier: float = 1.0, delimiter: str = "_", alpha: float = 1.0,
                 target_replace=DEFAULT_TARGET_REPLACE, prefix=LORA_PREFIX_UNET,
                 train_method: TRAINING_METHODS = "full") -> None:
        super().__init__()
        self.lora_scale = 1
        self.multiplier = multiplier
        self.lora_dim = rank
        self.alpha = alpha
        visits = select_targets(unet, train_method, target_replace, prefix, delimiter, with_duplicates=True)
        targets = [v[:3] for v in visits if not v[3]]
        n_down = sum(math.prod(lora_shapes(c, rank)[1]) for _, _, c in targets)
        n_up = sum(math.prod(lora_shapes(c, rank)[2]) for _, _, c in targets)
        self._n_down, self._n_up = n_down, n_up
        self.flat = nn.Parameter(torch.zeros(max(n_down + n_up, 1), dtype=torch.float32))
        self.unet_loras: List[LoRAModule] = []
        od = ou = 0
        by_name = {}
        for lora_name, path, child in targets:
            m = LoRAModule(self, lora_name, path, child, multiplier, rank, alpha, od, ou)
            self.unet_loras.append(m)
            by_name[lora_name] = m
            od += m.lora_down._numel
            ou += m.lora_up._numel
        lora_names = set()
        for lora in self.unet_loras:
            assert lora.lora_name not in lora_names, f"duplicated lora name: {lora.lora_name}. {lora_names}"
            lora_names.add(lora.lora_name)
        with torch.no_grad():  # lora.py:123-124: kaiming_uniform_(down, a=1); zeros_(up) -- same RNG draw order
            for lora_name, _path, child, dup in visits:
                # the reference builds nn.Linear / nn.Conv2d layers first (lora.py:97-98, 109-112); their default inits
                # consume the global RNG, so draw-and-discard the same amounts to stay seed-compatible -- also for the
                # modules it builds and then drops as duplicates (lora.py:243-249)
                _r, dshape, ushape = lora_shapes(child, rank)
                torch.empty(dshape).uniform_()
                torch.empty(ushape).uniform_()
                if dup:
                    nn.init.kaiming_uniform_(torch.empty(dshape), a=1)
                else:
                    nn.init.kaiming_uniform_(by_name[lora_name].lora_down.weight, a=1)
        # "apply_to": register with the engine-backed UNet instead of patching module forwards (lora.py:129-132)
        unet.__dict__["_lora_network"] = self  # plain attribute: must not become a registered child module

    @property
    def flat_down(self) -> torch.Tensor:
        return self.flat[:self._n_down]

    @property
    def flat_up(self) -> torch.Tensor:
        return self.flat[self._n_down:self._n_down + self._n_up]

    # ---- engine side ---------------------------------------------------------------------------------------------
    def engine_sites(self):
        # the engine tells Linear from Conv2d targets by the module path (conv1 / conv2 / *samplers.0.conv)
        return [{"target": l.target_path, "off_down": l.off_down, "off_up": l.off_up, "rank": l.lora_dim,
                 "scale": float(l.scale)} for l in self.unet_loras]

    def engine_params(self):
        """(flat parameter leaf, number of lora_down elements, current multiplier)"""
        mult = self.unet_loras[0].multiplier if self.unet_loras else 0.0
        return self.flat, self._n_down, float(mult)

    def _apply(self, fn, *a, **kw):
        super()._apply(fn, *a, **kw)
        if self.flat.dtype != torch.float32:  # fp32 master weights regardless of the requested dtype
            self.flat.data = self.flat.data.float()
        return self

    # ---- reference API ---------------------------------------------------------------------------------------------
    def prepare_optimizer_params(self):
        return [{"params": [self.flat]}] if self.unet_loras else []

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        for l in self.unet_loras:
            sd[f"{prefix}{l.lora_name}.alpha"] = l.alpha.clone()
            sd[f"{prefix}{l.lora_name}.lora_down.weight"] = l.lora_down.weight.detach()
            sd[f"{prefix}{l.lora_name}.lora_up.weight"] = l.lora_up.weight.detach()
        return sd

    def load_state_dict(self, state_dict, strict: bool = True):
        want = set(self.state_dict().keys())
        have = set(state_dict.keys())
        if strict and want != have:
            raise RuntimeError(f"Error(s) in loading state_dict for LoRANetwork: missing {sorted(want - have)[:4]}, "
                               f"unexpected {sorted(have - want)[:4]}")
        with torch.no_grad():
            for l in self.unet_loras:
                for part, view in (("lora_down", l.lora_down), ("lora_up", l.lora_up)):
                    k = f"{l.lora_name}.{part}.weight"
                    if k in state_dict:
                        view.weight.copy_(state_dict[k].to(view.weight.device, torch.float32))

    def get_state_dict(self, dtype=None):
        state_dict = self.state_dict()
        if dtype is not None:
            for key in list(state_dict.keys()):
                state_dict[key] = state_dict[key].detach().clone().to("cpu").to(dtype)
        return state_dict

    def save_weights(self, file, dtype=None, metadata: Optional[dict] = None):
        state_dict = self.state_dict()
        for key in list(state_dict.keys()):
            v = state_dict[key].detach().clone().to("cpu")
            state_dict[key] = v.to(dtype) if dtype is not None else v
        if os.path.splitext(str(file))[1] == ".safetensors":
            from safetensors.torch import save_file
            save_file(state_dict, str(file), metadata)
        else:
            torch.save(state_dict, file)

    def set_lora_slider(self, scale):
        self.lora_scale = scale

    def __enter__(self):
        for lora in self.unet_loras:
            lora.multiplier = 1.0 * self.lora_scale

    def __exit__(self, exc_type, exc_value, tb):
        for lora in self.unet_loras:
            lora.multiplier = 0
