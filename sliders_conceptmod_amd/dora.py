"""DoRA adaptor (`--peft_type dora`) with the reference's API (conceptmod/textsliders/dora.py:53-330) on the HIP engine.

    dW = (W + up down) * (dora_scale / ||W + up down||_col) - W      (column norm over the output dim, detached)
    y  = org(x) + F.linear(x, dW) * scale * multiplier                                        (dora.py:124-162)

Same surface as `DoRANetwork(unet, rank, multiplier, delimiter, alpha, prefix, target_replace, train_method)`:
`.unet_loras` (modules with `.lora_name / .lora_down / .lora_up / .dora_scale / .multiplier / .scale / .alpha`),
`.prepare_optimizer_params()`, `.save_weights()`, `.get_state_dict()`, context-manager semantics (multiplier 1.0 inside
`with network:`, 0 outside; no `lora_scale`, dora.py:322-328), state-dict keys per module in the reference's order
`<name>.dora_scale [1, in]`, `<name>.alpha`, `<name>.lora_down.weight`, `<name>.lora_up.weight`.

As in lora.py all parameters live in ONE flat fp32 leaf: `flat = [all lora_down | all lora_up | all dora_scale]`; the
engine reads them by offset.  Linear targets only (what train_lora_xl.py adapts: target_replace=["Attention"]).

Reference quirk kept: DoRANetwork passes its `alpha` into DoRAModule's `prefix` slot (dora.py:247-249 vs :57-65), so every
module runs with the default alpha = 1, i.e. scale = 1 / rank."""
from __future__ import annotations

from collections import OrderedDict
from typing import List

import torch
import torch.nn as nn

from .lora import (DEFAULT_TARGET_REPLACE, LORA_PREFIX_UNET, LoRANetwork, _WeightView, select_targets)


class _ScaleView:
    """`module.dora_scale`: a live [1, in] view into the flat leaf (tensor-like through `.data` / indexing via `()`)."""

    def __init__(self, owner, offset, n):
        self._owner, self._offset, self._n = owner, offset, n

    def __call__(self) -> torch.Tensor:
        o = self._owner
        base = o._n_down + o._n_up_only + self._offset
        return o.flat[base:base + self._n].view(1, self._n)

    @property
    def grad(self):
        g = self._owner.flat.grad
        if g is None:
            return None
        o = self._owner
        base = o._n_down + o._n_up_only + self._offset
        return g[base:base + self._n].view(1, self._n)


class DoRAModule:
    def __init__(self, owner, lora_name, target_path, org_module, multiplier, lora_dim, off_down, off_up, off_scale):
        if not isinstance(org_module, nn.Linear):
            raise NotImplementedError(f"{lora_name}: DoRA is built for Linear targets (train_lora_xl.py adapts "
                                      f"target_replace=['Attention'] only)")
        self.lora_name, self.target_path = lora_name, target_path
        self.lora_dim = lora_dim
        alpha = 1  # dora.py:247-249: the network's alpha lands in `prefix`; the module keeps its default 1
        self.scale = alpha / self.lora_dim
        self.alpha = torch.tensor(alpha)
        self.multiplier = multiplier
        self.is_conv = False
        self.in_dim, self.out_dim = org_module.in_features, org_module.out_features
        self.off_down, self.off_up, self.off_scale = off_down, off_up, off_scale
        self.lora_down = _WeightView(owner, "down", off_down, (lora_dim, self.in_dim))
        self.lora_up = _WeightView(owner, "up", off_up, (self.out_dim, lora_dim))
        self._scale_view = _ScaleView(owner, off_scale, self.in_dim)

    @property
    def dora_scale(self) -> torch.Tensor:
        return self._scale_view()

    @property
    def dora_scale_grad(self):
        return self._scale_view.grad

    def parameters(self):
        return [self.lora_down.weight, self.lora_up.weight, self.dora_scale]


class DoRANetwork(LoRANetwork):
    def __init__(self, unet, rank: int = 4, multiplier: float = 1.0, delimiter: str = "_", alpha: float = 1.0,
                 prefix: str = LORA_PREFIX_UNET, target_replace=DEFAULT_TARGET_REPLACE,
                 train_method: str = "full") -> None:
        nn.Module.__init__(self)
        self.lora_scale = 1
        self.multiplier = multiplier
        self.lora_dim = rank
        self.alpha = alpha
        visits = select_targets(unet, train_method, target_replace, prefix, delimiter, with_duplicates=True)
        targets = [v[:3] for v in visits if not v[3]]
        n_down = sum(rank * c.in_features for _, _, c in targets)
        n_up = sum(rank * c.out_features for _, _, c in targets)
        n_scale = sum(c.in_features for _, _, c in targets)
        self._n_down, self._n_up_only, self._n_scale = n_down, n_up, n_scale
        self._n_up = n_up + n_scale  # what the engine sees as the "up" buffer: [all lora_up | all dora_scale]
        self.flat = nn.Parameter(torch.zeros(max(n_down + n_up + n_scale, 1), dtype=torch.float32))
        self.unet_loras: List[DoRAModule] = []
        od = ou = osc = 0
        by_name = {}
        for lora_name, path, child in targets:
            m = DoRAModule(self, lora_name, path, child, multiplier, rank, od, ou, osc)
            self.unet_loras.append(m)
            by_name[lora_name] = m
            od += rank * m.in_dim
            ou += rank * m.out_dim
            osc += m.in_dim
        names = set()
        for lora in self.unet_loras:
            assert lora.lora_name not in names, f"duplicated lora name: {lora.lora_name}. {names}"
            names.add(lora.lora_name)
        with torch.no_grad():
            for lora_name, _path, child, dup in visits:  # RNG draw order of dora.py:71-120 (see lora.py)
                torch.empty(rank, child.in_features).uniform_()
                torch.empty(child.out_features, rank).uniform_()
                if dup:
                    nn.init.kaiming_uniform_(torch.empty(rank, child.in_features), a=1)
                    continue
                m = by_name[lora_name]
                nn.init.kaiming_uniform_(m.lora_down.weight, a=1)
                # dora_scale = column norms of the frozen weight (dora.py:95-105), in fp32 whatever the weight dtype
                m.dora_scale.copy_(torch.norm(child.weight.detach().float(), dim=0, keepdim=True).to(self.flat.device))
        unet.__dict__["_lora_network"] = self

    @property
    def flat_up(self) -> torch.Tensor:
        return self.flat[self._n_down:self._n_down + self._n_up_only]

    @property
    def flat_scale(self) -> torch.Tensor:
        return self.flat[self._n_down + self._n_up_only:self._n_down + self._n_up_only + self._n_scale]

    def engine_sites(self):
        return [{"target": l.target_path, "off_down": l.off_down, "off_up": l.off_up, "rank": l.lora_dim,
                 "scale": float(l.scale), "off_dora": self._n_up_only + l.off_scale} for l in self.unet_loras]

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        for l in self.unet_loras:  # the reference module's own order: parameter, buffer, then the child layers
            sd[f"{prefix}{l.lora_name}.dora_scale"] = l.dora_scale.detach()
            sd[f"{prefix}{l.lora_name}.alpha"] = l.alpha.clone()
            sd[f"{prefix}{l.lora_name}.lora_down.weight"] = l.lora_down.weight.detach()
            sd[f"{prefix}{l.lora_name}.lora_up.weight"] = l.lora_up.weight.detach()
        return sd

    def load_state_dict(self, state_dict, strict: bool = True):
        want, have = set(self.state_dict().keys()), set(state_dict.keys())
        if strict and want != have:
            raise RuntimeError(f"Error(s) in loading state_dict for DoRANetwork: missing {sorted(want - have)[:4]}, "
                               f"unexpected {sorted(have - want)[:4]}")
        with torch.no_grad():
            for l in self.unet_loras:
                for key, dst in ((f"{l.lora_name}.lora_down.weight", l.lora_down.weight),
                                 (f"{l.lora_name}.lora_up.weight", l.lora_up.weight),
                                 (f"{l.lora_name}.dora_scale", l.dora_scale)):
                    if key in state_dict:
                        dst.copy_(state_dict[key].to(dst.device, torch.float32))

    def __enter__(self):  # dora.py:322-324
        for lora in self.unet_loras:
            lora.multiplier = 1.0
