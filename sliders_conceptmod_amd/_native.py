"""ctypes binding of libsmi_hip.so (include/smi.h).  PyTorch is used only to own device memory and streams:
tensors are passed as raw device pointers.  There is NO fallback: if the library is missing or fails to load, every
entry point raises (the product path never computes on the CPU)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMI_LIB") or os.path.join(_HERE, "libsmi_hip.so")  # SMI_LIB: A/B runs of tools/ against another build
SMI_MAX_LEVELS = 8
DTYPE_CODE = {torch.float16: 0, torch.bfloat16: 1}


class SmiError(RuntimeError):
    pass


class UNetConfigC(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int), ("n_levels", C.c_int),
        ("block_out_channels", C.c_int * SMI_MAX_LEVELS), ("down_has_attn", C.c_int * SMI_MAX_LEVELS),
        ("up_has_attn", C.c_int * SMI_MAX_LEVELS), ("layers_per_block", C.c_int),
        ("transformer_layers", C.c_int * SMI_MAX_LEVELS), ("num_heads", C.c_int * SMI_MAX_LEVELS),
        ("mid_transformer_layers", C.c_int), ("cross_attention_dim", C.c_int), ("norm_num_groups", C.c_int),
        ("use_linear_projection", C.c_int), ("addition_embed", C.c_int), ("addition_time_embed_dim", C.c_int),
        ("projection_class_embeddings_input_dim", C.c_int),
    ]


class VaeConfigC(C.Structure):
    _fields_ = [("dtype", C.c_int), ("in_channels", C.c_int), ("latent_channels", C.c_int), ("n_levels", C.c_int),
                ("block_out_channels", C.c_int * SMI_MAX_LEVELS), ("layers_per_block", C.c_int),
                ("norm_num_groups", C.c_int)]


class ClipConfigC(C.Structure):
    _fields_ = [("dtype", C.c_int), ("vocab_size", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int),
                ("num_heads", C.c_int), ("intermediate_size", C.c_int), ("max_positions", C.c_int),
                ("hidden_act", C.c_int), ("projection_dim", C.c_int)]


class WeightC(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class LoraSiteC(C.Structure):
    _fields_ = [("target", C.c_char_p), ("off_down", C.c_int64), ("off_up", C.c_int64), ("rank", C.c_int),
                ("scale", C.c_float), ("off_dora", C.c_int64)]


_lib = None

_SIGS = {
    "smi_last_error": (C.c_char_p, []),
    "smi_workspace_bytes": (C.c_int, [C.POINTER(UNetConfigC), C.POINTER(LoraSiteC), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "smi_create": (C.c_int, [C.POINTER(UNetConfigC), C.POINTER(WeightC), C.c_int, C.POINTER(LoraSiteC), C.c_int,
                             C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
                             C.POINTER(C.c_void_p)]),
    "smi_unet_forward_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "smi_unet_forward_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "smi_destroy": (None, [C.c_void_p]),
    "smi_weights_bytes": (C.c_int, [C.POINTER(UNetConfigC), C.POINTER(LoraSiteC), C.c_int, C.POINTER(C.c_size_t)]),
    "smi_arena_bytes": (C.c_int, [C.POINTER(UNetConfigC), C.POINTER(LoraSiteC), C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "smi_replan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "smi_engine_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "smi_vae_workspace_bytes": (C.c_int, [C.POINTER(VaeConfigC), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "smi_vae_create": (C.c_int, [C.POINTER(VaeConfigC), C.POINTER(WeightC), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "smi_vae_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "smi_clip_workspace_bytes": (C.c_int, [C.POINTER(ClipConfigC), C.c_int, C.POINTER(C.c_size_t)]),
    "smi_clip_create": (C.c_int, [C.POINTER(ClipConfigC), C.POINTER(WeightC), C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                  C.c_void_p, C.POINTER(C.c_void_p)]),
    "smi_clip_encode": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5),
    "smi_unet_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]),
    "smi_unet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "smi_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "smi_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int64)]),
    "smi_cfg_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "smi_slider_loss": (C.c_int, [C.c_void_p] * 4 + [C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]),
    "smi_clip_adamw": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_float] * 5 + [C.c_int, C.c_float, C.c_void_p,
                                                                                    C.c_void_p]),
    "smi_sched_step": (C.c_int, [C.c_void_p] * 3 + [C.c_float] * 3 + [C.c_int64, C.c_void_p]),
    "smi_op_gemm_scratch": (C.c_int, [C.c_void_p, C.c_size_t]),
    "smi_op_gemm": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "smi_op_gemm_rows": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int,
                                   C.c_void_p]),
    "smi_op_conv3x3": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 10 +
                       [C.c_void_p]),
    "smi_op_attention_fwd": (C.c_int, [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "smi_op_attention_bwd": (C.c_int, [C.c_int] + [C.c_void_p] * 10 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "smi_gn_coop_timeouts": (C.c_int, []),
    "smi_op_groupnorm": (C.c_int, [C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_int, C.c_void_p]),
    "smi_op_layernorm": (C.c_int, [C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 2 + [C.c_float, C.c_void_p]),
    "smi_op_geglu": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 2 + [C.c_void_p]),
    "smi_op_gemm_geglu": (C.c_int, [C.c_int] + [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "smi_op_lora_down": (C.c_int, [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "smi_op_lora_skinny": (C.c_int, [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "smi_op_lora_wgrad": (C.c_int, [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_float, C.c_void_p,
                                                                                   C.c_void_p]),
}

EXPORTED_SYMBOLS = sorted(_SIGS)


def lib():
    """Loads the shared library (once).  Raises SmiError if it has not been built -- no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SmiError(f"{LIB_PATH} not found: build it with `python -m sliders_conceptmod_amd.build` "
                           f"(the HIP extension is mandatory; there is no CPU fallback)")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:
            raise SmiError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().smi_last_error()
        raise SmiError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_config(cfg, dtype: torch.dtype) -> UNetConfigC:
    """cfg: sliders_conceptmod_amd.unet.UNetConfig (the diffusers public config values)."""
    c = UNetConfigC()
    c.dtype = DTYPE_CODE[dtype]
    c.in_channels = cfg.in_channels
    c.out_channels = cfg.out_channels
    n = len(cfg.block_out_channels)
    c.n_levels = n
    for i in range(n):
        c.block_out_channels[i] = cfg.block_out_channels[i]
        c.down_has_attn[i] = int(cfg.down_block_types[i] == "CrossAttnDownBlock2D")
        c.up_has_attn[i] = int(cfg.up_block_types[i] == "CrossAttnUpBlock2D")
        c.transformer_layers[i] = cfg.transformer_layers_per_block[i]
        c.num_heads[i] = cfg.num_attention_heads[i]
    c.layers_per_block = cfg.layers_per_block
    c.mid_transformer_layers = cfg.mid_block_transformer_layers or cfg.transformer_layers_per_block[-1]
    c.cross_attention_dim = cfg.cross_attention_dim
    c.norm_num_groups = cfg.norm_num_groups
    c.use_linear_projection = int(cfg.use_linear_projection)
    c.addition_embed = int(cfg.addition_embed_type == "text_time")
    c.addition_time_embed_dim = cfg.addition_time_embed_dim
    c.projection_class_embeddings_input_dim = cfg.projection_class_embeddings_input_dim
    return c


def make_sites(sites: Sequence[dict]):
    """sites: [{'target': str, 'off_down': int, 'off_up': int, 'rank': int, 'scale': float}]"""
    arr = (LoraSiteC * max(1, len(sites)))()
    keep = []
    for i, s in enumerate(sites):
        b = s["target"].encode()
        keep.append(b)
        arr[i].target = b
        arr[i].off_down = s["off_down"]
        arr[i].off_up = s["off_up"]
        arr[i].rank = s["rank"]
        arr[i].scale = s["scale"]
        arr[i].off_dora = s.get("off_dora", -1)
    return arr, keep


def workspace_bytes(cfg_c: UNetConfigC, sites: Sequence[dict], batch: int, h: int, w: int, ctx_len: int,
                    batch_adapted: Optional[int] = None) -> int:
    arr, _keep = make_sites(sites)
    out = C.c_size_t(0)
    ba = batch if batch_adapted is None else batch_adapted
    check(lib().smi_workspace_bytes(C.byref(cfg_c), arr, len(sites), batch, ba, h, w, ctx_len, C.byref(out)),
          "smi_workspace_bytes")
    return out.value


def arena_bytes(cfg_c: UNetConfigC, sites: Sequence[dict], batch: int, h: int, w: int, ctx_len: int,
                batch_adapted: Optional[int] = None) -> int:
    arr, _keep = make_sites(sites)
    out = C.c_size_t(0)
    ba = batch if batch_adapted is None else batch_adapted
    check(lib().smi_arena_bytes(C.byref(cfg_c), arr, len(sites), batch, ba, h, w, ctx_len, C.byref(out)),
          "smi_arena_bytes")
    return out.value


class Engine:
    """Owns one smi_engine plus the torch tensors backing its workspace.

    The packed weights are shape-independent; `plan(batch, batch_adapted, h, w, ctx_len)` switches the engine to
    another shape by pointing it at another activation arena (smi_replan) -- no weight is re-packed and no GEMM tile is
    re-tuned for shapes already seen.  A small LRU of arenas keeps the `dynamic_resolution` buckets
    (T/train_util.py:1085-1097) resident; MAX_PLANS bounds it."""

    MAX_PLANS = 4

    def __init__(self, cfg, dtype: torch.dtype, state: dict, sites: Sequence[dict], batch: int, h: int, w: int,
                 ctx_len: int, device, batch_adapted: Optional[int] = None):
        self.cfg_c = make_config(cfg, dtype)
        self.dtype = dtype
        self.batch, self.h, self.w, self.ctx_len = batch, h, w, ctx_len
        self.batch_adapted = batch if batch_adapted is None else batch_adapted
        self.sites = list(sites)
        nbytes = workspace_bytes(self.cfg_c, sites, batch, h, w, ctx_len, self.batch_adapted)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._weights_keepalive = []
        warr = (WeightC * len(state))()
        names = []
        for i, (k, v) in enumerate(state.items()):
            if v.dtype != dtype or not v.is_contiguous() or v.device != self.workspace.device:
                raise SmiError(f"weight {k}: expected contiguous {dtype} on {self.workspace.device}")
            nb = k.encode()
            names.append(nb)
            warr[i].name = nb
            warr[i].data = v.data_ptr()
            warr[i].ndim = v.ndim
            for d in range(v.ndim):
                warr[i].shape[d] = v.shape[d]
            self._weights_keepalive.append(v)
        sarr, keep = make_sites(sites)
        handle = C.c_void_p()
        with torch.cuda.device(self.workspace.device):
            check(lib().smi_create(C.byref(self.cfg_c), warr, len(state), sarr, len(sites), batch, self.batch_adapted,
                                   h, w, ctx_len, ptr(self.workspace), nbytes, stream_ptr(), C.byref(handle)),
                  "smi_create")
        self.handle = handle
        self.stream = torch.cuda.current_stream().cuda_stream
        self._home = (batch, self.batch_adapted, h, w, ctx_len)  # the shape the creation workspace was sized for
        self._plans = {}                                         # shape -> arena tensor (LRU, most recent last)

    def plan(self, batch: int, batch_adapted: int, h: int, w: int, ctx_len: int):
        """Make (batch, batch_adapted, h, w, ctx_len) the engine's current shape (no-op when it already is)."""
        cur = (self.batch, self.batch_adapted, self.h, self.w, self.ctx_len)
        want = (batch, batch_adapted, h, w, ctx_len)
        if want == cur:
            return
        arena = None
        if want != self._home:
            arena = self._plans.pop(want, None)
            if arena is None:
                nbytes = arena_bytes(self.cfg_c, self.sites, batch, h, w, ctx_len, batch_adapted)
                while len(self._plans) >= self.MAX_PLANS:
                    self._plans.pop(next(iter(self._plans)))
                arena = torch.empty(nbytes, dtype=torch.uint8, device=self.workspace.device)
            self._plans[want] = arena
        check(lib().smi_replan(self.handle, batch, batch_adapted, h, w, ctx_len, ptr(arena),
                               0 if arena is None else arena.numel()), "smi_replan")
        self.batch, self.batch_adapted, self.h, self.w, self.ctx_len = want

    def stats(self) -> dict:
        out = (C.c_int64 * 4)()
        check(lib().smi_engine_stats(self.handle, out), "smi_engine_stats")
        return {"pack_launches": out[0], "replans": out[1], "tape_generation": out[2], "weights_bytes": out[3]}

    def tape_generation(self) -> int:
        return self.stats()["tape_generation"] if self.handle else 0

    def forward(self, sample: torch.Tensor, timestep: float, ctx: torch.Tensor, text_embeds, time_ids, lora_down,
                lora_up, multiplier, save: bool, n_adapted: Optional[int] = None) -> torch.Tensor:
        """n_adapted: the LAST n_adapted samples get the LoRA delta and are differentiated (default: all).
        `multiplier`: one float, or a sequence of n_adapted floats (one adaptor multiplier per adapted sample,
        smi_unet_forward_multi)."""
        n = sample.shape[0]
        na = min(n, self.batch_adapted) if n_adapted is None else n_adapted
        eps = torch.empty(sample.shape, dtype=torch.float32, device=sample.device)
        if isinstance(multiplier, (list, tuple)):
            if len(multiplier) != na:
                raise SmiError(f"{len(multiplier)} multipliers for {na} adapted samples")
            arr = (C.c_float * na)(*[float(m) for m in multiplier])
            check(lib().smi_unet_forward_multi(self.handle, n, na, ptr(sample), float(timestep), ptr(ctx),
                                               ptr(text_embeds), ptr(time_ids), ptr(lora_down), ptr(lora_up), arr,
                                               int(save), ptr(eps)), "smi_unet_forward_multi")
            return eps
        check(lib().smi_unet_forward_batched(self.handle, n, na, ptr(sample), float(timestep), ptr(ctx),
                                             ptr(text_embeds), ptr(time_ids), ptr(lora_down), ptr(lora_up),
                                             float(multiplier), int(save), ptr(eps)), "smi_unet_forward_batched")
        return eps

    def backward(self, d_eps: torch.Tensor, d_down: torch.Tensor, d_up: torch.Tensor):
        check(lib().smi_unet_backward(self.handle, ptr(d_eps), ptr(d_down), ptr(d_up)), "smi_unet_backward")

    PROF_CLASSES = ("gemm", "conv", "attention", "norm", "elementwise", "lora")

    def profile_enable(self, on: bool):
        check(lib().smi_profile_enable(self.handle, int(on)), "smi_profile_enable")

    def profile_read(self) -> dict:
        n = len(self.PROF_CLASSES)
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        la = (C.c_int64 * n)()
        check(lib().smi_profile_read(self.handle, ms, fl, by, la), "smi_profile_read")
        return {k: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "launches": la[i]}
                for i, k in enumerate(self.PROF_CLASSES)}

    def close(self):
        if getattr(self, "handle", None):
            lib().smi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _weight_table(state: dict, dtype, device):
    warr = (WeightC * len(state))()
    keep = []
    for i, (k, v) in enumerate(state.items()):
        if v.dtype != dtype or not v.is_contiguous() or v.device != device:
            raise SmiError(f"weight {k}: expected contiguous {dtype} on {device}")
        nb = k.encode()
        keep.append((nb, v))
        warr[i].name = nb
        warr[i].data = v.data_ptr()
        warr[i].ndim = v.ndim
        for d in range(v.ndim):
            warr[i].shape[d] = v.shape[d]
    return warr, keep


class VaeEngine:
    """AutoencoderKL encoder on the HIP engine for one image size (smi_vae_*): image -> posterior moments."""

    def __init__(self, cfg, dtype: torch.dtype, state: dict, batch: int, h: int, w: int, device):
        c = VaeConfigC()
        c.dtype = DTYPE_CODE[dtype]
        c.in_channels, c.latent_channels = cfg.in_channels, cfg.latent_channels
        c.n_levels = len(cfg.block_out_channels)
        for i, v in enumerate(cfg.block_out_channels):
            c.block_out_channels[i] = v
        c.layers_per_block, c.norm_num_groups = cfg.layers_per_block, cfg.norm_num_groups
        self.cfg_c, self.batch, self.h, self.w = c, batch, h, w
        self.latent_channels = cfg.latent_channels
        self.down = 2 ** (len(cfg.block_out_channels) - 1)
        out = C.c_size_t(0)
        check(lib().smi_vae_workspace_bytes(C.byref(c), batch, h, w, C.byref(out)), "smi_vae_workspace_bytes")
        self.workspace = torch.empty(out.value, dtype=torch.uint8, device=device)
        warr, self._keep = _weight_table(state, dtype, self.workspace.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.workspace.device):
            check(lib().smi_vae_create(C.byref(c), warr, len(state), batch, h, w, ptr(self.workspace), out.value,
                                       stream_ptr(), C.byref(handle)), "smi_vae_create")
        self.handle = handle

    def moments(self, image: torch.Tensor) -> torch.Tensor:
        n = image.shape[0]
        out = torch.empty((n, 2 * self.latent_channels, self.h // self.down, self.w // self.down), dtype=torch.float32,
                          device=image.device)
        check(lib().smi_vae_encode(self.handle, n, ptr(image), ptr(out)), "smi_vae_encode")
        return out

    def close(self):
        if getattr(self, "handle", None):
            lib().smi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ClipEngine:
    """CLIP text encoder on the HIP engine (smi_clip_*): token ids -> last hidden state, hidden_states[-2], pooled."""

    ACT = {"quick_gelu": 0, "gelu": 1}

    def __init__(self, cfg, dtype: torch.dtype, state: dict, batch: int, device):
        c = ClipConfigC()
        c.dtype = DTYPE_CODE[dtype]
        c.vocab_size, c.hidden_size, c.num_layers = cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers
        c.num_heads, c.intermediate_size = cfg.num_attention_heads, cfg.intermediate_size
        c.max_positions = cfg.max_position_embeddings
        if cfg.hidden_act not in self.ACT:
            raise SmiError(f"CLIP hidden_act '{cfg.hidden_act}' is not built (quick_gelu / gelu)")
        c.hidden_act = self.ACT[cfg.hidden_act]
        c.projection_dim = cfg.projection_dim or 0
        self.cfg_c, self.batch, self.dtype = c, batch, dtype
        out = C.c_size_t(0)
        check(lib().smi_clip_workspace_bytes(C.byref(c), batch, C.byref(out)), "smi_clip_workspace_bytes")
        self.workspace = torch.empty(out.value, dtype=torch.uint8, device=device)
        warr, self._keep = _weight_table(state, dtype, self.workspace.device)
        handle = C.c_void_p()
        with torch.cuda.device(self.workspace.device):
            check(lib().smi_clip_create(C.byref(c), warr, len(state), batch, ptr(self.workspace), out.value, stream_ptr(),
                                        C.byref(handle)), "smi_clip_create")
        self.handle = handle

    def encode(self, ids: torch.Tensor, eos_pos: torch.Tensor):
        n, L = ids.shape
        if L != self.cfg_c.max_positions:  # the engine always runs max_positions tokens per prompt (smi_clip_encode)
            raise SmiError(f"CLIP text encoder: token ids must be padded to {self.cfg_c.max_positions} positions "
                           f"(padding='max_length'), got {L}")
        dev = self.workspace.device
        d = self.cfg_c.hidden_size
        last = torch.empty((n, L, d), dtype=self.dtype, device=dev)
        pen = torch.empty((n, L, d), dtype=self.dtype, device=dev)
        pooled = torch.empty((n, self.cfg_c.projection_dim or d), dtype=self.dtype, device=dev)
        ids = ids.to(dev, torch.int32).contiguous()
        eos = eos_pos.to(dev, torch.int32).contiguous()
        check(lib().smi_clip_encode(self.handle, n, ptr(ids), ptr(eos), ptr(last), ptr(pen), ptr(pooled)),
              "smi_clip_encode")
        return last, pen, pooled

    def close(self):
        if getattr(self, "handle", None):
            lib().smi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
