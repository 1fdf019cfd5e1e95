"""Noise schedulers with the interface the reference's helpers call (conceptmod/textsliders/model_util.py:388-436
builds them; train_util.py:103,287,324,705 and train_lora.py:157-213 use them), plus the model factory.

The scheduler classes of the reference live in diffusers (absent here); these are native restatements of the four it
can construct -- DDIM (eta = 0), Euler-ancestral, DDPM and LMS (order 4) -- with the same constants
(beta 0.00085 -> 0.012 scaled-linear, 1000 train steps, clip_sample False) and both prediction types
(`epsilon`, and `v_prediction` for `pretrained_model.v_pred`, model_util.py:134).  Every update is affine in
(sample, model output, noise): the coefficients are computed on the host and the latent update runs on the GPU
through `smi_sched_step` when the latents are on a cuda device."""
from __future__ import annotations

import os
from typing import Literal

import numpy as np
import torch

from . import _native

AVAILABLE_SCHEDULERS = Literal["ddim", "ddpm", "lms", "euler_a"]


class _StepOutput:
    def __init__(self, prev_sample):
        self.prev_sample = prev_sample


def _affine(x: torch.Tensor, eps: torch.Tensor, noise, cx: float, ce: float, cn: float) -> torch.Tensor:
    if x.is_cuda and x.dtype == torch.float32 and eps.dtype == torch.float32:
        out = x.contiguous().clone()
        e = eps.contiguous()
        nz = None if noise is None else noise.to(x.device, torch.float32).contiguous()
        _native.check(_native.lib().smi_sched_step(_native.ptr(out), _native.ptr(e), _native.ptr(nz), cx, ce, cn,
                                                   out.numel(), _native.stream_ptr()), "smi_sched_step")
        return out
    out = cx * x + ce * eps
    return out if noise is None else out + cn * noise.to(x.device, x.dtype)


def _alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def _check_prediction_type(p):
    if p not in ("epsilon", "v_prediction"):
        raise ValueError(f"prediction_type must be epsilon or v_prediction, got {p}")
    return p


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, prediction_type="epsilon"):
        self.prediction_type = _check_prediction_type(prediction_type)
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy()
                                          .astype(np.int64))  # kept on the host: indexing it never syncs the GPU

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else float(self.final_alpha_cumprod)
        if self.prediction_type == "epsilon":
            # prev = sqrt(a_prev) * (x - sqrt(1-a_t) eps) / sqrt(a_t) + sqrt(1 - a_prev) eps
            cx = (a_prev / a_t) ** 0.5
            ce = (1 - a_prev) ** 0.5 - (a_prev * (1 - a_t) / a_t) ** 0.5
        else:
            # x0 = sqrt(a_t) x - sqrt(1-a_t) v ; eps = sqrt(a_t) v + sqrt(1-a_t) x ; prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
            cx = (a_prev * a_t) ** 0.5 + ((1 - a_prev) * (1 - a_t)) ** 0.5
            ce = ((1 - a_prev) * a_t) ** 0.5 - (a_prev * (1 - a_t)) ** 0.5
        return _StepOutput(_affine(sample, model_output, None, cx, ce, 0.0))

    def add_noise(self, original, noise, timesteps):
        t = torch.as_tensor(timesteps).reshape(-1).long().cpu()
        sa = (self.alphas_cumprod[t] ** 0.5).to(original.device, original.dtype)
        sb = ((1 - self.alphas_cumprod[t]) ** 0.5).to(original.device, original.dtype)
        while sa.ndim < original.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original + sb * noise


class DDPMScheduler(DDIMScheduler):
    """Ancestral sampling with the "fixed_small" posterior variance (diffusers' default), leading timestep spacing."""

    def step(self, model_output, timestep, sample, generator=None):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        b_t, b_prev = 1 - a_t, 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c0 = a_prev ** 0.5 * cur_b / b_t  # weight of the predicted x0
        c1 = cur_a ** 0.5 * b_prev / b_t  # weight of the current sample
        if self.prediction_type == "epsilon":  # x0 = (x - sqrt(b_t) eps) / sqrt(a_t)
            cx, ce = c1 + c0 / a_t ** 0.5, -c0 * (b_t / a_t) ** 0.5
        else:                                  # x0 = sqrt(a_t) x - sqrt(b_t) v
            cx, ce = c1 + c0 * a_t ** 0.5, -c0 * b_t ** 0.5
        noise, cn = None, 0.0
        if t > 0:
            rank, world = getattr(self, "dp_shard", (0, 1))
            from .parallel import shard_noise
            noise = shard_noise(lambda shp: torch.randn(shp, dtype=torch.float32, generator=generator),
                                tuple(model_output.shape), rank, world)
            cn = max(b_prev / b_t * cur_b, 1e-20) ** 0.5
        return _StepOutput(_affine(sample, model_output, noise, cx, ce, cn))


class EulerAncestralDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, prediction_type="epsilon"):
        self.prediction_type = _check_prediction_type(prediction_type)
        self.num_train_timesteps = num_train_timesteps
        ac = _alphas_cumprod(num_train_timesteps)
        self._train_sigmas = (((1 - ac) / ac) ** 0.5).numpy().astype(np.float64)
        self.sigmas = torch.from_numpy(np.concatenate([self._train_sigmas[::-1], [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(
            np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy())
        self.num_inference_steps = None

    @property
    def init_noise_sigma(self):
        return self.sigmas.max()

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.num_train_timesteps - 1, num_inference_steps, dtype=np.float32)[::-1].copy()
        sig = np.interp(ts, np.arange(0, len(self._train_sigmas)), self._train_sigmas)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)

    def _index(self, timestep) -> int:
        return int((self.timesteps == float(timestep)).nonzero()[0].item())

    def scale_model_input(self, sample, timestep):
        sigma = float(self.sigmas[self._index(timestep)])
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, generator=None):
        i = self._index(timestep)
        sigma, sigma_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        sigma_up = (sigma_to ** 2 * (sigma ** 2 - sigma_to ** 2) / sigma ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        # host torch RNG (global generator) keeps the draw order of the reference run: SURVEY.md section 7 "RNG parity"
        # under data parallelism (`dp_shard = (rank, world)`, set by the trainers) the GLOBAL batch's noise is drawn and
        # this rank's rows are taken, so W ranks consume the control RNG like one rank at the global batch
        rank, world = getattr(self, "dp_shard", (0, 1))
        from .parallel import shard_noise
        # `noise_on_device` (or a device generator) draws on the sample's device instead, as diffusers' randn_tensor does
        # when it is handed no CPU generator: no host draw and no 1 MB host-to-device copy per denoising step
        on_dev = model_output.is_cuda and (getattr(self, "noise_on_device", False)
                                           or (generator is not None and generator.device.type != "cpu"))
        dev = model_output.device if on_dev else None
        noise = shard_noise(lambda shp: torch.randn(shp, dtype=torch.float32, generator=generator, device=dev),
                            tuple(model_output.shape), rank, world)
        dt = sigma_down - sigma
        if self.prediction_type == "epsilon":
            # prev = x + eps * dt + noise * sigma_up       (derivative = eps for epsilon prediction)
            cx, ce = 1.0, dt
        else:
            # x0 = -sigma v / sqrt(sigma^2+1) + x / (sigma^2+1); derivative = (x - x0) / sigma
            cx, ce = 1.0 + dt * sigma / (sigma ** 2 + 1), dt / (sigma ** 2 + 1) ** 0.5
        return _StepOutput(_affine(sample, model_output, noise, cx, ce, sigma_up))

    def add_noise(self, original, noise, timesteps):
        i = self._index(torch.as_tensor(timesteps).reshape(-1)[0])
        return original + noise * float(self.sigmas[i])


class LMSDiscreteScheduler(EulerAncestralDiscreteScheduler):
    """Linear multistep (order 4, Katherine Crowson's k-diffusion `sample_lms`) on the same sigma ladder as Euler-a:
    prev = x + sum_j c_j d_{i-j}, d = (x - x0) / sigma, c_j = integral over [sigma_i, sigma_{i+1}] of the j-th Lagrange
    basis polynomial through the last `order` sigmas.  Stateful: the derivative history restarts at set_timesteps."""
    order = 4

    def set_timesteps(self, num_inference_steps: int, device=None):
        super().set_timesteps(num_inference_steps, device)
        self.derivatives = []

    def get_lms_coefficient(self, order, t, current_order):
        from scipy import integrate
        sig = self.sigmas.double().numpy()

        def basis(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - sig[t - k]) / (sig[t - current_order] - sig[t - k])
            return prod

        return integrate.quad(basis, sig[t], sig[t + 1], epsrel=1e-4)[0]

    def step(self, model_output, timestep, sample, order: int = 4, generator=None):
        i = self._index(timestep)
        sigma = float(self.sigmas[i])
        # derivative d = (x - x0) / sigma as an affine map of (x, model output)
        if self.prediction_type == "epsilon":
            dx, de = 0.0, 1.0
        else:
            dx, de = sigma / (sigma ** 2 + 1), 1.0 / (sigma ** 2 + 1) ** 0.5
        d = dx * sample.float() + de * model_output.float()
        hist = getattr(self, "derivatives", [])
        hist.append(d)
        if len(hist) > order:
            hist.pop(0)
        self.derivatives = hist
        order = min(i + 1, order, len(hist))
        coeffs = [self.get_lms_coefficient(order, i, j) for j in range(order)]
        prev = sample.float()
        for c, dj in zip(coeffs, reversed(hist)):
            prev = prev + float(c) * dj
        return _StepOutput(prev)


def create_noise_scheduler(scheduler_name: AVAILABLE_SCHEDULERS = "ddpm", prediction_type="epsilon"):
    """model_util.py:388-436: the four schedulers with the reference's constants; unknown names raise ValueError."""
    name = scheduler_name.lower().replace(" ", "_")
    if name == "ddim":
        return DDIMScheduler(prediction_type=prediction_type)
    if name == "ddpm":
        return DDPMScheduler(prediction_type=prediction_type)
    if name == "lms":
        return LMSDiscreteScheduler(prediction_type=prediction_type)
    if name == "euler_a":
        s = EulerAncestralDiscreteScheduler(prediction_type=prediction_type)
        s.noise_on_device = os.environ.get("SMI_DEVICE_NOISE", "0") == "1"
        return s
    raise ValueError(f"Unknown scheduler name: {name}")


# ----------------------------------------------------------------------------------------------------------------
# model factory (reference: conceptmod/textsliders/model_util.py:27-137, 170-385 -- hub / checkpoint loaders)
# ----------------------------------------------------------------------------------------------------------------
class SyntheticTextEncoder:
    """Offline stand-in for tokenizer + CLIP text encoder(s): a prompt maps to seeded normal embeddings
    ([1, 77, D] and, for SD-XL, a pooled [1, P]) keyed by a hash of the text, so equal prompts share embeddings
    exactly as the reference's PromptEmbedsCache assumes (train_lora.py:106-146).  The real front end (CLIP-L /
    OpenCLIP-bigG) is a 'next' row of the scope table."""

    def __init__(self, dim: int, pooled_dim: int = 0, seq_len: int = 77):
        self.dim, self.pooled_dim, self.seq_len = dim, pooled_dim, seq_len

    def encode(self, prompt: str):
        import hashlib
        seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:8], "little") % (2 ** 31)
        g = torch.Generator().manual_seed(seed)
        te = torch.randn(1, self.seq_len, self.dim, generator=g)
        if self.pooled_dim:
            return te, torch.randn(1, self.pooled_dim, generator=g)
        return te


def _unet_from_dir(path: str, dtype):
    """diffusers directory layout: <path>/unet/config.json + diffusion_pytorch_model.safetensors (safe loader only)."""
    import json
    import os
    from safetensors.torch import load_file
    from . import unet as PU
    cfgj = json.load(open(os.path.join(path, "unet", "config.json")))
    ahd = cfgj.get("attention_head_dim", 8)
    n = len(cfgj["block_out_channels"])
    tl = cfgj.get("transformer_layers_per_block", 1)
    cfg = PU.UNetConfig(
        in_channels=cfgj.get("in_channels", 4), out_channels=cfgj.get("out_channels", 4),
        block_out_channels=tuple(cfgj["block_out_channels"]), down_block_types=tuple(cfgj["down_block_types"]),
        up_block_types=tuple(cfgj["up_block_types"]), layers_per_block=cfgj.get("layers_per_block", 2),
        transformer_layers_per_block=tuple(tl) if isinstance(tl, (list, tuple)) else (tl,) * n,
        num_attention_heads=tuple(ahd) if isinstance(ahd, (list, tuple)) else (ahd,) * n,
        cross_attention_dim=cfgj["cross_attention_dim"], norm_num_groups=cfgj.get("norm_num_groups", 32),
        use_linear_projection=cfgj.get("use_linear_projection", False),
        addition_embed_type=cfgj.get("addition_embed_type"),
        addition_time_embed_dim=cfgj.get("addition_time_embed_dim") or 256,
        projection_class_embeddings_input_dim=cfgj.get("projection_class_embeddings_input_dim") or 2816)
    unet = PU.UNet2DConditionModel(cfg)
    sd = load_file(os.path.join(path, "unet", "diffusion_pytorch_model.safetensors"))
    unet.load_state_dict(sd)
    return unet


def _clip_from_dir(path: str, sub: str):
    """<path>/<sub>/config.json + model.safetensors (transformers layout, safe loader only) -> native CLIP text encoder;
    `architectures` tells CLIPTextModel from CLIPTextModelWithProjection."""
    import json
    from safetensors.torch import load_file
    from . import clip as PC
    cj = json.load(open(os.path.join(path, sub, "config.json")))
    proj = "CLIPTextModelWithProjection" in (cj.get("architectures") or [])
    cfg = PC.CLIPTextConfig(vocab_size=cj["vocab_size"], hidden_size=cj["hidden_size"],
                            intermediate_size=cj["intermediate_size"], num_hidden_layers=cj["num_hidden_layers"],
                            num_attention_heads=cj["num_attention_heads"],
                            max_position_embeddings=cj.get("max_position_embeddings", 77),
                            hidden_act=cj.get("hidden_act", "quick_gelu"),
                            projection_dim=cj.get("projection_dim") if proj else None,
                            eos_token_id=cj.get("eos_token_id", 2))
    enc = (PC.CLIPTextModelWithProjection if proj else PC.CLIPTextModel)(cfg)
    enc.load_state_dict(load_file(os.path.join(path, sub, "model.safetensors")))
    return enc


def load_vae(pretrained_model_name_or_path: str, xl: bool = False):
    """The AutoencoderKL the image-slider scripts get next to the UNet (I/model_util.py:75,179): `synthetic://...`
    builds the SD / SD-XL VAE encoder architecture with seeded weights; a local diffusers directory is read from its
    `vae/` sub-folder with the safetensors loader."""
    import json
    import os
    from . import vae as PV
    name = pretrained_model_name_or_path
    if name.startswith("synthetic://"):
        if "tiny" in name:
            cfg = PV.VAEConfig(block_out_channels=(64, 128, 128, 128), norm_num_groups=16,
                               scaling_factor=0.13025 if xl else 0.18215)
        else:
            cfg = PV.sdxl_vae_config() if xl else PV.sd_vae_config()
        return PV.init_synthetic_(PV.AutoencoderKL(cfg), seed=7)
    if os.path.isdir(name):
        from safetensors.torch import load_file
        cj = json.load(open(os.path.join(name, "vae", "config.json")))
        cfg = PV.VAEConfig(in_channels=cj.get("in_channels", 3), latent_channels=cj.get("latent_channels", 4),
                           block_out_channels=tuple(cj["block_out_channels"]),
                           layers_per_block=cj.get("layers_per_block", 2),
                           norm_num_groups=cj.get("norm_num_groups", 32),
                           scaling_factor=cj.get("scaling_factor", 0.18215))
        vae = PV.AutoencoderKL(cfg)
        vae.load_state_dict(load_file(os.path.join(name, "vae", "diffusion_pytorch_model.safetensors")))
        return vae
    raise ValueError(f"cannot load a VAE from '{name}': pass a local diffusers directory or synthetic://sd1x | sdxl")


def load_models(pretrained_model_name_or_path: str, scheduler_name: str = "ddim", v2: bool = False,
                v_pred: bool = False, weight_dtype=torch.float32, xl: bool = False):
    """Returns (tokenizer(s), text_encoder(s), unet, noise_scheduler) like the reference (model_util.py:112-137,
    :359-385).  `synthetic://sd1x|sdxl|tiny_sd1x|tiny_sdxl` builds the architecture with seeded random weights (no
    network here).  A local diffusers directory is loaded with safetensors; its text encoders through `transformers`."""
    from . import unet as PU
    if v2 and not os.path.isdir(pretrained_model_name_or_path):
        raise ValueError("SD-2.x needs a local diffusers directory (its UNet config is read from it)")
    # model_util.py:134: prediction_type = "v_prediction" if v_pred else "epsilon"
    scheduler = create_noise_scheduler(scheduler_name, prediction_type="v_prediction" if v_pred else "epsilon")
    name = pretrained_model_name_or_path
    if name.startswith("synthetic://"):
        kind = name[len("synthetic://"):]
        tiny = {"tiny_sd1x": PU.UNetConfig(block_out_channels=(64, 128, 256, 256), num_attention_heads=(4, 4, 4, 4),
                                           cross_attention_dim=64, norm_num_groups=16),
                "tiny_sdxl": PU.UNetConfig(
                    block_out_channels=(64, 128, 256),
                    down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                    up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                    transformer_layers_per_block=(1, 2, 3), num_attention_heads=(2, 4, 8), cross_attention_dim=64,
                    norm_num_groups=16, use_linear_projection=True, addition_embed_type="text_time",
                    addition_time_embed_dim=32, projection_class_embeddings_input_dim=256)}
        cfg = {"sd1x": PU.sd1x_config, "sdxl": PU.sdxl_config}.get(kind, lambda: tiny[kind])()
        unet = PU.init_synthetic_(PU.UNet2DConditionModel(cfg), seed=0)
        pooled = 0
        if cfg.addition_embed_type == "text_time":
            pooled = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        enc = SyntheticTextEncoder(cfg.cross_attention_dim, pooled)
        return None, enc, unet, scheduler
    if os.path.isdir(name):
        unet = _unet_from_dir(name, weight_dtype)
        # tokenisation is host text processing (transformers.CLIPTokenizer reads the directory's vocab files); the text
        # encoders are the native ones (clip.py: transformers parameter names, arithmetic on the HIP engine)
        from transformers import CLIPTokenizer
        toks = [CLIPTokenizer.from_pretrained(name, subfolder="tokenizer")]
        encs = [_clip_from_dir(name, "text_encoder")]
        if os.path.isdir(os.path.join(name, "text_encoder_2")):
            toks.append(CLIPTokenizer.from_pretrained(name, subfolder="tokenizer_2"))
            encs.append(_clip_from_dir(name, "text_encoder_2"))
        return toks, encs, unet, scheduler
    raise ValueError(f"cannot load '{name}': no network in this environment; pass a local diffusers directory or "
                     f"synthetic://sd1x | synthetic://sdxl")
