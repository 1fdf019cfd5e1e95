"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the two noise schedulers the reference
constructs at conceptmod/textsliders/model_util.py:388-436 (`create_noise_scheduler`): DDIM (eta = 0) and
Euler-ancestral, both with beta_start=0.00085, beta_end=0.012, "scaled_linear", 1000 train timesteps,
clip_sample=False, epsilon prediction.

The scheduler classes themselves live in the un-vendored dependency `diffusers` (requirements.txt:3,
diffusers==0.20.2, absent from this image), so the arithmetic below follows the published algorithms
(DDIM: Song et al. 2020 eq. 12 with sigma_t = 0; Euler-ancestral: Karras et al. 2022 / k-diffusion
`sample_euler_ancestral`) with the diffusers-0.20 defaults the reference relies on (DDIM
timestep_spacing="leading", steps_offset=0, set_alpha_to_one=True; Euler-a timestep_spacing="linspace").
PARITY STATUS: **parity unpinned** (no fixture in the reference covers them); self-consistency known-answer
tests are in tests/test_oracle_sched.py.

Call sites in the reference that define the interface: train_lora.py:157-159,209-213 (`set_timesteps`,
`timesteps[...]`), train_util.py:103 (`init_noise_sigma`), :287 (`scale_model_input`), :324 (`step(...).prev_sample`),
trainscripts/imagesliders/train_util.py:233 (`add_noise`).
"""
from __future__ import annotations

import numpy as np
import torch


def _alphas_cumprod(beta_start=0.00085, beta_end=0.012, n=1000) -> torch.Tensor:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _StepOut:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample


class DDIMSchedulerRef:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000):
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(n=num_train_timesteps)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = torch.tensor(1.0)
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        step_ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None):
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].to(sample.device)
        a_prev = (self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod).to(sample.device)
        b_t = 1 - a_t
        pred_x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        direction = (1 - a_prev) ** 0.5 * model_output  # eta = 0 -> sigma_t = 0
        return _StepOut(a_prev ** 0.5 * pred_x0 + direction, pred_x0)

    def add_noise(self, original, noise, timesteps):
        ac = self.alphas_cumprod.to(original.device)
        t = torch.as_tensor(timesteps, device=original.device).reshape(-1).long()
        sa = ac[t] ** 0.5
        sb = (1 - ac[t]) ** 0.5
        while sa.ndim < original.ndim:
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original + sb * noise


class EulerAncestralSchedulerRef:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000):
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(n=num_train_timesteps)
        sig = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self._train_sigmas = sig.numpy().astype(np.float64)
        self.sigmas = torch.from_numpy(np.concatenate([self._train_sigmas[::-1], [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps,
                                                      dtype=np.float32)[::-1].copy())
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
        if device is not None:
            self.sigmas = self.sigmas.to(device)
            self.timesteps = self.timesteps.to(device)

    def _index(self, timestep) -> int:
        t = float(timestep)
        idx = (self.timesteps.float().cpu() == t).nonzero()
        return int(idx[0].item())

    def scale_model_input(self, sample, timestep):
        sigma = self.sigmas[self._index(timestep)].to(sample.device)
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, generator=None):
        i = self._index(timestep)
        sigma = self.sigmas[i].to(sample.device)
        sigma_to = self.sigmas[i + 1].to(sample.device)
        pred_x0 = sample - sigma * model_output
        sigma_up = (sigma_to ** 2 * (sigma ** 2 - sigma_to ** 2) / sigma ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        derivative = (sample - pred_x0) / sigma
        prev = sample + derivative * (sigma_down - sigma)
        # fresh noise from the torch global generator, drawn on CPU so host RNG order is device independent
        noise = torch.randn(model_output.shape, dtype=torch.float32, generator=generator).to(
            device=model_output.device, dtype=model_output.dtype)
        return _StepOut(prev + noise * sigma_up, pred_x0)

    def add_noise(self, original, noise, timesteps):
        i = self._index(torch.as_tensor(timesteps).reshape(-1)[0])
        return original + noise * self.sigmas[i].to(original.device)


def create_noise_scheduler_ref(name: str):
    name = name.lower().replace(" ", "_")
    if name == "ddim":
        return DDIMSchedulerRef()
    if name == "euler_a":
        return EulerAncestralSchedulerRef()
    raise ValueError(f"Unknown scheduler name: {name}")
