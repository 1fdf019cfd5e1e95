"""The slider train step as one device-side sequence: 3 frozen UNet passes + 1 adapted pass + guidance loss +
backward + (all-reduce) + global-norm clip + AdamW, without a host sync and without autograd graph construction.

It is the same arithmetic, in the same order, as the reference loop body (conceptmod/textsliders/train_lora.py:
216-300 for SD-1.x; train_lora_xl.py:240-351 for SD-XL) -- the drop-in helpers in train_util.py + torch autograd give
identical results and stay available; this class only removes Python/autograd overhead from the hot loop:

    positive / neutral / negative|unconditional : predict_noise(_xl) with the adaptor off     (train_lora.py:216-252)
    target                                      : predict_noise(_xl) inside `with network`     (train_lora.py:261-273)
    loss = PromptEmbedsPair.loss(...)                                                          (prompt_util.py:134-174)
    loss.backward(); [clip_grad_norm_(0.2)]; optimizer.step()                    (train_lora_xl.py:348-350)

Data parallelism (no reference counterpart): each rank runs the step on its shard of the batch; the flat fp32
LoRA gradient is all-reduced (RCCL, sum / world) after backward and BEFORE the clip, so the clip sees the
global-batch gradient exactly as a single-GPU run of the global batch would."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _native, parallel


class SliderStep:
    def __init__(self, unet, network, scheduler, *, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: float = 0.0, cfg_scale: float = 1.0,
                 skip_dead_cfg_half: bool = False, process_group=None, batch_passes: bool = True,
                 dedup_uncond: bool = False, preroll_skip_dead_half: bool = True):
        self.unet, self.network, self.scheduler = unet, network, scheduler
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.cfg_scale = cfg_scale
        # With CFG scale 1 the unconditional half of the doubled batch is algebraically dead (u + 1*(t-u) == t).
        # Off by default: the reference computes it, so the headline number does too.
        self.skip_dead = bool(skip_dead_cfg_half and cfg_scale == 1.0)
        self.pg = process_group
        # Run the four guidance passes as ONE UNet pass (frozen samples first, adapted target samples last;
        # smi_unet_forward_batched): same per-sample arithmetic, 4x larger GEMM M, ~60 % fewer launches.
        self.batch_passes = batch_passes
        # The three frozen passes each carry the same unconditional half (same latents, timestep and "" prompt), and
        # the `unconditional` pass is that sample twice: 6B frozen samples, of which only 3B (4B with a real negative
        # prompt) are distinct.  With dedup_uncond the distinct ones run once and are shared -- bit-identical results
        # (per-sample arithmetic does not depend on batch composition: tests/test_engine_gpu.py), 8B -> 5B samples
        # per step.  Off by default: the reference runs all of them, so the headline number does too.
        self.dedup = bool(dedup_uncond and batch_passes and not self.skip_dead)
        # Pre-roll at guidance scale exactly 1 (the SD-XL trainer's default, `train.cfg` = 1.0, T/config_util.py:41,
        # T/train_lora_xl.py:66,209-231): predict_noise_xl forms u + 1 * (t - u) from the doubled batch, i.e. t up to one fp32
        # rounding (<= 6e-8 relative) -- the unconditional half only feeds that rounding.  The pre-roll then runs the
        # conditional half alone (UNet batch B instead of 2B).  Off: the doubled batch at every scale.
        import os
        self.preroll_skip_dead_half = bool(preroll_skip_dead_half) and os.environ.get("SMI_PREROLL_FULL") != "1"  # (A/B switch)
        flat = network.flat
        # ONE message per step: [flat fp32 LoRA gradient | loss scalar] -- the loss rides on the
        # gradient's all-reduce
        self.msg = torch.zeros(flat.numel() + 1, dtype=flat.dtype, device=flat.device)
        self.grad = self.msg[:flat.numel()]
        self.loss = self.msg[flat.numel():]
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scratch = torch.empty(4096, dtype=torch.float32, device=flat.device)
        self.step_count = 0
        self._lib = _native.lib()

    # ---- conditioning for one prompt pair, laid out as the reference's concat_embeddings does (train_util.py:267-272)
    def make_conditioning(self, emb: Dict[str, torch.Tensor], batch_size: int, pooled: Optional[dict] = None,
                          time_ids: Optional[torch.Tensor] = None) -> dict:
        dt, dev = self.unet.dtype, self.unet.device
        sub = "negative" if "negative" in emb else "unconditional"
        out = {"B": batch_size, "keys": {"positive": "positive", "neutral": "neutral", "negative": sub,
                                         "target": "target"}}
        for role, key in out["keys"].items():
            if self.skip_dead:
                te = emb[key].repeat_interleave(batch_size, dim=0)
            else:
                te = torch.cat([emb["unconditional"], emb[key]]).repeat_interleave(batch_size, dim=0)
            c = {"ctx": te.to(dev, dt).contiguous()}
            if pooled is not None:
                if self.skip_dead:
                    pe = pooled[key].repeat_interleave(batch_size, dim=0)
                    ti = time_ids.repeat_interleave(batch_size, dim=0)
                else:
                    pe = torch.cat([pooled["unconditional"], pooled[key]]).repeat_interleave(batch_size, dim=0)
                    ti = torch.cat([time_ids, time_ids]).repeat_interleave(batch_size, dim=0)
                c["text_embeds"] = pe.to(dev, dt).contiguous()
                c["time_ids"] = ti.to(dev, torch.float32).contiguous()
            out[role] = c
        # the pre-roll always runs the doubled [unconditional, target] batch (guidance 3 / train.cfg there)
        tf = {"ctx": torch.cat([emb["unconditional"], emb["target"]]).repeat_interleave(batch_size, dim=0)
              .to(dev, dt).contiguous()}
        if pooled is not None:
            tf["text_embeds"] = torch.cat([pooled["unconditional"], pooled["target"]]).repeat_interleave(
                batch_size, dim=0).to(dev, dt).contiguous()
            tf["time_ids"] = torch.cat([time_ids, time_ids]).repeat_interleave(batch_size, dim=0).to(
                dev, torch.float32).contiguous()
        out["target_full"] = tf
        order = ("positive", "neutral", "negative", "target")  # adapted (target) samples LAST
        out["all"] = {k: torch.cat([out[r][k] for r in order]).contiguous() for k in out["target"]}
        if self.dedup:
            # distinct frozen prompts, unconditional first; then the adapted pair [unconditional, target]
            uniq = ["unconditional"]
            for role in ("positive", "neutral", "negative"):
                if out["keys"][role] not in uniq:
                    uniq.append(out["keys"][role])
            out["uniq"] = uniq

            def rows(src, key):
                return src[key].repeat_interleave(batch_size, dim=0)

            seq = uniq + ["unconditional", "target"]
            d = {"ctx": torch.cat([rows(emb, k) for k in seq]).to(dev, dt).contiguous()}
            if pooled is not None:
                d["text_embeds"] = torch.cat([rows(pooled, k) for k in seq]).to(dev, dt).contiguous()
                d["time_ids"] = torch.cat([time_ids.repeat_interleave(batch_size, dim=0)] * len(seq)).to(
                    dev, torch.float32).contiguous()
            out["dedup"] = d
        return out

    def _pass(self, engine, x, t, c, lora: bool, save: bool):
        flat, n_down, mult = self.network.engine_params()
        down = up = None
        if lora:
            down, up = flat[:n_down], flat[n_down:]
        eps = engine.forward(x, t, c["ctx"], c.get("text_embeds"), c.get("time_ids"), down, up,
                             mult if lora else 0.0, save)
        if self.skip_dead:
            return eps
        out = torch.empty((eps.shape[0] // 2,) + tuple(eps.shape[1:]), dtype=torch.float32, device=eps.device)
        _native.check(self._lib.smi_cfg_combine(_native.ptr(eps), _native.ptr(out), out.numel(), self.cfg_scale,
                                                _native.stream_ptr()), "smi_cfg_combine")
        return out

    @torch.no_grad()
    def preroll(self, latents: torch.Tensor, cond: dict, total_timesteps: int, guidance_scale: float,
                start_timesteps: int = 0) -> torch.Tensor:
        """The no-grad pre-roll `diffusion(_xl)` (train_util.py:306-327, 677-708): for each of the first
        `total_timesteps` scheduler timesteps, predict_noise(_xl) with the adaptor ON at `guidance_scale` on the target
        prompt pair, then scheduler.step(...).prev_sample.  Same arithmetic as train_util.diffusion(_xl) on this
        engine, without autograd bookkeeping; returns the denoised latents (fp32)."""
        net = self.network
        lat = latents.float()
        c = cond["target_full"]
        B = lat.shape[0]
        net.__enter__()
        flat, n_down, mult = net.engine_params()
        net.__exit__(None, None, None)
        if self.preroll_skip_dead_half and float(guidance_scale) == 1.0:
            half = {k: v[B:].contiguous() for k, v in c.items()}  # rows [B, 2B) of concat_embeddings: the conditional ones
            for timestep in self.scheduler.timesteps[start_timesteps:total_timesteps]:
                x = self.scheduler.scale_model_input(lat, timestep).contiguous()
                _, _, h, w = x.shape
                engine = self.unet._ensure_engine(B, h, w, half["ctx"].shape[1])
                pred = engine.forward(x, float(timestep), half["ctx"], half.get("text_embeds"), half.get("time_ids"),
                                      flat[:n_down], flat[n_down:], mult, False)
                lat = self.scheduler.step(pred, timestep, lat).prev_sample
            return lat
        for timestep in self.scheduler.timesteps[start_timesteps:total_timesteps]:
            x = self.scheduler.scale_model_input(torch.cat([lat] * 2), timestep).contiguous()
            _, _, h, w = x.shape
            engine = self.unet._ensure_engine(2 * B, h, w, c["ctx"].shape[1])
            eps = engine.forward(x, float(timestep), c["ctx"], c.get("text_embeds"), c.get("time_ids"), flat[:n_down],
                                 flat[n_down:], mult, False)
            pred = torch.empty((B,) + tuple(eps.shape[1:]), dtype=torch.float32, device=eps.device)
            _native.check(self._lib.smi_cfg_combine(_native.ptr(eps), _native.ptr(pred), pred.numel(),
                                                    float(guidance_scale), _native.stream_ptr()), "smi_cfg_combine")
            lat = self.scheduler.step(pred, timestep, lat).prev_sample
        return lat

    def train_step(self, denoised_latents: torch.Tensor, timestep, cond: dict, action: str, eta: float,
                   lr: Optional[float] = None) -> torch.Tensor:
        """One optimisation step; returns the loss as a 1-element device tensor (no host sync)."""
        lat = denoised_latents.float()
        if self.dedup:
            return self._train_step_dedup(lat, timestep, cond, action, eta, lr)
        x = lat if self.skip_dead else torch.cat([lat] * 2)
        x = self.scheduler.scale_model_input(x, timestep).contiguous()
        t = float(timestep)
        n, _, h, w = x.shape
        engine = None if self.batch_passes else self.unet._ensure_engine(n, h, w, cond["target"]["ctx"].shape[1])
        net = self.network
        if self.batch_passes:
            return self._train_step_batched(x, t, cond, action, eta, lr)
        net.__exit__(None, None, None)
        positive = self._pass(engine, x, t, cond["positive"], False, False)
        neutral = self._pass(engine, x, t, cond["neutral"], False, False)
        negative = self._pass(engine, x, t, cond["negative"], False, False)
        net.__enter__()
        target = self._pass(engine, x, t, cond["target"], True, True)
        net.__exit__(None, None, None)

        return self._finish(engine, target, positive, neutral, negative, action, eta, lr)

    def _train_step_dedup(self, lat, timestep, cond, action, eta, lr):
        net = self.network
        uniq, c = cond["uniq"], cond["dedup"]
        B = lat.shape[0]
        nu = len(uniq)
        x = self.scheduler.scale_model_input(torch.cat([lat] * (nu + 2)), timestep).contiguous()
        _, _, h, w = x.shape
        engine = self.unet._ensure_engine((nu + 2) * B, h, w, c["ctx"].shape[1], n_adapted=2 * B)
        net.__enter__()
        flat, n_down, mult = net.engine_params()
        net.__exit__(None, None, None)
        eps = engine.forward(x, float(timestep), c["ctx"], c.get("text_embeds"), c.get("time_ids"), flat[:n_down],
                             flat[n_down:], mult, True, n_adapted=2 * B)
        e = {k: eps[i * B:(i + 1) * B] for i, k in enumerate(uniq)}

        def cfg(pair):
            o = torch.empty((B,) + tuple(pair.shape[1:]), dtype=torch.float32, device=pair.device)
            _native.check(self._lib.smi_cfg_combine(_native.ptr(pair), _native.ptr(o), o.numel(), self.cfg_scale,
                                                    _native.stream_ptr()), "smi_cfg_combine")
            return o

        outs = [cfg(torch.cat([e["unconditional"], e[cond["keys"][r]]])) for r in ("positive", "neutral", "negative")]
        target = cfg(eps[nu * B:].contiguous())
        return self._finish(engine, target, outs[0], outs[1], outs[2], action, eta, lr)

    def _train_step_batched(self, x, t, cond, action, eta, lr):
        net = self.network
        n, _, h, w = x.shape
        engine = self.unet._ensure_engine(4 * n, h, w, cond["target"]["ctx"].shape[1], n_adapted=n)
        net.__enter__()
        flat, n_down, mult = net.engine_params()
        net.__exit__(None, None, None)
        c = cond["all"]
        x4 = torch.cat([x] * 4)
        eps = engine.forward(x4, t, c["ctx"], c.get("text_embeds"), c.get("time_ids"), flat[:n_down], flat[n_down:],
                             mult, True, n_adapted=n)
        outs = []
        for i in range(4):
            e = eps[i * n:(i + 1) * n]
            if self.skip_dead:
                outs.append(e)
            else:
                o = torch.empty((n // 2,) + tuple(e.shape[1:]), dtype=torch.float32, device=e.device)
                _native.check(self._lib.smi_cfg_combine(_native.ptr(e), _native.ptr(o), o.numel(), self.cfg_scale,
                                                        _native.stream_ptr()), "smi_cfg_combine")
                outs.append(o)
        positive, neutral, negative, target = outs
        return self._finish(engine, target, positive, neutral, negative, action, eta, lr)

    def _finish(self, engine, target, positive, neutral, negative, action, eta, lr):
        net = self.network
        sign_eta = eta if action == "enhance" else -eta
        if action not in ("enhance", "erase"):
            raise ValueError("action must be erase or enhance")
        dtarget = torch.empty_like(target)
        _native.check(self._lib.smi_slider_loss(_native.ptr(target), _native.ptr(positive), _native.ptr(neutral),
                                                _native.ptr(negative), sign_eta, target.numel(),
                                                _native.ptr(self.loss), _native.ptr(dtarget),
                                                _native.ptr(self.scratch), _native.stream_ptr()), "smi_slider_loss")
        if self.skip_dead:
            d_eps = dtarget
        else:  # d(u + g (t - u)) = (1 - g) du + g dt
            d_eps = torch.cat([dtarget * (1.0 - self.cfg_scale), dtarget * self.cfg_scale])
        self.grad.zero_()
        n_down = net._n_down
        engine.backward(d_eps, self.grad[:n_down], self.grad[n_down:])
        parallel.allreduce_mean_(self.msg, self.pg)  # gradient + loss in one collective; no-op on a single rank
        self.step_count += 1
        flat = net.flat
        _native.check(self._lib.smi_clip_adamw(_native.ptr(flat), _native.ptr(self.grad), _native.ptr(self.exp_avg),
                                               _native.ptr(self.exp_avg_sq), flat.numel(),
                                               self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps,
                                               self.weight_decay, self.step_count, self.max_grad_norm,
                                               _native.ptr(self.scratch), _native.stream_ptr()), "smi_clip_adamw")
        return self.loss


class ImageSliderStep:
    """The image-slider step (trainscripts/imagesliders/train_lora-scale-xl.py:317-381; SD-1.x twin train_lora-scale.py:
    283-345) as one device-side sequence, without an autograd graph:

        slider +s : eps = predict_noise(_xl)(high_noised, positive prompt) -> MSE(eps, high_noise) -> backward
        slider -s : eps = predict_noise(_xl)(low_noised,  neutral prompt)  -> MSE(eps, low_noise)  -> backward
        (gradients accumulate in the flat fp32 buffer) -> all-reduce -> AdamW

    Same arithmetic and order as `train_lora_scale_xl.image_slider_step` + torch.optim.AdamW (tested against it).  The two
    sides differ in their adaptor multiplier (+s / -s); with Linear-only adaptors they still share ONE UNet pass and one
    backward through per-sample multipliers (`smi_unet_forward_multi`: twice the GEMM rows, half the launches); conv
    (c3lier) or DoRA adaptors fall back to one pass per side."""

    def __init__(self, unet, network, scheduler, *, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, max_grad_norm: float = 0.0, process_group=None, one_pass: bool = True):
        self.unet, self.network, self.scheduler = unet, network, scheduler
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.pg = process_group
        self.one_pass = one_pass  # both sides in one UNet pass where the adaptor set allows it (see _one_pass_ok)
        self._one_pass_cached = None
        flat = network.flat
        # one message per step: [flat fp32 LoRA gradient | loss of the high side, loss of the low side]
        self.msg = torch.zeros(flat.numel() + 2, dtype=flat.dtype, device=flat.device)
        self.grad = self.msg[:flat.numel()]
        self.losses = self.msg[flat.numel():]  # (high side, low side); mean over ranks after the step's all-reduce
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scratch = torch.empty(4096, dtype=torch.float32, device=flat.device)
        self.step_count = 0
        self._lib = _native.lib()

    def make_conditioning(self, text_embeds: torch.Tensor, batch_size: int, pooled: Optional[torch.Tensor] = None,
                          time_ids: Optional[torch.Tensor] = None, uncond: Optional[torch.Tensor] = None,
                          uncond_pooled: Optional[torch.Tensor] = None) -> dict:
        """CFG-doubled conditioning of one side: the prompt paired with the unconditional one (`uncond`,
        `uncond_pooled`), as the reference does (I/train_lora-scale-xl.py:321-337, I/train_lora-scale.py:283-318); without
        them the prompt is paired with itself (the same prediction at guidance 1)."""
        dt, dev = self.unet.dtype, self.unet.device
        first = text_embeds if uncond is None else uncond
        c = {"ctx": torch.cat([first, text_embeds]).repeat_interleave(batch_size, dim=0).to(dev, dt).contiguous()}
        if pooled is not None:
            first_p = pooled if uncond_pooled is None else uncond_pooled
            c["text_embeds"] = torch.cat([first_p, pooled]).repeat_interleave(batch_size, dim=0).to(dev, dt).contiguous()
            c["time_ids"] = torch.cat([time_ids, time_ids]).repeat_interleave(batch_size, dim=0).to(
                dev, torch.float32).contiguous()
        return c

    def _side(self, idx, sign_scale, noised, noise, timestep, c, guidance_scale):
        net = self.network
        x = self.scheduler.scale_model_input(torch.cat([noised.float()] * 2), timestep).contiguous()
        n, _, h, w = x.shape
        engine = self.unet._ensure_engine(n, h, w, c["ctx"].shape[1])
        net.set_lora_slider(scale=sign_scale)
        net.__enter__()
        flat, n_down, mult = net.engine_params()
        net.__exit__(None, None, None)
        eps = engine.forward(x, float(timestep), c["ctx"], c.get("text_embeds"), c.get("time_ids"), flat[:n_down],
                             flat[n_down:], mult, True)
        pred = torch.empty((n // 2,) + tuple(eps.shape[1:]), dtype=torch.float32, device=eps.device)
        _native.check(self._lib.smi_cfg_combine(_native.ptr(eps), _native.ptr(pred), pred.numel(),
                                                float(guidance_scale), _native.stream_ptr()), "smi_cfg_combine")
        diff = pred - noise.float()                       # MSE in fp32 (I/train_lora-scale-xl.py:338)
        self.losses[idx] = (diff * diff).mean()
        dpred = diff * (2.0 / diff.numel())
        d_eps = torch.cat([dpred * (1.0 - guidance_scale), dpred * guidance_scale])  # d(u + g (t - u))
        engine.backward(d_eps.contiguous(), self.grad[:n_down], self.grad[n_down:])  # accumulates

    def _one_pass_ok(self) -> bool:
        """Both sides can share ONE UNet pass (per-sample adaptor multipliers +s / -s, smi_unet_forward_multi) when every
        adaptor sits on a Linear layer: plain LoRA, no conv (c3lier) sites."""
        if self._one_pass_cached is None:
            net = self.network
            self._one_pass_cached = type(net).__name__ == "LoRANetwork" and all(
                not getattr(l, "is_conv", False) and len(l.lora_down._shape) == 2 for l in net.unet_loras)
        return self._one_pass_cached

    def _both_sides(self, scale, noised_high, noised_low, noise_high, noise_low, timestep, cond_pos, cond_neu,
                    guidance_scale):
        net = self.network
        B = noised_high.shape[0]
        x = torch.cat([noised_high.float()] * 2 + [noised_low.float()] * 2)       # [high CFG pair | low CFG pair]
        x = self.scheduler.scale_model_input(x, timestep).contiguous()
        n, _, h, w = x.shape
        c = {k: torch.cat([cond_pos[k], cond_neu[k]]).contiguous() for k in cond_pos}
        engine = self.unet._ensure_engine(n, h, w, c["ctx"].shape[1])
        net.set_lora_slider(scale=1)
        net.__enter__()
        flat, n_down, mult = net.engine_params()
        net.__exit__(None, None, None)
        mults = [mult * scale] * (2 * B) + [-mult * scale] * (2 * B)
        eps = engine.forward(x, float(timestep), c["ctx"], c.get("text_embeds"), c.get("time_ids"), flat[:n_down],
                             flat[n_down:], mults, True)
        d_parts = []
        for idx, (e, noise) in enumerate(((eps[:2 * B], noise_high), (eps[2 * B:], noise_low))):
            pred = torch.empty((B,) + tuple(e.shape[1:]), dtype=torch.float32, device=e.device)
            _native.check(self._lib.smi_cfg_combine(_native.ptr(e.contiguous()), _native.ptr(pred), pred.numel(),
                                                    float(guidance_scale), _native.stream_ptr()), "smi_cfg_combine")
            diff = pred - noise.float()
            self.losses[idx] = (diff * diff).mean()
            dpred = diff * (2.0 / diff.numel())
            d_parts += [dpred * (1.0 - guidance_scale), dpred * guidance_scale]
        engine.backward(torch.cat(d_parts).contiguous(), self.grad[:n_down], self.grad[n_down:])

    def train_step(self, noised_low, noised_high, noise_low, noise_high, timestep, cond_pos: dict, cond_neu: dict,
                   scale: float, guidance_scale: float = 1.0, lr: Optional[float] = None) -> torch.Tensor:
        """One optimisation step; returns the two side losses (high, low) as a device tensor (no host sync)."""
        net = self.network
        self.grad.zero_()
        if self.one_pass and self._one_pass_ok():
            self._both_sides(scale, noised_high, noised_low, noise_high, noise_low, timestep, cond_pos, cond_neu,
                             guidance_scale)
        else:
            self._side(0, +scale, noised_high, noise_high, timestep, cond_pos, guidance_scale)
            self._side(1, -scale, noised_low, noise_low, timestep, cond_neu, guidance_scale)
        net.set_lora_slider(scale=1)
        parallel.allreduce_mean_(self.msg, self.pg)
        self.step_count += 1
        flat = net.flat
        _native.check(self._lib.smi_clip_adamw(_native.ptr(flat), _native.ptr(self.grad), _native.ptr(self.exp_avg),
                                               _native.ptr(self.exp_avg_sq), flat.numel(),
                                               self.lr if lr is None else lr, self.betas[0], self.betas[1], self.eps,
                                               self.weight_decay, self.step_count, self.max_grad_norm,
                                               _native.ptr(self.scratch), _native.stream_ptr()), "smi_clip_adamw")
        return self.losses
