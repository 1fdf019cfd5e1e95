"""Host-side overhead of the fused CLI loop: `train_lora_xl.train(..., fused_step=True)` at the headline shape, per-iteration
wall time regressed on the number of pre-roll forwards the RNG drew:  t_iter = a + b * n_forwards.  b is the pre-roll
forward, a the train step + everything the loop does around the two (prompt pick, latents, scheduler set-up, logging)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliders_conceptmod_amd.config_util as CU
import sliders_conceptmod_amd.prompt_util as PRU
import sliders_conceptmod_amd.step as ST
from sliders_conceptmod_amd.train_lora_xl import train

iters = int(os.environ.get("SMI_CLI_ITERS", "40"))
cfg = CU.RootConfig(
    prompts_file="unused", pretrained_model=CU.PretrainedModelConfig(name_or_path="synthetic://sdxl"),
    network=CU.NetworkConfig(type="lierla", rank=4, alpha=1.0, training_method="noxattn"),
    train=CU.TrainConfig(precision="float16", noise_scheduler="euler_a", iterations=iters, lr=1e-4, optimizer="AdamW",
                         lr_scheduler="constant", max_denoising_steps=12, cfg=1.0),
    save=CU.SaveConfig(name="t", path="/tmp/smi_cli_timing", per_steps=100000), logging=CU.LoggingConfig(),
    other=CU.OtherConfig())
prompts = [PRU.PromptSettings(target="person", positive="old person", neutral="person", unconditional="",
                              negative="young person", action="enhance", guidance_scale=4.0, resolution=1024, batch_size=2)]
torch.manual_seed(0)
nfwd, stamps = [], []
orig = ST.SliderStep.preroll


def preroll(self, latents, cond, total_timesteps, guidance_scale, start_timesteps=0):
    nfwd.append(total_timesteps - start_timesteps)
    return orig(self, latents, cond, total_timesteps, guidance_scale, start_timesteps)


ST.SliderStep.preroll = preroll


def tick(i):
    torch.cuda.synchronize()
    stamps.append(time.perf_counter())


train(cfg, prompts, torch.device("cuda:0"), on_step_complete=tick, rank=4, save_file=False, fused_step=True)
t = np.diff(np.array(stamps))[2:]
n = np.array(nfwd[1:len(stamps)])[2:]
A = np.stack([np.ones_like(n, dtype=float), n.astype(float)], 1)
(a, b), *_ = np.linalg.lstsq(A, t, rcond=None)
print(json.dumps({"iterations": int(len(t)), "mean_forwards": float(n.mean()), "s_per_it": float(t.mean()),
                  "fit_ms": {"per_iteration_fixed": a * 1e3, "per_preroll_forward": b * 1e3},
                  "residual_ms_rms": float(np.sqrt(((A @ np.array([a, b]) - t) ** 2).mean()) * 1e3)}))
