"""End-to-end timing of the text-slider CLI code path at the headline shape: `train_lora_xl.train` --
SD-XL architecture (synthetic weights, synthetic://sdxl), 1024 x 1024, batch_size 2, rank 4, Euler-a, 12 denoising steps --
20 iterations INCLUDING the reference's stochastic pre-roll (timesteps_to ~ U{1..11} forwards per step) in three modes: the
CLI default (round 4: fused step, each distinct frozen sample once, conditional-half pre-roll at train.cfg = 1), round 3's
`--fused_step` (full 8B-sample step, doubled-batch pre-roll) and `--no_fused_step` (the reference-style autograd loop).
Prints one JSON line per mode: iterations/s over the iterations after the first two (engine set-up, tile
tuning), and the mean number of pre-roll forwards the RNG happened to draw."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sliders_conceptmod_amd.config_util as CU
import sliders_conceptmod_amd.prompt_util as PRU
from sliders_conceptmod_amd.train_lora_xl import train

iters = int(os.environ.get("SMI_CLI_ITERS", "20"))
for mode, fused, dedup, full_pre in (("default", None, True, False), ("round-3 --fused_step", True, False, True),
                                     ("--no_fused_step", False, False, True)):
    os.environ["SMI_PREROLL_FULL"] = "1" if full_pre else "0"
    cfg = CU.RootConfig(
        prompts_file="unused", pretrained_model=CU.PretrainedModelConfig(name_or_path="synthetic://sdxl"),
        network=CU.NetworkConfig(type="lierla", rank=4, alpha=1.0, training_method="noxattn"),
        train=CU.TrainConfig(precision="float16", noise_scheduler="euler_a", iterations=iters, lr=1e-4, optimizer="AdamW",
                             lr_scheduler="constant", max_denoising_steps=12, cfg=1.0),
        save=CU.SaveConfig(name="t", path="/tmp/smi_cli_timing", per_steps=100000), logging=CU.LoggingConfig(),
        other=CU.OtherConfig())
    prompts = [PRU.PromptSettings(target="person", positive="old person", neutral="person", unconditional="",
                                  negative="young person", action="enhance", guidance_scale=4.0, resolution=1024,
                                  batch_size=2)]
    torch.manual_seed(0)
    stamps = []

    def tick(i):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())

    train(cfg, prompts, torch.device("cuda:0"), on_step_complete=tick, rank=4, save_file=False, fused_step=fused,
          dedup_uncond=dedup)
    dt = stamps[-1] - stamps[1]
    n = len(stamps) - 2
    print(json.dumps({"cli": "train_lora_xl.train", "mode": mode, "fused_step": fused is not False, "dedup_uncond": dedup,
                      "preroll": "doubled batch" if full_pre else "conditional half (train.cfg = 1)", "shape": "SD-XL 1024^2, batch_size 2, rank 4, 12-step "
                      "Euler-a pre-roll", "iterations_timed": n, "it_per_s": n / dt, "s_per_it": dt / n}), flush=True)
