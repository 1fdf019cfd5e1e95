#!/usr/bin/env python3
"""bench.py -- slider train-steps/sec (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config sdxl_1024_b2_r4|sd15_512_b4_r4|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input: 3 frozen UNet
forwards + 1 LoRA-adapted forward + backward + guidance loss + global-norm clip + AdamW (+ all-reduce of the flat
LoRA gradient when N > 1), all on the CFG-doubled UNet batch 2B exactly as the reference computes it
(train_lora_xl.py:240-351).  The stochastic pre-roll (`diffusion_xl`) is excluded: synthetic `denoised_latents` are
supplied directly, resident in HBM before the timed region.  Weights are random-init of the real architecture
(no checkpoints offline) -- "data": "synthetic".

Extra objects on the JSON line:
  roofline     -- dominant kernel class (the MFMA GEMM kernels gemm_5ph_kernel / gemm_8ph_kernel / gemm_glds_kernel: Linear + implicit-GEMM conv launches), algorithmic
                  FLOPs / device time measured with HIP events on the launch stream in a separate profiled step
  cpu_baseline -- the CPU oracle (oracle/unet_ref.py, kind "port") timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (model, resolution, per-GPU batch B, rank, dtype, scheduler, max_grad_norm, lr, wd)
    "sdxl_1024_b2_r4": ("sdxl", 1024, 2, 4, "fp16", "euler_a", 0.2, 1e-4, 1e-6),   # BASELINE configs[2] (headline)
    "sdxl_1024_b2_r8": ("sdxl", 1024, 2, 8, "fp16", "euler_a", 0.2, 1e-4, 1e-6),   # configs[3] per-GPU shard
    "sd15_512_b4_r4": ("sd1x", 512, 4, 4, "bf16", "ddim", 0.0, 2e-4, 1e-2),        # configs[1]
    "sd14_512_b1_r4": ("sd1x", 512, 1, 4, "fp16", "ddim", 0.0, 2e-4, 1e-2),        # configs[0] shape on the GPU
    "tiny_sdxl": ("tiny_sdxl", 128, 2, 4, "fp16", "euler_a", 0.2, 1e-4, 1e-6),
    # the shipped SD-1.x default network type (T/data/config.yaml:7): conv + time_emb_proj + attention adaptors
    "sd14_512_b1_r4_c3lier": ("sd1x", 512, 1, 4, "fp16", "ddim", 0.0, 2e-4, 1e-2),
    # `--peft_type dora` of train_lora_xl.py (T/dora.py): the headline shape with DoRA modules instead of LoRA
    "sdxl_1024_b2_r4_dora": ("sdxl", 1024, 2, 4, "fp16", "euler_a", 0.2, 1e-4, 1e-6),
    # BASELINE configs[4] per-GPU unit: SD-XL image slider (train_lora-scale-xl.py), one before/after pair per step at
    # 1024^2, scales +1 / -1: VAE-encode both images, noise them, two adapted UNet passes (+s / -s) with their two backward
    # passes accumulating one LoRA gradient, AdamW.  Measured by the same JSON contract; `config.step` says what a step is.
    "image_sdxl_1024_b1_r4": ("sdxl", 1024, 1, 4, "fp16", "ddim", 0.0, 1e-4, 1e-2),
    "tiny_image_sdxl": ("tiny_sdxl", 128, 1, 4, "fp16", "ddim", 0.0, 1e-4, 1e-2),
}
IMAGE_WORKLOADS = ("image_sdxl_1024_b1_r4", "tiny_image_sdxl")
def pmc_traffic(config):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.py); None
    when no such pass exists for this configuration."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{config}.json")))
    if not paths:
        return None
    path = paths[-1]  # the latest round's passes (file names sort by round tag)
    rows = [v for k, v in json.load(open(path)).items() if "gemm_glds_kernel" in k or "gemm_8ph_kernel" in k or "gemm_5ph_kernel" in k]
    n = sum(v["launches"] for v in rows)
    return (sum(v["total_fetch_bytes"] + v["total_write_bytes"] for v in rows) / n) if n else None


MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0


def unet_cfg(model):
    import sliders_conceptmod_amd.unet as PU
    if model == "sdxl":
        return PU.sdxl_config()
    if model == "sd1x":
        return PU.sd1x_config()
    if model == "tiny_sdxl":
        return PU.UNetConfig(block_out_channels=(64, 128, 256),
                             down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                             up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
                             transformer_layers_per_block=(1, 2, 3), num_attention_heads=(2, 4, 8),
                             cross_attention_dim=64, norm_num_groups=16, use_linear_projection=True,
                             addition_embed_type="text_time", addition_time_embed_dim=32,
                             projection_class_embeddings_input_dim=256)
    raise ValueError(model)


@torch.no_grad()
def init_synthetic_on_device(module, seed):
    """Seeded fan-in-scaled weights generated directly in HBM (SURVEY.md section 8d 'Synthetic inputs')."""
    import math
    g = torch.Generator(device="cuda").manual_seed(seed)
    for name, p in module.named_parameters():
        if p.ndim >= 2:
            w = torch.randn(p.shape, generator=g, device="cuda") * (0.7 / math.sqrt(p[0].numel()))
        elif name.endswith("weight"):
            w = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device="cuda")
        else:
            w = 0.02 * torch.randn(p.shape, generator=g, device="cuda")
        p.copy_(w.to(p.dtype))


def algorithmic_step_flops(engine_profile):
    return sum(v["flops"] for v in engine_profile.values())


def host_cpu_info():
    """(model name, physical cores, logical CPUs this process may run on) from /proc/cpuinfo and the affinity mask."""
    name, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and name == "unknown":
                name = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
                cores.add((phys, core))
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return name, len(cores) or (os.cpu_count() or 1), usable


def cgroup_cpu_quota():
    """CPUs granted by the cgroup quota (v2 cpu.max, v1 cfs quota / period), or None when unlimited / unreadable."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, q // p)
    except (OSError, ValueError):
        return None


def cpu_baseline(model, lrank, step_flops, gpu_res, gpu_batch=2, seconds_budget=45.0):
    """The CPU oracle (kind "port": oracle/slider_ref.py + oracle/unet_ref.py, fp32 PyTorch on the host cores) timed on
    ONE WHOLE STEP in the reference's order: 3 frozen forwards + 1 adapted forward + guidance
    loss + backward + clip + AdamW at UNet batch 2B, B = 1, at a reduced resolution (SD-XL: 512^2 px; 256^2 if a short
    forward probe says 512^2 would not fit the time budget), then scaled to the benchmarked configuration by the
    algorithmic FLOP ratio -- stated as an extrapolation in `sample`."""
    from oracle import sched_ref as OS
    from oracle import slider_ref as R
    from oracle import unet_ref as OU
    cpu_name, phys, usable = host_cpu_info()
    # threads = the CPUs this process may actually run on (affinity mask, then the cgroup CPU quota when one is set: a GPU
    # box grants a share per GPU whatever os.cpu_count() says, and oversubscribing it is far slower), capped at 16
    nthreads = max(1, min(16, usable, cgroup_cpu_quota() or usable))
    torch.set_num_threads(nthreads)
    ocfg = {"sdxl": OU.sdxl_config, "sd1x": OU.sd1x_config, "tiny_sdxl": OU.tiny_sdxl_config}[model]()
    xl = ocfg.addition_embed_type == "text_time"
    ou = OU.UNet2DConditionModel(ocfg)
    with torch.no_grad():
        for p in ou.parameters():  # cheap deterministic fill: timing does not depend on the values
            p.uniform_(-0.02, 0.02)
    ou.requires_grad_(False).eval()
    net = R.LoRANetworkRef(ou, lrank, 1.0, 1.0, "noxattn")
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.normal_(0, 1e-2)
    params = [p for l in net.unet_loras for p in l.parameters()]
    keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if xl else [])
    emb = {k: torch.randn(1, 77, ocfg.cross_attention_dim) for k in keys}
    xlc = None
    if xl:
        pdim = ocfg.projection_class_embeddings_input_dim - 6 * ocfg.addition_time_embed_dim
        xlc = {"pooled": {k: torch.randn(1, pdim) for k in keys}}
    sched = OS.create_noise_scheduler_ref("euler_a" if xl else "ddim")
    sched.set_timesteps(1000)
    t = sched.timesteps[500]

    # FLOPs of what is timed, counted by hooks on Linear / Conv2d (+ attention analytically): forward hooks count the
    # forward; the backward is counted as the engine counts it (dX only, attention x2) via the same ratio Bk/F
    flops = [0.0]

    def lin_hook(m, inp, out):
        flops[0] += 2.0 * out.numel() * m.in_features

    def conv_hook(m, inp, out):
        flops[0] += 2.0 * out.numel() * m.in_channels * m.kernel_size[0] * m.kernel_size[1]

    def attn_hook(m, args, kwargs, out):
        xq = args[0]
        ctx_ = kwargs.get("context", args[1] if len(args) > 1 else None)
        nk = xq.shape[1] if ctx_ is None else ctx_.shape[1]
        flops[0] += 4.0 * xq.shape[0] * xq.shape[1] * nk * m.to_q.out_features

    def hooked_forward_flops(lat):
        hs = []
        for m in ou.modules():
            if isinstance(m, torch.nn.Linear):
                hs.append(m.register_forward_hook(lin_hook))
            elif isinstance(m, torch.nn.Conv2d):
                hs.append(m.register_forward_hook(conv_hook))
            elif m.__class__.__name__ == "Attention":
                hs.append(m.register_forward_hook(attn_hook, with_kwargs=True))
        flops[0] = 0.0
        x = torch.randn(2, 4, lat, lat)
        add = None
        if xl:
            add = {"text_embeds": torch.randn(2, pdim), "time_ids": torch.tensor([[8.0 * lat] * 2 + [0, 0] + [8.0 * lat] * 2] * 2)}
        t0 = time.time()
        with torch.no_grad():
            ou(x, 499.0, torch.randn(2, 77, ocfg.cross_attention_dim), add)
        dt = time.time() - t0
        for h in hs:
            h.remove()
        return flops[0], dt

    # probe (also the warm-up): one doubled-batch forward at 32x32 latents -> forward TFLOP/s of this host
    probe_lat = 16 if model == "tiny_sdxl" else 32
    hooked_forward_flops(probe_lat)
    pf, pdt = hooked_forward_flops(probe_lat)
    rate = pf / pdt
    lat = probe_lat
    for cand in ([64] if model != "tiny_sdxl" else []):
        f_c = pf * (cand / probe_lat) ** 2 * 1.15  # attention grows faster than the pixel count
        if 5.3 * f_c / rate <= seconds_budget:     # step ~ 4 forwards + backward (~1.3 forward)
            lat = cand
    fwd_flops, _ = hooked_forward_flops(lat)        # FLOPs of ONE forward of the doubled batch (2 samples)
    if xl:
        xlc["time_ids"] = torch.tensor([[8.0 * lat, 8.0 * lat, 0.0, 0.0, 8.0 * lat, 8.0 * lat]])
    denoised = torch.randn(1, 4, lat, lat)
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    t0 = time.time()
    loss, _ = R.slider_step(ou, net, sched, denoised, t, emb, 1, "enhance", 4.0, cfg_scale=1.0, xl=xlc)
    t_fwd = time.time() - t0
    loss.backward()
    t_bwd = time.time() - t0 - t_fwd
    if xl:
        R.clip_grad_norm_(params, 0.2)
    with torch.no_grad():
        for p, mm, vv in zip(params, m, v):
            R.adamw_step_(p, p.grad, mm, vv, 1, 1e-4, weight_decay=1e-6)
    dt = time.time() - t0
    # algorithmic FLOPs of the timed step with the engine's counting rule: 4 forwards + backward (dX only; the
    # benchmarked step's ratio backward/forward is taken from its own profile: step = 4F + Bk)
    sample_step_flops = fwd_flops * (4.0 + BWD_OVER_FWD.get(model, 1.12))
    steps_per_s_sample = 1.0 / dt
    same_size = (lat * 8 == gpu_res) and gpu_batch == 1
    if same_size:  # the identical workload on both sides (SD-1.x 512^2 B = 1): nothing is extrapolated
        return {
            "value": steps_per_s_sample, "unit": "steps/s", "cores": nthreads, "kind": "port",
            "host": {"cpu": cpu_name, "physical_cores": phys, "usable_logical_cpus": usable, "threads_used": nthreads},
            "measured": {"steps_per_s": steps_per_s_sample, "seconds_per_step": dt, "forward_s": t_fwd,
                         "backward_s": t_bwd, "resolution": lat * 8, "batch": 1, "unet_batch": 2,
                         "tflop_per_step": sample_step_flops / 1e12, "tflops": sample_step_flops / dt / 1e12},
            "sample": f"ONE whole oracle step (oracle/slider_ref.slider_step: 3 frozen + 1 adapted UNet forward, loss, "
                      f"backward, clip, AdamW; fp32, {model}, rank {lrank}) at the SAME size as the GPU line "
                      f"({gpu_res}x{gpu_res} px, B = 1, UNet batch 2): {dt:.1f} s on {nthreads} threads of '{cpu_name}' "
                      f"({phys} physical cores on the box, {usable} usable); `value` is measured, not extrapolated",
        }
    return {
        "value": steps_per_s_sample * sample_step_flops / step_flops, "unit": "steps/s", "cores": nthreads,
        "kind": "port",
        "host": {"cpu": cpu_name, "physical_cores": phys, "usable_logical_cpus": usable, "threads_used": nthreads},
        "measured": {"steps_per_s": steps_per_s_sample, "seconds_per_step": dt, "forward_s": t_fwd, "backward_s": t_bwd,
                     "resolution": lat * 8, "batch": 1, "unet_batch": 2,
                     "tflop_per_step": sample_step_flops / 1e12, "tflops": sample_step_flops / dt / 1e12},
        "sample": f"ONE whole oracle step (oracle/slider_ref.slider_step: 3 frozen + 1 adapted UNet forward, loss, "
                  f"backward, clip, AdamW; fp32, {model}, rank {lrank}) at {lat * 8}x{lat * 8} px, B = 1 (UNet batch 2): "
                  f"{dt:.1f} s = {sample_step_flops / 1e12:.2f} TFLOP at {sample_step_flops / dt / 1e12:.3f} TFLOP/s on "
                  f"{nthreads} threads of '{cpu_name}' ({phys} physical cores on the box, {usable} usable); `value` is "
                  f"EXTRAPOLATED to the benchmarked {gpu_res}x{gpu_res} step by the algorithmic FLOP ratio "
                  f"({step_flops / 1e12:.1f} TFLOP per step)",
    }


# backward / forward algorithmic FLOPs (dX only, attention counted twice): SD-XL 7.545 / 6.761, SD-1.x 0.929 / 0.803
# (SURVEY.md section 8d); the sample's FLOPs are counted as forward x (4 + this), the same rule bench's profile uses
BWD_OVER_FWD = {"sdxl": 7.545 / 6.761, "sd1x": 0.929 / 0.803}


def image_workload(args, world, rank, rank_devices, unet, net, sched, cfg, model, res, B, lrank, dt_name, lr, wd):
    """BASELINE configs[4]: the SD-XL image-slider step (trainscripts/imagesliders/train_lora-scale-xl.py:212-381) --
    per step and GPU: VAE-encode the before/after image pair (latent_dist.sample x scaling_factor), add_noise with one
    seed, then slider +s on the `high` image and -s on the `low` one (two adapted UNet forwards at UNet batch 2B, two
    backward passes accumulating one LoRA gradient), all-reduce, AdamW.  Inputs (two [B, 3, res, res] images in [-1, 1],
    prompt embeddings) are resident in HBM before the timed region; ranks take different pairs."""
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.vae as PV
    from sliders_conceptmod_amd.step import ImageSliderStep
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[dt_name]
    xl = cfg.addition_embed_type == "text_time"
    vcfg = PV.sdxl_vae_config() if model == "sdxl" else PV.VAEConfig(block_out_channels=(64, 128, 128, 128),
                                                                      norm_num_groups=16)
    with torch.device("cuda"):
        vae = PV.AutoencoderKL(vcfg).to(dtype)
    init_synthetic_on_device(vae, seed=5)
    vae.requires_grad_(False).eval()
    g = torch.Generator().manual_seed(40 + rank)
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in ("positive", "neutral", "unconditional")}
    pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
    pooled = {k: torch.randn(1, pdim, generator=g) for k in emb}
    time_ids = torch.tensor([[float(res), float(res), 0.0, 0.0, float(res), float(res)]])
    step = ImageSliderStep(unet, net, sched, lr=lr, weight_decay=wd)
    cpos, cneu = (step.make_conditioning(emb[k], B, pooled[k], time_ids, uncond=emb["unconditional"],
                                         uncond_pooled=pooled["unconditional"]) for k in ("positive", "neutral"))
    img = {k: (torch.rand(B, 3, res, res, generator=torch.Generator().manual_seed(50 + 2 * rank + i)) * 2 - 1).cuda()
           for i, k in enumerate(("low", "high"))}
    timestep = sched.timesteps[500]
    sf = vae.config.scaling_factor
    noise = torch.randn(B, 4, res // 8, res // 8, generator=torch.Generator().manual_seed(60 + rank)).cuda()
    t_vae = [0.0]

    def one_step(timed_vae=False):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if timed_vae else None
        if ev:
            ev[0].record()
        # I/train_util.py:213-222 per image; both images in ONE batched encode, sampled per image in the reference's order
        # (train_util.get_noisy_image_pair, what train_lora_scale_xl.py runs)
        keys = list(img)
        dist = vae.encode(torch.cat([img[k] for k in keys])).latent_dist
        lat = {k: sf * dist.rows(i * B, (i + 1) * B).sample(None) for i, k in enumerate(keys)}
        noised = {k: sched.add_noise(lat[k], noise, timestep.reshape(1)) for k in img}  # same seed -> same noise (:224-247)
        if ev:
            ev[1].record()
        out = step.train_step(noised["low"], noised["high"], noise, noise, timestep, cpos, cneu, 1.0)
        if ev:
            torch.cuda.synchronize()
            t_vae[0] = ev[0].elapsed_time(ev[1])
        return out

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(max(args.warmup, 1)):
        one_step()
        if i == 0:
            torch.cuda.synchronize()
            if rank == 0:
                log("first step done (engines created, weights packed, GEMM tiles tuned)")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        te = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        tl = [torch.zeros_like(te) for _ in range(world)]
        torch.distributed.all_gather(tl, te)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in tl]
        elapsed = max(float(x.item()) for x in tl)
    engine = unet._engine
    engine.profile_enable(True)
    one_step(timed_vae=True)
    prof = engine.profile_read()
    engine.profile_enable(False)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        step_flops = algorithmic_step_flops(prof)
        mm_ms = prof["gemm"]["ms"] + prof["conv"]["ms"]
        mm_fl = prof["gemm"]["flops"] + prof["conv"]["flops"]
        mm_la = prof["gemm"]["launches"] + prof["conv"]["launches"]
        achieved = mm_fl / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0
        out = {
            "metric": "slider train-steps/sec (image slider: 2 adapted UNet fwd+bwd + 2 VAE encodes)",
            "value": world * args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dt_name, "data": "synthetic",
            "config": {"workload": args.config, "unet": model, "resolution": res, "per_gpu_pairs": B, "unet_batch": 2 * B,
                       "lora_rank": lrank, "train_method": "noxattn", "lora_params": int(net.flat.numel()),
                       "scales": "1,-1", "scheduler": "ddim", "parallelism": f"dp{world}",
                       "step": "VAE-encode 2 images + add_noise, slider +s fwd+bwd on `high`, slider -s fwd+bwd on `low`, "
                               "all-reduce, AdamW (I/train_lora-scale-xl.py:212-381; the two frozen passes whose results "
                               "its loss never uses are not run)"},
            "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            "rccl_world_size": world, "rank_devices": rank_devices,
            "losses_high_low": [float(v) for v in losses.tolist()],
            "vae_encode_and_noise_ms": t_vae[0],
            "step_algorithmic_tflop_unet": step_flops / 1e12,
            "roofline": {"kernel": "gemm_5ph_kernel + gemm_8ph_kernel + gemm_glds_kernel (UNet Linear GEMMs + implicit-GEMM 3x3 convs)",
                         "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": None,
                         "launches_per_step": int(mm_la), "avg_launch_us": mm_ms * 1e3 / max(mm_la, 1),
                         "algorithmic_tflop_per_step": mm_fl / 1e12},
            "kernel_classes": {k: {"ms": round(v["ms"], 3), "launches": int(v["launches"]),
                                   "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] else None,
                                   "gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 and v["bytes"] else None}
                               for k, v in prof.items()},
            "cpu_baseline": None,
            "cpu_baseline_note": "the CPU oracle step is timed on the text-slider workloads (sd14_512_b1_r4 at the same size)",
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="sdxl_1024_b2_r4")
    ap.add_argument("--dedup-uncond", action="store_true",
                    help="NON-headline: run each distinct frozen sample once (the unconditional half is shared by the "
                         "three frozen passes): identical results, 8B -> 5B UNet samples per step")
    ap.add_argument("--skip-dead-cfg-half", action="store_true",
                    help="drop the algebraically dead unconditional half (CFG scale 1); reported separately")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="counter-collection passes (tools/collect_profiles.sh): warm-up + timed steps only -- no pre-roll, no "
                         "HIP-event profiled step (two event packets around every launch), no CPU baseline, no JSON line")
    ap.add_argument("--separate-passes", action="store_true",
                    help="run the four guidance passes as four UNet calls instead of one batched pass")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched through torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # rehearsal on a one-GPU box (tests only): SMI_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo, which
    # RCCL would refuse (duplicate GPU); the measured path is always one rank per GPU over RCCL
    rehearsal = os.environ.get("SMI_BENCH_ONE_DEVICE") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # --gpus N is a contract, not a hint: the RCCL-seen world must be N, one rank per DISTINCT device
    seen_world = torch.distributed.get_world_size() if world > 1 else 1
    if seen_world != args.gpus:
        print(f"bench.py --gpus {args.gpus}: the process group has {seen_world} rank(s); launch with "
              f"torch.distributed.run --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    rank_devices = None
    if world > 1:
        import socket
        props = torch.cuda.get_device_properties(local_rank)
        ident = f"{socket.gethostname()}:{getattr(props, 'uuid', None) or getattr(props, 'pci_bus_id', local_rank)}:{local_rank}"
        rank_devices = [None] * world
        torch.distributed.all_gather_object(rank_devices, ident)
        if not rehearsal and len(set(rank_devices)) != world:
            print(f"bench.py: ranks share a device {rank_devices}: one process per GPU is required", file=sys.stderr)
            sys.exit(2)

    from sliders_conceptmod_amd import build as smi_build
    if rank == 0:
        smi_build.build()
    if world > 1:
        torch.distributed.barrier()
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.unet as PU
    from sliders_conceptmod_amd.step import SliderStep

    model, res, B, lrank, dt_name, sched_name, max_norm, lr, wd = CONFIGS[args.config]
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[dt_name]
    cfg = unet_cfg(model)
    xl = cfg.addition_embed_type == "text_time"
    with torch.device("cuda"):
        unet = PU.UNet2DConditionModel(cfg).to(dtype)
    init_synthetic_on_device(unet, seed=0)
    unet.requires_grad_(False).eval()
    if rank == 0:
        log(f"{args.config}: weights initialised ({sum(p.numel() for p in unet.parameters()) / 1e6:.0f} M params)")
    torch.manual_seed(1)
    c3lier = args.config.endswith("_c3lier")
    targets = list(L.DEFAULT_TARGET_REPLACE) + (list(L.UNET_TARGET_REPLACE_MODULE_CONV) if c3lier else [])
    dora = args.config.endswith("_dora")
    if dora:
        from sliders_conceptmod_amd.dora import DoRANetwork
        net = DoRANetwork(unet, rank=lrank, multiplier=1.0, alpha=1.0, train_method="noxattn").to("cuda")
    else:
        net = L.LoRANetwork(unet, rank=lrank, multiplier=1.0, alpha=1.0, train_method="noxattn",
                            target_replace=targets).to("cuda")
    with torch.no_grad():  # non-zero up weights so no kernel can short-circuit
        net.flat_up.copy_(torch.randn(net.flat_up.shape, generator=torch.Generator().manual_seed(2)) * 1e-2)
    sched = MU.create_noise_scheduler(sched_name)
    sched.set_timesteps(1000)
    timestep = sched.timesteps[500]  # t = 499

    if args.config in IMAGE_WORKLOADS:
        return image_workload(args, world, rank, rank_devices, unet, net, sched, cfg, model, res, B, lrank, dt_name, lr, wd)

    g = torch.Generator().manual_seed(4 + rank)
    keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if xl else [])
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in keys}
    pooled = time_ids = None
    if xl:
        pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        pooled = {k: torch.randn(1, pdim, generator=g) for k in keys}
        time_ids = torch.tensor([[float(res), float(res), 0.0, 0.0, float(res), float(res)]])
    step = SliderStep(unet, net, sched, lr=lr, weight_decay=wd, max_grad_norm=max_norm, cfg_scale=1.0,
                      skip_dead_cfg_half=args.skip_dead_cfg_half, batch_passes=not args.separate_passes,
                      dedup_uncond=args.dedup_uncond)
    cond = step.make_conditioning(emb, B, pooled, time_ids)
    lat = res // 8
    denoised = torch.randn(B, 4, lat, lat, generator=torch.Generator().manual_seed(3 + rank)).cuda()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # one-time set-up that is not part of any step: the engine is created (weights packed into the workspace) and the
    # GEMM tile choices are measured on first use.  With --warmup >= 1 that happens inside the first warm-up step; with
    # --warmup 0 a set-up step is run anyway so the K timed steps never contain it.
    setup_steps = 1 if args.warmup == 0 else 0
    for i in range(args.warmup + setup_steps):
        step.train_step(denoised, timestep, cond, "enhance", 4.0)
        if i == 0:
            torch.cuda.synchronize()
            if rank == 0:
                log("first step done (engine created, weights packed, GEMM tiles tuned)")
    sync()
    if rank == 0:
        log("warm-up done; timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step.train_step(denoised, timestep, cond, "enhance", 4.0)
    sync()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    allreduce_us = None
    if world > 1:
        te = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        tl = [torch.zeros_like(te) for _ in range(world)]
        torch.distributed.all_gather(tl, te)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in tl]
        elapsed = max(float(x.item()) for x in tl)  # MAX over ranks
        # the step's one collective on its own: HIP events around the flat fp32 LoRA-gradient all-reduce
        from sliders_conceptmod_amd import parallel
        buf = step.msg.clone()  # [gradient | loss scalar]: the step's one message
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        parallel.allreduce_mean_(buf)
        torch.cuda.synchronize()
        torch.distributed.barrier()
        ev[0].record()
        for _ in range(10):
            parallel.allreduce_mean_(buf)
        ev[1].record()
        torch.cuda.synchronize()
        allreduce_us = ev[0].elapsed_time(ev[1]) * 1e3 / 10
    loss_val = float(loss.item())
    if args.timed_only:
        if rank == 0:
            log(f"timed {args.steps} steps: {elapsed / args.steps * 1e3:.1f} ms/step (--timed-only: done)")
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    if rank == 0:
        log(f"timed {args.steps} steps: {elapsed / args.steps * 1e3:.1f} ms/step; profiling one step")

    # ---- second number: the no-grad pre-roll the reference runs before every step
    # (diffusion / diffusion_xl, adaptor on, CFG batch 2B), with timesteps_to fixed at its mean
    n_pre = 6 if xl else 25
    pre_sched = MU.create_noise_scheduler(sched_name)
    pre_sched.set_timesteps(12 if xl else 50)
    pre_step = SliderStep(unet, net, pre_sched, cfg_scale=1.0)
    torch.manual_seed(7)
    pre_step.preroll(denoised, cond, 1, 3.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pre_step.preroll(denoised, cond, n_pre, 3.0)
    torch.cuda.synchronize()
    preroll_ms = (time.perf_counter() - t0) * 1e3
    # the SD-XL trainer's DEFAULT pre-roll runs at guidance scale train.cfg = 1.0 (T/config_util.py:41), where the
    # unconditional half of the doubled batch is algebraically dead and SliderStep.preroll runs the conditional half alone
    preroll_cfg1_ms = None
    if xl:
        pre_step.preroll(denoised, cond, 1, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pre_step.preroll(denoised, cond, n_pre, 1.0)
        torch.cuda.synchronize()
        preroll_cfg1_ms = (time.perf_counter() - t0) * 1e3
    if os.environ.get("SMI_BENCH_PROFILE_PREROLL") in ("1", "g1"):  # per-shape table of ONE pre-roll forward (with SMI_PROF_DUMP=1)
        unet._engine.profile_enable(True)                             # "g1": at guidance 1 (the conditional half alone)
        pre_step.preroll(denoised, cond, 1, 1.0 if os.environ["SMI_BENCH_PROFILE_PREROLL"] == "g1" else 3.0)
        pp = unet._engine.profile_read()
        unet._engine.profile_enable(False)
        if rank == 0:
            log("one pre-roll forward by class: " + json.dumps({k: round(v["ms"], 3) for k, v in pp.items()}))
    if rank == 0:
        log(f"pre-roll of {n_pre} forwards: {preroll_ms:.1f} ms")

    # ---- separate profiled step: per-kernel-class device time from HIP events on the launch stream
    engine = unet._engine
    engine.profile_enable(True)
    step.train_step(denoised, timestep, cond, "enhance", 4.0)
    prof = engine.profile_read()
    engine.profile_enable(False)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        step_flops = algorithmic_step_flops(prof)
        mm_ms = prof["gemm"]["ms"] + prof["conv"]["ms"]
        mm_fl = prof["gemm"]["flops"] + prof["conv"]["flops"]
        mm_la = prof["gemm"]["launches"] + prof["conv"]["launches"]
        achieved = mm_fl / (mm_ms * 1e-3) / 1e12 if mm_ms > 0 else 0.0
        out = {
            "metric": "slider train-steps/sec (4-pass UNet fwd+bwd)",
            "value": world * args.steps / elapsed,
            "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dt_name, "data": "synthetic",
            "config": {"workload": args.config, "unet": model, "resolution": res, "per_gpu_batch": B,
                       "unet_batch": B if args.skip_dead_cfg_half else 2 * B, "global_batch": B * world,
                       "lora_rank": lrank, "train_method": "noxattn", "network_type": "dora" if dora else ("c3lier" if c3lier else "lierla"),
                       "lora_params": int(net.flat.numel()),
                       "scheduler": sched_name, "parallelism": f"dp{world}", "pre_roll": "excluded",
                       "skip_dead_cfg_half": bool(args.skip_dead_cfg_half),
                       "dedup_uncond": bool(args.dedup_uncond),
                       "guidance_passes": "4 separate UNet calls" if args.separate_passes else
                       "1 batched UNet call (3 frozen + 1 adapted sub-batches)"},
            "samples_per_s": args.steps * B * world / elapsed,
            "per_rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
            "rccl_world_size": seen_world, "rank_devices": rank_devices,
            "allreduce": None if allreduce_us is None else {
                "us": allreduce_us, "bytes": int(step.msg.numel() * 4),
                "what": "flat fp32 LoRA gradient + the loss scalar in ONE message, all-reduce(sum) + divide, on the compute stream (RCCL), issued after "
                        "the deferred grouped weight-gradient launches of the backward: nothing overlaps it (it is "
                        "~0.1 % of a step)"},
            "loss": loss_val,
            "preroll": {"forwards": n_pre, "ms": preroll_ms, "unet_batch": 2 * B,
                        "note": "no-grad diffusion(_xl) pre-roll at its mean length, adaptor on, guidance scale 3 (doubled "
                                "batch); NOT part of `value`",
                        "ms_at_guidance_1": preroll_cfg1_ms,
                        "note_guidance_1": "the SD-XL trainer's default train.cfg = 1.0: the conditional half alone "
                                           "(UNet batch B), the unconditional half being algebraically dead there",
                        "steps_per_s_with_preroll": world / ((ms_per_step + preroll_ms) * 1e-3)},
            "step_algorithmic_tflop": step_flops / 1e12,
            "step_tflops_achieved": step_flops / 1e12 / (ms_per_step * 1e-3),
            "step_frac_of_mfma_peak": step_flops / 1e12 / (ms_per_step * 1e-3) / MFMA_PEAK_TFLOPS,
            "roofline": {"kernel": "gemm_5ph_kernel + gemm_8ph_kernel + gemm_glds_kernel (Linear GEMMs + implicit-GEMM 3x3 convs)",
                         "bound": "mfma",
                         "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(args.config),
                         "traffic_note": "bytes/launch at the L2's memory side (FETCH_SIZE x2 + WRITE_SIZE, KiB -> B), "
                                         "launch-weighted over the GEMM kernels; from the committed rocprofv3 --pmc "
                                         "passes in profiles/ (not collected live)",
                         "launches_per_step": int(mm_la), "avg_launch_us": mm_ms * 1e3 / max(mm_la, 1),
                         "algorithmic_tflop_per_step": mm_fl / 1e12,
                         "share_of_step_device_time": mm_ms / max(sum(v["ms"] for v in prof.values()), 1e-9)},
            "kernel_classes": {k: {"ms": round(v["ms"], 3), "launches": int(v["launches"]),
                                   "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 and v["flops"] else None,
                                   "gbs": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 and v["bytes"] else None}
                               for k, v in prof.items()},
            "kernel_classes_note": "HIP events around every launch of ONE separately profiled step: the events add about 2 % "
                                   "(the classes sum to more than ms_per_step); the rocprofv3 kernel trace cut into "
                                   "steady-state steps (profiles/r*_trace_steps_*.json) is the better per-class source",
        }
        # weak scaling: a unit is one 4-pass step on one per-GPU batch B; all ranks together process world*steps units
        if world == 1 and not args.no_cpu_baseline:
            log("GPU part done; timing the CPU oracle sample: " + json.dumps({k: out[k] for k in ("value", "ms_per_step")}))
            try:
                out["cpu_baseline"] = cpu_baseline(model, lrank, step_flops, res, B)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
