"""Generates tests/golden/*.safetensors|json by running the REFERENCE's own harness code
(/root/reference/conceptmod/textsliders/{lora,train_util,prompt_util,config_util}.py) in this container.

Run:  python tests/golden/make_goldens.py          (needs /root/reference; never runs on the GPU box)

The reference imports `diffusers` only for type annotations in these four modules (lora.py:11,
train_util.py:7-10,14); diffusers is not installed, so a type-only stub is registered first.  No model arithmetic
comes from the stub: the UNet the reference code drives is oracle/unet_ref.py and the schedulers are
oracle/sched_ref.py (both "parity unpinned" restatements of the absent dependency).  What these fixtures pin is
everything the reference itself computes on the path: adaptor selection/naming, LoRAModule.forward, CFG batching in
predict_noise(_xl), PromptEmbedsPair.loss, the optimiser/LR-schedule helpers, config default filling, and a full
multi-step training trajectory in the reference's step order (train_lora.py:155-300 / train_lora_xl.py:165-351).

Fixtures are data only (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import sys
import types
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

warnings.filterwarnings("ignore")
# Adam eps of the "smooth" trajectories (section 7): >> the largest |gradient| of the run (recorded in the manifest)
SMOOTH_EPS = {"tiny_sd1x": 1e-3, "tiny_sdxl": 1e-3}


def install_diffusers_stub():
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            cls = type(name, (), {})
            setattr(self, name, cls)
            return cls

    for mod in ["diffusers", "diffusers.models", "diffusers.models.transformers", "diffusers.models.embeddings",
                "diffusers.utils", "diffusers.utils.torch_utils", "diffusers.pipelines", "diffusers.pipelines.flux",
                "diffusers.pipelines.flux.pipeline_flux", "diffusers.schedulers", "diffusers.image_processor",
                "diffusers.pipelines.stable_diffusion", "diffusers.pipelines.stable_diffusion.convert_from_ckpt"]:
        m = _Any(mod)
        m.__path__ = []
        sys.modules[mod] = m


def main():
    from safetensors.torch import save_file

    install_diffusers_stub()
    sys.path.insert(0, REF)
    from conceptmod.textsliders import lora as ref_lora
    from conceptmod.textsliders import train_util as ref_tu
    from conceptmod.textsliders import prompt_util as ref_pu
    from conceptmod.textsliders import config_util as ref_cu

    from oracle import unet_ref as U
    from oracle import sched_ref as S

    manifest = {}

    # ------------------------------------------------------------------ (1) adaptor selection / naming manifests
    sel = {}
    for model, cfg in [("sd1x", U.sd1x_config()), ("sdxl", U.sdxl_config()),
                       ("tiny_sd1x", U.tiny_sd1x_config()), ("tiny_sdxl", U.tiny_sdxl_config())]:
        for method in ["noxattn", "innoxattn", "selfattn", "xattn", "full", "xattn-strict"]:
            with torch.device("meta"):
                unet = U.UNet2DConditionModel(cfg)
            net = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, target_replace=["Attention"],
                                       train_method=method)
            names = [l.lora_name for l in net.unet_loras]
            shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
            nparam = sum(p.numel() for p in net.parameters())
            sel[f"{model}/{method}"] = {
                "n_modules": len(names), "n_params": nparam,
                "first": names[:6], "last": names[-3:],
                "names_sha": __import__("hashlib").sha256("\n".join(names).encode()).hexdigest(),
                "shapes_sha": __import__("hashlib").sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest(),
            }
            if model.startswith("tiny"):
                sel[f"{model}/{method}"]["names"] = names
    manifest["selection"] = sel
    try:
        with torch.device("meta"):
            unet = U.UNet2DConditionModel(U.tiny_sd1x_config())
        ref_lora.LoRANetwork(unet, train_method="bogus")
    except NotImplementedError as e:
        manifest["bad_method_error"] = type(e).__name__

    # ------------------------------------------------------------------ (2) LoRAModule forward + grads
    tensors = {}
    for seed in range(3):
        torch.manual_seed(100 + seed)
        lin = torch.nn.Linear(24, 40)
        mod = ref_lora.LoRAModule("lin", lin, multiplier=1.0, lora_dim=4, alpha=1.0 + seed)
        down0 = mod.lora_down.weight.detach().clone()
        mod.lora_up.weight.data.normal_(0, 0.05)
        mod.apply_to()
        mod.multiplier = 0.75
        x = torch.randn(3, 7, 24, requires_grad=True)
        y = lin(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        for k, v in {"w": lin.weight, "b": lin.bias, "down": mod.lora_down.weight, "down_init": down0,
                     "up": mod.lora_up.weight, "x": x, "y": y, "gy": gy, "gx": x.grad,
                     "gdown": mod.lora_down.weight.grad, "gup": mod.lora_up.weight.grad}.items():
            tensors[f"lora_linear/{seed}/{k}"] = v.detach().clone().contiguous()
        manifest[f"lora_linear/{seed}"] = {"multiplier": 0.75, "alpha": 1.0 + seed, "rank": 4, "scale": mod.scale}
    torch.manual_seed(7)
    conv = torch.nn.Conv2d(8, 12, 3, 1, 1)
    mod = ref_lora.LoRAModule("conv", conv, multiplier=1.0, lora_dim=4, alpha=2.0)
    mod.lora_up.weight.data.normal_(0, 0.05)
    mod.apply_to()
    x = torch.randn(2, 8, 6, 6, requires_grad=True)
    y = conv(x)
    gy = torch.randn_like(y)
    y.backward(gy)
    for k, v in {"w": conv.weight, "b": conv.bias, "down": mod.lora_down.weight, "up": mod.lora_up.weight, "x": x,
                 "y": y, "gy": gy, "gx": x.grad, "gdown": mod.lora_down.weight.grad,
                 "gup": mod.lora_up.weight.grad}.items():
        tensors[f"lora_conv/{k}"] = v.detach().clone().contiguous()
    manifest["lora_conv"] = {"alpha": 2.0, "rank": 4, "scale": mod.scale}

    # ------------------------------------------------------------------ (3) loss
    class _S:  # duck-typed PromptSettings
        guidance_scale = 4.0
        resolution = 512
        dynamic_resolution = False
        batch_size = 1
        dynamic_crops = False
        action = "enhance"
    torch.manual_seed(11)
    lat = {k: torch.randn(2, 4, 8, 8) for k in ["target", "positive", "neutral", "negative"]}
    for action in ["enhance", "erase"]:
        s = _S()
        s.action = action
        tgt = lat["target"].clone().requires_grad_(True)
        pair = ref_pu.PromptEmbedsPair(torch.nn.MSELoss(), None, None, None, None, None, s)
        loss = pair.loss(target_latents=tgt, positive_latents=lat["positive"], neutral_latents=lat["neutral"],
                         negative_latents=lat["negative"])
        loss.backward()
        tensors[f"loss/{action}/value"] = loss.detach().reshape(1)
        tensors[f"loss/{action}/gtarget"] = tgt.grad.clone()
    for k, v in lat.items():
        tensors[f"loss/{k}"] = v
    manifest["loss"] = {"guidance_scale": 4.0}

    # ------------------------------------------------------------------ (4) predict_noise(_xl) through the reference code
    def synth_embeds(cfg, keys, seed):
        g = torch.Generator().manual_seed(seed)
        emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in keys}
        pooled = None
        if cfg.addition_embed_type == "text_time":
            pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
            pooled = {k: torch.randn(1, pdim, generator=g) for k in keys}
        return emb, pooled

    def run_trajectory(model, cfg, sched, xl, tag, opt_kwargs, full_sd):
        """Six steps of the reference loop (train_lora.py:155-300 / train_lora_xl.py:165-351) driven through the
        reference's own helper functions on the oracle UNet."""
        torch.manual_seed(1234)
        unet = U.init_synthetic_(U.UNet2DConditionModel(cfg), seed=0)
        unet.requires_grad_(False)
        unet.eval()
        torch.manual_seed(1)  # LoRA down init uses the global RNG (lora.py:123)
        net = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, target_replace=["Attention"],
                                   train_method="noxattn")
        for l in (net.unet_loras[0], net.unet_loras[-1]):
            tensors[f"{tag}/init/{l.lora_name}.lora_down.weight"] = l.lora_down.weight.detach().clone()
        keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if xl else [])
        emb, pooled = synth_embeds(cfg, keys, seed=4)
        okw = opt_kwargs or {}
        if not xl:
            opt = ref_tu.get_optimizer("AdamW")(net.prepare_optimizer_params(), lr=2e-3, **okw)
            lrs = ref_tu.get_lr_scheduler("constant", opt, max_iterations=8, lr_min=2e-5)
        else:  # hard-coded in train_lora_xl.py:104-105 (only eps differs in the `smooth` variant of section 7)
            opt = torch.optim.AdamW(net.parameters(), lr=1e-4, weight_decay=1e-6, **okw)
            lrs = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=1e-6)
        s = _S()
        s.action = "enhance"
        s.guidance_scale = 4.0
        s.batch_size = 1
        pair = ref_pu.PromptEmbedsPair(torch.nn.MSELoss(), None, None, None, None, None, s)
        max_steps = 8
        losses, gmax = [], []
        # NOTE: no reseed here -- the loop continues the global RNG stream left by the LoRA init (seed 1 above),
        # exactly as a real run of the reference script does; the product's train() is checked against this.
        for it in range(6):
            with torch.no_grad():
                sched.set_timesteps(max_steps)
                opt.zero_grad()
                _pair_idx = torch.randint(0, 1, (1,)).item()
                timesteps_to = torch.randint(1, max_steps, (1,)).item()
                latents = ref_tu.get_initial_latents(sched, 1, 128, 128, 1)
                tid = ref_tu.get_add_time_ids(128, 128, dynamic_crops=False, dtype=torch.float32) if xl else None

                def cat(key):
                    te = ref_tu.concat_embeddings(emb["unconditional"], emb[key], 1)
                    if not xl:
                        return dict(text_embeddings=te)
                    return dict(text_embeddings=te,
                                add_text_embeddings=ref_tu.concat_embeddings(pooled["unconditional"], pooled[key], 1),
                                add_time_ids=ref_tu.concat_embeddings(tid, tid, 1))

                with net:
                    if not xl:
                        denoised = ref_tu.diffusion(unet, sched, latents, cat("target")["text_embeddings"],
                                                    start_timesteps=0, total_timesteps=timesteps_to, guidance_scale=3)
                    else:
                        denoised = ref_tu.diffusion_xl(unet, sched, latents, **cat("target"), start_timesteps=0,
                                                       total_timesteps=timesteps_to, guidance_scale=1.0)
                sched.set_timesteps(1000)
                cur_t = sched.timesteps[int(timesteps_to * 1000 / max_steps)]
                pn = ref_tu.predict_noise_xl if xl else ref_tu.predict_noise
                positive = pn(unet, sched, cur_t, denoised, **cat("positive"), guidance_scale=1)
                neutral = pn(unet, sched, cur_t, denoised, **cat("neutral"), guidance_scale=1)
                negative = pn(unet, sched, cur_t, denoised, **cat("negative" if xl else "unconditional"),
                              guidance_scale=1)
            with net:
                target = pn(unet, sched, cur_t, denoised, **cat("target"), guidance_scale=1)
            loss = pair.loss(target_latents=target, positive_latents=positive, neutral_latents=neutral,
                             negative_latents=negative)
            losses.append(loss.item())
            loss.backward()
            gmax.append(max(float(p.grad.abs().max()) for p in net.parameters()))
            if xl:
                torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=0.2)
            opt.step()
            lrs.step()
            if it == 0:
                for k, v in net.state_dict().items():
                    if "alpha" not in k:
                        pass
                tensors[f"{tag}/step0_timestep"] = torch.tensor([float(cur_t)])
                tensors[f"{tag}/step0_denoised"] = denoised.contiguous()
                tensors[f"{tag}/step0_target"] = target.detach().contiguous()
        sd = net.state_dict()
        # keep the fixture small: store three full modules + per-tensor norms of all
        keep = [k for k in sd if any(s_ in k for s_ in ("down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q",
                                                        "mid_block_attentions_0_transformer_blocks_0_attn1_to_out_0",
                                                        "up_blocks_1_attentions_2_transformer_blocks_0_attn1_to_v"))]
        if full_sd:
            keep = [k for k in sd if "alpha" not in k]
        for k in keep:
            tensors[f"{tag}/sd/{k}"] = sd[k].detach().clone().contiguous()
        manifest[tag] = {
            "losses": losses, "iterations": 6, "max_denoising_steps": max_steps,
            "norms": {k: float(v.float().norm()) for k, v in sd.items()},
            "seeds": {"lora_init_then_loop": 1, "unet": 0, "emb": 4},
            "optimizer": "AdamW lr 2e-3 constant" if not xl else "AdamW lr 1e-4 wd 1e-6 cosine(50,1e-6) clip 0.2",
        }
        if opt_kwargs:
            manifest[tag]["optimizer_kwargs"] = dict(opt_kwargs)
            manifest[tag]["grad_absmax_per_step"] = gmax


    for model, cfg, sched_name in [("tiny_sd1x", U.tiny_sd1x_config(), "ddim"),
                                   ("tiny_sdxl", U.tiny_sdxl_config(), "euler_a")]:
        xl = cfg.addition_embed_type == "text_time"
        unet = U.init_synthetic_(U.UNet2DConditionModel(cfg), seed=0)
        unet.requires_grad_(False)
        unet.eval()
        torch.manual_seed(1)  # LoRA down init draws from the global RNG (lora.py:123)
        net = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, target_replace=["Attention"],
                                   train_method="noxattn" if not xl else "full")
        g = torch.Generator().manual_seed(2)
        for l in net.unet_loras:
            l.lora_up.weight.data.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.05)
        sched = S.create_noise_scheduler_ref(sched_name)
        sched.set_timesteps(1000)
        emb, pooled = synth_embeds(cfg, ["unconditional", "target"], seed=4)
        B = 2
        g3 = torch.Generator().manual_seed(3)
        latents = torch.randn(B, 4, 16, 16, generator=g3)
        t = sched.timesteps[500]
        te = ref_tu.concat_embeddings(emb["unconditional"], emb["target"], B)
        res = {}
        for lora_on in [False, True]:
            if lora_on:
                net.__enter__()
            else:
                net.__exit__(None, None, None)
            for gs in [1.0, 3.0]:
                with torch.no_grad():
                    if not xl:
                        out = ref_tu.predict_noise(unet, sched, t, latents, te, guidance_scale=gs)
                    else:
                        tid = ref_tu.get_add_time_ids(128, 128, dynamic_crops=False, dtype=torch.float32)
                        out = ref_tu.predict_noise_xl(
                            unet, sched, t, latents, te,
                            ref_tu.concat_embeddings(pooled["unconditional"], pooled["target"], B),
                            ref_tu.concat_embeddings(tid, tid, B), guidance_scale=gs)
                res[f"lora{int(lora_on)}_g{gs}"] = out
        for k, v in res.items():
            tensors[f"predict/{model}/{k}"] = v.contiguous()
        tensors[f"predict/{model}/latents"] = latents
        manifest[f"predict/{model}"] = {"timestep": float(t), "B": B, "lora_up_seed": 2, "emb_seed": 4,
                                        "latent_seed": 3, "unet_seed": 0, "scheduler": sched_name}

        # -------------------------------------------------------------- (5) training trajectory, reference step order
        run_trajectory(model, cfg, sched, xl, f"traj/{model}", None, full_sd=False)

    # ------------------------------------------------------------------ (6) helpers: add_time_ids, LR traces, config
    tensors["add_time_ids"] = ref_tu.get_add_time_ids(1024, 768, dynamic_crops=False, dtype=torch.float32)
    traces = {}
    for name in ["cosine", "cosine_with_restarts", "step", "constant", "linear"]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = ref_tu.get_optimizer("adamw")([p], lr=2e-4)
        try:
            sch = ref_tu.get_lr_scheduler(name, opt, max_iterations=1000, lr_min=2e-6)
        except TypeError as e:  # "linear": the reference passes LinearLR(factor=...), not a torch kwarg
            manifest[f"error/lr_scheduler_{name}"] = type(e).__name__
            continue
        tr = []
        for _ in range(100):
            tr.append(sch.get_last_lr()[0])
            opt.step()
            sch.step()
        traces[name] = tr
    manifest["lr_traces"] = traces
    for bad, fn in [("optimizer", lambda: ref_tu.get_optimizer("sgd")),
                    ("lr_scheduler", lambda: ref_tu.get_lr_scheduler("bogus", None, 10, 0.0)),
                    ("precision", lambda: ref_cu.parse_precision("fp8"))]:
        try:
            fn()
        except Exception as e:  # noqa
            manifest[f"error/{bad}"] = type(e).__name__
    cfgs = {}
    for path in ["conceptmod/textsliders/data/config.yaml", "conceptmod/textsliders/data/config-xl.yaml"]:
        root = ref_cu.load_config_from_yaml(os.path.join(REF, path))
        cfgs[os.path.basename(path)] = json.loads(root.json())
    manifest["config_defaults"] = {
        "train": json.loads(ref_cu.TrainConfig().json()), "save": json.loads(ref_cu.SaveConfig().json()),
        "network": json.loads(ref_cu.NetworkConfig().json()), "shipped": cfgs,
    }
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        ps = ref_pu.load_prompts_from_yaml(os.path.join(REF, "conceptmod/textsliders/data/prompts.yaml"))
    manifest["prompts_shipped"] = [json.loads(p.json()) for p in ps]
    manifest["prompt_defaults"] = json.loads(ref_pu.PromptSettings(target="x").json())
    torch.manual_seed(5)
    manifest["random_resolution_bucket_seed5"] = [list(ref_tu.get_random_resolution_in_bucket(512)) for _ in range(8)]

    # ------------------------------------------------------------------ (7) "smooth" trajectories: the same loop with
    # Adam's eps raised far above the gradient scale, so the update is ~ lr * g / eps (proportional to the gradient)
    # instead of ~ lr * sign(g): 16-bit gradient noise is then NOT amplified and the final LoRA tensors of an fp16
    # engine run can be compared element-wise.  Full state dicts are stored.  (Placed last: fixtures (1)-(6) unchanged.)
    for model, cfg, sched_name in [("tiny_sd1x", U.tiny_sd1x_config(), "ddim"),
                                   ("tiny_sdxl", U.tiny_sdxl_config(), "euler_a")]:
        xl = cfg.addition_embed_type == "text_time"
        sched = S.create_noise_scheduler_ref(sched_name)
        run_trajectory(model, cfg, sched, xl, f"traj_smooth/{model}", {"eps": SMOOTH_EPS[model]}, full_sd=True)

    # ------------------------------------------------------------------ (8) c3lier: conv / time-embedding adaptors
    # The shipped SD-1.x config trains `type: c3lier` (data/config.yaml:7): train_lora.py:44-46 appends the conv classes to
    # the target list.  Pinned here: selection incl. the duplicate-name pass (DownBlock2D / UpBlock2D reach the same
    # convs as ResnetBlock2D, lora.py:243-249), shapes with the rank clamp, and the RNG-dependent init of modules that
    # come AFTER such duplicates (the reference builds -- and seeds -- the dropped duplicates too).
    c3 = list(ref_lora.UNET_TARGET_REPLACE_MODULE_TRANSFORMER) + [
        "ResnetBlock2D", "Downsample2D", "Upsample2D", "DownBlock2D", "UpBlock2D"]  # lora.py:19-26
    sel3 = {}
    for model, cfg in [("sd1x", U.sd1x_config()), ("sdxl", U.sdxl_config()),
                       ("tiny_sd1x", U.tiny_sd1x_config()), ("tiny_sdxl", U.tiny_sdxl_config())]:
        for method in ["noxattn", "full"]:
            with torch.device("meta"):
                unet = U.UNet2DConditionModel(cfg)
            net = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, target_replace=c3, train_method=method)
            names = [l.lora_name for l in net.unet_loras]
            shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
            sel3[f"{model}/{method}"] = {
                "n_modules": len(names), "n_params": sum(p.numel() for p in net.parameters()),
                "names_sha": __import__("hashlib").sha256("\n".join(names).encode()).hexdigest(),
                "shapes_sha": __import__("hashlib").sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest(),
            }
            if model.startswith("tiny"):
                sel3[f"{model}/{method}"]["names"] = names
    manifest["selection_c3lier"] = sel3
    for model, cfg in [("tiny_sd1x", U.tiny_sd1x_config()), ("tiny_sdxl", U.tiny_sdxl_config())]:
        unet = U.init_synthetic_(U.UNet2DConditionModel(cfg), seed=0)
        torch.manual_seed(1)
        net = ref_lora.LoRANetwork(unet, rank=4, multiplier=1.0, alpha=1.0, target_replace=c3, train_method="noxattn")
        picks = [net.unet_loras[0], net.unet_loras[len(net.unet_loras) // 2], net.unet_loras[-1]]
        picks += [l for l in net.unet_loras if "conv_shortcut" in l.lora_name][:1]
        picks += [l for l in net.unet_loras if "time_emb_proj" in l.lora_name][-1:]
        picks += [l for l in net.unet_loras if l.lora_name.endswith("samplers_0_conv")][-1:]
        for l in picks:
            tensors[f"c3lier_init/{model}/{l.lora_name}.lora_down.weight"] = l.lora_down.weight.detach().clone()
        manifest[f"c3lier_init/{model}"] = {"seed": 1, "after_init_randint": int(torch.randint(0, 2 ** 31 - 1, (1,)))}

    # ------------------------------------------------------------------ (9) DoRA (`--peft_type dora`, T/dora.py:53-162)
    from conceptmod.textsliders import dora as ref_dora
    for seed in range(2):
        torch.manual_seed(200 + seed)
        lin = torch.nn.Linear(24, 40)
        mod = ref_dora.DoRAModule("lin", lin, multiplier=1.0, lora_dim=4, alpha=1.0 + seed)
        scale0 = mod.dora_scale.detach().clone()
        mod.lora_up.weight.data.normal_(0, 0.05)
        mod.dora_scale.data.mul_(1.0 + 0.1 * torch.randn(mod.dora_scale.shape))
        mod.apply_to()
        mod.multiplier = 0.75
        x = torch.randn(3, 7, 24, requires_grad=True)
        y = lin(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        for k, v in {"w": lin.weight, "b": lin.bias, "down": mod.lora_down.weight, "up": mod.lora_up.weight,
                     "dora_scale_init": scale0, "dora_scale": mod.dora_scale, "x": x, "y": y, "gy": gy, "gx": x.grad,
                     "gdown": mod.lora_down.weight.grad, "gup": mod.lora_up.weight.grad,
                     "gscale": mod.dora_scale.grad}.items():
            tensors[f"dora_linear/{seed}/{k}"] = v.detach().clone().contiguous()
        manifest[f"dora_linear/{seed}"] = {"multiplier": 0.75, "alpha": 1.0 + seed, "rank": 4, "scale": mod.scale,
                                           "state_dict_keys": list(mod.state_dict().keys())}
    dsel = {}
    for model, cfg in [("sdxl", U.sdxl_config()), ("tiny_sdxl", U.tiny_sdxl_config()), ("tiny_sd1x", U.tiny_sd1x_config())]:
        for method in ["noxattn", "full"]:
            unet = U.UNet2DConditionModel(cfg) if model.startswith("tiny") else None
            if unet is None:
                continue
            net = ref_dora.DoRANetwork(unet, rank=4, multiplier=1.0, delimiter="_", target_replace=["Attention"],
                                       prefix="lora_unet", train_method=method)
            names = [l.lora_name for l in net.unet_loras]
            shapes = {k: list(v.shape) for k, v in net.state_dict().items()}
            dsel[f"{model}/{method}"] = {
                "n_modules": len(names), "n_params": sum(p.numel() for p in net.parameters()),
                "names_sha": __import__("hashlib").sha256("\n".join(names).encode()).hexdigest(),
                "shapes_sha": __import__("hashlib").sha256(json.dumps(shapes, sort_keys=True).encode()).hexdigest(),
                "keys_head": list(net.state_dict().keys())[:4],
            }
    manifest["selection_dora"] = dsel

    save_file(tensors, os.path.join(OUT, "harness_goldens.safetensors"))
    with open(os.path.join(OUT, "harness_goldens.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    sz = os.path.getsize(os.path.join(OUT, "harness_goldens.safetensors"))
    print(f"wrote {len(tensors)} tensors ({sz/1e3:.1f} kB) and manifest with {len(manifest)} entries")


if __name__ == "__main__":
    main()
