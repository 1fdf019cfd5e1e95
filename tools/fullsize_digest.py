"""Full-size slider steps of a bench.py configuration (default: the headline, SD-XL 1024^2, B = 2, rank 4; synthetic weights) -> loss, LoRA gradient and updated
parameters written to a .pt file.  Run under different SMI_GEMM settings to cross-check the kernel
generations against each other at the real shapes (tests/test_fullsize_gpu.py)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default="sdxl_1024_b2_r4")
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import bench
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.model_util as MU
    import sliders_conceptmod_amd.unet as PU
    from sliders_conceptmod_amd.step import SliderStep
    model, res, B, lrank, dt_name, sched_name, max_norm, lr, wd = bench.CONFIGS[args.config]
    dtype = {"fp16": torch.float16, "bf16": torch.bfloat16}[dt_name]
    cfg = bench.unet_cfg(model)
    xl = cfg.addition_embed_type == "text_time"
    with torch.device("cuda"):
        unet = PU.UNet2DConditionModel(cfg).to(dtype)
    bench.init_synthetic_on_device(unet, seed=0)
    unet.requires_grad_(False).eval()
    torch.manual_seed(1)
    targets = list(L.DEFAULT_TARGET_REPLACE) + (list(L.UNET_TARGET_REPLACE_MODULE_CONV) if args.config.endswith("_c3lier") else [])
    net = L.LoRANetwork(unet, rank=lrank, multiplier=1.0, alpha=1.0, train_method="noxattn", target_replace=targets).to("cuda")
    with torch.no_grad():
        net.flat_up.copy_(torch.randn(net.flat_up.shape, generator=torch.Generator().manual_seed(2)) * 1e-2)
    sched = MU.create_noise_scheduler(sched_name)
    sched.set_timesteps(1000)
    timestep = sched.timesteps[500]
    g = torch.Generator().manual_seed(4)
    keys = ["target", "positive", "neutral", "unconditional"] + (["negative"] if xl else [])
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g) for k in keys}
    pooled = time_ids = None
    if xl:
        pdim = cfg.projection_class_embeddings_input_dim - 6 * cfg.addition_time_embed_dim
        pooled = {k: torch.randn(1, pdim, generator=g) for k in keys}
        time_ids = torch.tensor([[float(res), float(res), 0.0, 0.0, float(res), float(res)]])
    step = SliderStep(unet, net, sched, lr=lr, weight_decay=wd, max_grad_norm=max_norm, cfg_scale=1.0)
    cond = step.make_conditioning(emb, B, pooled, time_ids)
    lat = res // 8
    denoised = torch.randn(B, 4, lat, lat, generator=torch.Generator().manual_seed(3)).cuda()
    losses = [float(step.train_step(denoised, timestep, cond, "enhance", 4.0).item()) for _ in range(args.steps)]
    torch.cuda.synchronize()
    torch.save({"losses": losses, "grad": step.grad.cpu(), "flat": net.flat.detach().cpu()}, args.out)
    print("digest written:", args.out, losses)


if __name__ == "__main__":
    main()
