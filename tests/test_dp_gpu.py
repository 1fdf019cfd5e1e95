"""The SURVEY.md section 8(e) determinism contract ON THE HIP PATH: W ranks, each running the product step on its
shard of the global batch, all-reducing the flat LoRA gradient before the clip, must land on the same gradient and
the same updated parameters as ONE rank running the global batch -- up to fp32 reduction order.

Two ranks share this box's single GPU (gloo carries the CUDA tensors through the host; RCCL refuses two ranks on one
device).  The collective call sites are the product's own (step.SliderStep / parallel.allreduce_mean_); on an 8-GPU
node the same code runs over RCCL."""
import dataclasses
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import unet_ref as OU


def _build(rank_lora=4):
    import sliders_conceptmod_amd.lora as L
    import sliders_conceptmod_amd.unet as PU
    ocfg = OU.tiny_sdxl_config()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0)
    pu = PU.UNet2DConditionModel(PU.UNetConfig(**dataclasses.asdict(ocfg)))
    pu.load_state_dict(ou.state_dict())
    pu = pu.to("cuda", torch.float16).requires_grad_(False).eval()
    torch.manual_seed(1)
    net = L.LoRANetwork(pu, rank=rank_lora, alpha=1.0, train_method="noxattn").to("cuda")
    with torch.no_grad():
        net.flat_up.copy_(torch.randn(net.flat_up.shape, generator=torch.Generator().manual_seed(2)) * 2e-2)
    return pu, net


def _text_step(rank, world, global_b=2, steps=2):
    """SliderStep on this rank's slice of the GLOBAL latent batch (same control RNG on every rank)."""
    import sliders_conceptmod_amd.model_util as MU
    from sliders_conceptmod_amd import parallel
    from sliders_conceptmod_amd.step import SliderStep
    pu, net = _build()
    sched = MU.create_noise_scheduler("euler_a")
    sched.set_timesteps(1000)
    t = sched.timesteps[400]
    g = torch.Generator().manual_seed(4)
    keys = ["target", "positive", "neutral", "unconditional", "negative"]
    emb = {k: torch.randn(1, 77, 64, generator=g) for k in keys}
    pooled = {k: torch.randn(1, 64, generator=g) for k in keys}
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]])
    lat = torch.randn(global_b, 4, 16, 16, generator=torch.Generator().manual_seed(3))
    local = lat[parallel.shard_slice(global_b, rank, world)].cuda()
    # Adam eps >> |gradient|: the update is proportional to the gradient, so fp32 reduction-order differences are not
    # amplified into the second step (with eps = 1e-8 the first steps are sign-like)
    step = SliderStep(pu, net, sched, lr=1e-3, weight_decay=1e-6, max_grad_norm=0.2, eps=1e-3)
    cond = step.make_conditioning(emb, local.shape[0], pooled, tid)
    losses, grad1 = [], None
    for i in range(steps):
        losses.append(float(step.train_step(local, t, cond, "enhance", 2.0).item()))
        if i == 0:
            grad1 = step.grad.cpu().clone()  # all-reduced gradient of the FIRST step: same parameters on both sides
    return {"losses": torch.tensor(losses), "grad": grad1, "flat": net.flat.detach().cpu()}


def _image_step(rank, world, pairs=2):
    """Image-slider sharding (BASELINE config 5): rank r takes image pair r; both +/-s backward passes run locally
    before the exchange.  One rank accumulates all pairs and divides by their number."""
    import sliders_conceptmod_amd.model_util as MU
    from sliders_conceptmod_amd import parallel
    from sliders_conceptmod_amd.prompt_util import PromptEmbedsXL
    from sliders_conceptmod_amd.train_lora_scale_xl import image_slider_step
    pu, net = _build()
    sched = MU.create_noise_scheduler("ddim")
    sched.set_timesteps(1000)
    t = sched.timesteps[600]
    g = torch.Generator().manual_seed(5)
    emb = {k: PromptEmbedsXL(torch.randn(1, 77, 64, generator=g).cuda().half(),
                             torch.randn(1, 64, generator=g).cuda().half()) for k in ("pos", "neu")}
    lat = torch.randn(pairs, 2, 4, 16, 16, generator=g)  # [pair][low | high]
    noise = torch.randn(pairs, 4, 16, 16, generator=g)
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]]).cuda()
    mine = range(pairs)[parallel.shard_slice(pairs, rank, world)]
    for i in mine:
        nz = noise[i:i + 1]
        nl = sched.add_noise(lat[i, 0:1], nz, t).cuda()
        nh = sched.add_noise(lat[i, 1:2], nz, t).cuda()
        image_slider_step(pu, net, sched, nl, nh, nz.cuda(), nz.cuda(), t, emb["pos"], emb["neu"], tid, 2.0)
    grad = net.flat.grad / len(mine)
    parallel.allreduce_mean_(grad)
    return {"grad": grad.cpu()}


def _cli_train(rank, world, tmp):
    """The CLI code path itself (train_lora_xl.train, autograd loop with Euler-a pre-roll) at global batch 2.  Ranks
    start from DIFFERENT torch seeds on purpose: train() must put them on one control RNG (parallel.sync_control_rng)
    before the adaptor init and broadcast rank 0's parameters, or replicas diverge silently."""
    from test_train_gpu import make
    from sliders_conceptmod_amd.train_lora_xl import train
    cfg, prompts, models = make("tiny_sdxl", tmp, True)
    cfg.train.iterations = 3
    prompts[0].batch_size = 2
    torch.manual_seed(1 if rank == 0 else 999)
    import random
    random.seed(rank)
    # Adam eps >> |gradient|: updates proportional to the gradient, so fp32 reduction-order noise is not amplified
    net = train(cfg, prompts, torch.device("cuda:0"), rank=4, save_file=False, models=models,
                optimizer_kwargs={"eps": 1e-3})
    # save_file=False returns the state dict (train_lora_xl.py:384-389); the flat view is what replicas must share
    return {"grad": torch.cat([v.float().flatten() for k, v in sorted(net.items()) if "alpha" not in k])}


def _worker(rank, world, port, out, which):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if which == "cli":
            res = _cli_train(rank, world, os.path.dirname(out))
        else:
            res = (_text_step if which == "text" else _image_step)(rank, world)
        flats = [torch.zeros_like(res["grad"]) for _ in range(world)]
        dist.all_gather(flats, res["flat"] if "flat" in res else res["grad"])
        res["replicas_equal"] = torch.tensor(all(torch.equal(flats[0], f) for f in flats))
        if rank == 0:
            torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["text", "image", "cli"])
def test_two_ranks_equal_one_rank_on_the_global_batch(tmp_path, which):
    out = str(tmp_path / "dp.pt")
    port = 29600 + (os.getpid() % 1500)
    mp.spawn(_worker, args=(2, port, out, which), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    assert bool(got["replicas_equal"]), "replicas diverged: ranks hold different parameters / gradients"
    if which == "cli":
        ref = _cli_train(0, 1, str(tmp_path))
        # saved in the train dtype (fp16): one storage rounding of the final values on each side
        rel = float((got["grad"] - ref["grad"]).norm() / ref["grad"].norm())
        assert rel < 1e-3, f"2-rank CLI run vs 1-rank global batch after 3 steps: {rel:.2e}"
        return
    ref = (_text_step if which == "text" else _image_step)(0, 1)
    rel = float((got["grad"] - ref["grad"]).norm() / ref["grad"].norm())
    assert rel < 1e-5, f"{which}: 2-rank all-reduced LoRA gradient vs 1-rank global batch: {rel:.2e}"
    if which == "text":
        # per-rank losses are shard means; their all-reduced mean (what SliderStep reports) is the global-batch loss
        torch.testing.assert_close(got["losses"], ref["losses"], rtol=1e-5, atol=1e-9)
        relp = float((got["flat"] - ref["flat"]).norm() / ref["flat"].norm())
        assert relp < 1e-5, f"updated parameters after 2 steps: {relp:.2e}"


def _image_main_rank(rank, world, port, root):
    """One rank of `python -m torch.distributed.run ... -m sliders_conceptmod_amd.train_lora_scale_xl`: the environment
    torch.distributed.run would set, then the script's own main() -- which must create the process group itself and put
    the rank on its device (VERDICT r2 missing 1).  gloo + one shared GPU for the rehearsal (SMI_DIST_BACKEND)."""
    import random
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SMI_DIST_BACKEND="gloo")
    from sliders_conceptmod_amd import train_lora_scale_xl as T
    args = T.build_parser().parse_args(["--config_file", os.path.join(root, "config.yaml"), "--alpha", "1.0", "--rank", "4",
                                        "--device", "0", "--name", "img", "--folder_main", os.path.join(root, "data"),
                                        "--folders", "big, small", "--scales", "1, -1", "--fused_step"])
    torch.manual_seed(100 + rank)  # ranks start from DIFFERENT seeds: main() has to synchronise the control RNG
    random.seed(100 + rank)
    assert not dist.is_initialized()
    net = T.main(args)
    assert dist.is_initialized() and dist.get_world_size() == world
    flat = net.flat.detach().cpu()
    flats = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(flats, flat)
    if rank == 0:
        torch.save({"equal": all(torch.equal(flats[0], f) for f in flats), "moved": float(flat.abs().max())},
                   os.path.join(root, "dp_image_main.pt"))
    dist.destroy_process_group()


def test_image_slider_main_launches_multi_rank(tmp_path):
    import yaml
    g = torch.Generator().manual_seed(5)
    for folder, bias in (("big", 0.3), ("small", -0.3)):
        (tmp_path / "data" / folder).mkdir(parents=True)
        for i in range(4):
            torch.save(torch.randn(4, 16, 16, generator=g) + bias, tmp_path / "data" / folder / f"p{i}.pt")
    (tmp_path / "prompts.yaml").write_text(yaml.safe_dump([{"target": "t", "positive": "big", "neutral": "small",
                                                            "unconditional": "", "resolution": 128, "batch_size": 1}]))
    (tmp_path / "config.yaml").write_text(yaml.safe_dump({
        "prompts_file": str(tmp_path / "prompts.yaml"),
        "pretrained_model": {"name_or_path": "synthetic://tiny_sdxl"},
        "network": {"type": "lierla", "rank": 4, "alpha": 1.0, "training_method": "noxattn"},
        "train": {"precision": "float16", "noise_scheduler": "ddim", "iterations": 3, "lr": 1e-3, "optimizer": "AdamW",
                  "lr_scheduler": "constant", "max_denoising_steps": 10},
        "save": {"name": "x", "path": str(tmp_path / "out"), "per_steps": 1000}}))
    port = 29600 + ((os.getpid() + 7) % 1500)
    mp.spawn(_image_main_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "dp_image_main.pt", weights_only=True)
    assert got["equal"], "ranks launched through main() hold different LoRA parameters"
    assert got["moved"] > 0
    outs = list((tmp_path / "out").rglob("*_last.pt"))
    assert len(outs) == 1, outs  # rank 0 alone writes
