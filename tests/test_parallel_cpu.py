"""World-size-2 data parallelism on CPU (gloo): the product's DP helpers (shard_slice / allreduce_mean_) with the
CPU oracle as the compute.  Checks the determinism contract of SURVEY.md section 8e: two ranks on a sharded global
batch == one process on the whole batch."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sched_ref as S
from oracle import slider_ref as R
from oracle import unet_ref as OU
from sliders_conceptmod_amd import parallel


def _grads(rank, world, global_b=2):
    torch.set_num_threads(2)
    cfg = OU.tiny_sd1x_config()
    unet = OU.init_synthetic_(OU.UNet2DConditionModel(cfg), seed=0).requires_grad_(False).eval()
    torch.manual_seed(1)
    net = R.LoRANetworkRef(unet, 4, 1.0, 1.0, "noxattn")
    g = torch.Generator().manual_seed(2)
    for l in net.unet_loras:
        l.lora_up.weight.data.copy_(torch.randn(l.lora_up.weight.shape, generator=g) * 0.05)
    g4 = torch.Generator().manual_seed(4)
    emb = {k: torch.randn(1, 77, cfg.cross_attention_dim, generator=g4)
           for k in ["target", "positive", "neutral", "unconditional"]}
    sched = S.create_noise_scheduler_ref("ddim")
    sched.set_timesteps(1000)
    torch.manual_seed(99)  # identical control RNG on every rank: the GLOBAL latent batch is drawn, then sliced
    latents = R.get_initial_latents(sched, global_b, 64, 64, 1)
    sl = parallel.shard_slice(global_b, rank, world)
    local = latents[sl]
    loss, _ = R.slider_step(unet, net, sched, local, sched.timesteps[500], emb, local.shape[0], "enhance", 4.0)
    loss.backward()
    flat = torch.cat([p.grad.flatten() for l in net.unet_loras for p in l.parameters()])
    return flat, loss.detach().reshape(1)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        flat, loss = _grads(rank, world)
        assert parallel.world_info() == (rank, world)
        parallel.allreduce_mean_(flat)
        parallel.allreduce_mean_(loss)
        if rank == 0:
            torch.save({"flat": flat, "loss": loss}, out)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process_global_batch(tmp_path):
    out = str(tmp_path / "dp.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    ref_flat, ref_loss = _grads(0, 1)
    torch.testing.assert_close(got["loss"], ref_loss, rtol=1e-5, atol=1e-8)
    rel = float((got["flat"] - ref_flat).norm() / ref_flat.norm())
    assert rel < 1e-5, f"2-rank all-reduced LoRA gradient differs from the global-batch gradient: {rel:.2e}"


def test_shard_slice_and_single_rank_noop():
    assert parallel.shard_slice(16, 3, 8) == slice(6, 8)
    with pytest.raises(ValueError):
        parallel.shard_slice(3, 0, 2)
    x = torch.ones(4)
    assert parallel.allreduce_mean_(x) is x and parallel.world_info() == (0, 1)
