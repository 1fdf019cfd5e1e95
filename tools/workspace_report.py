"""Workspace (packed weights + activation arenas) the engine plans for the BASELINE configurations and for the author's
own driver setting (`batch_size: 12` at 512^2, reference conceptmod/textsliders/make_config.py:17-20), from the engine's dry
run (`smi_workspace_bytes` / `smi_arena_bytes`: no GPU needed).  Batched 4-pass step: UNet batch 8B, of which 2B adapted."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
import sliders_conceptmod_amd.lora as L
import sliders_conceptmod_amd.unet as PU

for name, cfg, res, B, rank in (("C1 SD-1.4 512^2 B=1", PU.sd1x_config(), 512, 1, 4), ("C2 SD-1.5 512^2 B=4", PU.sd1x_config(), 512, 4, 4),
                                ("C3 SD-XL 1024^2 B=2", PU.sdxl_config(), 1024, 2, 4), ("C4 SD-XL 1024^2 B=2 rank 8", PU.sdxl_config(), 1024, 2, 8),
                                ("SD-XL 512^2 batch_size 12", PU.sdxl_config(), 512, 12, 4), ("SD-1.x 512^2 batch_size 12", PU.sd1x_config(), 512, 12, 4),
                                ("C5 SD-XL image slider 1024^2, 1 pair", PU.sdxl_config(), 1024, 0, 4)):
    with torch.device("meta"):
        unet = PU.UNet2DConditionModel(cfg)
    net = L.LoRANetwork(unet, rank=rank, alpha=1.0, train_method="noxattn")
    cc = _native.make_config(cfg, torch.float16)
    n, na = (4, 4) if B == 0 else (8 * B, 2 * B)  # image slider: both sides of one pair in one pass (2 CFG pairs)
    tot = _native.workspace_bytes(cc, net.engine_sites(), n, res // 8, res // 8, 77, na)
    arena = _native.arena_bytes(cc, net.engine_sites(), n, res // 8, res // 8, 77, na)
    print(f"{name:40s} UNet batch {n:3d} ({na:2d} adapted): workspace {tot / 1e9:7.1f} GB = packed weights {(tot - arena) / 1e9:5.1f} GB "
          f"+ arenas {arena / 1e9:7.1f} GB   [{'fits' if tot < 270e9 else 'DOES NOT FIT'} 288 GB HBM]")
