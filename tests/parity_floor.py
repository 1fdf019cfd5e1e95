"""Where the fp16 parity floor comes from -- two oracle-only experiments quoted in DESIGN.md section 6 (CPU, about a
minute each on 8-16 threads; not collected by pytest):

    python tests/parity_floor.py grad  [sdxl|sd1x]   LoRA-gradient error caused by FORWARD storage rounding alone
                                                      (oracle with storage_dtype = fp16, exact fp32 autograd backward)
    python tests/parity_floor.py sites [sdxl|sd1x]   eps error of each of the oracle's rounding sites alone (ablation), and of
                                                      all of them with the transformer residual stream kept in fp32
    python tests/parity_floor.py blocks [sdxl|sd1x]  eps error of the rounding sites of ONE top-level block at a time, and
                                                      with the last up block / the top level kept exact

Real SD-1.x / SD-XL widths (synthetic weights) at 32 x 32 latents, rank 4, noxattn -- the configuration of
tests/test_fullsize_gpu.py, so the numbers sit next to the engine's measured distances."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import slider_ref as R  # noqa: E402
from oracle import unet_ref as OU  # noqa: E402


def setup(model, n):
    ocfg = {"sdxl": OU.sdxl_config, "sd1x": OU.sd1x_config}[model]()
    ou = OU.init_synthetic_(OU.UNet2DConditionModel(ocfg), seed=0).requires_grad_(False).eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, 4, 32, 32, generator=g)
    ctx = torch.randn(n, 77, ocfg.cross_attention_dim, generator=g)
    add = None
    if model == "sdxl":
        add = {"text_embeds": torch.randn(n, 1280, generator=g), "time_ids": torch.tensor([[256.0, 256, 0, 0, 256, 256]] * n)}
    return ocfg, ou, x, ctx, add, g


def grad_floor(model):
    ocfg, ou, x, ctx, add, g = setup(model, 2)
    torch.manual_seed(1)
    net = R.LoRANetworkRef(ou, 4, 1.0, 1.0, "noxattn")
    g2 = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for l in net.unet_loras:
            l.lora_up.weight.copy_(torch.randn(l.lora_up.weight.shape, generator=g2) * 0.02)
    gy = torch.randn(2, 4, 32, 32, generator=g) * 1e-4

    def grads(sd):
        ou.storage_dtype = sd
        for p in net.parameters():
            p.grad = None
        with net:
            y = ou(x, 499.0, ctx, add).sample
        (y * gy).sum().backward()
        ou.storage_dtype = None
        return y.detach(), torch.cat([p.grad.flatten() for l in net.unet_loras for p in l.parameters()])

    y0, g0 = grads(None)
    y1, g1 = grads(torch.float16)
    print(f"{model}: eps rel err from fp16 forward storage {float((y1 - y0).norm() / y0.norm()):.3e}; "
          f"LoRA-gradient rel err from FORWARD storage rounding alone {float((g1 - g0).norm() / g0.norm()):.3e}")


def sites(model):
    ocfg, ou, x, ctx, add, _ = setup(model, 1)
    counts = collections.Counter()
    only = [None]

    def tagged_q(t):
        fr = sys._getframe(1)
        slf = fr.f_locals.get("self")
        tag = (type(slf).__name__ if slf is not None else "?") + ":" + str(fr.f_lineno)
        counts[tag] += 1
        return t.to(torch.float16).to(t.dtype) if only[0] is None or only[0](tag) else t

    OU.q = tagged_q

    def run(sel):
        only[0] = sel
        with torch.no_grad():
            return ou(x, 499.0, ctx, add).sample

    ref = run(lambda tag: False)
    e_all = float((run(None) - ref).norm() / ref.norm())
    e_res = float((run(lambda tag: not tag.startswith("BasicTransformerBlock")) - ref).norm() / ref.norm())
    print(f"{model}: all sites {e_all:.3e}; all but the transformer blocks' LayerNorm / residual sites {e_res:.3e}")
    tot = 0.0
    for t in sorted(counts):
        e = float((run(lambda tag, t=t: tag == t) - ref).norm() / ref.norm())
        tot += e * e
        print(f"  {t:36s} x{counts[t]:4d}  {e:.3e}")
    print(f"  root-sum-square of the single-site errors {tot ** 0.5:.3e}")


def blocks(model):
    """eps error with fp16 storage rounding in ONE top-level block at a time (down_blocks.i / mid_block / up_blocks.i; "top" =
    conv_in, the time-embedding MLPs, conv_norm_out / conv_out), and with the last blocks kept exact -- what fp32 storage of
    "the last few tensors" could buy."""
    ocfg, ou, x, ctx, add, _ = setup(model, 1)
    cur = ["top"]

    def wrap(name, m):
        f = m.forward

        def g(*a, **k):
            prev, cur[0] = cur[0], name
            try:
                return f(*a, **k)
            finally:
                cur[0] = prev
        m.forward = g

    for i, b in enumerate(ou.down_blocks):
        wrap(f"down{i}", b)
    wrap("mid", ou.mid_block)
    for i, b in enumerate(ou.up_blocks):
        wrap(f"up{i}", b)
    counts = collections.Counter()
    only = [None]

    def tagged_q(t):
        counts[cur[0]] += 1
        return t.to(torch.float16).to(t.dtype) if only[0] is None or only[0](cur[0]) else t

    OU.q = tagged_q

    def run(sel):
        only[0] = sel
        with torch.no_grad():
            return ou(x, 499.0, ctx, add).sample

    ref = run(lambda b: False)
    print(f"{model}: all sites {float((run(None) - ref).norm() / ref.norm()):.3e}")
    for b in sorted(counts):
        print(f"  only {b:6s} x{counts[b]:4d}  {float((run(lambda bb, b=b: bb == b) - ref).norm() / ref.norm()):.3e}", flush=True)
    last = sorted(counts)[-1]
    for excl in (["top"], [last], ["top", last]):
        print(f"  all but {excl}: {float((run(lambda bb: bb not in excl) - ref).norm() / ref.norm()):.3e}", flush=True)


if __name__ == "__main__":
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    what = sys.argv[1] if len(sys.argv) > 1 else "grad"
    model = sys.argv[2] if len(sys.argv) > 2 else "sdxl"
    {"grad": grad_floor, "sites": sites, "blocks": blocks}[what](model)
