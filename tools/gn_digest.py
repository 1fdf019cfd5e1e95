"""SHA-256 of GroupNorm forward + backward outputs (y, dx, the saved statistics) for a list of maps, under whatever
SMI_GN_COOP / SMI_GN_FUSED_HW the process was started with: the one-launch cooperative form must give the bits of the
two-launch form (tests/test_kernels_gpu.py::test_groupnorm_one_launch_forms_are_bit_identical).  Repeats every map a few
times -- a lost or early wake-up in the per-sample barrier shows up as differing repeats -- and prints the library's
count of workgroups that gave up waiting (must be 0)."""
import ctypes as C
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native  # noqa: E402

lib = _native.lib()
P = _native.ptr


def dig(*ts):
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()[:16]


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    out = {}
    # (samples, pixels, channels, groups): SD-XL / SD-1.x maps at pre-roll and step batches, a ragged one, a many-sample one
    for nb, HW, Cc, G in ((2, 16384, 320, 32), (16, 4096, 640, 32), (2, 4096, 320, 32), (4, 1024, 1280, 32),
                          (16, 1024, 2560, 32), (3, 1000, 320, 32), (200, 64 * 5, 64, 32), (1, 16384, 960, 32)):
        for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
            for silu in (0, 1):
                x = (torch.randn(nb, HW, Cc, device="cuda", generator=g) * 1.3 + 0.2).to(dt)
                dy = torch.randn(nb, HW, Cc, device="cuda", generator=g).to(dt)
                gamma = (1 + 0.1 * torch.randn(Cc, device="cuda", generator=g)).to(dt)
                beta = (0.1 * torch.randn(Cc, device="cuda", generator=g)).to(dt)
                nchunk = 64 if HW <= 16384 else (HW + 255) // 256 + 64  # (upper bound of norm.hip's gn_num_chunks)
                n_s = 2 * nb * Cc + nb * G * 2 + nb * nchunk * G * 2 + 2 * nb * Cc + 64
                reps = []
                for _ in range(3):
                    y, dx = torch.empty_like(x), torch.empty_like(x)
                    scratch = torch.zeros(n_s, device="cuda")
                    _native.check(lib.smi_op_groupnorm(code, P(x), P(gamma), P(beta), P(y), P(dy), P(dx), P(scratch), nb, HW,
                                                       Cc, G, 1e-5, silu, None), "groupnorm")
                    reps.append(dig(y, dx, scratch[:2 * nb * Cc + nb * G * 2]))
                assert len(set(reps)) == 1, ("repeats differ", nb, HW, Cc, dt, silu, reps)
                out[f"{nb}x{HW}x{Cc}_{'f16' if code == 0 else 'bf16'}_silu{silu}"] = reps[0]
    out["timeouts"] = lib.smi_gn_coop_timeouts()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
