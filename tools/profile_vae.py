"""Per-class / per-shape device time of ONE VAE encode (the image-slider step runs two): SD-XL VAE encoder, 1024 x 1024.

    SMI_PROF_DUMP=1 python tools/profile_vae.py [res] [batch]      # on the GPU box"""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
import sliders_conceptmod_amd.vae as PV
from bench import init_synthetic_on_device

res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dtype = torch.float16
with torch.device("cuda"):
    vae = PV.AutoencoderKL(PV.sdxl_vae_config()).to(dtype)
init_synthetic_on_device(vae, seed=5)
vae.requires_grad_(False).eval()
img = torch.rand(n, 3, res, res, device="cuda", dtype=dtype) * 2 - 1
for _ in range(2):
    vae.encode(img)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5):
    vae.encode(img)
e.record()
torch.cuda.synchronize()
print(f"VAE encode {n} x 3 x {res} x {res}: {s.elapsed_time(e) / 5:.2f} ms")
eng = vae._engine(n, res, res)
lib = _native.lib()
_native.check(lib.smi_profile_enable(eng.handle, 1), "profile")
vae.encode(img)
torch.cuda.synchronize()
k = len(_native.Engine.PROF_CLASSES)
ms, fl, by = (C.c_double * k)(), (C.c_double * k)(), (C.c_double * k)()
la = (C.c_int64 * k)()
_native.check(lib.smi_profile_read(eng.handle, ms, fl, by, la), "read")
print(json.dumps({c: {"ms": round(ms[i], 3), "launches": la[i], "TF/s": round(fl[i] / max(ms[i], 1e-9) / 1e9, 1),
                      "TB/s": round(by[i] / max(ms[i], 1e-9) / 1e9, 2)} for i, c in enumerate(_native.Engine.PROF_CLASSES)}))
