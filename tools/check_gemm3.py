"""Race / correctness screen for the 8-phase GEMM (SMI_GEMM=8ph): every shape is run many times in one process and
compared (a) with an fp32 torch matmul, (b) bit-for-bit with its own first run."""
import ctypes as C, os, sys, torch
os.environ.setdefault("SMI_GEMM", "8ph")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sliders_conceptmod_amd import _native
lib = _native.lib()
P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
torch.manual_seed(0)
bad = 0
shapes = [(256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 512, 256), (1000, 520, 640), (4096, 1280, 1280),
          (16384, 3840, 1280), (65536, 640, 640), (16384, 1280, 5120), (4096, 4096, 4096), (300, 264, 2048),
          (65536, 1920, 640), (8192, 8192, 1024)]
for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
    for (M, N, K) in shapes:
        a = torch.randn(M, K, device="cuda").to(dt)
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
        bias = torch.randn(N, device="cuda").to(dt)
        res = torch.randn(M, N, device="cuda").to(dt)
        ref = (a.float() @ w.float().t() + bias.float() + res.float())
        first = None
        nrep = 30 if M * N * K < 2e11 else 8
        for rep in range(nrep):
            c = torch.empty(M, N, device="cuda", dtype=dt)
            rc = lib.smi_op_gemm(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), None, None, 0, 0.0, 0, None)
            assert rc == 0, _native.last_error()
            torch.cuda.synchronize()
            if first is None:
                first = c.clone()
                err = float((c.float() - ref).abs().max() / ref.abs().max())
                tol = 2e-3 if dt == torch.float16 else 1.2e-2
                ok = err < tol
                print(f"{str(dt):15s} {M}x{N}x{K}: rel err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
                bad += 0 if ok else 1
            elif not torch.equal(c, first):
                nd = int((c != first).sum())
                print(f"   run {rep}: {nd} elements differ from the first run -> RACE", flush=True)
                bad += 1
                break
# implicit-GEMM 3x3 convs (NHWC activations, [Cout, 9 Cin] tap-major filters), incl. ragged M and image borders
for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
    for (nb, H, W, Cin, Cout) in [(16, 64, 64, 64, 256), (3, 40, 56, 128, 520), (16, 128, 128, 320, 320),
                                  (16, 32, 32, 1280, 1280), (5, 33, 47, 192, 1024)]:
        x = torch.randn(nb, H, W, Cin, device="cuda").to(dt)
        w = (torch.randn(Cout, 9 * Cin, device="cuda") * (9 * Cin) ** -0.5).to(dt)
        b = torch.randn(Cout, device="cuda").to(dt)
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2),
                                         w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.float(),
                                         padding=1).permute(0, 2, 3, 1)
        first = None
        for rep in range(10):
            y = torch.empty(nb, H, W, Cout, device="cuda", dtype=dt)
            rc = lib.smi_op_conv3x3(code, P(x), P(w), P(b), P(y), nb, H, W, Cin, Cout, 1, 0, 0, H, W, None)
            assert rc == 0
            torch.cuda.synchronize()
            if first is None:
                first = y.clone()
                err = float((y.float() - ref).abs().max() / ref.abs().max())
                ok = err < (2e-3 if dt == torch.float16 else 1.2e-2)
                print(f"{str(dt):15s} conv nb={nb} {H}x{W} {Cin}->{Cout}: rel err {err:.2e} {'ok' if ok else 'FAIL'}", flush=True)
                bad += 0 if ok else 1
            elif not torch.equal(y, first):
                print(f"   run {rep}: differs from the first run -> RACE", flush=True)
                bad += 1
                break
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
