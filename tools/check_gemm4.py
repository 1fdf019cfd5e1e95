"""Race / correctness screen and A/B timing for the 256x320 persistent GEMM (gemm4.hip).

Every shape is run (a) on the default selection with tuning off (gemm2 / gemm3 heuristics) and (b) forced onto gemm4
(SMI_GEMM is read once per process, so the two arms are child processes writing their outputs' digests), compared with
an fp32 torch matmul, bit-for-bit between the arms, and bit-for-bit across repeated runs of the same arm.

    python tools/check_gemm4.py            # parent: runs both arms, compares digests, prints timings side by side
"""
import ctypes as C, hashlib, json, os, subprocess, sys

SHAPES = [(256, 320, 128), (256, 320, 192), (512, 640, 320), (16384, 1280, 1280), (16384, 1280, 5120),
          (16384, 3840, 1280), (65536, 640, 640), (65536, 640, 2560), (65536, 1920, 640), (4096, 1280, 1280),
          (16384, 640, 5120), (16384, 5120, 1280), (65536, 320, 960), (16384, 10240, 1280), (4096, 1280, 10240)]


QUICK = os.environ.get("SMI_CHECK_QUICK") == "1"  # the pytest screen: every mode and epilogue class, fewer big shapes / repeats


def child(arm):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sliders_conceptmod_amd import _native
    lib = _native.lib()
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    out = {}
    shapes = [sh for sh in SHAPES if not QUICK or sh in ((256, 320, 128), (512, 640, 320), (16384, 1280, 1280),
                                                        (16384, 3840, 1280), (65536, 640, 640), (4096, 1280, 1280))]
    for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
        for (M, N, K) in shapes:
            if QUICK and code == 1 and M * N * K > 1e10:
                continue
            for epi in (0, 1, 2):  # 0: plain, 1: bias + res, 2: bias + res + LoRA rank 4 on the rows >= M/2? (all rows)
                if dt == torch.bfloat16 and epi == 2:
                    continue
                g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K + epi)
                a = torch.randn(M, K, device="cuda", generator=g).to(dt)
                w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
                bias = torch.randn(N, device="cuda", generator=g).to(dt) if epi else None
                res = torch.randn(M, N, device="cuda", generator=g).to(dt) if epi else None
                xa = torch.randn(M, 4, device="cuda", generator=g) if epi == 2 else None
                up = torch.randn(N, 4, device="cuda", generator=g) if epi == 2 else None
                ref = a.float() @ w.float().t()
                if epi:
                    ref = ref + bias.float() + res.float()
                if epi == 2:
                    ref = ref + 0.5 * (xa @ up.t())
                first = None
                nrep = (12 if M * N * K < 2e11 else 5) if not QUICK else 4
                for rep in range(nrep):
                    c = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
                    rc = lib.smi_op_gemm(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up),
                                         4 if epi == 2 else 0, 0.5 if epi == 2 else 0.0, 0, None)
                    assert rc == 0, lib.smi_last_error()
                    torch.cuda.synchronize()
                    if first is None:
                        first = c.clone()
                        err = float((c.float() - ref).abs().max() / ref.abs().max())
                    elif not torch.equal(c, first):
                        err = float("inf")  # race
                        break
                # timing
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                c = torch.empty(M, N, device="cuda", dtype=dt)
                s.record()
                for _ in range(10):
                    lib.smi_op_gemm(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up),
                                    4 if epi == 2 else 0, 0.5 if epi == 2 else 0.0, 0, None)
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) * 100
                dig = hashlib.sha256(first.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
                out[f"{code}:{M}x{N}x{K}:e{epi}"] = [err, dig, us]
                print(f"[{arm}] {str(dt):15s} {M}x{N}x{K} epi={epi}: err {err:.2e}  {us:8.1f} us  "
                      f"{2*M*N*K/us/1e6:7.1f} TF/s", flush=True)
    # batched-pass form: LoRA delta on the last quarter of the rows only (gemm4's interleaved "mix" tiles), plain and
    # fused-segment (q|k|v) epilogues, rank 4 and 8
    for (M, N, K, seg, r) in [(1024, 640, 256, 0, 4), (16384, 1280, 1280, 0, 4), (16384, 3840, 1280, 1280, 4),
                              (65536, 1920, 640, 640, 8), (65536, 640, 640, 0, 4)]:
        dt, code = torch.float16, 0
        g = torch.Generator(device="cuda").manual_seed(M + N + K + r)
        row0 = 3 * M // 4
        nseg = N // seg if seg else 1
        a = torch.randn(M, K, device="cuda", generator=g).to(dt)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
        bias = torch.randn(N, device="cuda", generator=g).to(dt)
        res = torch.randn(M, N, device="cuda", generator=g).to(dt)
        xa = torch.randn(M - row0, nseg * r, device="cuda", generator=g)
        up = torch.randn(N, r, device="cuda", generator=g)
        ref = a.float() @ w.float().t() + bias.float() + res.float()
        for sg in range(nseg):
            cs = N // nseg
            ref[row0:, sg * cs:(sg + 1) * cs] += 0.5 * (xa[:, sg * r:(sg + 1) * r] @ up[sg * cs:(sg + 1) * cs].t())
        first, err = None, 0.0
        for rep in range(6):
            c = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
            rc = lib.smi_op_gemm_rows(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.5, row0, seg,
                                      None)
            assert rc == 0, lib.smi_last_error()
            torch.cuda.synchronize()
            if first is None:
                first = c.clone()
                err = float((c.float() - ref).abs().max() / ref.abs().max())
            elif not torch.equal(c, first):
                err = float("inf")
                break
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            lib.smi_op_gemm_rows(code, P(a), P(w), P(c), M, N, K, P(bias), P(res), P(xa), P(up), r, 0.5, row0, seg, None)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        dig = hashlib.sha256(first.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
        out[f"{code}:{M}x{N}x{K}:rows{seg}r{r}"] = [err, dig, us]
        print(f"[{arm}] rows-form {M}x{N}x{K} seg={seg} r={r}: err {err:.2e}  {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s",
              flush=True)
    # fused GEGLU: out = proj[:, :N/2] * gelu(proj[:, N/2:]), projection kept for the rows >= row0
    for (M, N, K) in [(256, 1280, 128), (1024, 2560, 320), (16384, 10240, 1280), (65536, 5120, 640)]:
        if QUICK and M == 65536:
            continue
        for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
            if QUICK and code == 1 and M > 1024:
                continue
            g = torch.Generator(device="cuda").manual_seed(M + N + K)
            row0 = 3 * M // 4
            a = torch.randn(M, K, device="cuda", generator=g).to(dt)
            w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dt)
            bias = torch.randn(N, device="cuda", generator=g).to(dt)
            pr = (a.float() @ w.float().t() + bias.float()).to(dt).float()
            ref = pr[:, :N // 2] * torch.nn.functional.gelu(pr[:, N // 2:])
            first, err = None, 0.0
            for rep in range(5):
                o = torch.full((M, N // 2), float("nan"), device="cuda", dtype=dt)
                pj = torch.zeros(M, N, device="cuda", dtype=dt)
                rc = lib.smi_op_gemm_geglu(code, P(a), P(w), P(bias), P(o), P(pj), M, N, K, row0, None)
                assert rc == 0, lib.smi_last_error()
                torch.cuda.synchronize()
                both = torch.cat([o, pj[row0:].reshape(-1, N // 2)], 0)
                if first is None:
                    first = both.clone()
                    err = float((o.float() - ref).abs().max() / ref.abs().max())
                    err = max(err, float((pj[row0:].float() - pr[row0:]).abs().max() / pr.abs().max()))
                    if float(pj[:row0].abs().max()) != 0.0:
                        err = float("inf")  # frozen rows of the projection must not be written
                elif not torch.equal(both, first):
                    err = float("inf")
                    break
            del ref, pr
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                lib.smi_op_gemm_geglu(code, P(a), P(w), P(bias), P(o), P(pj), M, N, K, row0, None)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 100
            dig = hashlib.sha256(first.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
            out[f"{code}:geglu{M}x{N}x{K}"] = [err, dig, us]
            print(f"[{arm}] geglu {M}x{N}x{K} dt={code}: err {err:.2e}  {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF/s",
                  flush=True)
    # implicit-GEMM 3x3 convs (NHWC activations, [Cout, 9 Cin] tap-major filters): image borders, several images per tile
    for (nb, H, W, Cin, Cout) in [(4, 16, 16, 64, 320), (3, 32, 64, 128, 640), (16, 128, 128, 320, 320),
                                  (16, 64, 64, 640, 640), (16, 32, 32, 1280, 1280), (16, 64, 64, 1920, 640),
                                  (16, 128, 128, 960, 320), (16, 32, 32, 2560, 1280), (16, 64, 64, 320, 640)]:
        if QUICK and nb * H * W * Cout * Cin > 16 * 64 * 64 * 640 * 640:
            continue
        for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
            if code == 1 and (Cin > 640 or (QUICK and nb * H * W > 4096)):
                continue
            g = torch.Generator(device="cuda").manual_seed(nb + H + Cin + Cout)
            x = torch.randn(nb, H, W, Cin, device="cuda", generator=g).to(dt)
            w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(dt)
            b = torch.randn(Cout, device="cuda", generator=g).to(dt)
            ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2),
                                             w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.float(),
                                             padding=1).permute(0, 2, 3, 1)
            first, err = None, 0.0
            for rep in range(5):
                y = torch.full((nb, H, W, Cout), float("nan"), device="cuda", dtype=dt)
                rc = lib.smi_op_conv3x3(code, P(x), P(w), P(b), P(y), nb, H, W, Cin, Cout, 1, 0, 0, H, W, None)
                assert rc == 0, lib.smi_last_error()
                torch.cuda.synchronize()
                if first is None:
                    first = y.clone()
                    err = float((y.float() - ref).abs().max() / ref.abs().max())
                elif not torch.equal(y, first):
                    err = float("inf")
                    break
            del ref
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                lib.smi_op_conv3x3(code, P(x), P(w), P(b), P(y), nb, H, W, Cin, Cout, 1, 0, 0, H, W, None)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 100
            dig = hashlib.sha256(first.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
            out[f"{code}:conv{nb}x{H}x{W}x{Cin}->{Cout}"] = [err, dig, us]
            print(f"[{arm}] conv {nb}x{H}x{W} {Cin}->{Cout} dt={code}: err {err:.2e}  {us:8.1f} us  "
                  f"{2*nb*H*W*Cout*9*Cin/us/1e6:7.1f} TF/s", flush=True)
    # up-sampler convs: nearest-2x up-sampling folded into the gather (Wout % 64 == 0 on gemm4)
    for (nb, H, W, Cin, Cout) in [(2, 32, 32, 64, 320), (1, 32, 64, 128, 640), (16, 64, 64, 640, 640),
                                  (16, 32, 32, 1280, 1280)]:
        if QUICK and Cin > 640:
            continue
        for dt, code in ((torch.float16, 0), (torch.bfloat16, 1)):
            if code == 1 and (Cin > 640 or (QUICK and nb > 2)):
                continue
            g = torch.Generator(device="cuda").manual_seed(nb + H + Cin + Cout + 1)
            x = torch.randn(nb, H, W, Cin, device="cuda", generator=g).to(dt)
            w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(dt)
            b = torch.randn(Cout, device="cuda", generator=g).to(dt)
            xu = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
            ref = torch.nn.functional.conv2d(xu, w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), b.float(),
                                             padding=1).permute(0, 2, 3, 1)
            del xu
            first, err = None, 0.0
            for rep in range(5):
                y = torch.full((nb, 2 * H, 2 * W, Cout), float("nan"), device="cuda", dtype=dt)
                rc = lib.smi_op_conv3x3(code, P(x), P(w), P(b), P(y), nb, H, W, Cin, Cout, 1, 1, 0, 2 * H, 2 * W, None)
                assert rc == 0, lib.smi_last_error()
                torch.cuda.synchronize()
                if first is None:
                    first = y.clone()
                    err = float((y.float() - ref).abs().max() / ref.abs().max())
                elif not torch.equal(y, first):
                    err = float("inf")
                    break
            del ref
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                lib.smi_op_conv3x3(code, P(x), P(w), P(b), P(y), nb, H, W, Cin, Cout, 1, 1, 0, 2 * H, 2 * W, None)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 100
            dig = hashlib.sha256(first.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
            out[f"{code}:upconv{nb}x{H}x{W}x{Cin}->{Cout}"] = [err, dig, us]
            print(f"[{arm}] upconv {nb}x{H}x{W} {Cin}->{Cout} dt={code}: err {err:.2e}  {us:8.1f} us  "
                  f"{2*nb*4*H*W*Cout*9*Cin/us/1e6:7.1f} TF/s", flush=True)
    json.dump(out, open(f"gpurun_out/check_gemm4_{arm}.json", "w"))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    os.makedirs("gpurun_out", exist_ok=True)
    for arm, env in (("base", {"SMI_GEMM_TUNE": "0"}), ("v4", {"SMI_GEMM": "5ph"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), arm], env={**os.environ, **env})
        if r.returncode != 0:
            print(f"arm {arm} failed rc={r.returncode}")
            sys.exit(1)
    a = json.load(open("gpurun_out/check_gemm4_base.json"))
    b = json.load(open("gpurun_out/check_gemm4_v4.json"))
    bad = 0
    for k in a:
        ea, da, ta = a[k]
        eb, db, tb = b[k]
        tol = 4e-3 if k.startswith("0:") else 2.5e-2
        ok = ea < tol and eb < tol and da == db
        bad += 0 if ok else 1
        print(f"{k:28s} base {ta:8.1f} us  v4 {tb:8.1f} us  x{ta/tb:5.2f}  err {ea:.1e}/{eb:.1e}  "
              f"{'bit-identical' if da == db else 'DIFFERENT'}  {'ok' if ok else 'FAIL'}")
    print("FAILED" if bad else "ALL OK")
    sys.exit(1 if bad else 0)
