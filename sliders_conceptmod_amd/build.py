"""Builds sliders_conceptmod_amd/libsmi_hip.so (gfx950 code object + host launchers + C ABI) in-tree with hipcc.

    python -m sliders_conceptmod_amd.build        # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmi_hip.so")
SOURCES = ["gemm.hip", "gemm2.hip", "gemm3.hip", "gemm4.hip", "attention.hip", "norm.hip", "elementwise.hip", "lora.hip", "engine.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# attention: the softmax works on MFMA results every tile; with the default AGPR-form MFMA hipcc shuttles every
# accumulator through v_accvgpr_read/write (481 moves per key tile, the kernel was VALU-bound at 91 % VALU busy).
# gfx950's unified register file lets MFMA accumulate in VGPRs directly.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("smi_common.h", "kernels.h")] + [
        os.path.join(os.path.dirname(HERE), "include", "smi.h")]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
