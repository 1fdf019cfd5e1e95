"""Cut a rocprofv3 --kernel-trace CSV of `bench.py` into train steps and total the kernels of the STEADY-STATE steps.

    python tools/trace_steps.py <kernel_trace.csv> <bench.json> [<out.json>]

A step ends with its `clip_adamw_kernel` launch (the optimiser is the last kernel of SliderStep.train_step), so the trace
is cut at those launches.  The first segment (engine creation, weight packing and -- without SMI_TUNE_FILE -- the tile
tuner's candidate launches) and everything after the last timed step (pre-roll, the HIP-event profiled step) are dropped;
what is left are the warm-up steps >= 2 and the timed steps.  Per kernel class the script prints the mean device time per
step, and recomputes `roofline.frac` = algorithmic FLOPs of the GEMM / conv launches (from the bench line, which gets
them from the engine's shape walk) / the traced duration of the gemm_*_kernel launches -- the profiler-observed
counterpart of the HIP-event figure on the bench line."""
import collections
import csv
import json
import sys


def klass(name):
    n = name
    if "gemm_5ph_kernel" in n or "gemm_8ph_kernel" in n or "gemm_glds_kernel" in n or "gemm_nt_kernel" in n:
        return "gemm+conv"
    if "conv3x3_small" in n:
        return "gemm+conv"
    if "attn_" in n:
        return "attention"
    if "gn_" in n or "ln_" in n or "groupnorm" in n or "layernorm" in n:
        return "norm"
    if "lora_" in n or "dora_" in n or "wgrad" in n:
        return "lora"
    if n.startswith("void at::") or "rocclr" in n or "at::native" in n:
        return "torch"
    return "elementwise"


def main():
    trace, bench = sys.argv[1], sys.argv[2]
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "clip_adamw_kernel" in r[2]]
    line = [l for l in open(bench).read().splitlines() if l.startswith("{")][-1]
    b = json.loads(line)
    n_steady = b["warmup"] + b["steps"]          # steps before the pre-roll / profiled step
    segs = []
    lo = 0
    for c in cuts:
        segs.append(rows[lo:c + 1])
        lo = c + 1
    steady = segs[1:n_steady]                     # drop step 1 (set-up) and everything after the timed steps
    per = []
    for s in steady:
        d = collections.Counter()
        n = collections.Counter()
        for st, en, name in s:
            d[klass(name)] += (en - st) * 1e-6
            n[klass(name)] += 1
        per.append({"wall_ms": (s[-1][1] - s[0][0]) * 1e-6, "ms": dict(d), "launches": dict(n)})
    if not per:
        print("no steady-state step found", file=sys.stderr)
        sys.exit(1)
    classes = sorted({k for p in per for k in p["ms"]})
    mean = {k: sum(p["ms"].get(k, 0.0) for p in per) / len(per) for k in classes}
    launches = {k: sum(p["launches"].get(k, 0) for p in per) / len(per) for k in classes}
    wall = sum(p["wall_ms"] for p in per) / len(per)
    mm_tflop = b["roofline"]["algorithmic_tflop_per_step"]
    frac_traced = mm_tflop / (mean["gemm+conv"] * 1e-3) / b["roofline"]["peak"]
    out = {"steady_steps": len(per), "segments_in_trace": len(segs), "wall_ms_per_step": wall,
           "device_ms_per_step_by_class": mean, "launches_per_step_by_class": launches,
           "gemm_conv_algorithmic_tflop_per_step": mm_tflop,
           "roofline_frac_from_trace": frac_traced, "roofline_frac_on_bench_line": b["roofline"]["frac"],
           "bench_ms_per_step": b["ms_per_step"], "workload": b["config"]["workload"]}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
